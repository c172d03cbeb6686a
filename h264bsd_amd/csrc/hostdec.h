/*
 * hostdec.h — internal declarations of the host-side H.264 baseline parser.
 *
 * The host keeps everything that is bit-serial or control-plane (SURVEY.md §2 rows marked "host"):
 * Annex-B/NAL extraction, SPS/PPS/VUI, slice headers, CAVLC, macroblock-layer syntax, motion-vector
 * and intra-mode prediction, DPB bookkeeping.  It never touches a pixel; its product is the packed
 * frame job of framejob.h, handed to a JobSink (the HIP engine, or a capture callback).
 *
 * Written from the H.264 specification (clause numbers cited inline); behaviour that is specific to
 * the reference decoder (return-code protocol, output timing) cites /root/reference file:line.
 */
#ifndef H264BSD_AMD_HOSTDEC_H
#define H264BSD_AMD_HOSTDEC_H

#include <stdint.h>
#include <stddef.h>
#include "framejob.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- bit reader */
typedef struct BitReader {
    const uint8_t *buf;   /* RBSP bytes; at least 8 readable zero bytes follow buf[size-1] */
    uint32_t size_bits;
    uint32_t pos;         /* bits consumed; may run past size_bits, callers test br_overrun() */
} BitReader;

static inline uint32_t br_peek32(const BitReader *b)
{
    const uint8_t *p = b->buf + (b->pos >> 3);
    uint64_t v = ((uint64_t)p[0] << 56) | ((uint64_t)p[1] << 48) | ((uint64_t)p[2] << 40) |
                 ((uint64_t)p[3] << 32) | ((uint64_t)p[4] << 24) | ((uint64_t)p[5] << 16) |
                 ((uint64_t)p[6] << 8) | (uint64_t)p[7];
    return (uint32_t)((v << (b->pos & 7)) >> 32);
}
static inline int br_overrun(const BitReader *b) { return b->pos > b->size_bits; }
static inline uint32_t br_left(const BitReader *b) { return b->pos >= b->size_bits ? 0 : b->size_bits - b->pos; }
static inline void br_skip(BitReader *b, uint32_t n)
{
    /* saturate well past the end so an overrun stays an overrun and peeks stay in the pad */
    if (b->pos <= b->size_bits) b->pos += n;
    if (b->pos > b->size_bits) b->pos = b->size_bits + 1;
}
static inline uint32_t br_get(BitReader *b, uint32_t n) /* 0 <= n <= 32 */
{
    if (n == 0) return 0;
    uint32_t v = br_peek32(b) >> (32 - n);
    br_skip(b, n);
    return v;
}
static inline uint32_t br_get1(BitReader *b) { return br_get(b, 1); }
/* Exp-Golomb ue(v), clause 9.1; code numbers up to 2^32-2, anything longer flags an overrun */
static inline uint32_t br_ue(BitReader *b)
{
    uint32_t w = br_peek32(b);
    if (w & 0x80000000u) { br_skip(b, 1); return 0; }
    if (w == 0) {
        /* 32 or more leading zeros: only the 33-bit/65-bit forms are legal, treat as corrupt */
        b->pos = b->size_bits + 1;
        return 0xFFFFFFFFu;
    }
    uint32_t lz = (uint32_t)__builtin_clz(w);
    if (lz <= 15) {
        uint32_t v = (w >> (31 - 2 * lz)) - 1;
        br_skip(b, 2 * lz + 1);
        return v;
    }
    br_skip(b, lz + 1);
    uint32_t suffix = br_get(b, lz);
    return ((1u << lz) - 1) + suffix;
}
static inline int32_t br_se(BitReader *b)
{
    uint32_t k = br_ue(b);
    return (k & 1) ? (int32_t)((k + 1) >> 1) : -(int32_t)(k >> 1);
}
/* more_rbsp_data(), clause 7.2: false when only the stop bit and alignment zeros remain */
static inline int br_more_rbsp_data(const BitReader *b)
{
    uint32_t left = br_left(b);
    if (left == 0) return 0;
    if (left > 8) return 1;
    return (br_peek32(b) >> (32 - left)) != (1u << (left - 1));
}

/* ---------------------------------------------------------------- parameter sets */
#define HD_MAX_SPS 32
#define HD_MAX_PPS 256

typedef struct Sps {
    uint8_t  valid;
    uint8_t  profile_idc, level_idc, constraint_flags;
    uint8_t  sps_id;
    uint32_t max_frame_num;           /* 1 << log2_max_frame_num                     */
    uint8_t  poc_type;
    uint32_t max_poc_lsb;
    uint8_t  delta_pic_order_always_zero;
    int32_t  offset_for_non_ref_pic, offset_for_top_to_bottom_field;
    uint32_t num_ref_frames_in_poc_cycle;
    int32_t  offset_for_ref_frame[255];
    uint32_t num_ref_frames;
    uint8_t  gaps_in_frame_num_allowed;
    uint32_t width_mbs, height_mbs;
    uint8_t  cropping;
    uint32_t crop_left, crop_right, crop_top, crop_bottom;
    uint32_t max_dpb_size;
    /* VUI subset */
    uint8_t  vui_present;
    uint8_t  aspect_ratio_present, aspect_ratio_idc;
    uint32_t sar_width, sar_height;
    uint8_t  video_signal_type_present, video_full_range, colour_description_present;
    uint8_t  matrix_coefficients;
    uint8_t  bitstream_restriction;
    uint32_t num_reorder_frames, max_dec_frame_buffering;
} Sps;

typedef struct Pps {
    uint8_t  valid;
    uint8_t  pps_id, sps_id;
    uint8_t  pic_order_present;
    uint32_t num_slice_groups;
    uint8_t  slice_group_map_type;
    uint32_t run_length[8];
    uint32_t top_left[8], bottom_right[8];
    uint8_t  slice_group_change_direction;
    uint32_t slice_group_change_rate;
    uint32_t pic_size_in_map_units;
    uint8_t *slice_group_id;          /* type 6 */
    uint32_t num_ref_idx_l0_active;
    int32_t  pic_init_qp;             /* 26 + pic_init_qp_minus26                    */
    int32_t  chroma_qp_index_offset;
    uint8_t  deblocking_filter_control_present;
    uint8_t  constrained_intra_pred;
    uint8_t  redundant_pic_cnt_present;
} Pps;

/* ---------------------------------------------------------------- slice header */
typedef struct ReorderCmd { uint8_t idc; uint32_t val; } ReorderCmd;
typedef struct MmcoCmd { uint8_t op; uint32_t a, b; } MmcoCmd;

typedef struct SliceHdr {
    uint32_t first_mb;
    uint8_t  is_p;                    /* slice_type%5 == 0 (P) else I                */
    uint32_t pps_id;
    uint32_t frame_num;
    uint32_t idr_pic_id;
    uint32_t poc_lsb;
    int32_t  delta_poc_bottom;
    int32_t  delta_poc[2];
    uint32_t redundant_pic_cnt;
    uint32_t num_ref_idx_active;
    uint8_t  reorder_flag;
    uint32_t n_reorder;
    ReorderCmd reorder[18];
    /* dec_ref_pic_marking */
    uint8_t  no_output_of_prior_pics, long_term_reference_flag;
    uint8_t  adaptive_marking;
    uint32_t n_mmco;
    MmcoCmd  mmco[36];
    int32_t  slice_qp_delta;
    uint8_t  disable_deblocking_filter_idc;
    int32_t  alpha_off, beta_off;     /* already x2                                  */
    uint32_t slice_group_change_cycle;
} SliceHdr;

/* ---------------------------------------------------------------- per-MB persistent metadata */
typedef struct MbInfo {
    uint8_t  kind;          /* FJ_MB_* of the last decode of this MB                             */
    uint8_t  mb_type;       /* reference numbering: 0 P_Skip, 1..5 P, 6 I4x4, 7..30 I16x16, 31 PCM */
    uint8_t  qp;
    uint8_t  dbk_idc;       /* disable_deblocking_filter_idc ... */
    int8_t   alpha_off, beta_off, cqp_off;   /* ... FilterOffsetA / B and chroma_qp_index_offset of the last slice that STARTED on this
                               macroblock, in this picture or an earlier one: the reference stamps them on its mbStorage_t before it
                               parses the macroblock (SetMbParams, src/h264bsd_slice_data.c:53-66,140) and filters with whatever is there */
    uint8_t  tc[24];        /* total_coeff per 4x4 block, H.264 block order (luma 0-15, Cb, Cr)   */
    int8_t   i4mode[16];    /* Intra4x4PredMode, H.264 block order                                */
    /* ref_idx / ref_slot / mv persist like the reference's mbStorage_t.refPic / refAddr / mv: across decodes and across
     * pictures, written only where MvPrediction writes them (hd_mb.c restore_unwritten) */
    int8_t   ref_idx[4];
    uint8_t  ref_slot[4];   /* frame buffer per quadrant; 0xFF = none (refAddr NULL: never written, or written as missing) */
    int16_t  mv[16][2];     /* H.264 block order                                                  */
} MbInfo;

/* ---------------------------------------------------------------- DPB bookkeeping */
enum { DPB_UNUSED = 0, DPB_NON_EXISTING, DPB_SHORT, DPB_LONG };

typedef struct DpbPic {
    uint8_t  status;
    uint8_t  to_be_displayed;
    uint8_t  is_idr;
    uint32_t frame_num;
    int32_t  pic_num;       /* FrameNumWrap-derived PicNum, or LongTermPicNum for long-term */
    int32_t  poc;
    uint32_t pic_id, num_err_mbs;
    uint8_t  slot;          /* frame buffer (DPB slot in HBM) that holds / will hold this picture's pixels */
} DpbPic;

typedef struct OutPic { uint8_t slot; uint8_t is_idr; uint32_t pic_id, num_err_mbs; } OutPic;

typedef struct Dpb {
    uint32_t n_slots;       /* dpb_size + 1 */
    uint32_t dpb_size;
    uint32_t max_ref_frames, max_frame_num;
    uint32_t max_long_term_idx; /* 0xFFFF = no long-term indices */
    uint8_t  no_reordering;
    uint32_t num_ref_frames, fullness;
    uint32_t prev_ref_frame_num;
    uint8_t  last_contains_mmco5;
    int32_t  cur;           /* position of the picture being decoded (always dpb_size), -1 none */
    DpbPic   pic[FJ_MAX_SLOTS]; /* by POSITION, like the reference's dpbStorage_t.buffer[] (see hd_dpb.c) */
    int8_t   list[33];          /* RefPicList0: index -> position, -1 = none; persists between pictures */
    OutPic   out[FJ_MAX_SLOTS + 1];
    uint32_t n_out, out_idx;
} Dpb;

/* ---------------------------------------------------------------- job sink (device boundary) */
typedef struct JobSink {
    void *user;
    /* (re)configure for a sequence: n_slots frames of frame_bytes each. 0 = ok */
    int  (*configure)(void *user, uint32_t width_mbs, uint32_t height_mbs, uint32_t n_slots);
    /* optional: a buffer of at least `bytes` for the NEXT frame job to be built in place (the engine hands out pinned
     * staging memory, so that no copy is needed between the parser and the H2D transfer).  The buffer returns to the
     * sink with submit() — or with the next acquire() when the picture is abandoned. */
    uint8_t *(*acquire)(void *user, uint32_t bytes);
    /* one finished picture; blob is only valid during the call unless it came from acquire(). 0 = ok */
    int  (*submit)(void *user, const uint8_t *blob, uint32_t bytes);
    /* make slot's pixels available at host address; returns pointer or NULL */
    uint8_t *(*fetch)(void *user, uint32_t slot);
    /* optional: fetch() in two steps, for callers that have other work between them (the batch calls of api.c: a parser thread starts the
     * next instance's picture on its way before it waits for this one's).  begin enqueues and returns at once (0 = ok); end waits and
     * returns what fetch() would have.  One begin per sink at a time, every begin is followed by its end. */
    int  (*fetch_begin)(void *user, uint32_t slot);
    uint8_t *(*fetch_end)(void *user);
    /* colour conversion of a slot into a host buffer of width*height u32; fmt 0 RGBA 1 BGRA 2 YCbCrA */
    uint32_t *(*fetch_converted)(void *user, uint32_t slot, int fmt);
    /* slot (cropped to x0,y0,w,h; fmt 0..2 converted, 3 = I420) as a DEVICE pointer; *stream = the HIP stream used */
    void *(*fetch_device)(void *user, uint32_t slot, int fmt, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h, void **stream);
    void (*close)(void *user);
    /* optional: how often the sink's device has reported an error so far, as far as it knows without waiting (the engine's
     * tripwire counter, folded in wherever the host waits for the device anyway; monotonic).  More than when the decoder was
     * created = pictures may not have been produced as submitted: the parser stops relying on what the frame buffers hold
     * (copy elision, hd_job_finish) */
    uint32_t (*errors)(void *user);
} JobSink;

/* ---------------------------------------------------------------- decoder instance */
typedef struct PocState {
    uint32_t prev_poc_lsb; int32_t prev_poc_msb;
    uint32_t prev_frame_num, prev_frame_num_offset;
    uint8_t  contains_mmco5;
} PocState;

typedef struct HostDec {
    Sps *sps[HD_MAX_SPS];
    Pps *pps[HD_MAX_PPS];
    int  active_sps_id, active_pps_id, old_sps_id;   /* -1 = none */
    Sps *active_sps; Pps *active_pps;
    uint8_t pending_activation;
    uint8_t no_reordering_app;
    uint8_t input_readonly;         /* h264bsdmiSetInputReadOnly: never write to the caller's buffer (hd_extract_nal) */

    uint32_t pic_size_mbs, width_mbs, height_mbs;
    uint32_t width_magic;           /* floor(2^32 / width_mbs) + 1 while pic_size_mbs < 2^16 (every level of the standard up to 5.2), else 0:
                                       addr / width_mbs = (addr * width_magic) >> 32 — hd_mb_row() */
    MbInfo  *mb;
    uint32_t *slice_group_map;
    /* per-picture state that is reset for every picture lives in compact arrays of its own, not in MbInfo: the
     * reset touches 5 bytes per macroblock instead of a cache line (the parser is memory-bound at scale) */
    uint8_t  *mb_decoded;   /* times decoded in the current picture                                 */
    uint32_t *mb_slice_id;  /* slice that last touched the macroblock; 0 = none in this picture       */
    uint32_t *mb_rec_sid;   /* slice whose decode wrote the macroblock's record in the job; 0 = none  */
    uint32_t num_decoded_mbs, slice_id;
    uint32_t last_mb_addr;

    /* access-unit tracking, clause 7.4.1.2.4 */
    uint8_t  aub_first_call;
    uint8_t  prev_nal_type, prev_nal_ref_idc;
    uint32_t aub_prev_frame_num, aub_prev_idr_pic_id, aub_prev_poc_lsb;
    int32_t  aub_prev_delta_poc_bottom, aub_prev_delta_poc[2];
    uint8_t  pic_started, valid_slice_in_au, skip_redundant;
    uint8_t  cur_nal_type, cur_nal_ref_idc;   /* of the last stored slice */
    uint32_t current_pic_id;
    uint32_t pic_seq;

    SliceHdr slice;        /* last successfully decoded slice header */
    PocState poc;
    Dpb      dpb;

    /* NAL staging: unescaped payload with 16 bytes of zero padding */
    uint8_t *nal_buf; uint32_t nal_cap, nal_size;
    const uint8_t *prev_buf_ptr; uint32_t prev_bytes_consumed; uint8_t prev_buf_not_finished;

    /* frame job under construction */
    uint8_t *job; uint32_t job_cap;
    uint8_t  job_from_sink;  /* job points into memory handed out by sink.acquire (not ours to free) */
    uint32_t coef_blocks;  /* blocks written so far */
    uint32_t coef_cap_blocks; /* blocks the coefficient section of the job can hold */
    uint32_t n_inter, n_intra;
    uint8_t  job_open;

    /* "ghost" pixels (damaged streams only, see hd_core.c mark_slice_corrupted): the macroblocks of slices that were
     * rolled back after they had been reconstructed, kept so that their pixels can be reproduced if a macroblock that is
     * never reconstructed (FJ_MB_STALE) ends up showing them */
    uint8_t *ghost_buf; size_t ghost_len, ghost_cap;
    uint8_t *mb_ghost;      /* per macroblock: pixels written by a slice that was rolled back; allocated on first use */
    uint8_t  ghost_dirty, ghost_needed;

    /* macroblocks decoded again by a redundant slice (only possible while the primary picture is incomplete, i.e. in
     * damaged streams): the reference keeps the pixels of the first decode and the metadata of the last one
     * (macroblock_layer.c:985-1046 run again, the writes are skipped: :1006, :1110, intra_prediction.c:526,
     * inter_prediction.c:468).  The first decode's record and motion vectors are kept here; at the end of the picture
     * they go into a reconstruction-only job, and the picture's own job only deblocks (FjHeader.dbk_only). */
    struct RedoMb { uint32_t addr; FjMbRec rec; int16_t mv[32]; } *redo;
    uint32_t n_redo, redo_cap;
    /* ... and pixel-making versions written ON TOP of a first decode (FJ_PRED_PHASE2: a macroblock that a failed slice
     * un-decoded, decoded anew) whose metadata a still later decode replaced: they get a reconstruction-only job of their
     * own between the two */
    struct RedoMb *redo2;
    uint32_t n_redo2, redo2_cap;
    uint8_t *mb_redone;     /* per macroblock, allocated on first use */
    uint8_t  pic_irregular;         /* a slice of this picture was rolled back (mark_slice_corrupted) */
    uint8_t  slice_ids_rewritten;   /* a redundant slice ran over macroblocks of this picture (it restamps their slice id even when it fails) */

    JobSink sink;
    uint8_t  sink_configured;
    uint32_t *conv_buf; size_t conv_cap;

    /* Copy elision (hd_core.c, hd_tiles_*): what every tile of every DPB slot holds, as the serial number of the job that
     * last produced NEW content for it; a whole-tile copy that nothing filters afterwards inherits the number of its source.
     * A copy whose destination already carries the source's number would write the bytes that are there: it is left out
     * of the job's copy list.  tile_ver = [n_slots][pic_size_mbs]; tile_pending = the numbers of the job being submitted,
     * committed when the sink has taken it. */
    uint8_t  copy_elision;          /* on for decoders bound to a device, off in capture mode unless asked for (h264bsdmiSetCopyElision) */
    uint32_t sink_errors_at_start;  /* sink.errors() when this decoder was bound to its sink: only LATER device errors switch its elision off */
    uint8_t  tile_uncommitted;      /* a job was finalised but its submission was never confirmed: nothing is known any more */
    uint32_t *tile_ver, *tile_pending;
    uint32_t tile_slots, tile_mbs, tile_serial, tile_pending_slot;
    uint32_t n_elided;              /* macroblocks left out of the last job's copy list */
} HostDec;

/* the row of a macroblock address without a division (the parser derives it for every macroblock) */
static inline uint32_t hd_mb_row(const HostDec *d, uint32_t addr)
{
    return d->width_magic ? (uint32_t)(((uint64_t)addr * d->width_magic) >> 32) : addr / d->width_mbs;
}

/* what fj_finalize_ex needs to leave copies out (NULL: a frame job is a pure function of records, vectors and coefficients) */
typedef struct FjElide {
    const uint32_t *ver;            /* [n_slots][n_mbs] */
    uint32_t n_slots, cur_slot;
    uint32_t serial;                /* number for the tiles this job gives new content */
    uint32_t *out;                  /* [n_mbs]: the numbers of cur_slot after this job */
    uint32_t n_elided;
} FjElide;

/* return codes shared with the public API (reference src/h264bsd_decoder.h:45-52) */
enum { HD_RDY = 0, HD_PIC_RDY = 1, HD_HDRS_RDY = 2, HD_ERROR = 3, HD_PARAM_SET_ERROR = 4, HD_MEMALLOC_ERROR = 5 };

/* hd_nal.c */
int hd_extract_nal(HostDec *d, uint8_t *stream, uint32_t len, uint32_t *read_bytes);
/* hd_params.c */
int hd_parse_sps(BitReader *br, Sps *sps);
int hd_parse_pps(BitReader *br, Pps *pps);
void hd_free_pps(Pps *pps);
int hd_sps_equal(const Sps *a, const Sps *b);
int hd_check_pps(const Pps *p, const Sps *s);
/* hd_slice.c */
int hd_parse_slice_header(BitReader *br, SliceHdr *sh, const Sps *sps, const Pps *pps, int nal_type, int nal_ref_idc);
int hd_peek_pps_id(const BitReader *br, uint32_t *pps_id);
int32_t hd_decode_poc(PocState *st, const Sps *sps, const SliceHdr *sh, int nal_type, int nal_ref_idc);
void hd_slice_group_map(uint32_t *map, const Pps *pps, uint32_t change_cycle, uint32_t w, uint32_t h);
/* hd_dpb.c */
int  hd_dpb_reset(Dpb *dpb, uint32_t dpb_size, uint32_t max_ref_frames, uint32_t max_frame_num, int no_reordering);
int  hd_dpb_alloc_current(Dpb *dpb);
int  hd_dpb_cur_slot(const Dpb *dpb);
int  hd_dpb_ref_slot(const Dpb *dpb, uint32_t ref_idx);
void hd_dpb_init_ref_list(Dpb *dpb);
int  hd_dpb_reorder_ref_list(Dpb *dpb, const SliceHdr *sh);
int  hd_dpb_check_gaps(Dpb *dpb, uint32_t frame_num, int is_ref, int gaps_allowed);
int  hd_dpb_mark_current(Dpb *dpb, const SliceHdr *sh, int is_ref, int is_idr, int32_t poc, uint32_t pic_id, uint32_t err_mbs);
void hd_dpb_flush(Dpb *dpb);
const OutPic *hd_dpb_next_output(Dpb *dpb);
/* hd_cavlc.c */
extern int hd_no_fast_skip;   /* HD_NO_FAST_SKIP in the environment: hd_mb.c takes the general path for every macroblock (A/B of the fast paths) */
extern int hd_trace;   /* HD_TRACE in the environment, read once by hd_cavlc_init() (debugging aid) */
void hd_cavlc_init(void);
/* Decodes one residual block.  coef[] (raster 4x4 via zig-zag, or plain order for chroma DC) must
 * be zeroed by the caller; returns total_coeff or -1 on a bitstream error.
 * max_coeff: 16, 15 (AC: scan positions 1..15) or 4 (chroma DC, nc == -1).
 * *spill (may be NULL): the level a damaged 15-coefficient block places one element past its end (else 0). */
int hd_cavlc_block(BitReader *br, int nc, int max_coeff, int16_t *coef, int *spill);
int hd_cavlc_block_sum(BitReader *br, int nc, int max_coeff, int16_t *coef, int *spill, uint32_t *abs_sum);
/* ... given the sums of the level magnitudes of the macroblock's luma blocks, chroma DC block and chroma AC blocks: 1 = the
 * bound proves every residual sample in range (no Intra16x16 DC block), 0 = hd_residual_out_of_range() has to look */
int hd_residual_bound_ok(uint32_t sum_luma, uint32_t sum_cdc, uint32_t sum_cac, int qp_y, int qp_c);
int hd_residual_bound_ok4(uint32_t sum_luma, uint32_t sum_cdc, uint32_t sum_cac, uint32_t sum_ldc, int qp_y, int qp_c);
/* hd_resid.c */
int hd_residual_out_of_range(const int16_t *blk, uint32_t coded, int qp_y, int qp_c, int is_i16);
/* hd_mb.c */
uint32_t hd_next_mb_in_group(const uint32_t *map, uint32_t n, uint32_t addr);
int hd_decode_slice_data(HostDec *d, BitReader *br, const SliceHdr *sh, int nal_ref_idc);
/* hd_core.c: set the first decode's record of a macroblock aside before a redundant slice's decode replaces it (0 = ok) */
int hd_redo_keep_first(HostDec *d, uint32_t addr, const FjMbRec *rec, const int16_t *mv);
void hd_dense_mv_read(const FjMbRec *r, const int16_t *dense, int16_t out[32]);
/* hd_api.c helpers used across files */
int  hd_job_begin(HostDec *d);
int  hd_job_finish(HostDec *d, int is_idr, int single_job);
int  fj_finalize(uint8_t *job, uint32_t cap, uint32_t coef_blocks);
int  fj_finalize_ex(uint8_t *job, uint32_t cap, uint32_t coef_blocks, FjElide *elide);

#ifdef __cplusplus
}
#endif
#endif
