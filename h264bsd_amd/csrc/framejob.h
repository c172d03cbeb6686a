/*
 * framejob.h — the packed per-picture work description ("frame job") that crosses the
 * host → device boundary.
 *
 * The host parser (hostdec_*.c) turns one coded picture into ONE contiguous blob: a fixed header,
 * a 32-byte record per macroblock, 16 motion vectors per macroblock, the intra dependency-level
 * schedule and a dense stream of 4x4 coefficient blocks.  The HIP kernels (kernels.hip) and the
 * CPU test oracle (oracle/pixel_oracle.c) both consume exactly this format, so a blob captured on
 * the host can be replayed on either.
 *
 * What the blob replaces in the reference: the per-MB hand-off at seam A/B of SURVEY.md §1 —
 * macroblockLayer_t + mbStorage_t (reference src/h264bsd_macroblock_layer.h:140-185) as consumed by
 * h264bsdDecodeMacroblock (src/h264bsd_macroblock_layer.c:965) and h264bsdFilterPicture
 * (src/h264bsd_deblocking.c:575).  Motion vectors and the levels inside a coefficient block are in RASTER
 * order (blk = 4*by + bx; c = 4*row + col); the per-block bit fields (`coded`, `i4mode`) keep the H.264
 * block order (z-order: z = 8*(by>>1) + 4*(bx>>1) + 2*(by&1) + (bx&1)), which is also the order in which a
 * macroblock's coefficient blocks are stored.
 *
 * All integers little-endian; every section starts 32-byte aligned.
 * Section order in the blob: header | records | motion vectors | coefficients | level starts |
 * intra index | copy list (runs of <= 4 MBs) | general-inter list | non-trivial deblocking index.
 */
#ifndef H264BSD_AMD_FRAMEJOB_H
#define H264BSD_AMD_FRAMEJOB_H

#include <stdint.h>

#define FJ_MAGIC      0x314A4648u /* "HFJ1" */
#define FJ_MAX_SLOTS  17          /* max_dec_frame_buffering(16) + the picture being decoded */

/* FjMbRec.kind */
#define FJ_MB_INTER   0   /* P_Skip and all P partitions: per-4x4 mv + per-8x8 reference slot   */
#define FJ_MB_I4x4    1
#define FJ_MB_I16x16  2
#define FJ_MB_IPCM    3
#define FJ_MB_CONCEAL_I 4 /* lost macroblock, synthesised from its decoded neighbours (reference ConcealMb,
                             src/h264bsd_conceal.c:266-560): avail = FJ_CONC_* neighbours used, coef_idx = position
                             in the concealment order (later ones may read earlier ones)                      */
#define FJ_MB_CONCEAL_P 5 /* lost macroblock of a P picture: copy of the co-located macroblock of DPB slot
                             ref_slot[0] (conceal.c:318-343); travels in the copy list like a mv-0 P_Skip       */
#define FJ_MB_STALE   6   /* intra macroblock that counts as decoded but was never reconstructed (its reconstruction
                             failed and the reference's slice roll-back does not reach it, src/h264bsd_slice_data.c:
                             318-333): the pixels stay what the frame buffer held, the deblocking filter treats it as
                             the intra macroblock it was parsed as (qp_y, dbk, offsets valid; nothing else)        */
#define FJ_MB_ABSENT  255 /* macroblock not covered by any decoded slice: pixels left untouched */

/* FjMbRec.avail of a FJ_MB_CONCEAL_I macroblock: which neighbours were decoded (or already concealed) when the
 * reference's concealment loop reaches it */
#define FJ_CONC_LEFT  1
#define FJ_CONC_ABOVE 2
#define FJ_CONC_RIGHT 4
#define FJ_CONC_BELOW 8

/* FjMbRec.avail bits: neighbour usable for intra prediction (in picture, same slice, and not an
 * inter MB when constrained_intra_pred is on) — reference src/h264bsd_neighbour.c:370-381,
 * src/h264bsd_intra_prediction.c:644-655 */
#define FJ_AVAIL_A 1  /* left       */
#define FJ_AVAIL_B 2  /* above      */
#define FJ_AVAIL_C 4  /* above-right*/
#define FJ_AVAIL_D 8  /* above-left */

/* FjMbRec.dbk bits — reference GetMbFilteringFlags, src/h264bsd_deblocking.c:289-320 */
#define FJ_PRED_PHASE2 0x80u  /* in a deblock-only job (FjHeader.dbk_only): this macroblock is nevertheless reconstructed, on top of
                                 the pixels of the job before — it replaced pixels that other macroblocks had already predicted
                                 from (concealment of, or a later slice over, a macroblock that a failed redundant slice un-decoded) */
#define FJ_PRED_UNIFORM_MV 0x40u /* inter macroblocks: the 16 motion vectors are 16 copies of one vector and the four references are equal.
                                  Before fj_finalize() it is a statement of whoever built the record (the parser: P_Skip, P_L0_16x16):
                                  the one vector is ALREADY in FjMbRec.mv and the macroblock's entry of the dense array at mv_off is not
                                  written and will not be read; a job built elsewhere may leave it clear and fill the dense entry —
                                  fj_finalize() then compares the sixteen vectors itself.  fj_finalize() sets the bit wherever it holds,
                                  and in a FINISHED job it says where the vectors are: set -> FjMbRec.mv is the macroblock's one vector,
                                  clear -> the 16 vectors are entry FjMbRec.mvx of the sparse section at FjHeader.mvx_off. */
#define FJ_PRED_PARTS_SHIFT 4  /* inter macroblocks, pred bits 4-5: where the macroblock TYPE allows motion to differ inside it — the
                                 deblocking filter compares motion vectors / references only there (reference
                                 deblocking.c:1266-1345) */
#define FJ_PARTS_8x8   0      /* P_8x8, P_8x8ref0: every inner edge */
#define FJ_PARTS_16x16 1      /* P_L0_16x16, P_Skip: none           */
#define FJ_PARTS_16x8  2      /* the middle horizontal edge         */
#define FJ_PARTS_8x16  3      /* the middle vertical edge           */
#define FJ_DBK_LEFT  1
#define FJ_DBK_TOP   2
#define FJ_DBK_INNER 4

/* FjMbRec.ref_slot[0] of a macroblock in the intra schedule (kinds I4x4, I16x16, IPCM, CONCEAL_I): the neighbours whose
 * samples it reads.  It may start as soon as those of them that are themselves in the intra schedule are done
 * (everything else was reconstructed by the earlier kernels).  Bit b and bit b^4 are opposite directions. */
#define FJ_NEED_L   0x01  /* (x-1, y)   */
#define FJ_NEED_UL  0x02  /* (x-1, y-1) */
#define FJ_NEED_U   0x04  /* (x,   y-1) */
#define FJ_NEED_UR  0x08  /* (x+1, y-1) */
#define FJ_NEED_R   0x10  /* (x+1, y)   */
#define FJ_NEED_DR  0x20  /* (x+1, y+1) */
#define FJ_NEED_D   0x40  /* (x,   y+1) */
#define FJ_NEED_DL  0x80  /* (x-1, y+1) */

/* FjMbRec.coded bits */
#define FJ_CODED_LUMA_DC   (1u << 24) /* Intra16x16 DC block present                        */
#define FJ_CODED_CHROMA_DC (1u << 25) /* chroma DC block (Cb[0..3], Cr[4..7]) present        */
#define FJ_CODED_LUMA_DC_RAW (1u << 26) /* with FJ_CODED_LUMA_DC: the block holds the FINAL DC of luma block (bx,by) at
                                         raster position 4*by+bx — no Hadamard, no scaling.  Only a damaged stream
                                         produces it (a chroma AC block whose last coefficient lands in the luma DC
                                         array of an Intra16x16 macroblock that coded no DC, hd_mb.c parse_residual) */

#define FJ_CODED_WIDE (1u << 27)     /* the cheap magnitude bound of hd_resid.c (sum of the level magnitudes x the largest scale <= 32735)
                                         does NOT hold for this macroblock: some intermediate of its inverse transforms may leave 16
                                         bits.  Clear (almost always): every intermediate fits 16 bits and the residual lies in
                                         [-512, 511], so the inter kernels run luma and chroma as ONE packed 16-bit transform
                                         (kernels.hip.h mb_residual_pk).  Set by the parser; h264bsdmiJobFinalize() sets it for
                                         hand-built jobs.  Inter macroblocks only (the intra kernel always computes in 32 bits) */

typedef struct FjHeader {
    uint32_t magic;
    uint32_t total_bytes;     /* whole blob                                                   */
    uint16_t width_mbs;
    uint16_t height_mbs;
    uint32_t n_mbs;
    uint8_t  cur_slot;        /* DPB slot that receives this picture                          */
    uint8_t  is_idr;
    uint8_t  n_slots;         /* DPB slots of the owning decoder (dpbSize+1)                  */
    uint8_t  any_deblock;     /* 0: no MB of the picture is filtered (idc==1 everywhere)      */
    uint32_t rec_off;         /* FjMbRec[n_mbs]                                               */
    uint32_t mv_off;          /* INPUT of fj_finalize(), host only: int16 mv[n_mbs][16][2] (x,y) quarter-pel, raster 4x4 order — the dense
                                 array the parser writes.  The parser keeps it behind everything else in the job buffer, past
                                 total_bytes: it is never transferred.  Finished jobs carry vectors as FjMbRec.mv / mvx_off. */
    uint32_t lvl_off;         /* uint32 lvl_start[n_intra_levels+1]  (indices into intra_idx) */
    uint32_t idx_off;         /* uint16 intra_idx[n_intra]  MB addresses sorted by level      */
    uint32_t coef_off;        /* int16 coef[n_coef_blocks][16]                                */
    uint32_t n_intra;
    uint32_t n_intra_levels;
    uint32_t n_coef_blocks;
    uint32_t n_inter;         /* statistics for the byte accounting of the bench              */
    uint32_t pic_seq;         /* running picture number of the stream                         */
    uint32_t copy_off;        /* FjCopy[n_copy]: inter MBs that are pure whole-sample copies   */
    uint32_t n_copy;
    uint32_t gen_off;         /* FjGen[n_gen]: all other inter MBs                             */
    uint32_t n_gen;
    uint32_t dbk_off;         /* uint16 dbk_idx[n_dbk]: MBs whose boundary strengths are not trivially all zero */
    uint32_t n_dbk;
    uint32_t n_gen_uniform;   /* the first n_gen_uniform entries of the general-inter list have ONE motion vector and
                                 reference for the whole macroblock (FjGen.uniform == 1), the rest are partitioned */
    uint32_t ghost;           /* 1: pre-pass of the picture that follows in the same slot (pixels of a rolled-back slice that a
                                 FJ_MB_STALE macroblock of that picture shows): reconstruction only, not a picture of its own */
    uint32_t dbk_only;        /* 1: the pixels of this picture were reconstructed by the job before it (a ghost job with the records
                                 they were made from); this job only deblocks, with the records the reference ends up with
                                 (macroblocks re-decoded by a redundant slice: it keeps the first pixels, the last metadata) */
    uint32_t intra_down_deps; /* 1: the intra schedule holds concealed macroblocks, which may wait for the macroblock BELOW them
                                 (FJ_NEED_D) and read all four neighbours: the picture's intra reconstruction must not be split
                                 into row bands (k_frame_intra) */
    uint32_t n_gen_quad;      /* of the partitioned entries behind the first n_gen_uniform, the first n_gen_quad have one motion vector
                                 per 8x8 quadrant (FjGen.uniform == 2: 16x8, 8x16, 8x8 partitions), the rest finer partitions */
    uint32_t mvx_off;         /* int16 mvx[n_mvx][16][2]: the vectors of the inter macroblocks that have more than one (raster 4x4
                                 order), in macroblock address order; FjMbRec.mvx is the entry's index.  64 bytes for one macroblock
                                 in six of the bundled 1080p stream instead of 64 bytes for every macroblock */
    uint32_t n_mvx;
    uint32_t reserved[4];
} FjHeader;                   /* 128 bytes */

/* An inter macroblock with no coefficients whose 16 motion vectors are equal and whole-sample for
 * luma AND chroma (mv multiple of 8 quarter-samples; 66 % of the inter MBs of the bundled 1080p
 * stream, almost all of them P_Skip with mv 0): reconstruction is a 384-byte copy. */
#ifndef FJ_COPY_RUN
#define FJ_COPY_RUN 8         /* longest run of a copy-list entry.  Frames are stored as macroblock tiles (384 contiguous bytes
                                 per macroblock, address order), so a run with zero displacement is ONE contiguous block of
                                 count x 384 bytes in both frames — also across the end of a macroblock row */
#endif
typedef struct FjCopy {
    uint16_t mb;              /* address of the first macroblock of the run                   */
    uint8_t  slot;            /* reference DPB slot                                           */
    uint8_t  count;           /* 1..FJ_COPY_RUN MBs with consecutive addresses, the same slot and mv (a displaced run stays inside one row) */
    int16_t  dx, dy;          /* displacement in luma samples (even)                          */
} FjCopy;                     /* 8 bytes */

/* Every other inter macroblock.  The entry repeats what the reconstruction needs first (reference,
 * motion vector when all 16 agree, where the coefficients are) so that the reference-window loads
 * can be issued one dependent load after the launch descriptor instead of three. */
typedef struct FjGen {
    uint16_t mb;
    uint8_t  uniform;         /* 1: 16 equal motion vectors and one reference slot; 2: one motion vector per 8x8
                                 quadrant (references in FjMbRec.ref_slot); 0: anything else                   */
    uint8_t  slot;            /* reference slot when uniform                                  */
    int16_t  mvx, mvy;        /* the motion vector when uniform (quarter samples)             */
    uint32_t coef_idx;        /* bits 0..19 = FjMbRec.coef_idx (at most 27 blocks x 36864 macroblocks), bits 20..25 = FjMbRec.qp_y,
                                 bits 26..31 = FjMbRec.qp_c: a one-vector macroblock is reconstructed from its list entry alone — the
                                 record, a second dependent scalar load in front of every macroblock of k_recon_inter, is not fetched */
    uint32_t coded;           /* = FjMbRec.coded                                              */
} FjGen;                      /* 16 bytes */
#define FJ_GEN_COEF(idx, qp_y, qp_c) (((uint32_t)(idx) & 0xFFFFFu) | ((uint32_t)((qp_y) & 63u) << 20) | ((uint32_t)((qp_c) & 63u) << 26))
#define FJ_GEN_COEF_IDX(w) ((w) & 0xFFFFFu)
#define FJ_GEN_QP_Y(w) (((w) >> 20) & 63u)
#define FJ_GEN_QP_C(w) ((w) >> 26)

typedef struct FjMbRec {
    uint8_t  kind;
    uint8_t  qp_y;
    uint8_t  qp_c;            /* QPc of THIS macroblock (chroma dequant)                      */
    uint8_t  avail;
    uint8_t  pred;            /* bits0-1 Intra16x16 mode (0 V,1 H,2 DC,3 plane);
                                 bits2-3 intra chroma mode (0 DC,1 H,2 V,3 plane);
                                 inter: bits4-5 FJ_PARTS_*;  bit7 FJ_PRED_PHASE2               */
    uint8_t  dbk;
    int8_t   alpha_off;       /* FilterOffsetA = 2*slice_alpha_c0_offset_div2                 */
    int8_t   beta_off;
    uint32_t coded;           /* bit z (0..15): luma 4x4 block z (H.264 block order) has coefficients;
                                 16..19 Cb AC, 20..23 Cr AC (raster 2x2); 24 luma DC; 25 chroma DC */
    uint32_t coef_idx;        /* first coefficient block of this MB (16 x int16 each)         */
    uint8_t  ref_slot[4];     /* inter: reference DPB slot per 8x8 quadrant (raster); intra schedule: [0] = FJ_NEED_* */
    int8_t   cqp_off;         /* chroma_qp_index_offset of the MB's PPS (deblock QPc)         */
    uint8_t  dbk_trivial;     /* 1: every boundary strength of this MB is zero by construction (host-proved) */
    uint16_t intra_level;     /* dependency level among intra MBs of the picture              */
    union {
        uint8_t  i4mode[8];   /* Intra4x4: 16 nibbles, Intra4x4PredMode of block z (H.264 block order) */
        struct {              /* inter macroblocks, written by fj_finalize():                          */
            int16_t  mv[2];   /*   the vector of block 0 = THE vector when FJ_PRED_UNIFORM_MV          */
            uint32_t mvx;     /*   else: index of the macroblock's 16 vectors in the section at mvx_off */
        };
    };
} FjMbRec;                    /* 32 bytes */

/* Order of a macroblock's coefficient blocks starting at coef_idx:
 *   [luma DC 4x4 (raster c[i][j])]            if FJ_CODED_LUMA_DC
 *   luma blocks z with bit z set, ascending z (H.264 block order); each block is RASTER 4x4
 *        (row-major), un-dequantised levels, position 0 forced to 0 for Intra16x16 AC blocks
 *   [chroma DC: Cb c[0..3] raster 2x2, Cr c[4..7], 8 pad]   if FJ_CODED_CHROMA_DC
 *   chroma AC blocks 16..23 with bit set, ascending (position 0 = 0)
 * I_PCM: 12 blocks = 384 raw samples as bytes: Y[256] raster, Cb[64], Cr[64].
 */

static inline uint32_t fj_align32(uint32_t x) { return (x + 31u) & ~31u; }

/* Byte size of one decoded frame (I420, uncropped, 16-aligned) */
static inline uint32_t fj_frame_bytes(uint32_t width_mbs, uint32_t height_mbs)
{
    return width_mbs * height_mbs * 384u;
}

#endif
