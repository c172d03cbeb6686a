/*
 * hd_mb.c — slice_data() and macroblock_layer() parsing (H.264 7.3.4, 7.3.5), motion-vector
 * prediction (8.4.1), Intra4x4PredMode derivation (8.3.1.1), CAVLC nC derivation (9.2.1) and the
 * packing of every macroblock into its FjMbRec + motion vectors + coefficient blocks.
 *
 * This is the host half of seam A in SURVEY.md §1: everything h264bsdDecodeMacroblockLayer
 * (reference src/h264bsd_macroblock_layer.c:134) and the metadata half of h264bsdDecodeMacroblock
 * (:965-1131: QP carry, totalCoeff bookkeeping), h264bsdInterPrediction's MvPrediction*
 * (src/h264bsd_inter_prediction.c:494-1026) and DetermineIntra4x4PredMode
 * (src/h264bsd_intra_prediction.c:1886) decide — without producing a single pixel.
 *
 * Reference-specific behaviour kept on purpose (differs from a literal reading of the standard):
 *  - an I_PCM macroblock stores QP 0 for deblocking but does NOT reset the running slice QP
 *    (macroblock_layer.c:989-1022 never writes *qpY);
 *  - motion vectors outside [-8192,8191] x [-2048,2047] quarter-pels are a decode error
 *    (inter_prediction.c:538-544).
 */
#include <stddef.h>
#include <string.h>
#include "hostdec.h"

/* H.264 4x4 block order (z-order) <-> position inside the macroblock */
static const uint8_t Z_X[16] = { 0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3 };
static const uint8_t Z_Y[16] = { 0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3 };
static inline int z_of(int x, int y) { return ((y >> 1) << 3) | ((x >> 1) << 2) | ((y & 1) << 1) | (x & 1); }

/* Table 9-4: coded_block_pattern from codeNum, for Intra4x4 and Inter macroblocks */
static const uint8_t cbp_intra[48] = {
    47, 31, 15, 0, 23, 27, 29, 30, 7, 11, 13, 14, 39, 43, 45, 46, 16, 3, 5, 10, 12, 19, 21, 26,
    28, 35, 37, 42, 44, 1, 2, 4, 8, 17, 18, 20, 24, 6, 9, 22, 25, 32, 33, 34, 36, 40, 38, 41 };
static const uint8_t cbp_inter[48] = {
    0, 16, 1, 2, 4, 8, 32, 3, 5, 10, 12, 15, 47, 7, 11, 13, 14, 6, 9, 31, 35, 37, 42, 44,
    33, 34, 36, 40, 39, 43, 45, 46, 17, 18, 20, 24, 19, 21, 26, 28, 23, 27, 29, 30, 22, 25, 38, 41 };
/* Table 8-15: QPc as a function of qPi */
static const uint8_t qpc_table[52] = {
    0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25,
    26, 27, 28, 29, 29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39 };

typedef struct MbCtx {
    HostDec *d;
    BitReader *br;
    const SliceHdr *sh;
    const Pps *pps;
    uint32_t addr, mbx, mby;
    MbInfo *cur, *A, *B, *C, *D;   /* NULL when outside the picture or another slice */
    uint16_t done;                 /* raster bit per 4x4 block of cur whose mv/ref is final */
    uint8_t  ok_quads;             /* P_8x8: quadrants whose reference was written before the first such error — the reference
                                      writes refPic / refAddr of a sub-macroblock BEFORE it checks them
                                      (inter_prediction.c:805-811), so the failing quadrant carries "no picture" */
    uint16_t ok_blocks;            /* ... and was final before the first reconstruction-time error (p2err): the reference
                                      writes motion vectors and references partition by partition and stops at the first
                                      one that fails (inter_prediction.c:520-565 and the partitioned variants) */
    uint32_t coef_start;           /* first coefficient block of this macroblock */
    uint8_t  have_old;             /* old_* below are filled in (parse_inter) */
    int p2err;                     /* an error the reference only finds when it RECONSTRUCTS the macroblock
                                      (h264bsdDecodeMacroblock: missing reference picture, motion vector range, intra
                                      mode without its neighbours) — i.e. after the whole macroblock_layer() has been
                                      parsed and after mb.decoded was incremented (macroblock_layer.c:988).  Recorded
                                      here, acted upon once parsing of the macroblock has succeeded. */
    /* What the macroblock carried before this decode (the reference's mbStorage_t keeps mv / refPic / refAddr across
     * decodes AND across pictures, only MvPrediction writes them, inter_prediction.c:520-821): the parts a failing
     * decode did not reach get it back (restore_unwritten). */
    int8_t   old_ref_idx[4];
    uint8_t  old_ref_slot[4];
    int16_t  old_mv[16][2];
} MbCtx;

static inline MbInfo *usable(HostDec *d, uint32_t idx, uint32_t slice_id) { return d->mb_slice_id[idx] == slice_id ? &d->mb[idx] : NULL; }
static inline int is_inter(const MbInfo *m) { return m->kind == FJ_MB_INTER; }

/* ---------------------------------------------------------------- nC, 9.2.1 */
static int nc_luma(const MbCtx *c, int z)
{
    const int x = Z_X[z], y = Z_Y[z];
    int na = -1, nb = -1;
    if (x > 0) na = c->cur->tc[z_of(x - 1, y)]; else if (c->A) na = c->A->tc[z_of(3, y)];
    if (y > 0) nb = c->cur->tc[z_of(x, y - 1)]; else if (c->B) nb = c->B->tc[z_of(x, 3)];
    if (na >= 0 && nb >= 0) return (na + nb + 1) >> 1;
    return na >= 0 ? na : nb >= 0 ? nb : 0;
}
static int nc_chroma(const MbCtx *c, int plane, int k) /* k = 2*cy+cx */
{
    const int base = 16 + 4 * plane, x = k & 1, y = k >> 1;
    int na = -1, nb = -1;
    if (x > 0) na = c->cur->tc[base + k - 1]; else if (c->A) na = c->A->tc[base + 2 * y + 1];
    if (y > 0) nb = c->cur->tc[base + k - 2]; else if (c->B) nb = c->B->tc[base + 2 + x];
    if (na >= 0 && nb >= 0) return (na + nb + 1) >> 1;
    return na >= 0 ? na : nb >= 0 ? nb : 0;
}

/* ---------------------------------------------------------------- motion vector prediction */
typedef struct Nb { int avail, ref; int16_t mx, my; } Nb;
/* (always inlined: returned by value from a real call, the 12-byte struct goes through the stack as two 4-byte stores and one
 * 8-byte load, which the store buffer cannot forward — a dozen cycles per neighbour, 7 % of the parser's time) */
#define HD_INLINE static inline __attribute__((always_inline))

HD_INLINE Nb nb_from(const MbInfo *m, int x, int y)
{
    Nb n = { 0, -1, 0, 0 };
    if (!m) return n;
    n.avail = 1;
    if (is_inter(m)) {
        const int z = z_of(x, y);
        n.ref = m->ref_idx[z >> 2];
        n.mx = m->mv[z][0];
        n.my = m->mv[z][1];
    }
    return n;
}
/* neighbouring 4x4 block at (x,y) relative to the current MB, 6.4.11.7 + decoding-order rule */
HD_INLINE Nb nb_at(const MbCtx *c, int x, int y)
{
    if (y < 0) {
        if (x < 0) return nb_from(c->D, 3, 3);
        if (x < 4) return nb_from(c->B, x, 3);
        return nb_from(c->C, 0, 3);
    }
    if (x < 0) return nb_from(c->A, 3, y);
    Nb n = { 0, -1, 0, 0 };
    if (x >= 4 || !(c->done & (1u << (4 * y + x)))) return n;   /* right MB / later partition */
    return nb_from(c->cur, x, y);
}
static inline int16_t median3(int a, int b, int cc)
{
    int mn = a < b ? a : b, mx = a < b ? b : a;
    return (int16_t)(cc < mn ? mn : cc > mx ? mx : cc);
}
/* shape: 0 generic, 1 = upper 16x8, 2 = lower 16x8, 3 = left 8x16, 4 = right 8x16 (8.4.1.3) */
static void predict_mv(const MbCtx *c, int x, int y, int w, int ref, int shape, int16_t out[2])
{
    Nb a = nb_at(c, x - 1, y), b = nb_at(c, x, y - 1), cc = nb_at(c, x + w, y - 1);
    if (!cc.avail) cc = nb_at(c, x - 1, y - 1);
    if (shape == 1 && b.ref == ref) { out[0] = b.mx; out[1] = b.my; return; }
    if (shape == 2 && a.ref == ref) { out[0] = a.mx; out[1] = a.my; return; }
    if (shape == 3 && a.ref == ref) { out[0] = a.mx; out[1] = a.my; return; }
    if (shape == 4 && cc.ref == ref) { out[0] = cc.mx; out[1] = cc.my; return; }
    if (!b.avail && !cc.avail && a.avail) { out[0] = a.mx; out[1] = a.my; return; }
    const int ma = a.ref == ref, mb = b.ref == ref, mc = cc.ref == ref;
    if (ma + mb + mc == 1) {
        const Nb *s = ma ? &a : mb ? &b : &cc;
        out[0] = s->mx; out[1] = s->my;
        return;
    }
    out[0] = median3(a.mx, b.mx, cc.mx);
    out[1] = median3(a.my, b.my, cc.my);
}

static int set_partition(MbCtx *c, int x, int y, int w, int h, int ref, const int16_t mv[2])
{
    /* horizontal [-2048, 2047.75], vertical [-512, 511.75] luma samples */
    if ((uint32_t)(mv[0] + 8192) >= 16384u || (uint32_t)(mv[1] + 2048) >= 4096u) return -1;
    if (w == 4 && h == 4) {                                  /* the whole macroblock (P_Skip, 16x16): most inter macroblocks */
        uint32_t one;
        memcpy(&one, mv, 4);
        const uint64_t two = (uint64_t)one << 32 | one;
        uint8_t *dst = (uint8_t *)c->cur->mv;               /* 4-byte aligned only */
        for (int i = 0; i < 8; i++) memcpy(dst + 8 * i, &two, 8);
        c->done = 0xFFFF;
        if (!c->p2err) c->ok_blocks = 0xFFFF;
        return 0;
    }
    for (int yy = y; yy < y + h; yy++)
        for (int xx = x; xx < x + w; xx++) {
            const int z = z_of(xx, yy);
            c->cur->mv[z][0] = mv[0];
            c->cur->mv[z][1] = mv[1];
            c->done |= (uint16_t)(1u << (4 * yy + xx));
            if (!c->p2err) c->ok_blocks |= (uint16_t)(1u << (4 * yy + xx));
        }
    (void)ref;
    return 0;
}

/* the same reference for all four quadrants (P_Skip, 16x16): one DPB lookup */
static int resolve_ref_all(MbCtx *c, int ref_idx)
{
    if (ref_idx < 0) return -1;
    const int slot = hd_dpb_ref_slot(&c->d->dpb, (uint32_t)ref_idx);
    if (slot < 0) return -1;
    memset(c->cur->ref_idx, ref_idx, 4);
    memset(c->cur->ref_slot, slot, 4);
    return 0;
}

static int resolve_ref(MbCtx *c, int quadrant, int ref_idx)
{
    if (ref_idx < 0) return -1;
    const int slot = hd_dpb_ref_slot(&c->d->dpb, (uint32_t)ref_idx);
    if (slot < 0) return -1;
    c->cur->ref_idx[quadrant] = (int8_t)ref_idx;
    c->cur->ref_slot[quadrant] = (uint8_t)slot;
    return 0;
}

/* error exits below report where they happened when HD_TRACE is set in the environment (debugging aid) */
#include <stdio.h>
#include <stdlib.h>
#define FAIL do { if (hd_trace) fprintf(stderr, "TRACE hd_mb fail at line %d\n", __LINE__); return -1; } while (0)
#define P2ERR(c_) do { if (hd_trace) fprintf(stderr, "TRACE hd_mb p2err at line %d\n", __LINE__); (c_)->p2err = 1; } while (0)

static uint32_t read_te(BitReader *br, uint32_t n_active)
{
    if (n_active > 2) return br_ue(br);
    return br_get1(br) ^ 1u;
}

/* P macroblock prediction syntax + mv reconstruction. p_type: 0 16x16, 1 16x8, 2 8x16, 3 8x8, 4 8x8ref0 */
static int parse_inter(MbCtx *c, int p_type)
{
    BitReader *br = c->br;
    const uint32_t n_active = c->sh->num_ref_idx_active;
    int16_t mvd[16][2], mvp[2], mv[2];
    c->done = 0;
    c->have_old = 1;
    memcpy(c->old_ref_idx, c->cur->ref_idx, 4);
    memcpy(c->old_ref_slot, c->cur->ref_slot, 4);
    memcpy(c->old_mv, c->cur->mv, 64);

    if (p_type <= 2) {
        const int nparts = p_type == 0 ? 1 : 2;
        uint32_t ref[2] = { 0, 0 };
        if (n_active > 1)
            for (int i = 0; i < nparts; i++) {
                ref[i] = read_te(br, n_active);
                if (br_overrun(br) || ref[i] >= n_active) FAIL;
            }
        for (int i = 0; i < nparts; i++) { mvd[i][0] = (int16_t)br_se(br); mvd[i][1] = (int16_t)br_se(br); }
        if (br_overrun(br)) FAIL;
        for (int i = 0; i < nparts; i++) {
            int x = 0, y = 0, w = 4, h = 4, shape = 0;
            if (p_type == 1) { y = 2 * i; h = 2; shape = 1 + i; }
            if (p_type == 2) { x = 2 * i; w = 2; shape = 3 + i; }
            /* reference of the quadrants covered by this partition must be known before prediction
             * of the next partition looks at it */
            if (p_type == 0) { if (resolve_ref_all(c, (int)ref[0])) P2ERR(c); }
            else if (p_type == 1) { if (resolve_ref(c, 2 * i, (int)ref[i]) || resolve_ref(c, 2 * i + 1, (int)ref[i])) P2ERR(c); }
            else { if (resolve_ref(c, i, (int)ref[i]) || resolve_ref(c, i + 2, (int)ref[i])) P2ERR(c); }
            predict_mv(c, x, y, w, (int)ref[i], shape, mvp);
            mv[0] = (int16_t)(mvp[0] + mvd[i][0]);
            mv[1] = (int16_t)(mvp[1] + mvd[i][1]);
            if (set_partition(c, x, y, w, h, (int)ref[i], mv)) P2ERR(c);
        }
        return 0;
    }

    /* P_8x8 / P_8x8ref0 */
    uint32_t sub[4], ref[4] = { 0, 0, 0, 0 };
    for (int i = 0; i < 4; i++) {
        sub[i] = br_ue(br);
        if (br_overrun(br) || sub[i] > 3) FAIL;
    }
    if (n_active > 1 && p_type != 4)
        for (int i = 0; i < 4; i++) {
            ref[i] = read_te(br, n_active);
            if (br_overrun(br) || ref[i] >= n_active) FAIL;
        }
    static const uint8_t nsub[4] = { 1, 2, 2, 4 };
    int k = 0;
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < nsub[sub[i]]; j++, k++) { mvd[k][0] = (int16_t)br_se(br); mvd[k][1] = (int16_t)br_se(br); }
    if (br_overrun(br)) FAIL;
    k = 0;
    for (int i = 0; i < 4; i++) {
        if (!c->p2err) c->ok_quads |= (uint8_t)(1u << i);
        if (resolve_ref(c, i, (int)ref[i])) { if (!c->p2err) c->cur->ref_slot[i] = 0xFF; P2ERR(c); }   /* 0xFF: "no picture", unequal to every slot */
        const int bx = (i & 1) * 2, by = (i >> 1) * 2;
        const int sw = (sub[i] == 0 || sub[i] == 1) ? 2 : 1;   /* 8x8, 8x4 are 2 blocks wide */
        const int shh = (sub[i] == 0 || sub[i] == 2) ? 2 : 1;  /* 8x8, 4x8 are 2 blocks high */
        for (int j = 0; j < nsub[sub[i]]; j++, k++) {
            int x = bx, y = by;
            if (sub[i] == 1) y += j;
            else if (sub[i] == 2) x += j;
            else if (sub[i] == 3) { x += j & 1; y += j >> 1; }
            predict_mv(c, x, y, sw, (int)ref[i], 0, mvp);
            mv[0] = (int16_t)(mvp[0] + mvd[k][0]);
            mv[1] = (int16_t)(mvp[1] + mvd[k][1]);
            if (set_partition(c, x, y, sw, shh, (int)ref[i], mv)) P2ERR(c);
        }
    }
    return 0;
}

static int infer_skip(MbCtx *c)
{
    int16_t mv[2] = { 0, 0 };
    c->done = 0;
    if (resolve_ref_all(c, 0)) { P2ERR(c); return 0; }     /* nothing written (inter_prediction.c:547-551) */
    Nb a = nb_at(c, -1, 0), b = nb_at(c, 0, -1);
    if (a.avail && b.avail && !(a.ref == 0 && a.mx == 0 && a.my == 0) && !(b.ref == 0 && b.mx == 0 && b.my == 0))
        predict_mv(c, 0, 0, 4, 0, 0, mv);
    if (set_partition(c, 0, 0, 4, 4, 0, mv)) P2ERR(c);
    return 0;
}

/* A decode that failed while the reference reconstructed it (p2err): motion vectors and references the reference had
 * not written by then are what the macroblock carried before — possibly from an earlier picture. */
static void restore_unwritten(MbCtx *c)
{
    if (!c->have_old) return;
    MbInfo *m = c->cur;
    for (int q = 0; q < 4; q++) {
        const uint16_t quad = (uint16_t)(0x33u << (2 * (q & 1) + 8 * (q >> 1)));
        if ((c->ok_blocks & quad) != quad && !(c->ok_quads & (1u << q))) { m->ref_idx[q] = c->old_ref_idx[q]; m->ref_slot[q] = c->old_ref_slot[q]; }
    }
    for (int z = 0; z < 16; z++)
        if (!(c->ok_blocks & (1u << (4 * Z_Y[z] + Z_X[z])))) { m->mv[z][0] = c->old_mv[z][0]; m->mv[z][1] = c->old_mv[z][1]; }
}

/* ---------------------------------------------------------------- Intra4x4PredMode, 8.3.1.1 */
static int i4_neighbour_mode(const MbCtx *c, const MbInfo *m, int z)
{
    /* -1: "not available" (forces DC prediction of the mode), else the neighbour's mode, 2 if not I4x4 */
    if (!m) return -1;
    if (c->pps->constrained_intra_pred && is_inter(m)) return -1;
    return m->kind == FJ_MB_I4x4 ? m->i4mode[z] : 2;
}
static int parse_i4_modes(MbCtx *c)
{
    BitReader *br = c->br;
    for (int z = 0; z < 16; z++) {
        const int x = Z_X[z], y = Z_Y[z];
        int ma = x > 0 ? c->cur->i4mode[z_of(x - 1, y)] : i4_neighbour_mode(c, c->A, z_of(3, y));
        int mb = y > 0 ? c->cur->i4mode[z_of(x, y - 1)] : i4_neighbour_mode(c, c->B, z_of(x, 3));
        int pred = (ma < 0 || mb < 0) ? 2 : (ma < mb ? ma : mb);
        if (!br_get1(br)) {
            int rem = (int)br_get(br, 3);
            pred = rem < pred ? rem : rem + 1;
        }
        c->cur->i4mode[z] = (int8_t)pred;
    }
    return br_overrun(br) ? -1 : 0;
}

/* ---------------------------------------------------------------- residual, 7.3.5.3 */
/* The coefficient section holds job_capacity()'s 27 blocks per macroblock (+2).  A macroblock owns at most 26 live
 * blocks and dead ones are reclaimed (decode_mb on failure, mark_slice_corrupted), so a valid stream never gets
 * near the end; the check keeps a hostile one (slices repeated over the same macroblocks) inside the buffer. */
static inline int16_t *next_block(HostDec *d, int16_t *coefs)
{
    if (d->coef_blocks >= d->coef_cap_blocks) return NULL;
    int16_t *p = coefs + 16u * d->coef_blocks;
    memset(p, 0, 32);
    return p;
}

static int parse_residual(MbCtx *c, int is_i16, uint32_t cbp, int16_t *coefs, uint32_t *coded_out, uint32_t sums[4])
{
    HostDec *d = c->d;
    BitReader *br = c->br;
    MbInfo *m = c->cur;
    uint32_t coded = 0;
    int n;
    sums[0] = sums[1] = sums[2] = sums[3] = 0;       /* level magnitudes, the LARGEST block's of each kind: luma blocks, chroma DC (per plane), chroma AC (hd_residual_bound_ok:
                                                        the bound is a property of one block — the macroblock's total cleared 130 macroblocks per 1080p picture fewer) */

    if (is_i16) {
        int16_t *blk = next_block(d, coefs);
        if (!blk) FAIL;
        sums[3] = 0;
        n = hd_cavlc_block_sum(br, nc_luma(c, 0), 16, blk, NULL, &sums[3]);      /* sums[3]: the Intra16x16 DC levels (every DC is a signed sum of them) */
        if (n < 0) FAIL;
        if (n) { coded |= FJ_CODED_LUMA_DC; d->coef_blocks++; }
    }
    for (int z = 0; z < 16; z++) {
        if (!(cbp & (1u << (z >> 2)))) { m->tc[z] = 0; continue; }
        int16_t *blk = next_block(d, coefs);
        if (!blk) FAIL;
        uint32_t bsum = 0;
        n = hd_cavlc_block_sum(br, nc_luma(c, z), is_i16 ? 15 : 16, blk, NULL, &bsum);
        if (n < 0) FAIL;
        if (bsum > sums[0]) sums[0] = bsum;
        m->tc[z] = (uint8_t)n;
        if (n) { coded |= 1u << z; d->coef_blocks++; }
    }
    memset(m->tc + 16, 0, 8);
    if (cbp & 0x30) {
        int16_t *blk = next_block(d, coefs);
        if (!blk) FAIL;
        uint32_t s0 = 0, s1 = 0;
        int n0 = hd_cavlc_block_sum(br, -1, 4, blk, NULL, &s0);
        if (n0 < 0) FAIL;
        int n1 = hd_cavlc_block_sum(br, -1, 4, blk + 4, NULL, &s1);
        if (n1 < 0) FAIL;
        sums[1] = s0 > s1 ? s0 : s1;
        if (n0 || n1) { coded |= FJ_CODED_CHROMA_DC; d->coef_blocks++; }
    }
    if (cbp & 0x20) {
        for (int k = 0; k < 8; k++) {
            int16_t *blk = next_block(d, coefs);
            int spill;
            if (!blk) FAIL;
            uint32_t bsum = 0;
            n = hd_cavlc_block_sum(br, nc_chroma(c, k >> 2, k & 3), 15, blk, &spill, &bsum);
            if (n < 0) FAIL;
            if (bsum > sums[2]) sums[2] = bsum;
            m->tc[16 + k] = (uint8_t)n;
            if (n) { coded |= 1u << (16 + k); d->coef_blocks++; }
            if (k == 7 && spill && is_i16) {
                /* Damaged stream: the last Cr block put a level one element past its end, which in the reference's
                 * residual_t is level[24][0] — the first Intra16x16 DC coefficient, parsed earlier
                 * (macroblock_layer.c:720-792; hd_cavlc_block).  With a coded DC block the level replaces its first
                 * coefficient before the DC transform; without one (totalCoeff[24] == 0) no transform runs and the
                 * level IS the DC of luma block 0 (macroblock_layer.c:1366-1374). */
                int16_t *first = coefs + 16u * c->coef_start;
                if (coded & FJ_CODED_LUMA_DC) first[0] = (int16_t)spill;
                else {
                    if (d->coef_blocks >= d->coef_cap_blocks) FAIL;
                    memmove(first + 16, first, (size_t)(d->coef_blocks - c->coef_start) * 32u);
                    memset(first, 0, 32);
                    first[0] = (int16_t)spill;
                    d->coef_blocks++;
                    coded |= FJ_CODED_LUMA_DC | FJ_CODED_LUMA_DC_RAW;
                }
            }
        }
    }
    *coded_out = coded;
    return 0;
}

/* Does every prediction mode of this intra macroblock have the neighbour samples it reads?  The reference
 * checks this while predicting (src/h264bsd_intra_prediction.c:655-680 Intra16x16, :770-830 Intra4x4 per block,
 * :880-900 chroma) and fails the macroblock otherwise. */
static int intra_modes_have_neighbours(const FjMbRec *rec, const MbInfo *m, int is_i16, int chroma_mode)
{
    const int A = (rec->avail & FJ_AVAIL_A) != 0, B = (rec->avail & FJ_AVAIL_B) != 0, D = (rec->avail & FJ_AVAIL_D) != 0;
    if (is_i16) {
        const int mode = rec->pred & 3;
        if ((mode == 0 && !B) || (mode == 1 && !A) || (mode == 3 && !(A && B && D))) return 0;
    } else {
        for (int z = 0; z < 16; z++) {
            const int bx = Z_X[z], by = Z_Y[z], mode = m->i4mode[z];
            const int a = bx > 0 || A, b = by > 0 || B;
            const int dd = (bx > 0 && by > 0) ? 1 : bx > 0 ? B : by > 0 ? A : D;
            switch (mode) {
            case 0: case 3: case 7: if (!b) return 0; break;
            case 1: case 8: if (!a) return 0; break;
            case 4: case 5: case 6: if (!(a && b && dd)) return 0; break;
            default: break;
            }
        }
    }
    if ((chroma_mode == 1 && !A) || (chroma_mode == 2 && !B) || (chroma_mode == 3 && !(A && B && D))) return 0;
    return 1;
}

/* ---------------------------------------------------------------- one macroblock */
static int decode_mb_body(HostDec *d, BitReader *br, const SliceHdr *sh, const Pps *pps, uint32_t addr,
                          int skipped, int *qp)
{
    FjHeader *hdr = (FjHeader *)d->job;
    FjMbRec *recs = (FjMbRec *)(d->job + hdr->rec_off);
    int16_t (*mvs)[16][2] = (int16_t (*)[16][2])(d->job + hdr->mv_off);
    int16_t *coefs = (int16_t *)(d->job + hdr->coef_off);

    MbCtx c;
    memset(&c, 0, offsetof(MbCtx, old_ref_idx));    /* (old_* are written before they are read: have_old) */
    c.d = d; c.br = br; c.sh = sh; c.pps = pps;
    c.coef_start = d->coef_blocks;
    c.addr = addr; c.mby = hd_mb_row(d, addr); c.mbx = addr - c.mby * d->width_mbs;
    MbInfo *m = c.cur = &d->mb[addr];
    const uint32_t sid = d->slice_id;
    c.A = c.mbx ? usable(d, addr - 1, sid) : NULL;
    c.B = c.mby ? usable(d, addr - d->width_mbs, sid) : NULL;
    c.C = (c.mby && c.mbx + 1 < d->width_mbs) ? usable(d, addr - d->width_mbs + 1, sid) : NULL;
    c.D = (c.mby && c.mbx) ? usable(d, addr - d->width_mbs - 1, sid) : NULL;

    const int first_decode = d->mb_decoded[addr] == 0;
    const uint32_t coef_start = d->coef_blocks;
    uint32_t level_sums[4] = { 0, 0, 0, 0 };
    FjMbRec rec;
    memset(&rec, 0, sizeof(rec));
    rec.coef_idx = coef_start;
    rec.cqp_off = (int8_t)pps->chroma_qp_index_offset;
    rec.alpha_off = (int8_t)sh->alpha_off;
    rec.beta_off = (int8_t)sh->beta_off;
    m->dbk_idc = (uint8_t)sh->disable_deblocking_filter_idc;
    if (sh->disable_deblocking_filter_idc != 1) {
        rec.dbk = FJ_DBK_INNER;
        if (c.mbx && (sh->disable_deblocking_filter_idc != 2 || c.A)) rec.dbk |= FJ_DBK_LEFT;
        if (c.mby && (sh->disable_deblocking_filter_idc != 2 || c.B)) rec.dbk |= FJ_DBK_TOP;
    }

    if (skipped) {
        m->mb_type = 0;
        m->kind = FJ_MB_INTER;
        if (infer_skip(&c)) FAIL;
        memset(m->tc, 0, sizeof(m->tc));
        m->qp = (uint8_t)*qp;
        rec.kind = FJ_MB_INTER;
        rec.pred = FJ_PARTS_16x16 << FJ_PRED_PARTS_SHIFT;
    } else {
        uint32_t t = br_ue(br);
        if (br_overrun(br)) FAIL;
        int itype = -1, ptype = -1;
        if (sh->is_p) { if (t < 5) ptype = (int)t; else itype = (int)t - 5; }
        else itype = (int)t;
        if (itype > 25) FAIL;
        m->mb_type = (uint8_t)(ptype >= 0 ? ptype + 1 : itype + 6);

        if (itype == 25) {                                   /* I_PCM */
            while (br->pos & 7) if (br_get1(br)) FAIL;
            if (d->coef_blocks + 12u > d->coef_cap_blocks) FAIL;
            uint8_t *dst = (uint8_t *)(coefs + 16u * d->coef_blocks);
            for (int i = 0; i < 384; i++) dst[i] = (uint8_t)br_get(br, 8);
            if (br_overrun(br)) FAIL;
            d->coef_blocks += 12;
            m->kind = FJ_MB_IPCM;
            m->qp = 0;
            memset(m->tc, 16, sizeof(m->tc));
            rec.kind = FJ_MB_IPCM;
        } else {
            uint32_t cbp;
            int is_i16 = 0;
            if (ptype >= 0) {
                m->kind = FJ_MB_INTER;
                if (parse_inter(&c, ptype)) FAIL;
                rec.kind = FJ_MB_INTER;
                rec.pred = (uint8_t)((ptype == 0 ? FJ_PARTS_16x16 : ptype == 1 ? FJ_PARTS_16x8 : ptype == 2 ? FJ_PARTS_8x16 : FJ_PARTS_8x8) << FJ_PRED_PARTS_SHIFT);
            } else {
                /* intra: availability of the four neighbours for prediction */
                const int cip = pps->constrained_intra_pred;
                if (c.A && !(cip && is_inter(c.A))) rec.avail |= FJ_AVAIL_A;
                if (c.B && !(cip && is_inter(c.B))) rec.avail |= FJ_AVAIL_B;
                if (c.C && !(cip && is_inter(c.C))) rec.avail |= FJ_AVAIL_C;
                if (c.D && !(cip && is_inter(c.D))) rec.avail |= FJ_AVAIL_D;
                if (itype == 0) {
                    /* the mode derivation must see the neighbours' kinds but the current MB's own
                     * previous kind must not leak in: set kind after the modes are known */
                    if (parse_i4_modes(&c)) FAIL;
                    m->kind = FJ_MB_I4x4;
                    rec.kind = FJ_MB_I4x4;
                    for (int z = 0; z < 16; z++) rec.i4mode[z >> 1] |= (uint8_t)(m->i4mode[z] << ((z & 1) * 4));
                } else {
                    is_i16 = 1;
                    m->kind = FJ_MB_I16x16;
                    rec.kind = FJ_MB_I16x16;
                    rec.pred = (uint8_t)((itype - 1) & 3);
                }
                uint32_t cm = br_ue(br);
                if (br_overrun(br) || cm > 3) FAIL;
                if (!intra_modes_have_neighbours(&rec, m, is_i16, (int)cm)) P2ERR(&c);
                rec.pred |= (uint8_t)(cm << 2);
            }
            if (is_i16) {
                cbp = (((uint32_t)(itype - 1) >> 2) % 3) << 4;
                if (itype >= 13) cbp |= 15;
            } else {
                uint32_t code = br_ue(br);
                if (br_overrun(br) || code > 47) FAIL;
                cbp = ptype >= 0 ? cbp_inter[code] : cbp_intra[code];
            }
            memset(m->tc, 0, sizeof(m->tc));
            if (cbp || is_i16) {
                int32_t dq = br_se(br);
                if (br_overrun(br) || dq < -26 || dq > 25) FAIL;
                if (parse_residual(&c, is_i16, cbp, coefs, &rec.coded, level_sums)) FAIL;
                *qp += dq;
                if (*qp < 0) *qp += 52; else if (*qp >= 52) *qp -= 52;
            }
            m->qp = (uint8_t)*qp;
        }
    }
    if (d->mb_decoded[addr] != 255) d->mb_decoded[addr]++;   /* saturates: a hostile stream cannot wrap it back to "undecoded" */
    /* residual outside [-512,511]: the reference's first check when it reconstructs the macroblock, redundant copies
     * included (macroblock_layer.c:1090-1094 runs before the decoded > 1 tests of the prediction) */
    if (!c.p2err && rec.coded && rec.kind != FJ_MB_IPCM) {
        int qi = (int)m->qp + pps->chroma_qp_index_offset;
        qi = qi < 0 ? 0 : qi > 51 ? 51 : qi;
        /* (almost every macroblock is cleared by the bound on the level magnitudes the parse has summed up) */
        if (!((rec.coded & FJ_CODED_LUMA_DC_RAW) == 0 && hd_residual_bound_ok4(level_sums[0], level_sums[1], level_sums[2], level_sums[3], m->qp, qpc_table[qi]))) {
            /* the bound does not clear it: the kernels must not run this macroblock's transforms in 16 bits (framejob.h) */
            if (rec.kind == FJ_MB_INTER) rec.coded |= FJ_CODED_WIDE;
            if (hd_residual_out_of_range(coefs + 16u * coef_start, rec.coded, m->qp, qpc_table[qi], rec.kind == FJ_MB_I16x16)) {
                P2ERR(&c);
                c.ok_blocks = 0; c.ok_quads = 0;   /* the residual is processed before the prediction: no motion vector was written */
            }
        }
    }
    if (c.p2err) restore_unwritten(&c);
    if (c.p2err && first_decode) {
        {
            /* Counted as decoded, never reconstructed.  The reference has by now stored the macroblock type, the
             * coefficient counts and the updated QP in its mbStorage_t (macroblock_layer.c:985-1046), and if the slice
             * roll-back does not reach this macroblock h264bsdFilterPicture filters it with them: an intra macroblock
             * keeps a record that says so (FJ_MB_STALE).  An inter macroblock is always rolled back with its P slice. */
            FjMbRec keep;
            memset(&keep, 0, sizeof(keep));
            keep.kind = FJ_MB_ABSENT;
            if (rec.kind == FJ_MB_I4x4 || rec.kind == FJ_MB_I16x16) {
                keep.kind = FJ_MB_STALE;
                keep.qp_y = m->qp;
                keep.dbk = rec.dbk; keep.alpha_off = rec.alpha_off; keep.beta_off = rec.beta_off; keep.cqp_off = rec.cqp_off;
                keep.coef_idx = coef_start;
                if (d->mb_ghost && d->mb_ghost[addr]) d->ghost_needed = 1;   /* it shows what a rolled-back slice left there */
            }
            /* (a macroblock that a failed redundant slice left "not decoded" with an earlier slice's pixels under it:
             * those pixels stay, this record only speaks for the deblocking filter — RedoMb) */
            if (d->mb_rec_sid[addr] && keep.kind == FJ_MB_STALE && hd_redo_keep_first(d, addr, &recs[addr], &mvs[addr][0][0])) FAIL;
            if (!(d->mb_rec_sid[addr] && keep.kind == FJ_MB_ABSENT)) {
                recs[addr] = keep;
                d->mb_rec_sid[addr] = sid;
            }
        }
        FAIL;
    }

    if (first_decode && d->mb_redone && d->mb_redone[addr]) rec.pred |= FJ_PRED_PHASE2;   /* an earlier version is already set aside: this one is written on top of it */
    else if (first_decode && d->mb_rec_sid[addr] && recs[addr].kind != FJ_MB_ABSENT && recs[addr].kind != FJ_MB_STALE) {
        /* A macroblock that a failed redundant slice un-decoded (hd_core.c, mark_slice_corrupted) is decoded anew: the
         * reference writes it again, over pixels that other macroblocks may have predicted from.  Those are reconstructed
         * first (RedoMb), this decode on top of them. */
        if (!hd_redo_keep_first(d, addr, &recs[addr], &mvs[addr][0][0])) rec.pred |= FJ_PRED_PHASE2;
    }
    if (!first_decode) {
        /* Decoded again by a redundant slice: the pixels of the first decode stay (its record is set aside for the
         * reconstruction), everything the deblocking filter looks at becomes what this decode says (hostdec.h, RedoMb).
         * The coefficients of this decode are never used: only their positions (rec.coded) matter. */
        d->coef_blocks = coef_start;
        if (hd_redo_keep_first(d, addr, &recs[addr], &mvs[addr][0][0])) return 0;     /* out of memory: first metadata stays */
        rec.coef_idx = 0;
    }
    rec.qp_y = m->qp;
    {
        int qi = (int)m->qp + pps->chroma_qp_index_offset;
        qi = qi < 0 ? 0 : qi > 51 ? 51 : qi;
        rec.qp_c = qpc_table[qi];
    }
    if (rec.kind == FJ_MB_INTER) {
        memcpy(rec.ref_slot, m->ref_slot, 4);
        int16_t (*dst)[2] = mvs[addr];
        /* (a redundant decode that failed while it reconstructed: m->mv / m->ref_slot are what the reference's
         * mbStorage_t holds, restore_unwritten) */
        /* one vector: it travels in the record, the dense entry stays unwritten (framejob.h, FJ_PRED_UNIFORM_MV) */
        if (!c.p2err && ((rec.pred >> FJ_PRED_PARTS_SHIFT) & 3) == FJ_PARTS_16x16) { rec.mv[0] = m->mv[0][0]; rec.mv[1] = m->mv[0][1]; rec.pred |= FJ_PRED_UNIFORM_MV; }
        else
        for (int z = 0; z < 16; z++) {
            const int r = 4 * Z_Y[z] + Z_X[z];
            dst[r][0] = m->mv[z][0];
            dst[r][1] = m->mv[z][1];
        }
        d->n_inter += (uint32_t)first_decode;
    } else {
        d->n_intra += (uint32_t)first_decode;
    }
    recs[addr] = rec;
    d->mb_rec_sid[addr] = sid;
    /* (a redundant decode that fails while it reconstructs has by then replaced the macroblock's type, coefficient
     * counts and QP in the reference's mbStorage_t, macroblock_layer.c:985-1046: the record above says the same) */
    if (c.p2err) FAIL;
    return 0;
}

/* Coefficient blocks of a macroblock that failed are given back: nothing refers to them (no record was written, or
 * an inert one), and a hostile stream must not be able to grow the section by repeating broken slices. */
/* P_Skip, the common case (6 of 10 macroblocks of the bundled 1080p stream) without the generality of decode_mb_body: first
 * decode of the macroblock in this picture, nothing set aside for it, reference 0 present, motion vector in range — i.e. no
 * error path can be reached.  Writes exactly what decode_mb_body writes for such a macroblock (MbInfo, record, vectors,
 * counters: tests/test_parser_fast_paths.py compares the frame jobs of whole streams with and without it); returns 0 when it
 * declined and the general path has to run. */
static int decode_skip_fast(HostDec *d, const SliceHdr *sh, const Pps *pps, uint32_t addr, int qp, int slot)
{
    /* slot: the frame buffer behind RefPicList0[0], the same for every macroblock of the slice (hd_decode_slice_data) */
    if (d->mb_decoded[addr] || d->mb_rec_sid[addr] || (d->mb_redone && d->mb_redone[addr])) return 0;
    if (slot < 0) return 0;
    MbCtx c;                                          /* only what the motion vector prediction looks at */
    const uint32_t sid = d->slice_id, w = d->width_mbs, mby = hd_mb_row(d, addr), mbx = addr - mby * w;
    MbInfo *m = c.cur = &d->mb[addr];
    c.done = 0;
    c.A = mbx ? usable(d, addr - 1, sid) : NULL;
    c.B = mby ? usable(d, addr - w, sid) : NULL;
    c.C = (mby && mbx + 1 < w) ? usable(d, addr - w + 1, sid) : NULL;
    c.D = (mby && mbx) ? usable(d, addr - w - 1, sid) : NULL;
    int16_t mv[2] = { 0, 0 };
    {
        const Nb a = nb_from(c.A, 3, 0), b = nb_from(c.B, 0, 3);
        if (a.avail && b.avail && !(a.ref == 0 && a.mx == 0 && a.my == 0) && !(b.ref == 0 && b.mx == 0 && b.my == 0)) {
            predict_mv(&c, 0, 0, 4, 0, 0, mv);
            if ((uint32_t)(mv[0] + 8192) >= 16384u || (uint32_t)(mv[1] + 2048) >= 4096u) return 0;
        }
    }
    FjHeader *hdr = (FjHeader *)d->job;
    FjMbRec *recs = (FjMbRec *)(d->job + hdr->rec_off);
    /* MbInfo */
    m->dbk_idc = (uint8_t)sh->disable_deblocking_filter_idc;
    m->mb_type = 0;
    m->kind = FJ_MB_INTER;
    memset(m->ref_idx, 0, 4);
    memset(m->ref_slot, slot, 4);
    uint32_t one;
    memcpy(&one, mv, 4);
    const uint64_t two = (uint64_t)one << 32 | one;
    uint8_t *mdst = (uint8_t *)m->mv;                 /* 4-byte aligned only */
    for (int i = 0; i < 8; i++) memcpy(mdst + 8 * i, &two, 8);
    memset(m->tc, 0, sizeof(m->tc));
    m->qp = (uint8_t)qp;
    d->mb_decoded[addr] = 1;
    /* record: four 64-bit words composed in registers and stored once each (a struct filled byte by byte and then copied
     * with two 16-byte moves waits for its own stores: they cannot be forwarded to a wider load) */
    {
        int qi = qp + pps->chroma_qp_index_offset;
        qi = qi < 0 ? 0 : qi > 51 ? 51 : qi;
        uint32_t dbk = 0;
        if (sh->disable_deblocking_filter_idc != 1) {
            dbk = FJ_DBK_INNER;
            if (mbx && (sh->disable_deblocking_filter_idc != 2 || c.A)) dbk |= FJ_DBK_LEFT;
            if (mby && (sh->disable_deblocking_filter_idc != 2 || c.B)) dbk |= FJ_DBK_TOP;
        }
        _Static_assert(offsetof(FjMbRec, coded) == 8 && offsetof(FjMbRec, ref_slot) == 16 && offsetof(FjMbRec, cqp_off) == 20 &&
                       offsetof(FjMbRec, mv) == 24 && offsetof(FjMbRec, mvx) == 28 && sizeof(FjMbRec) == 32, "FjMbRec layout");
        const uint64_t w0 = (uint64_t)FJ_MB_INTER | (uint64_t)(uint8_t)qp << 8 | (uint64_t)qpc_table[qi] << 16 /* avail 0 */ |
                            (uint64_t)((FJ_PARTS_16x16 << FJ_PRED_PARTS_SHIFT) | FJ_PRED_UNIFORM_MV) << 32 | (uint64_t)dbk << 40 |
                            (uint64_t)(uint8_t)sh->alpha_off << 48 | (uint64_t)(uint8_t)sh->beta_off << 56;
        const uint64_t w1 = (uint64_t)d->coef_blocks << 32;                                     /* coded 0 | coef_idx */
        const uint64_t w2 = (uint64_t)((uint32_t)(uint8_t)slot * 0x01010101u) | (uint64_t)(uint8_t)pps->chroma_qp_index_offset << 32;   /* ref_slot x 4 | cqp_off | dbk_trivial 0 | intra_level 0 */
        const uint64_t w3 = (uint64_t)one;                                                       /* mv | mvx 0: the one vector travels in the record (FJ_PRED_UNIFORM_MV) */
        uint8_t *rw = (uint8_t *)&recs[addr];
        memcpy(rw, &w0, 8); memcpy(rw + 8, &w1, 8); memcpy(rw + 16, &w2, 8); memcpy(rw + 24, &w3, 8);
    }
    d->mb_rec_sid[addr] = sid;
    d->n_inter++;
    return 1;
}

static int decode_mb(HostDec *d, BitReader *br, const SliceHdr *sh, const Pps *pps, uint32_t addr, int skipped, int *qp, int ref0_slot)
{
    if (skipped && !hd_no_fast_skip && decode_skip_fast(d, sh, pps, addr, *qp, ref0_slot)) return 0;
    const uint32_t coef_start = d->coef_blocks;
    const int rc = decode_mb_body(d, br, sh, pps, addr, skipped, qp);
    if (rc) d->coef_blocks = coef_start;
    return rc;
}

/* ---------------------------------------------------------------- slice_data(), 7.3.4 */
uint32_t hd_next_mb_in_group(const uint32_t *map, uint32_t n, uint32_t addr)
{
    const uint32_t g = map[addr];
    for (uint32_t i = addr + 1; i < n; i++) if (map[i] == g) return i;
    return 0;
}

int hd_decode_slice_data(HostDec *d, BitReader *br, const SliceHdr *sh, int nal_ref_idc)
{
    (void)nal_ref_idc;
    const Pps *pps = d->active_pps;
    uint32_t addr = sh->first_mb, skip_run = 0, count = 0;
    int prev_skipped = 0, more;
    int qp = pps->pic_init_qp + sh->slice_qp_delta;
    d->slice_id++;
    d->last_mb_addr = 0;
    const int ref0_slot = sh->is_p ? hd_dpb_ref_slot(&d->dpb, 0) : -1;      /* P_Skip predicts from it (decode_skip_fast) */
    if (sh->redundant_pic_cnt) d->slice_ids_rewritten = 1;

    do {
        if (!sh->redundant_pic_cnt && d->mb_decoded[addr]) FAIL;
        d->mb_slice_id[addr] = d->slice_id;
        {
            MbInfo *m = &d->mb[addr];               /* SetMbParams: stamped whether the parse below succeeds or not */
            m->dbk_idc = (uint8_t)sh->disable_deblocking_filter_idc;
            m->alpha_off = (int8_t)sh->alpha_off; m->beta_off = (int8_t)sh->beta_off;
            m->cqp_off = (int8_t)pps->chroma_qp_index_offset;
        }
        if (d->mb_decoded[addr] || d->mb_rec_sid[addr]) {
            /* a redundant slice over a decoded macroblock (or over one that an earlier redundant slice un-decoded and whose
             * pixels are still in the picture): the slice-level parameters the deblocking filter uses are
             * restamped before the macroblock is parsed (SetMbParams, slice_data.c:53-66,140), so they change even when
             * the parse then fails; edge flags across slice boundaries are settled at the end of the picture */
            FjMbRec *r = (FjMbRec *)(d->job + ((const FjHeader *)d->job)->rec_off) + addr;
            r->alpha_off = (int8_t)sh->alpha_off; r->beta_off = (int8_t)sh->beta_off;
            r->cqp_off = (int8_t)pps->chroma_qp_index_offset;
            r->dbk = (uint8_t)(sh->disable_deblocking_filter_idc == 1 ? 0 :
                               FJ_DBK_INNER | (addr % d->width_mbs ? FJ_DBK_LEFT : 0) | (addr >= d->width_mbs ? FJ_DBK_TOP : 0));
            d->mb[addr].dbk_idc = (uint8_t)sh->disable_deblocking_filter_idc;
        }
        if (sh->is_p && !prev_skipped) {
            skip_run = br_ue(br);
            if (br_overrun(br) || skip_run > d->pic_size_mbs - addr) FAIL;
            if (skip_run) prev_skipped = 1;
        }
        int skipped = 0;
        if (skip_run) { skip_run--; skipped = 1; }
        else prev_skipped = 0;
        if (decode_mb(d, br, sh, pps, addr, skipped, &qp, ref0_slot)) FAIL;
        if (d->mb_decoded[addr] == 1) count++;
        more = br_more_rbsp_data(br) || skip_run;
        if (!sh->is_p) d->last_mb_addr = addr;
        addr = hd_next_mb_in_group(d->slice_group_map, d->pic_size_mbs, addr);
        if (more && !addr) FAIL;
    } while (more);

    if (d->num_decoded_mbs + count > d->pic_size_mbs) FAIL;
    d->num_decoded_mbs += count;
    return 0;
}
