/*
 * hd_resid.c — the one decode error the reference finds AFTER dequantisation: a residual sample outside
 * [-512, 511] makes h264bsdProcessBlock return HANTRO_NOK (reference src/h264bsd_transform.c:184-188, 199, 224-228),
 * which ProcessResidual / h264bsdDecodeMacroblock hand up (src/h264bsd_macroblock_layer.c:1374-1421, 1090-1094) and
 * h264bsdDecodeSliceData turns into a corrupt slice (src/h264bsd_slice_data.c:186-193 -> decoder.c marks the slice,
 * conceals at the end of the access unit).
 *
 * The error decides how the REST OF THE BITSTREAM is consumed (the slice stops at this macroblock, the call returns
 * H264BSD_ERROR), so it has to be known while parsing — it cannot wait for the device, whose pixels materialise only
 * when the application pulls the picture.  The host therefore decides it here, from the coefficient blocks it has just
 * written into the frame job: a bound that proves almost every block in range with one pass over its levels (the
 * inverse transform never amplifies: every output is a sum of the 16 dequantised inputs with weights of magnitude
 * <= 1), and the exact integer transform for the few blocks the bound cannot clear.  The kernels carry the same test as
 * a tripwire (FrameDesc error word): it must never fire for a job this file has passed.
 *
 * Arithmetic mirrors the device code (kernels.hip.h idct_quad / mb_residual_compute), which is bit-exact against the
 * reference; all intermediates fit 32 bits for every level CAVLC can code (|level| <= 2530, scale <= 29 << 8).
 */
#include <stdlib.h>
#include "hostdec.h"

static const uint8_t level_scale[6][3] = {
    { 10, 13, 16 }, { 11, 14, 18 }, { 13, 16, 20 }, { 14, 18, 23 }, { 16, 20, 25 }, { 18, 23, 29 } };

/* exact: does the residual of this block (raster levels c, element 0 replaced by dc when use_dc) leave [-512,511]? */
static int block_out_of_range(const int16_t *c, int qp, int use_dc, int32_t dc)
{
    const int m = qp % 6, sh = qp / 6;
    const int32_t ls[3] = { level_scale[m][0], level_scale[m][1], level_scale[m][2] };
    int32_t d[16];
    for (int r = 0; r < 4; r++)
        for (int k = 0; k < 4; k++)
            d[4 * r + k] = ((int32_t)c[4 * r + k] * ls[(r & 1) + (k & 1)]) << sh;
    if (use_dc) d[0] = dc;
    for (int r = 0; r < 4; r++) {
        int32_t *p = d + 4 * r;
        const int32_t e0 = p[0] + p[2], e1 = p[0] - p[2], e2 = (p[1] >> 1) - p[3], e3 = p[1] + (p[3] >> 1);
        p[0] = e0 + e3; p[1] = e1 + e2; p[2] = e1 - e2; p[3] = e0 - e3;
    }
    for (int k = 0; k < 4; k++) {
        const int32_t *p = d + k;
        const int32_t e0 = p[0] + p[8], e1 = p[0] - p[8], e2 = (p[4] >> 1) - p[12], e3 = p[4] + (p[12] >> 1);
        const int32_t o[4] = { (e0 + e3 + 32) >> 6, (e1 + e2 + 32) >> 6, (e1 - e2 + 32) >> 6, (e0 - e3 + 32) >> 6 };
        for (int r = 0; r < 4; r++) if ((uint32_t)(o[r] + 512) > 1023u) return 1;
    }
    return 0;
}

/* bound first, exact transform only when the bound cannot prove the block in range */
static inline int block_check(const int16_t *c, int qp, int use_dc, int32_t dc)
{
    uint32_t sum = 0;
    for (int i = use_dc ? 1 : 0; i < 16; i++) sum += (uint32_t)abs(c[i]);
    const uint64_t bound = (uint64_t)sum * ((uint32_t)level_scale[qp % 6][2] << (qp / 6)) + (uint64_t)(use_dc ? llabs((long long)dc) : 0);
    if (bound <= 32735u) return 0;          /* -32800 <= x <= 32735  <=>  -512 <= (x + 32) >> 6 <= 511 */
    return block_out_of_range(c, qp, use_dc, dc);
}

int hd_residual_bound_ok(uint32_t sum_l, uint32_t sum_d, uint32_t sum_c, int qp_y, int qp_c)
{
    /* the bound of block_check() for the macroblock's WORST blocks: sum_l / sum_c = the largest sum of level magnitudes of one
     * luma / one chroma AC block, sum_d = the larger of the two planes' chroma DC sums (every DC of a plane is a signed sum of
     * its four levels) — collected by the parser on the way.  Holds -> every block's residual is in range AND every intermediate
     * of its transforms fits 16 bits (FJ_CODED_WIDE stays clear, framejob.h) */
    const uint64_t bound_l = (uint64_t)sum_l * ((uint32_t)level_scale[qp_y % 6][2] << (qp_y / 6));
    const int q6c = qp_c / 6;
    const uint64_t dc_max = ((uint64_t)sum_d * level_scale[qp_c % 6][0]) << (q6c >= 1 ? q6c - 1 : 0);
    const uint64_t bound_c = (uint64_t)sum_c * ((uint32_t)level_scale[qp_c % 6][2] << q6c) + dc_max;
    return bound_l <= 32735u && bound_c <= 32735u;
}

/* ... with an Intra16x16 DC block: every luma DC is a signed sum of its sixteen levels, scaled as in 8.5.10 (sum_ldc = the
 * sum of their magnitudes; 0 without the block) and replaces element 0 of its 4x4 block */
int hd_residual_bound_ok4(uint32_t sum_l, uint32_t sum_d, uint32_t sum_c, uint32_t sum_ldc, int qp_y, int qp_c)
{
    if (!sum_ldc) return hd_residual_bound_ok(sum_l, sum_d, sum_c, qp_y, qp_c);
    const int q6 = qp_y / 6;
    const uint64_t f = (uint64_t)sum_ldc * level_scale[qp_y % 6][0];
    const uint64_t ldc_max = q6 >= 2 ? f << (q6 - 2) : ((f + (1u << (1 - q6))) >> (2 - q6));
    const uint64_t bound_l = (uint64_t)sum_l * ((uint32_t)level_scale[qp_y % 6][2] << q6) + ldc_max;
    return bound_l <= 32735u && hd_residual_bound_ok(0, sum_d, sum_c, qp_y, qp_c);
}

static const int16_t zero_block[16];

/* blk: the macroblock's coefficient blocks in frame-job order (framejob.h); coded: FjMbRec.coded.
 * Returns 1 when the reference's ProcessResidual would fail on this macroblock. */
int hd_residual_out_of_range(const int16_t *blk, uint32_t coded, int qp_y, int qp_c, int is_i16)
{
    int32_t ydc[16];
    const int has_ldc = (coded & FJ_CODED_LUMA_DC) != 0;
    if (!has_ldc) {
        /* One pass over the whole macroblock before any per-block work (all macroblocks but Intra16x16): the sum of
         * the magnitudes of ALL its levels bounds the sum of any one block, so the per-block bound holds for every
         * block at once when it holds for the totals.  Almost every coded macroblock leaves here. */
        const uint32_t n_luma = (uint32_t)__builtin_popcount(coded & 0xFFFFu), n_cac = (uint32_t)__builtin_popcount((coded >> 16) & 0xFFu);
        const int16_t *p = blk;
        uint32_t sum_l = 0, sum_d = 0, sum_c = 0;
        for (uint32_t i = 0; i < 16u * n_luma; i++) sum_l += (uint32_t)abs(p[i]);
        p += 16u * n_luma;
        if (coded & FJ_CODED_CHROMA_DC) { for (int i = 0; i < 8; i++) sum_d += (uint32_t)abs(p[i]); p += 16; }
        for (uint32_t i = 0; i < 16u * n_cac; i++) sum_c += (uint32_t)abs(p[i]);
        const uint64_t bound_l = (uint64_t)sum_l * ((uint32_t)level_scale[qp_y % 6][2] << (qp_y / 6));
        const int q6c = qp_c / 6;
        const uint64_t dc_max = ((uint64_t)sum_d * level_scale[qp_c % 6][0]) << (q6c >= 1 ? q6c - 1 : 0);   /* |every chroma DC| after 8.5.11 */
        const uint64_t bound_c = (uint64_t)sum_c * ((uint32_t)level_scale[qp_c % 6][2] << q6c) + dc_max;
        if (bound_l <= 32735u && bound_c <= 32735u) return 0;
    }
    if (has_ldc) {
        if (coded & FJ_CODED_LUMA_DC_RAW) {
            for (int i = 0; i < 16; i++) ydc[i] = blk[i];
        } else {
            /* 8.5.10: f = A c A with the 4x4 Hadamard matrix, then the DC scaling (transform.c:255-338) */
            int32_t t[16];
            for (int r = 0; r < 4; r++) {
                const int16_t *p = blk + 4 * r;
                const int32_t a = p[0] + p[2], b = p[0] - p[2], cc = p[1] - p[3], dd = p[1] + p[3];
                t[4 * r] = a + dd; t[4 * r + 1] = b + cc; t[4 * r + 2] = b - cc; t[4 * r + 3] = a - dd;
            }
            const int32_t ls = level_scale[qp_y % 6][0], q6 = qp_y / 6;
            for (int k = 0; k < 4; k++) {
                const int32_t a = t[k] + t[8 + k], b = t[k] - t[8 + k], cc = t[4 + k] - t[12 + k], dd = t[4 + k] + t[12 + k];
                const int32_t f[4] = { a + dd, b + cc, b - cc, a - dd };
                for (int r = 0; r < 4; r++)
                    ydc[4 * r + k] = q6 >= 2 ? (f[r] * ls) << (q6 - 2) : (f[r] * ls + (1 << (1 - q6))) >> (2 - q6);
            }
        }
        blk += 16;
    }
    /* luma: z ascending is the storage order; the DC of block z sits at its raster position in the DC block */
    for (int z = 0; z < 16; z++) {
        const int has = (coded >> z) & 1;
        if (is_i16) {
            const int bx = ((z >> 2) & 1) * 2 + (z & 1), by = (z >> 3) * 2 + ((z >> 1) & 1);
            const int32_t dc = has_ldc ? ydc[4 * by + bx] : 0;
            if ((dc || has) && block_check(has ? blk : zero_block, qp_y, 1, dc)) return 1;
        } else if (has && block_check(blk, qp_y, 0, 0)) return 1;
        if (has) blk += 16;
    }
    if (!(coded & (FJ_CODED_CHROMA_DC | 0x00FF0000u))) return 0;
    int32_t cdc[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (coded & FJ_CODED_CHROMA_DC) {
        const int32_t ls = level_scale[qp_c % 6][0], q6 = qp_c / 6;
        for (int p = 0; p < 2; p++) {
            const int16_t *c = blk + 4 * p;
            const int32_t f[4] = { c[0] + c[1] + c[2] + c[3], c[0] - c[1] + c[2] - c[3],
                                   c[0] + c[1] - c[2] - c[3], c[0] - c[1] - c[2] + c[3] };
            for (int i = 0; i < 4; i++) cdc[4 * p + i] = q6 >= 1 ? (f[i] * ls) << (q6 - 1) : (f[i] * ls) >> 1;
        }
        blk += 16;
    }
    for (int k = 0; k < 8; k++) {
        const int has = (coded >> (16 + k)) & 1;
        if ((cdc[k] || has) && block_check(has ? blk : zero_block, qp_c, 1, cdc[k])) return 1;
        if (has) blk += 16;
    }
    return 0;
}
