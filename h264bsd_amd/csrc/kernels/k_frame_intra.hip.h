/* kernels/k_frame_intra.hip.h — k_frame_intra: intra and concealed macroblocks, one workgroup per picture (or row band).  Part of kernels.hip.h (which see); not a stand-alone header. */
#pragma once
namespace h264k {
/* ------------------------------------------------------------------ intra macroblocks */
constexpr int TS = 32;   /* intra luma tile: row 0 = row above, rows 1..16 = MB; byte 3 = left column / corner,
                            bytes 4..19 = MB columns (dword aligned), bytes 20..23 of row 0 = above-right */

/* ---- concealment of a lost macroblock from its neighbours (reference ConcealMb, src/h264bsd_conceal.c:346-560) ----
 * Per plane the block is rebuilt from three numbers: t0 (mean of the border samples of the usable sides), t1 (left-
 * right slope) and v (top-bottom slope), pushed through the reference's 3-coefficient inverse transform; every
 * (size/4)x(size/4) sub-block is constant.  S = sum of a side's border samples, D = first half minus second half. */
__device__ __forceinline__ void conceal_coeffs(int SA, int DA, int SB, int DB, int SL, int DL, int SR, int DR,
                                               bool A, bool B, bool L, bool R, int sh, int &t0, int &t1, int &v)
{
    const int hor = (int)A + (int)B, ver = (int)L + (int)R, j = hor + ver;
    int f0 = (A ? SA : 0) + (B ? SB : 0) + (L ? SL : 0) + (R ? SR : 0);
    int f1 = (A ? DA : 0) + (B ? DB : 0), f4 = (L ? DL : 0) + (R ? DR : 0);
    if (!hor && L && R) f1 = (SL - SR) >> (5 - sh);
    else if (hor) f1 >>= (3 - sh + hor);
    if (!ver && A && B) f4 = (SA - SB) >> (5 - sh);
    else if (ver) f4 >>= (3 - sh + ver);
    f0 = j == 1 ? f0 >> (4 - sh) : j == 2 ? f0 >> (5 - sh) : j == 3 ? (21 * f0) >> (10 - sh) : f0 >> (6 - sh);
    t0 = f0; t1 = f1; v = f4;
}
/* value of sub-block (bx, by) after the reference's Transform() (conceal.c:589-637) */
__device__ __forceinline__ int conceal_value(int t0, int t1, int v, int bx, int by)
{
    const int h = bx == 0 ? t0 + t1 : bx == 1 ? t0 + (t1 >> 1) : bx == 2 ? t0 - (t1 >> 1) : t0 - t1;
    return clip255(by == 0 ? h + v : by == 1 ? h + (v >> 1) : by == 2 ? h - (v >> 1) : h - v);
}

__device__ __noinline__ void conceal_mb(const FrameDesc &fd, uint32_t mb, int lane, unsigned used)
{
    const int wmb = fd.wmb;
    /* the macroblock's tile; the neighbours' tiles lie wmb tiles above / below and one tile to either side */
    uint8_t *T = fd.cur + (size_t)mb * TILE;
    const ptrdiff_t up = -(ptrdiff_t)wmb * TILE, down = (ptrdiff_t)wmb * TILE;
    const bool A = used & FJ_CONC_ABOVE, B = used & FJ_CONC_BELOW, L = used & FJ_CONC_LEFT, R = used & FJ_CONC_RIGHT;
    /* luma: lanes 0-15 above, 16-31 below, 32-47 left, 48-63 right, one border sample each */
    {
        const int side = lane >> 4, k = lane & 15;
        uint8_t *Y = T;
        int s = 0;
        if (side == 0 && A) s = T[up + 15 * 16 + k];
        if (side == 1 && B) s = T[down + k];
        if (side == 2 && L) s = T[-TILE + k * 16 + 15];
        if (side == 3 && R) s = T[TILE + k * 16];
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4);
        const int o = __shfl_xor(s, 8);
        const int S = s + o, D = (lane & 8) ? o - s : s - o;
        int t0, t1, v;
        conceal_coeffs(__shfl(S, 0), __shfl(D, 0), __shfl(S, 16), __shfl(D, 16), __shfl(S, 32), __shfl(D, 32), __shfl(S, 48),
                       __shfl(D, 48), A, B, L, R, 0, t0, t1, v);
        const int blk = lane >> 2, row = lane & 3, bx = blk & 3, by = blk >> 2;
        const uint32_t px = (uint32_t)conceal_value(t0, t1, v, bx, by) * 0x01010101u;
        *reinterpret_cast<uint32_t *>(Y + (by * 4 + row) * 16 + bx * 4) = px;
    }
    /* chroma: lane = 32*plane + 8*side + k */
    {
        const int plane = lane >> 5, side = (lane >> 3) & 3, k = lane & 7;
        const uint8_t *P = T + T_CB + plane * 64;
        int s = 0;
        if (side == 0 && A) s = P[up + 7 * 8 + k];
        if (side == 1 && B) s = P[down + k];
        if (side == 2 && L) s = P[-TILE + k * 8 + 7];
        if (side == 3 && R) s = P[TILE + k * 8];
        s += __shfl_xor(s, 1); s += __shfl_xor(s, 2);
        const int o = __shfl_xor(s, 4);
        const int S = s + o, D = (lane & 4) ? o - s : s - o;
        int t0[2], t1[2], v[2];
#pragma unroll
        for (int p = 0; p < 2; p++)
            conceal_coeffs(__shfl(S, 32 * p), __shfl(D, 32 * p), __shfl(S, 32 * p + 8), __shfl(D, 32 * p + 8), __shfl(S, 32 * p + 16),
                           __shfl(D, 32 * p + 16), __shfl(S, 32 * p + 24), __shfl(D, 32 * p + 24), A, B, L, R, 1, t0[p], t1[p], v[p]);
        if (lane < 32) {
            /* lane -> plane (lane>>4), row y = (lane>>1)&7, half = lane&1: four samples = two 2x2 sub-block values */
            const int pl = lane >> 4, y = (lane >> 1) & 7, half = lane & 1;
            uint8_t *Q = T + T_CB + pl * 64 + y * 8 + half * 4;
            const int a0 = conceal_value(t0[pl], t1[pl], v[pl], half * 2, y >> 1);
            const int a1 = conceal_value(t0[pl], t1[pl], v[pl], half * 2 + 1, y >> 1);
            *reinterpret_cast<uint32_t *>(Q) = (uint32_t)a0 * 0x00000101u | (uint32_t)a1 * 0x01010000u;
        }
    }
}

/* ---- Intra4x4 prediction, table-driven ----
 * Every sample of the eight directional modes is (a + 2b + c + 2) >> 2 or (a + b + 1) >> 1 over three of the block's
 * 13 neighbour samples n[0] = corner, n[1..8] = above 0..7 (above-right replaced by above[3] when it is not available),
 * n[9..12] = left 0..3 (8.3.1.2.1-9; reference Intra4x4*Prediction, src/h264bsd_intra_prediction.c:1493-1830).  The table
 * holds, per (mode, row, sample): a byte selector for v_perm_b32 (the three neighbours out of n[0..7] resp. n[8..12]) and the
 * byte mask that picks between the two.  Weights and rounding are the same for every sample — v_dot4_u32_u8 with (1, 2, 1), + 2,
 * >> 2 — because the two-tap form is written as (a + 2 b + a + 2) >> 2 = (a + b + 1) >> 1: the selector names a twice.  One table
 * row (the four samples of a block row) is 32 bytes, two ds_read_b128; a lane that owns a block fetches its row ONCE, before the
 * ten dependent steps of the macroblock (round 4 fetched four 16-byte entries — selector, mask, weights, shift — inside every
 * step, a second LDS round trip on each link of the chain).  Five instructions per sample, ONE instruction stream for all lanes whatever their modes are (a switch over the
 * modes executes every mode that occurs among the active lanes — up to eight when four macroblocks are predicted
 * together).  DC (mode 2) is the only special case. */
__constant__ uint2 c_i4tab[36][4] = {
    { { 0x0C010101u, 0x00000000u }, { 0x0C020202u, 0x00000000u }, { 0x0C030303u, 0x00000000u }, { 0x0C040404u, 0x00000000u } },
    { { 0x0C010101u, 0x00000000u }, { 0x0C020202u, 0x00000000u }, { 0x0C030303u, 0x00000000u }, { 0x0C040404u, 0x00000000u } },
    { { 0x0C010101u, 0x00000000u }, { 0x0C020202u, 0x00000000u }, { 0x0C030303u, 0x00000000u }, { 0x0C040404u, 0x00000000u } },
    { { 0x0C010101u, 0x00000000u }, { 0x0C020202u, 0x00000000u }, { 0x0C030303u, 0x00000000u }, { 0x0C040404u, 0x00000000u } },
    { { 0x0C010101u, 0x00FFFFFFu }, { 0x0C010101u, 0x00FFFFFFu }, { 0x0C010101u, 0x00FFFFFFu }, { 0x0C010101u, 0x00FFFFFFu } },
    { { 0x0C020202u, 0x00FFFFFFu }, { 0x0C020202u, 0x00FFFFFFu }, { 0x0C020202u, 0x00FFFFFFu }, { 0x0C020202u, 0x00FFFFFFu } },
    { { 0x0C030303u, 0x00FFFFFFu }, { 0x0C030303u, 0x00FFFFFFu }, { 0x0C030303u, 0x00FFFFFFu }, { 0x0C030303u, 0x00FFFFFFu } },
    { { 0x0C040404u, 0x00FFFFFFu }, { 0x0C040404u, 0x00FFFFFFu }, { 0x0C040404u, 0x00FFFFFFu }, { 0x0C040404u, 0x00FFFFFFu } },
    { { 0x0C000000u, 0x00000000u }, { 0x0C000000u, 0x00000000u }, { 0x0C000000u, 0x00000000u }, { 0x0C000000u, 0x00000000u } },
    { { 0x0C000000u, 0x00000000u }, { 0x0C000000u, 0x00000000u }, { 0x0C000000u, 0x00000000u }, { 0x0C000000u, 0x00000000u } },
    { { 0x0C000000u, 0x00000000u }, { 0x0C000000u, 0x00000000u }, { 0x0C000000u, 0x00000000u }, { 0x0C000000u, 0x00000000u } },
    { { 0x0C000000u, 0x00000000u }, { 0x0C000000u, 0x00000000u }, { 0x0C000000u, 0x00000000u }, { 0x0C000000u, 0x00000000u } },
    { { 0x0C030201u, 0x00000000u }, { 0x0C040302u, 0x00000000u }, { 0x0C050403u, 0x00000000u }, { 0x0C060504u, 0x00000000u } },
    { { 0x0C040302u, 0x00000000u }, { 0x0C050403u, 0x00000000u }, { 0x0C060504u, 0x00000000u }, { 0x0C070605u, 0x00000000u } },
    { { 0x0C050403u, 0x00000000u }, { 0x0C060504u, 0x00000000u }, { 0x0C070605u, 0x00000000u }, { 0x0C000706u, 0x00FF0000u } },
    { { 0x0C060504u, 0x00000000u }, { 0x0C070605u, 0x00000000u }, { 0x0C000706u, 0x00FF0000u }, { 0x0C000007u, 0x00FFFF00u } },
    { { 0x0C010001u, 0x00FF0000u }, { 0x0C020100u, 0x00000000u }, { 0x0C030201u, 0x00000000u }, { 0x0C040302u, 0x00000000u } },
    { { 0x0C020100u, 0x00FFFF00u }, { 0x0C010001u, 0x00FF0000u }, { 0x0C020100u, 0x00000000u }, { 0x0C030201u, 0x00000000u } },
    { { 0x0C030201u, 0x00FFFFFFu }, { 0x0C020100u, 0x00FFFF00u }, { 0x0C010001u, 0x00FF0000u }, { 0x0C020100u, 0x00000000u } },
    { { 0x0C040302u, 0x00FFFFFFu }, { 0x0C030201u, 0x00FFFFFFu }, { 0x0C020100u, 0x00FFFF00u }, { 0x0C010001u, 0x00FF0000u } },
    { { 0x0C000100u, 0x00000000u }, { 0x0C010201u, 0x00000000u }, { 0x0C020302u, 0x00000000u }, { 0x0C030403u, 0x00000000u } },
    { { 0x0C010001u, 0x000000FFu }, { 0x0C020100u, 0x00000000u }, { 0x0C030201u, 0x00000000u }, { 0x0C040302u, 0x00000000u } },
    { { 0x0C000102u, 0x0000FFFFu }, { 0x0C000100u, 0x00000000u }, { 0x0C010201u, 0x00000000u }, { 0x0C020302u, 0x00000000u } },
    { { 0x0C010203u, 0x00FFFFFFu }, { 0x0C010001u, 0x000000FFu }, { 0x0C020100u, 0x00000000u }, { 0x0C030201u, 0x00000000u } },
    { { 0x0C000100u, 0x0000FF00u }, { 0x0C010001u, 0x000000FFu }, { 0x0C000102u, 0x00000000u }, { 0x0C010203u, 0x00000000u } },
    { { 0x0C010201u, 0x00FFFFFFu }, { 0x0C020100u, 0x00FFFF00u }, { 0x0C000100u, 0x0000FF00u }, { 0x0C010001u, 0x000000FFu } },
    { { 0x0C020302u, 0x00FFFFFFu }, { 0x0C030201u, 0x00FFFFFFu }, { 0x0C010201u, 0x00FFFFFFu }, { 0x0C020100u, 0x00FFFF00u } },
    { { 0x0C030403u, 0x00FFFFFFu }, { 0x0C040302u, 0x00FFFFFFu }, { 0x0C020302u, 0x00FFFFFFu }, { 0x0C030201u, 0x00FFFFFFu } },
    { { 0x0C010201u, 0x00000000u }, { 0x0C020302u, 0x00000000u }, { 0x0C030403u, 0x00000000u }, { 0x0C040504u, 0x00000000u } },
    { { 0x0C030201u, 0x00000000u }, { 0x0C040302u, 0x00000000u }, { 0x0C050403u, 0x00000000u }, { 0x0C060504u, 0x00000000u } },
    { { 0x0C020302u, 0x00000000u }, { 0x0C030403u, 0x00000000u }, { 0x0C040504u, 0x00000000u }, { 0x0C050605u, 0x00000000u } },
    { { 0x0C040302u, 0x00000000u }, { 0x0C050403u, 0x00000000u }, { 0x0C060504u, 0x00000000u }, { 0x0C070605u, 0x00000000u } },
    { { 0x0C010201u, 0x00FFFFFFu }, { 0x0C030201u, 0x00FFFFFFu }, { 0x0C020302u, 0x00FFFFFFu }, { 0x0C040302u, 0x00FFFFFFu } },
    { { 0x0C020302u, 0x00FFFFFFu }, { 0x0C040302u, 0x00FFFFFFu }, { 0x0C030403u, 0x00FFFFFFu }, { 0x0C040403u, 0x00FFFFFFu } },
    { { 0x0C030403u, 0x00FFFFFFu }, { 0x0C040403u, 0x00FFFFFFu }, { 0x0C040404u, 0x00FFFFFFu }, { 0x0C040404u, 0x00FFFFFFu } },
    { { 0x0C040404u, 0x00FFFFFFu }, { 0x0C040404u, 0x00FFFFFFu }, { 0x0C040404u, 0x00FFFFFFu }, { 0x0C040404u, 0x00FFFFFFu } },
};
constexpr int I4TAB_BYTES = 36 * 4 * 8;

/* the table row of (mode, block row y): selector and mask of its four samples */
struct I4Row { uint4 a, b; };                        /* { sel0, mask0, sel1, mask1 }, { sel2, mask2, sel3, mask3 } */
__device__ __forceinline__ I4Row intra4_entries(const uint2 *i4tab, int mode, int y)
{
    const uint4 *ent = reinterpret_cast<const uint4 *>(i4tab + ((mode & 15) * 4 + y) * 4);
    I4Row r;
    r.a = ent[0]; r.b = ent[1];
    return r;
}
/* One row (4 samples) of the Intra4x4 prediction of the block at (bx4, by4) of the macroblock whose LDS tile is `tile`.
 * e: the block row's table entries (intra4_entries).  Lanes without a block pass any valid mode and ignore the result. */
__device__ __forceinline__ void intra4_row(const uint8_t *tile, int bx4, int by4, int mode, bool has_left, bool has_top, bool has_tr,
                                           const I4Row &e, int vv[4])
{
    /* the 13 neighbour samples in seven INDEPENDENT LDS reads: corner | above 0..7 | left 0..3 */
    const uint8_t *trow = &tile[by4 * TS + bx4];
    const uint32_t w0 = *reinterpret_cast<const uint32_t *>(trow), w1 = *reinterpret_cast<const uint32_t *>(trow + 4),
                   w2 = *reinterpret_cast<const uint32_t *>(trow + 8);
    const uint32_t l0 = tile[(by4 + 1) * TS + 3 + bx4], l1 = tile[(by4 + 2) * TS + 3 + bx4],
                   l2 = tile[(by4 + 3) * TS + 3 + bx4], l3 = tile[(by4 + 4) * TS + 3 + bx4];
    const uint32_t tr = has_tr ? w2 : (w1 >> 24) * 0x01010101u;
    const uint32_t N0 = (w0 >> 24) | (w1 << 8), N1 = (w1 >> 24) | (tr << 8);          /* n[0..3], n[4..7] */
    const uint32_t N2 = (tr >> 24) | (l0 << 8) | (l1 << 16) | (l2 << 24), N3 = l3;      /* n[8..11], n[12] */
    const uint32_t sel[4] = { e.a.x, e.a.z, e.b.x, e.b.z }, msk[4] = { e.a.y, e.a.w, e.b.y, e.b.w };
#pragma unroll
    for (int x = 0; x < 4; x++) {
        const uint32_t lo = perm(N1, N0, sel[x]), hi = perm(N3, N2, sel[x]);
        const uint32_t v = (uint32_t)__builtin_amdgcn_bitop3_b32(hi, lo, msk[x], 0xE4);      /* (hi & mask) | (lo & ~mask) */
        vv[x] = (int)(__builtin_amdgcn_udot4(v, 0x00010201u, 2u, false) >> 2);
    }
    if (__ballot(mode == 2) != 0ull) {
        const int st = (int)((w1 & 255u) + ((w1 >> 8) & 255u) + ((w1 >> 16) & 255u) + (w1 >> 24)), sl = (int)(l0 + l1 + l2 + l3);
        const int dc = (has_top && has_left) ? (st + sl + 4) >> 3 : has_left ? (sl + 2) >> 2 : has_top ? (st + 2) >> 2 : 128;
        if (mode == 2) vv[0] = vv[1] = vv[2] = vv[3] = dc;
    }
}

/* Bytes of the LDS luma tile that the macroblock's samples never use carry what the joint Intra4x4 pass needs to know
 * about a macroblock prepared earlier (intra_mb with res_defer): byte 0 = availability flags, bytes 24..31 = the 16 modes */
/* one intra macroblock by one wavefront; tile = 17*TS bytes, ctile = 2 x 9*16 bytes (wave-private LDS).
 * res_defer != nullptr: an Intra4x4 macroblock is only PREPARED — neighbours in the tile, residual (16 x 16 int16) in
 * res_defer, chroma done — and its luma prediction is left to intra4_joint(); other kinds are done completely. */
struct IntraLoads { int nb_y, nb_c; ResidRows rows; };

__device__ __forceinline__ FjMbRec rec_from_lds(const uint32_t *rec_lds)
{
    /* the record was fetched together with those of the other macroblocks this wavefront claimed (one round trip for all
     * of them) and parked in LDS; it is wave-uniform: back into scalar registers */
    FjMbRec rec;
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 8; i++) w[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec_lds[i]);
    __builtin_memcpy(&rec, w, 32);
    return rec;
}

/* The global loads of one intra macroblock — neighbour samples of the un-deblocked current picture and the coefficient
 * rows — issued one macroblock AHEAD of their use (k_frame_intra: while the previous macroblock of the group is being
 * reconstructed), so that the round trip hides behind that work. */
/* cross: the macroblock lies in the first row of a row band (k_frame_intra): the tiles above were written by another
 * workgroup and are read past the L1 (ld_agent_u8). */
/* Which neighbour sample a lane fetches for a macroblock is the same for every macroblock of a picture: byte offsets from
 * the macroblock's tile (the row above lies wmb tiles back) and the availability bit that gates the load, worked out once
 * per wavefront.  y: lanes 0..20 = corner, 16 above, 4 above-right; lanes 32..47 = the column to the left.  c: lanes 0..17 =
 * corner + 8 above of both planes; lanes 32..47 = the columns to the left.  (intra_issue used to derive them per macroblock
 * with a dozen selects per lane class: issuing the loads was 1.4 of the 12.7 k cycles an intra macroblock takes.) */
struct IntraLaneOffs { int y_off, c_off; uint32_t y_bit, c_bit; int y_at, c_at; };    /* y_at / c_at: where the fetched sample goes in the LDS tiles (-1: nowhere) */
__device__ __forceinline__ IntraLaneOffs intra_lane_offs(int wmb, int lane)
{
    IntraLaneOffs o;
    const int up = -wmb * TILE;
    o.y_off = 0; o.c_off = 0; o.y_bit = 0u; o.c_bit = 0u;
    o.y_at = lane < 21 ? 3 + lane : (lane >= 32 && lane < 48) ? (lane - 32 + 1) * TS + 3 : -1;
    o.c_at = lane < 18 ? (lane / 9) * 144 + lane % 9 : (lane >= 32 && lane < 48) ? ((lane - 32) >> 3) * 144 + (((lane - 32) & 7) + 1) * 16 : -1;
    if (lane < 21) {
        const int c = lane;
        o.y_bit = c == 0 ? FJ_AVAIL_D : c <= 16 ? FJ_AVAIL_B : FJ_AVAIL_C;
        o.y_off = c == 0 ? up - TILE + 255 : c <= 16 ? up + 240 + (c - 1) : up + TILE + 240 + (c - 17);
    } else if (lane >= 32 && lane < 48) {
        o.y_bit = FJ_AVAIL_A;
        o.y_off = -TILE + (lane - 32) * 16 + 15;
    }
    if (lane < 18) {
        const int plane = lane / 9, c = lane % 9;
        o.c_bit = c == 0 ? FJ_AVAIL_D : FJ_AVAIL_B;
        o.c_off = T_CB + plane * 64 + (c == 0 ? up - TILE + 63 : up + 56 + (c - 1));
    } else if (lane >= 32 && lane < 48) {
        const int plane = (lane - 32) >> 3, r = (lane - 32) & 7;
        o.c_bit = FJ_AVAIL_A;
        o.c_off = T_CB + plane * 64 - TILE + r * 8 + 7;
    }
    return o;
}

__device__ __forceinline__ void intra_issue(const FrameDesc &fd, uint32_t mb, const uint32_t *rec_lds, int lane, IntraLoads &L, const IntraLaneOffs &lo, bool cross = false)
{
    /* only the head of the record (kind, availability) and its coefficient fields are needed here */
    const uint32_t head = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec_lds[0]);
    const uint32_t coded = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec_lds[2]), coef_idx = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec_lds[3]);
    const uint32_t kind = head & 255u, avail = head >> 24;
    L.nb_y = L.nb_c = 128;
    L.rows.y = L.rows.c = L.rows.cdc = make_int2(0, 0);
    L.rows.ldc = 0;
    if (kind == FJ_MB_IPCM || kind == FJ_MB_CONCEAL_I) return;
    const uint8_t *Y = fd.cur + (size_t)mb * TILE;
    if (avail & lo.y_bit) L.nb_y = cross && lane < 21 ? (int)ld_agent_u8(Y + lo.y_off) : (int)Y[lo.y_off];
    if (avail & lo.c_bit) L.nb_c = cross && lane < 18 ? (int)ld_agent_u8(Y + lo.c_off) : (int)Y[lo.c_off];
    L.rows = mb_residual_fetch(coded, fd.coefs + 16 * (size_t)coef_idx, lane);
}

__device__ __forceinline__ void intra_mb(const FrameDesc &fd, uint32_t mb, int lane, uint8_t *tile, uint8_t *ctile0,
                                         const uint2 *i4tab, const uint32_t *rec_lds, const IntraLoads &L, const IntraLaneOffs &lo, bool wt,
                                         int16_t *res_defer = nullptr, unsigned long long *tp = nullptr)
{
#define ITICK() (tp ? __builtin_readcyclecounter() : 0ull)
    const unsigned long long i0 = ITICK();
    const FjMbRec rec = rec_from_lds(rec_lds);
    const int16_t *coef = fd.coefs + 16 * (size_t)rec.coef_idx;
    /* the macroblock's tile (Y 16x16 | Cb 8x8 | Cr 8x8) */
    uint8_t *Y = fd.cur + (size_t)mb * TILE;
    const int blk = lane >> 2, row = lane & 3, bx = blk & 3, by = blk >> 2;

    if (rec.kind == FJ_MB_IPCM) {
        /* the 384 raw samples arrive in tile order (Y raster, Cb, Cr: macroblock_layer.c:992-1022) */
        const uint8_t *s = reinterpret_cast<const uint8_t *>(coef);
        put4(Y + 4 * lane, *reinterpret_cast<const uint32_t *>(s + 4 * lane), wt);
        if (lane < 32) put4(Y + 256 + 4 * lane, *reinterpret_cast<const uint32_t *>(s + 256 + 4 * lane), wt);
        return;
    }

    const bool av_a = rec.avail & FJ_AVAIL_A, av_b = rec.avail & FJ_AVAIL_B, av_c = rec.avail & FJ_AVAIL_C;
    /* where the prefetched neighbour samples (intra_issue) go in the tiles */
    const int nb_y_at = lo.y_at, nb_c_at = lo.c_at;
    const int nb_y = L.nb_y, nb_c = L.nb_c;

    int ry[4], rc[4];
    report_residual_range(fd, mb_residual_compute(rec.coded, rec.qp_y, rec.qp_c, rec.kind == FJ_MB_I16x16, coef, lane, L.rows, ry, rc), lane);

    if (nb_y_at >= 0) tile[nb_y_at] = (uint8_t)nb_y;
    if (nb_c_at >= 0) ctile0[nb_c_at] = (uint8_t)nb_c;
    wave_sync();
    const unsigned long long i1 = ITICK();

    if (rec.kind == FJ_MB_I16x16) {
        const int mode = rec.pred & 3;
        const int y = by * 4 + row, x0 = bx * 4;
        const uint8_t *top = tile + 4, *left = tile + TS + 3;   /* top[x], left[y * TS]; corner = tile[3] */
        int pr[4];
        if (mode == 0) {
#pragma unroll
            for (int i = 0; i < 4; i++) pr[i] = top[x0 + i];
        } else if (mode == 1) {
            pr[0] = pr[1] = pr[2] = pr[3] = left[y * TS];
        } else if (mode == 2) {
            /* the sixteen samples above as four dwords summed by byte dot products; of the sixteen to the left every lane of a
             * 16-lane row reads ONE and the row adds them up (four rotating DPP adds): 14 instructions where 32 byte reads and
             * 32 adds per lane used to produce the same number in all 64 lanes */
            const uint32_t *tw = reinterpret_cast<const uint32_t *>(top);
            uint32_t stu = __builtin_amdgcn_udot4(tw[0], 0x01010101u, 0u, false);
            stu = __builtin_amdgcn_udot4(tw[1], 0x01010101u, stu, false);
            stu = __builtin_amdgcn_udot4(tw[2], 0x01010101u, stu, false);
            stu = __builtin_amdgcn_udot4(tw[3], 0x01010101u, stu, false);
            int sl = (int)left[(lane & 15) * TS];
            sl += __builtin_amdgcn_update_dpp(0, sl, 0x128, 0xF, 0xF, false);      /* row_ror:8 */
            sl += __builtin_amdgcn_update_dpp(0, sl, 0x124, 0xF, 0xF, false);      /* row_ror:4 */
            sl += __builtin_amdgcn_update_dpp(0, sl, 0x122, 0xF, 0xF, false);      /* row_ror:2 */
            sl += __builtin_amdgcn_update_dpp(0, sl, 0x121, 0xF, 0xF, false);      /* row_ror:1 */
            const int st = (int)stu;
            const int dc = (av_a && av_b) ? (st + sl + 16) >> 5 : av_a ? (sl + 8) >> 4 : av_b ? (st + 8) >> 4 : 128;
            pr[0] = pr[1] = pr[2] = pr[3] = dc;
        } else {
            int Hh = 0, Vv = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                Hh += (k + 1) * ((int)top[8 + k] - (int)(k == 7 ? tile[3] : top[6 - k]));
                Vv += (k + 1) * ((int)left[(8 + k) * TS] - (int)(k == 7 ? tile[3] : left[(6 - k) * TS]));
            }
            const int a = 16 * ((int)left[15 * TS] + (int)top[15]), b = (5 * Hh + 32) >> 6, c = (5 * Vv + 32) >> 6;
#pragma unroll
            for (int i = 0; i < 4; i++) pr[i] = clip255((a + b * (x0 + i - 7) + c * (y - 7) + 16) >> 5);
        }
        put4(Y + y * 16 + x0, pack4(clip255(pr[0] + ry[0]), clip255(pr[1] + ry[1]), clip255(pr[2] + ry[2]), clip255(pr[3] + ry[3])), wt);
    } else {
        /* Intra4x4.  Block (bx,by) needs the blocks left, above, above-left and above-right of it, so the
         * blocks with bx + 2*by == d are independent: 10 steps instead of 16, two blocks (8 lanes) at a time.
         * (The above-right AVAILABILITY stays the decoding-order rule of 8.3.1.2: it does not depend on when
         * we compute.)  The 4 lanes that own a block's rows do the work; results go to the LDS tile and are
         * written to the picture once at the end. */
        uint64_t i4modes;
        __builtin_memcpy(&i4modes, rec.i4mode, 8);
        if (res_defer) {
            /* joint pass later: residual rows and the per-macroblock facts go to LDS */
            *reinterpret_cast<uint2 *>(res_defer + (by * 4 + row) * 16 + bx * 4) =
                make_uint2((uint32_t)(ry[0] & 0xFFFF) | ((uint32_t)ry[1] << 16), (uint32_t)(ry[2] & 0xFFFF) | ((uint32_t)ry[3] << 16));
            if (lane == 0) { tile[0] = rec.avail; *reinterpret_cast<uint2 *>(&tile[24]) = make_uint2((uint32_t)i4modes, (uint32_t)(i4modes >> 32)); }
        } else {
        const int z = z_of(bx, by);
        const int mode = (int)((i4modes >> (4 * z)) & 15u);
        const int bx4 = bx * 4, by4 = by * 4, y = row;
        const bool has_left = bx > 0 || av_a, has_top = by > 0 || av_b;
        bool has_tr;
        if (by == 0) has_tr = bx < 3 ? av_b : av_c;
        else has_tr = bx < 3 && z_of(bx + 1, by - 1) < z;
        const I4Row ent = intra4_entries(i4tab, mode, y);          /* the lane's table row: once, not inside the ten steps */
        for (int d = 0; d < 10; d++) {
            if (bx + 2 * by == d) {
                int vv[4];
                intra4_row(tile, bx4, by4, mode, has_left, has_top, has_tr, ent, vv);
                int pr[4];
#pragma unroll
                for (int x = 0; x < 4; x++) pr[x] = clip255(vv[x] + ry[x]);
                /* the block's own samples are not inputs of its own prediction: writing is safe */
                *reinterpret_cast<uint32_t *>(&tile[(by4 + 1 + y) * TS + 4 + bx4]) = pack4(pr[0], pr[1], pr[2], pr[3]);
            }
            wave_sync();
        }
        put4(Y + (by * 4 + row) * 16 + bx * 4, *reinterpret_cast<const uint32_t *>(&tile[(by * 4 + 1 + row) * TS + 4 + bx * 4]), wt);
        }
    }

    const unsigned long long i2 = ITICK();
    /* chroma: lanes 0..31, lane = 4*k + row */
    if (lane < 32) {
        const int k = lane >> 2, plane = k >> 2, cbx = k & 1, cby = (k >> 1) & 1;
        const int y = cby * 4 + row, x0 = cbx * 4;
        const uint8_t *t = ctile0 + plane * 144;
        const int mode = (rec.pred >> 2) & 3;
        int pr[4];
        if (mode == 0) {
            int st = 0, sl = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) { st += t[1 + x0 + i]; sl += t[(1 + cby * 4 + i) * 16]; }
            int dc = 128;
            const int kk = cby * 2 + cbx;
            if (kk == 0 || kk == 3) {
                if (av_a && av_b) dc = (st + sl + 4) >> 3; else if (av_b) dc = (st + 2) >> 2; else if (av_a) dc = (sl + 2) >> 2;
            } else if (kk == 1) {
                if (av_b) dc = (st + 2) >> 2; else if (av_a) dc = (sl + 2) >> 2;
            } else {
                if (av_a) dc = (sl + 2) >> 2; else if (av_b) dc = (st + 2) >> 2;
            }
            pr[0] = pr[1] = pr[2] = pr[3] = dc;
        } else if (mode == 1) {
            pr[0] = pr[1] = pr[2] = pr[3] = t[(y + 1) * 16];
        } else if (mode == 2) {
#pragma unroll
            for (int i = 0; i < 4; i++) pr[i] = t[1 + x0 + i];
        } else {
            int Hh = 0, Vv = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                Hh += (i + 1) * ((int)t[1 + 4 + i] - (int)t[1 + 2 - i]);
                Vv += (i + 1) * ((int)t[(1 + 4 + i) * 16] - (int)t[(1 + 2 - i) * 16]);
            }
            const int a = 16 * ((int)t[8 * 16] + (int)t[8]), b = (34 * Hh + 32) >> 6, c = (34 * Vv + 32) >> 6;
#pragma unroll
            for (int i = 0; i < 4; i++) pr[i] = clip255((a + b * (x0 + i - 3) + c * (y - 3) + 16) >> 5);
        }
        put4(Y + T_CB + plane * 64 + y * 8 + x0, pack4(clip255(pr[0] + rc[0]), clip255(pr[1] + rc[1]), clip255(pr[2] + rc[2]), clip255(pr[3] + rc[3])), wt);
    }
    wave_sync();          /* the tiles are reused by this wave's next macroblock */
    if (tp && lane == 0) { const unsigned long long i3 = ITICK(); tp[5] += i1 - i0; tp[6] += i2 - i1; tp[7] += i3 - i2; }
#undef ITICK
}
#undef I4_T
#undef I4_L

/* Joint luma pass of up to FOUR prepared Intra4x4 macroblocks by one wavefront: 16 lanes per macroblock (group g =
 * lane >> 4, tile and residual of slot g).  Inside a macroblock the blocks with bx + 2*by == d are independent (10 steps
 * for 16 blocks), at most two per step: lanes 4a + y (a = 0, 1; y = row) of the group predict row y of the a-th of them —
 * 8 of 16 lanes busy, 32 of 64 with four macroblocks, against 8 of 64 when a wavefront walks one macroblock alone.  The
 * prediction is table-driven (intra4_row), so four macroblocks' worth of different modes cost one instruction stream.
 * my_mb < 0: the group has no macroblock.  Afterwards lane s of a group stores row s of the finished macroblock. */
constexpr int INTRA_SLOT = 1024;                     /* LDS per prepared macroblock: luma tile 17 x TS + chroma tiles 2 x 144 */
constexpr int INTRA_WAVE_LDS = 4 * INTRA_SLOT + 4 * 512 + 128;   /* four slots + four residual blocks of 16 x 16 int16 + four records */
__device__ __forceinline__ void intra4_joint(const FrameDesc &fd, int my_mb, int lane, uint8_t *wave_lds, const uint2 *i4tab, bool wt)
{
    const int g = lane >> 4, sub = lane & 15, a = sub >> 2, y = sub & 3;
    uint8_t *tile = wave_lds + g * INTRA_SLOT;
    const int16_t *res = reinterpret_cast<const int16_t *>(wave_lds + 4 * INTRA_SLOT + g * 512);
    const bool on = my_mb >= 0;
    const uint32_t avail = on ? tile[0] : 0u;
    const uint2 mw = on ? *reinterpret_cast<const uint2 *>(&tile[24]) : make_uint2(0u, 0u);
    const unsigned long long i4modes = (unsigned long long)mw.x | ((unsigned long long)mw.y << 32);
    const bool av_a = avail & FJ_AVAIL_A, av_b = avail & FJ_AVAIL_B, av_c = avail & FJ_AVAIL_C;
    /* what a lane does in step d — which block, its mode, its table row and its residual row — depends on nothing the steps produce:
     * it is worked out, and its two LDS reads are issued, one step AHEAD, so that a step's own chain is neighbour reads -> 20
     * instructions -> one LDS write */
    /* (plain scalars, no struct: the compiler keeps a struct with bool members in scratch memory) */
    int c_bx, c_by, c_mode, c_flags;                          /* flags: 1 active, 2 has_left, 4 has_top, 8 has_tr */
    I4Row c_ent; uint2 c_rr;
    auto setup = [&](int d, int &o_bx, int &o_by, int &o_mode, int &o_flags, I4Row &o_ent, uint2 &o_rr) {
        const int by = min(3, d >> 1) - a, bx = d - 2 * by;
        const bool act = on && a < 2 && by >= 0 && bx >= 0 && bx <= 3;
        o_bx = act ? bx : 0; o_by = act ? by : 0;
        const int z = z_of(o_bx, o_by);
        o_mode = act ? (int)((i4modes >> (4 * z)) & 15u) : 0;
        const bool has_left = o_bx > 0 || av_a, has_top = o_by > 0 || av_b;
        const bool has_tr = o_by == 0 ? (o_bx < 3 ? av_b : av_c) : (o_bx < 3 && z_of(o_bx + 1, o_by - 1) < z);
        o_flags = (act ? 1 : 0) | (has_left ? 2 : 0) | (has_top ? 4 : 0) | (has_tr ? 8 : 0);
        o_ent = intra4_entries(i4tab, o_mode, y);
        o_rr = *reinterpret_cast<const uint2 *>(res + (o_by * 4 + y) * 16 + o_bx * 4);
    };
    setup(0, c_bx, c_by, c_mode, c_flags, c_ent, c_rr);
    for (int d = 0; d < 10; d++) {
        int n_bx = 0, n_by = 0, n_mode = 0, n_flags = 0;
        I4Row n_ent = c_ent; uint2 n_rr = c_rr;
        if (d < 9) setup(d + 1, n_bx, n_by, n_mode, n_flags, n_ent, n_rr);
        if (__ballot(c_flags & 1) != 0ull) {
            int vv[4];
            intra4_row(tile, c_bx * 4, c_by * 4, c_mode, (c_flags & 2) != 0, (c_flags & 4) != 0, (c_flags & 8) != 0, c_ent, vv);
            if (c_flags & 1) {
                const uint2 rr = c_rr;
                const int r0 = (int16_t)(rr.x & 0xFFFFu), r1 = (int32_t)rr.x >> 16, r2 = (int16_t)(rr.y & 0xFFFFu), r3 = (int32_t)rr.y >> 16;
                *reinterpret_cast<uint32_t *>(&tile[(c_by * 4 + 1 + y) * TS + 4 + c_bx * 4]) =
                    pack4(clip255(vv[0] + r0), clip255(vv[1] + r1), clip255(vv[2] + r2), clip255(vv[3] + r3));
            }
        }
        wave_sync();
        c_bx = n_bx; c_by = n_by; c_mode = n_mode; c_flags = n_flags; c_ent = n_ent; c_rr = n_rr;
    }
    if (on) {
        const uint32_t *src = reinterpret_cast<const uint32_t *>(&tile[(sub + 1) * TS + 4]);
        put16(fd.cur + (size_t)my_mb * TILE + sub * 16, make_uint4(src[0], src[1], src[2], src[3]), wt);
    }
    wave_sync();
}

/* Intra (and concealed) macroblocks of one picture, dataflow-scheduled.  A macroblock of the
 * intra schedule waits for those of the neighbours named by its FJ_NEED_* mask that are themselves in the schedule
 * (inter macroblocks were reconstructed by the earlier kernels).  LDS: dep[mb] = outstanding predecessors (0xFF = not
 * scheduled), need[mb] = the mask, a ready queue with claim / publish cursors.  A free wavefront takes up to four ready
 * macroblocks, reconstructs them (intra_mb / conceal_mb), waits for its stores and then releases the neighbours that
 * wait for them.  No level barriers: the picture's time is its dependency critical path, not levels x slowest wave.
 *
 * ROW BANDS as in k_frame_dbk (which see): up to max_bands workgroups per picture, band-local state for rows r0-1 .. r1-1.
 * Intra prediction only looks up and to the left, so the only dependencies that cross a band boundary are those of a band's
 * first row on the last row of the band above (FJ_NEED_UL / U / UR): the producers write their tiles write-through and set
 * a "done" byte (scratch_done(fd, SCRATCH_DONE_INTRA)), an idle wavefront of the band below polls, the consumers read the row above past
 * the L1 (intra_issue, cross).  Pictures with concealed macroblocks (which may wait for the macroblock BELOW them) are never
 * split (FjHeader.intra_down_deps -> FrameDesc.intra_bands = 1).
 * Dynamic LDS: per wavefront INTRA_WAVE_LDS (4 macroblock slots + deferred residuals + records) | need | dep |
 * queue u16 | counters | seen bits | Intra4x4 table (intra_lds_bytes). */
__host__ __device__ inline size_t intra_lds_bytes(uint32_t waves, uint32_t wmb, uint32_t band_rows)
{
    const size_t n_loc16 = (((size_t)band_rows + 1) * wmb + 15) & ~(size_t)15, nq8 = ((size_t)band_rows * wmb + 7) & ~(size_t)7;
    return (size_t)waves * INTRA_WAVE_LDS + 2 * n_loc16 + 2 * nq8 + 32 + 4 * ((((size_t)wmb + 31) / 32 + 3) & ~(size_t)3) + I4TAB_BYTES;
}
/* BANDED = false: the launch gives every picture one workgroup (max_bands == 1, blockIdx.x = picture): no tickets, no
 * hand-over code in the loop. */
#ifndef INTRA_OCC
#define INTRA_OCC 3
#endif
template <bool BANDED>
__global__ __launch_bounds__(64 * TAIL_WAVES, INTRA_OCC) void k_frame_intra(const FrameDesc *__restrict__ frames, unsigned long long *prof,
                                                                 uint32_t *tickets, uint32_t max_bands, uint32_t rows_cap, uint32_t light_cap)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ uint32_t s_misc[4];
    /* these wavefronts walk dependency chains: whatever shares their SIMDs (k_dbk of the next tick, other lanes' list
     * kernels) takes the issue slots they leave, not the ones they need */
    __builtin_amdgcn_s_setprio(3);
    const uint32_t ticket = BANDED ? take_ticket(tickets, &s_misc[0]) : blockIdx.x;
    const uint32_t pic = BANDED ? ticket / max_bands : ticket, band = BANDED ? ticket - pic * max_bands : 0u;
    const FrameDesc &fd = FD_REF(frames, pic);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wmb = fd.wmb, hmb = fd.hmb;
    int R = hmb, nb = 1;
    if (BANDED) band_split(hmb, fd.intra_bands, fd.heavy, max_bands, light_cap, rows_cap, R, nb);
    if (!fd.n_levels || (int)band >= nb) { if (BANDED) return_ticket(tickets); return; }
    const uint32_t total_all = fd.lvl[fd.n_levels];
    const int r0 = (int)band * R, r1 = min(hmb, r0 + R);
    const int base = (r0 - 1) * wmb;                        /* band-local index of macroblock mb: mb - base (row r0-1 first) */
    const int lo = r0 * wmb, hi = r1 * wmb;                 /* the band's own macroblocks */
    const int n_loc = (R + 1) * wmb, n_loc16 = (n_loc + 15) & ~15, nq8 = (R * wmb + 7) & ~7;
    const bool has_up = BANDED && band > 0, has_down = BANDED && r1 < hmb;
    uint8_t *my = lds + wave * INTRA_WAVE_LDS;
    uint8_t *need = lds + (blockDim.x >> 6) * INTRA_WAVE_LDS;
    uint8_t *dep = need + n_loc16;
    uint16_t *queue = reinterpret_cast<uint16_t *>(dep + n_loc16);
    uint32_t *ctr = reinterpret_cast<uint32_t *>(queue + nq8);   /* [0] head, [1] tail, [2] total, [3] producers awaited, [4] producers seen, [5] poll lock */
    uint32_t *seen = ctr + 8;
    uint2 *i4tab = reinterpret_cast<uint2 *>(seen + ((((wmb + 31) >> 5) + 3) & ~3));
    uint8_t *done_g = scratch_done(fd, SCRATCH_DONE_INTRA);

    for (int i = tid; i < n_loc16 / 4; i += blockDim.x) reinterpret_cast<uint32_t *>(dep)[i] = 0xFFFFFFFFu;
    for (int i = tid; i < nq8 / 2; i += blockDim.x) reinterpret_cast<uint32_t *>(queue)[i] = 0xFFFFFFFFu;
    if (tid < 8) ctr[tid] = 0;
    for (int i = tid; i < (wmb + 31) >> 5; i += blockDim.x) seen[i] = 0;
    for (int i = tid; i < 144; i += blockDim.x) i4tab[i] = c_i4tab[i >> 2][i & 3];      /* (huge pictures run with as few as 1 wavefront) */
    __syncthreads();
    for (uint32_t i = tid; i < total_all; i += blockDim.x) {
        const int mb = fd.idx[i];
        if (mb < base || mb >= hi) continue;               /* (base < 0 for band 0: every mb >= 0 passes) */
        need[mb - base] = fd.recs[mb].ref_slot[0];
        dep[mb - base] = 0xFE;                              /* scheduled, count pending */
    }
    __syncthreads();
    /* neighbour b of (x,y): b = 0 L, 1 UL, 2 U, 3 UR, 4 R, 5 DR, 6 D, 7 DL  (b ^ 4 = opposite direction) */
    auto neighbour = [&](int mb, int b) -> int {
        const int y = (int)mb_row(fd, (uint32_t)mb), x = mb - y * wmb;
        const int dx = (b == 2 || b == 6) ? 0 : (b >= 3 && b <= 5) ? 1 : -1;
        const int dy = (b >= 1 && b <= 3) ? -1 : (b >= 5) ? 1 : 0;
        const int nx = x + dx, ny = y + dy;
        return (nx < 0 || ny < 0 || nx >= wmb || ny >= hmb) ? -1 : ny * wmb + nx;
    };
    for (uint32_t i = tid; i < total_all; i += blockDim.x) {
        const int mb = fd.idx[i];
        if (mb < lo || mb >= hi) continue;
        const uint32_t nd = need[mb - base];
        int cnt = 0;
#pragma unroll
        for (int b = 0; b < 8; b++)
            if ((nd >> b) & 1u) {
                const int s = neighbour(mb, b);
                /* (a neighbour below the band can only be named by a concealed macroblock, and those pictures have one band) */
                if (s >= 0 && s >= base && s < hi && dep[s - base] != 0xFF) cnt++;
            }
        dep[mb - base] = (uint8_t)cnt;                     /* byte store: other threads only test != 0xFF */
        atomicAdd(&ctr[2], 1u);
        if (cnt == 0) queue[atomicAdd(&ctr[1], 1u)] = (uint16_t)mb;
    }
    if (has_up)
        for (int x = tid; x < wmb; x += blockDim.x)
            if (dep[x] != 0xFF) atomicAdd(&ctr[3], 1u);
    __syncthreads();
    const uint32_t total = ctr[2], n_await = ctr[3];

    auto release = [&](int li) {
        uint32_t *w = reinterpret_cast<uint32_t *>(dep + (li & ~3));
        const uint32_t sh = 8u * (li & 3);
        const uint32_t old = atomicSub(w, 1u << sh);
        if (((old >> sh) & 255u) == 1u) queue[atomicAdd(&ctr[1], 1u)] = (uint16_t)(li + base);
    };

    volatile H264K_LDS uint16_t *vq = (volatile H264K_LDS uint16_t *)queue;      /* (a generic volatile pointer would read LDS through flat_load) */
    volatile H264K_LDS uint32_t *vctr = (volatile H264K_LDS uint32_t *)ctr;
    uint32_t spins = 0;                  /* safety net: a scheduling bug must end in a reported error (DEVERR_*), never in a hung GPU */
    /* debug accounting (h264bsdmiDebugTailProfile, second half of the buffer): band 0 of picture 0, per wavefront:
     * [0] cycles with nothing ready, [1] cycles reconstructing, [2] cycles waiting for stores + release, [3] MBs */
#ifdef H264K_TAIL_PROFILE
    unsigned long long *tp = (prof && ticket == 0) ? prof + 256 + wave * 8 : nullptr;
#else
    unsigned long long *const tp = nullptr;        /* (the cycle accounting costs registers in a loop that has none to spare: -DH264K_TAIL_PROFILE builds it, tools/prof_tail.py) */
    (void)prof;
#endif
    const IntraLaneOffs lane_offs = intra_lane_offs(wmb, lane);
    unsigned long long t_idle = 0, t_work = 0, t_rel = 0, t_rec = 0, n_done = 0, t_mark = tp ? __builtin_readcyclecounter() : 0ull;
    /* Pull model: a free wavefront takes up to FOUR ready macroblocks at once.  Each is prepared by the whole wavefront
     * in turn (neighbours, residual, chroma; Intra16x16 / I_PCM / concealed macroblocks completely); the luma of the
     * Intra4x4 ones among them — 10 dependent steps with at most two blocks each — is then predicted jointly, one quarter
     * of the wavefront per macroblock (intra4_joint).  A lone ready macroblock takes the single-macroblock path. */
    for (;;) {
        uint32_t cbase = 0, k = 0;
        if (lane == 0) {
            const uint32_t h = vctr[0], t = vctr[1];
            if (t > h) {
                /* several at once only when there is more ready work than wavefronts: with few ready macroblocks (P
                 * pictures) one per wavefront finishes them sooner than one wavefront preparing four in turn */
                const uint32_t share = (t - h) / (blockDim.x >> 6);
                k = share < 1u ? 1u : share > 4u ? 4u : share;
                if (atomicCAS(&ctr[0], h, h + k) != h) k = 0;       /* lost the race: look again */
                cbase = h;
            } else if (h >= total) k = 0xFFFFFFFFu;                /* everything has been claimed */
        }
        cbase = __shfl(cbase, 0); k = __shfl(k, 0);
        if (k == 0xFFFFFFFFu) break;
        if (++spins > (1u << 24)) { if (lane == 0) report_device_error(fd, DEVERR_INTRA_SCHED); break; }
        if (k == 0) {
            /* nothing ready: have macroblocks of the band above, which the first row waits for, finished? (k_frame_dbk) */
            bool polled = false;
            if (has_up && vctr[4] < n_await) {
                uint32_t got = 0;
                if (lane == 0) got = atomicCAS(&ctr[5], 0u, 1u) == 0u;
                got = __shfl(got, 0);
                if (got) {
                    polled = true;
                    for (int x = lane; x < wmb; x += 64) {
                        const uint32_t bit = 1u << (x & 31);
                        if (dep[x] == 0xFF || (seen[x >> 5] & bit)) continue;
                        if (!ld_agent_u8(done_g + base + x)) continue;
                        if (atomicOr(&seen[x >> 5], bit) & bit) continue;
                        atomicAdd(&ctr[4], 1u);
                        /* (x, r0-1) is the UR / U / UL neighbour of (x-1, r0) / (x, r0) / (x+1, r0) */
#pragma unroll
                        for (int d = -1; d <= 1; d++) {
                            const int cx = x + d;
                            if (cx < 0 || cx >= wmb) continue;
                            const int li = wmb + cx;
                            const uint32_t wants = d < 0 ? FJ_NEED_UR : d == 0 ? FJ_NEED_U : FJ_NEED_UL;
                            if (li < n_loc && r0 < r1 && dep[li] != 0xFF && (need[li] & wants)) release(li);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) atomicExch(&ctr[5], 0u);
                }
            }
            if (polled) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(1);
            continue;
        }
        /* lane j < k fetches queue slot base + j (the publisher bumps the cursor, then writes the slot) */
        int v = 0;
        if ((uint32_t)lane < k) do { v = vq[cbase + lane]; } while (v == 0xFFFF);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (tp) { const unsigned long long t = __builtin_readcyclecounter(); t_idle += t - t_mark; t_mark = t; }
        int joint_mb = -1;                                          /* per 16-lane group: its Intra4x4 macroblock, if any */
        /* the records of all claimed macroblocks in ONE vector load (lane 8j + w: dword w of record j), parked in LDS:
         * one memory round trip per group instead of one per macroblock in front of the neighbour / coefficient loads */
        uint32_t *rec_lds = reinterpret_cast<uint32_t *>(my + 4 * INTRA_SLOT + 4 * 512);
        if ((uint32_t)lane < 8u * k) {
            const int mbj = __shfl(v, lane >> 3);
            rec_lds[lane] = reinterpret_cast<const uint32_t *>(&fd.recs[mbj])[lane & 7];
        }
        wave_sync();
        if (tp) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t = __builtin_readcyclecounter(); t_rec += t - t_mark; }
        /* first row of the band: the row above comes from another workgroup; last row: the band below reads this one */
        const int cross_lo = has_up ? lo : -1, cross_hi = has_up ? lo + wmb : -1, wt_lo = has_down ? hi - wmb : 0x7FFFFFFF;
        /* software pipeline over the group: the loads of macroblock j + 1 are in flight while macroblock j is reconstructed */
        IntraLoads cur_loads, next_loads;
        {
            const int mb0 = __builtin_amdgcn_readfirstlane(__shfl(v, 0));
            intra_issue(fd, (uint32_t)mb0, rec_lds, lane, cur_loads, lane_offs, BANDED && mb0 >= cross_lo && mb0 < cross_hi);
        }
        for (uint32_t j = 0; j < k; j++) {
            const uint32_t mb = (uint32_t)__builtin_amdgcn_readfirstlane(__shfl(v, (int)j));
            const uint32_t head = (uint32_t)__builtin_amdgcn_readfirstlane((int)rec_lds[8 * j]);     /* kind, qp_y, qp_c, avail */
            const uint32_t kind = head & 255u;
            const bool wt = BANDED && (int)mb >= wt_lo;
            uint8_t *slot = my + j * INTRA_SLOT;
            if (j + 1 < k) {
                const int mbn = __builtin_amdgcn_readfirstlane(__shfl(v, (int)j + 1));
                intra_issue(fd, (uint32_t)mbn, rec_lds + 8 * (j + 1), lane, next_loads, lane_offs, BANDED && mbn >= cross_lo && mbn < cross_hi);
            }
            /* lost macroblocks (error path) are a call, so that they cost the intra path no registers */
            if (kind == FJ_MB_CONCEAL_I) conceal_mb(fd, mb, lane, head >> 24);
            else if (kind == FJ_MB_I4x4 && k > 1) {
                intra_mb(fd, mb, lane, slot, slot + 17 * TS, i4tab, rec_lds + 8 * j, cur_loads, lane_offs, wt, reinterpret_cast<int16_t *>(my + 4 * INTRA_SLOT + j * 512), tp);
                if ((uint32_t)(lane >> 4) == j) joint_mb = (int)mb;
            } else intra_mb(fd, mb, lane, slot, slot + 17 * TS, i4tab, rec_lds + 8 * j, cur_loads, lane_offs, wt, nullptr, tp);
            if (j + 1 < k) cur_loads = next_loads;
        }
        if (__ballot(joint_mb >= 0) != 0ull) intra4_joint(fd, joint_mb, lane, my, i4tab, BANDED && joint_mb >= wt_lo);
        if (tp) { const unsigned long long t = __builtin_readcyclecounter(); t_work += t - t_mark; t_mark = t; n_done += k; }
        /* release: stores done -> the neighbours that wait for these macroblocks (lanes 16j + b: neighbour b of macroblock j) */
        release_stores(BANDED && v >= wt_lo && (uint32_t)lane < k);
        {
            const int j = lane >> 4, b = lane & 15;
            const int mbj = __shfl(v, j);
            if (BANDED && (uint32_t)j < k && b == 8 && mbj >= wt_lo) st_agent_u8(done_g + mbj, 1u);      /* hand-over to the band below */
            if ((uint32_t)j < k && b < 8) {
                const int s = neighbour(mbj, b);
                if (s >= lo && s < hi && dep[s - base] != 0xFF && ((need[s - base] >> (b ^ 4)) & 1u)) release(s - base);
            }
        }
        if (tp) { const unsigned long long t = __builtin_readcyclecounter(); t_rel += t - t_mark; t_mark = t; }
    }
    if (tp && lane == 0) { tp[0] += t_idle; tp[1] += t_work; tp[2] += t_rel; tp[3] += n_done; tp[4] += t_rec; }
    /* the last band of the picture to leave zeroes the done bytes for the next picture of this stream */
    if (BANDED && nb > 1) {
        __syncthreads();
        if (tid == 0) s_misc[1] = atomicAdd(scratch_exits(fd, 1), 1u);
        __syncthreads();
        if (s_misc[1] == (uint32_t)nb - 1u) {
            uint32_t *z = reinterpret_cast<uint32_t *>(done_g);
            for (int i = tid; i < (int)((fd.n_mbs + 3u) >> 2); i += blockDim.x) z[i] = 0;
            if (tid == 0) atomicExch(scratch_exits(fd, 1), 0u);
        }
    }
    if (BANDED) return_ticket(tickets);
}

} // namespace h264k
