/* kernels/k_recon_inter.hip.h — k_recon_inter: every other inter macroblock (interpolation, residual, write-back).  Part of kernels.hip.h (which see); not a stand-alone header. */
#pragma once
namespace h264k {
/* ------------------------------------------------------------------ inter prediction */
__device__ __forceinline__ int tap6(int a, int b, int c, int d, int e, int f) { return a - 5 * (b + e) + 20 * (c + d) + f; }

/* Register window of one lane: rows y-2..y+3, columns x-2..x+9 of the reference plane (9 columns used),
 * rw[r][k] = dword k of window row r.  Filled either straight from global memory (clamp-to-edge on the
 * slow path = h264bsdFillBlock, src/h264bsd_reconstruct.c:2244) or from the wave's LDS-staged window. */
__device__ __forceinline__ void luma_window_global(const uint8_t *__restrict__ p, int wmb, int w, int h, int x, int y, uint32_t rw[6][3])
{
    if (x >= 2 && x + 9 < w && y >= 2 && y + 3 < h) {
#pragma unroll
        for (int r = 0; r < 6; r++) {
            rw[r][0] = luma4_at(p, wmb, x - 2, y - 2 + r); rw[r][1] = luma4_at(p, wmb, x + 2, y - 2 + r); rw[r][2] = luma4_at(p, wmb, x + 6, y - 2 + r);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 6; r++) {
            const int yy = clip3(0, h - 1, y - 2 + r);
            uint32_t a = 0, b = 0, c = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) {
                a |= (uint32_t)p[luma_at(wmb, clip3(0, w - 1, x - 2 + i), yy)] << (8 * i);
                b |= (uint32_t)p[luma_at(wmb, clip3(0, w - 1, x + 2 + i), yy)] << (8 * i);
            }
            c = (uint32_t)p[luma_at(wmb, clip3(0, w - 1, x + 6), yy)];
            rw[r][0] = a; rw[r][1] = b; rw[r][2] = c;
        }
    }
}

/* 4 luma samples (x..x+3, y) of the prediction at quarter-sample fraction (fx,fy) from the window (8.4.2.2.1) */
__device__ __forceinline__ void luma_from_window(const uint32_t rw[6][3], int fx, int fy, int out[4])
{
#define GW(r, c) ((int)((rw[(r)][(c) >> 2] >> (8 * ((c) & 3))) & 255u))
    if ((fx | fy) == 0) {                            /* G: whole-sample */
#pragma unroll
        for (int i = 0; i < 4; i++) out[i] = GW(2, i + 2);
        return;
    }
    /* The one-dimensional and diagonal classes run on PAIRS of output samples (packed 16-bit: a six-tap sum of bytes lies
     * in [-2550, 10710]).  CP(r, c) = (sample c, sample c+1) of window row r; the selectors are compile-time constants. */
#define CP(r, c) as_s2(perm(rw[(r)][((c) + 1) >> 2], rw[(r)][(c) >> 2], \
                      0x0C000C00u | (uint32_t)((c) & 3) | ((uint32_t)(((((c) + 1) >> 2) != ((c) >> 2)) ? 4 + (((c) + 1) & 3) : (((c) + 1) & 3)) << 16)))
#define HT2(r, i) (CP(r, i) + CP(r, (i) + 5) - pk(5) * (CP(r, (i) + 1) + CP(r, (i) + 4)) + pk(20) * (CP(r, (i) + 2) + CP(r, (i) + 3)))
#define VT2(c) (CP(0, c) + CP(5, c) - pk(5) * (CP(1, c) + CP(4, c)) + pk(20) * (CP(2, c) + CP(3, c)))
#define RND5(x) pk_clip(pk(0), pk(255), ((x) + pk(16)) >> pk(5))
    if (fy == 0) {                                   /* a, b, c: horizontal only (window row 2) */
        s2 o01 = RND5(HT2(2, 0)), o23 = RND5(HT2(2, 2));
        if (fx == 1) { o01 = (o01 + CP(2, 2) + pk(1)) >> pk(1); o23 = (o23 + CP(2, 4) + pk(1)) >> pk(1); }
        else if (fx == 3) { o01 = (o01 + CP(2, 3) + pk(1)) >> pk(1); o23 = (o23 + CP(2, 5) + pk(1)) >> pk(1); }
        out[0] = o01.x; out[1] = o01.y; out[2] = o23.x; out[3] = o23.y;
        return;
    }
    if (fx == 0) {                                   /* d, h, n: vertical only */
        s2 o01 = RND5(VT2(2)), o23 = RND5(VT2(4));
        if (fy == 1) { o01 = (o01 + CP(2, 2) + pk(1)) >> pk(1); o23 = (o23 + CP(2, 4) + pk(1)) >> pk(1); }
        else if (fy == 3) { o01 = (o01 + CP(3, 2) + pk(1)) >> pk(1); o23 = (o23 + CP(3, 4) + pk(1)) >> pk(1); }
        out[0] = o01.x; out[1] = o01.y; out[2] = o23.x; out[3] = o23.y;
        return;
    }
    if (fx != 2 && fy != 2) {                        /* e, g, p, r: average of the nearest horizontal and vertical half samples */
        s2 b01, b23, h01, h23;
        if (fy == 1) { b01 = HT2(2, 0); b23 = HT2(2, 2); } else { b01 = HT2(3, 0); b23 = HT2(3, 2); }
        if (fx == 1) { h01 = VT2(2); h23 = VT2(4); } else { h01 = VT2(3); h23 = VT2(5); }
        const s2 o01 = (RND5(b01) + RND5(h01) + pk(1)) >> pk(1), o23 = (RND5(b23) + RND5(h23) + pk(1)) >> pk(1);
        out[0] = o01.x; out[1] = o01.y; out[2] = o23.x; out[3] = o23.y;
        return;
    }
    /* vertical 6-tap sums at window columns 2..6 (sample columns x .. x+4): h and m candidates */
#define VH1(c) tap6(GW(0, c), GW(1, c), GW(2, c), GW(3, c), GW(4, c), GW(5, c))
#define HB1(r, i) tap6(GW(r, i), GW(r, (i) + 1), GW(r, (i) + 2), GW(r, (i) + 3), GW(r, (i) + 4), GW(r, (i) + 5))
    if (fx == 2 || fy == 2) {                        /* j, f, q, i, k */
        /* j is the 6-tap filter over un-rounded intermediate sums, and it may run over the vertical sums of nine columns
         * just as well as over the horizontal sums of six rows (8.4.2.2.1: both orders are equal): 9 + 4 filters
         * instead of 24 + 4, and the vertical sums are exactly what i / k need */
        int v1[9];
#pragma unroll
        for (int c = 0; c < 9; c++) v1[c] = VH1(c);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int j = clip255((tap6(v1[i], v1[i + 1], v1[i + 2], v1[i + 3], v1[i + 4], v1[i + 5]) + 512) >> 10);
            int v = j;
            if (fy != 2) {                           /* f / q: with b (row y) or s (row y+1) */
                const int b = clip255(((fy == 1 ? HB1(2, i) : HB1(3, i)) + 16) >> 5);
                v = (j + b + 1) >> 1;
            } else if (fx != 2) {                    /* i / k: with h (col x) or m (col x+1) */
                const int hh = clip255(((fx == 1 ? v1[i + 2] : v1[i + 3]) + 16) >> 5);
                v = (j + hh + 1) >> 1;
            }
            out[i] = v;
        }
        return;
    }
#undef VH1
#undef HB1
#undef GW
#undef CP
#undef HT2
#undef VT2
#undef RND5
}

/* The same prediction for a lane whose window lies in LDS (k_recon_inter, staged windows): src = window row 0 at the dword that
 * holds window column 0, sh = 8 * (byte of that column in its dword).  The interpolation class is wave-uniform (one motion
 * vector per wavefront or quadrant... per lane in the quadrant path, still few classes per wavefront) and every class reads
 * only the window rows and dwords it uses — whole-sample: one row, two dwords; horizontal: one row; vertical: six rows of two
 * dwords; only the centre classes need all 6 x 3 — instead of 24 LDS dwords and 18 funnel shifts for every macroblock.
 * Result: the four samples as two packed pairs (o01, o23), 0..255 each. */
__device__ __forceinline__ void luma_pred_lds(const uint8_t *src, int stride, int sh, int fx, int fy, s2 &o01, s2 &o23)
{
    uint32_t rw[6][3];
    auto row = [&](int r, int nd) {                  /* dwords 0 .. nd-1 of window row r */
        const uint32_t *q = reinterpret_cast<const uint32_t *>(src + r * stride);   /* 4-byte aligned only */
        uint32_t d[4];
#pragma unroll
        for (int k = 0; k < 4; k++) if (k <= nd) d[k] = q[k];
#pragma unroll
        for (int k = 0; k < 3; k++) if (k < nd) rw[r][k] = (uint32_t)(((unsigned long long)d[k + 1] << 32 | d[k]) >> sh);
    };
#define CP(r, c) as_s2(perm(rw[(r)][((c) + 1) >> 2], rw[(r)][(c) >> 2], \
                      0x0C000C00u | (uint32_t)((c) & 3) | ((uint32_t)(((((c) + 1) >> 2) != ((c) >> 2)) ? 4 + (((c) + 1) & 3) : (((c) + 1) & 3)) << 16)))
#define HT2(r, i) (CP(r, i) + CP(r, (i) + 5) - pk(5) * (CP(r, (i) + 1) + CP(r, (i) + 4)) + pk(20) * (CP(r, (i) + 2) + CP(r, (i) + 3)))
#define VT2(c) (CP(0, c) + CP(5, c) - pk(5) * (CP(1, c) + CP(4, c)) + pk(20) * (CP(2, c) + CP(3, c)))
#define RND5(x) pk_clip(pk(0), pk(255), ((x) + pk(16)) >> pk(5))
#define GW(r, c) ((int)((rw[(r)][(c) >> 2] >> (8 * ((c) & 3))) & 255u))
    if ((fx | fy) == 0) {                            /* G: whole-sample */
        row(2, 2);
        o01 = CP(2, 2); o23 = CP(2, 4);
        return;
    }
    if (fy == 0) {                                   /* a, b, c: horizontal only (window row 2) */
        row(2, 3);
        o01 = RND5(HT2(2, 0)); o23 = RND5(HT2(2, 2));
        if (fx == 1) { o01 = (o01 + CP(2, 2) + pk(1)) >> pk(1); o23 = (o23 + CP(2, 4) + pk(1)) >> pk(1); }
        else if (fx == 3) { o01 = (o01 + CP(2, 3) + pk(1)) >> pk(1); o23 = (o23 + CP(2, 5) + pk(1)) >> pk(1); }
        return;
    }
    if (fx == 0) {                                   /* d, h, n: vertical only (columns 2..5) */
        row(0, 2); row(1, 2); row(2, 2); row(3, 2); row(4, 2); row(5, 2);
        o01 = RND5(VT2(2)); o23 = RND5(VT2(4));
        if (fy == 1) { o01 = (o01 + CP(2, 2) + pk(1)) >> pk(1); o23 = (o23 + CP(2, 4) + pk(1)) >> pk(1); }
        else if (fy == 3) { o01 = (o01 + CP(3, 2) + pk(1)) >> pk(1); o23 = (o23 + CP(3, 4) + pk(1)) >> pk(1); }
        return;
    }
    if (fx != 2 && fy != 2) {                        /* e, g, p, r: average of the nearest horizontal and vertical half samples */
        row(0, 2); row(1, 2); row(4, 2); row(5, 2);
        s2 b01, b23, h01, h23;
        if (fy == 1) { row(2, 3); row(3, 2); b01 = HT2(2, 0); b23 = HT2(2, 2); } else { row(2, 2); row(3, 3); b01 = HT2(3, 0); b23 = HT2(3, 2); }
        if (fx == 1) { h01 = VT2(2); h23 = VT2(4); } else { h01 = VT2(3); h23 = VT2(5); }
        o01 = (RND5(b01) + RND5(h01) + pk(1)) >> pk(1); o23 = (RND5(b23) + RND5(h23) + pk(1)) >> pk(1);
        return;
    }
    /* j, f, q, i, k: the 6-tap filter over un-rounded intermediate sums — over the vertical sums of nine columns (8.4.2.2.1: both
     * orders are equal): 9 + 4 filters instead of 24 + 4, and the vertical sums are exactly what i / k need */
    row(0, 3); row(1, 3); row(2, 3); row(3, 3); row(4, 3); row(5, 3);
#define VH1(c) tap6(GW(0, c), GW(1, c), GW(2, c), GW(3, c), GW(4, c), GW(5, c))
#define HB1(r, i) tap6(GW(r, i), GW(r, (i) + 1), GW(r, (i) + 2), GW(r, (i) + 3), GW(r, (i) + 4), GW(r, (i) + 5))
    int v1[9], out[4];
#pragma unroll
    for (int c = 0; c < 9; c++) v1[c] = VH1(c);
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int j = clip255((tap6(v1[i], v1[i + 1], v1[i + 2], v1[i + 3], v1[i + 4], v1[i + 5]) + 512) >> 10);
        int v = j;
        if (fy != 2) {                               /* f / q: with b (row y) or s (row y+1) */
            const int b = clip255(((fy == 1 ? HB1(2, i) : HB1(3, i)) + 16) >> 5);
            v = (j + b + 1) >> 1;
        } else if (fx != 2) {                        /* i / k: with h (col x) or m (col x+1) */
            const int hh = clip255(((fx == 1 ? v1[i + 2] : v1[i + 3]) + 16) >> 5);
            v = (j + hh + 1) >> 1;
        }
        out[i] = v;
    }
    o01 = as_s2((uint32_t)out[0] | ((uint32_t)out[1] << 16)); o23 = as_s2((uint32_t)out[2] | ((uint32_t)out[3] << 16));
#undef VH1
#undef HB1
#undef GW
#undef CP
#undef HT2
#undef VT2
#undef RND5
}

/* 2 chroma samples from the two rows a[0..2], b[0..2] at eighth-sample fraction (fx,fy), 8.4.2.2.2 */
__device__ __forceinline__ void chroma_from_rows(const int a[3], const int b[3], int fx, int fy, int out[2])
{
    const int w00 = (8 - fx) * (8 - fy), w10 = fx * (8 - fy), w01 = (8 - fx) * fy, w11 = fx * fy;
    out[0] = (w00 * a[0] + w10 * a[1] + w01 * b[0] + w11 * b[1] + 32) >> 6;
    out[1] = (w00 * a[1] + w10 * a[2] + w01 * b[1] + w11 * b[2] + 32) >> 6;
}
/* The same two samples from STAGED rows in packed 16-bit: s0 points at the first of three consecutive bytes of the upper row, s1 at
 * those of the row below (any byte alignment; the dword behind them belongs to the row's stride).  Weights are at most 64 and
 * samples at most 255: every partial sum fits a signed 16-bit half.  Round 6: the chroma prediction of a macroblock used to be
 * 32 lanes x 4 samples in 32-bit arithmetic from ten ds_read_u8 each — about 56 of the 257 vector instructions of a one-vector
 * macroblock; it is 64 lanes x 2 samples now (lanes 32..63 hand their pair to lanes 0..31 with v_permlane32_swap). */
__device__ __forceinline__ s2 chroma_pair_lds(const uint8_t *s0, const uint8_t *s1, int fx, int fy)
{
    auto three = [](const uint8_t *p) -> uint32_t {
        const uint32_t sh = (uint32_t)(uintptr_t)p & 3u;
        const uint32_t *q = reinterpret_cast<const uint32_t *>(p - sh);
        return __builtin_amdgcn_alignbyte(q[1], q[0], sh);
    };
    const uint32_t ra = three(s0), rb = three(s1);
    const s2 P0 = as_s2(perm(0u, ra, 0x0C010C00u)), P1 = as_s2(perm(0u, ra, 0x0C020C01u));
    const s2 Q0 = as_s2(perm(0u, rb, 0x0C010C00u)), Q1 = as_s2(perm(0u, rb, 0x0C020C01u));
    const int w00 = (8 - fx) * (8 - fy), w10 = fx * (8 - fy), w01 = (8 - fx) * fy, w11 = fx * fy;
    return (P0 * pk(w00) + P1 * pk(w10) + Q0 * pk(w01) + Q1 * pk(w11) + pk(32)) >> pk(6);
}
/* lanes 0..31 receive the pair of lane + 32 (theirs is samples 0, 1 of the row, that one samples 2, 3) */
__device__ __forceinline__ s2 pair_from_upper_half(s2 mine)
{
    const uint32_t x = as_u32(mine);
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
    return as_s2(r[1]);
}
/* 2 chroma samples (x, x+1 ; y) of plane `plane` straight from global memory (w, h: chroma plane size) */
__device__ __forceinline__ void chroma_pred2(const uint8_t *__restrict__ f, int wmb, int plane, int w, int h, int x, int y, int fx, int fy, int out[2])
{
    int a[3], b[3];
    const int y0 = clip3(0, h - 1, y), y1 = clip3(0, h - 1, y + 1);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const int xx = clip3(0, w - 1, x + i);
        a[i] = f[chroma_at(wmb, plane, xx, y0)]; b[i] = f[chroma_at(wmb, plane, xx, y1)];
    }
    chroma_from_rows(a, b, fx, fy, out);
}

/* ------------------------------------------------------------------ inter macroblocks */
/* General inter macroblocks, one wavefront each.  When the 16 motion vectors and the four references of
 * the macroblock agree (82 % of the general MBs, everything but sub-partitioned ones) the 21x21 luma and two
 * 9x9 chroma reference windows are staged ONCE in LDS with row-wide coalesced dword loads and every lane
 * cuts its 6x12-byte register window out of LDS; otherwise every lane fetches its own window from global
 * memory.  Both feed the same textbook interpolation (luma_from_window / chroma_from_rows). */
/* 16 / 8 bytes at a 4-byte aligned address (global_load_dwordx4 / dwordx2 need dword alignment only) */
struct __attribute__((packed, aligned(4))) U4a4 { uint32_t x, y, z, w; };
struct __attribute__((packed, aligned(4))) U2a4 { uint32_t x, y; };

/* Reference windows are staged tile row by tile row: a window row is the 16-byte rows of the 2-3 tiles it crosses,
 * loaded whole (aligned 16-byte / 8-byte requests) and laid side by side in LDS, so byte 0 of a staged row is the first
 * column of the window's first tile. */
constexpr int IW_STRIDE = 52;                        /* luma window: 21 rows x 3 tiles x 16 bytes; 13-dword stride: no bank conflicts for row-per-lane reads */
constexpr int IC_STRIDE = 20;                        /* chroma windows: 9 rows x 2 tiles x 8 bytes, two planes */
constexpr int QW_STRIDE = 36;                        /* quadrant luma windows: 13 rows x 2 tiles x 16 bytes, 9-dword stride */
constexpr int QC_STRIDE = 20;                        /* quadrant chroma windows: 5 rows x 2 tiles x 8 bytes, two planes */
constexpr int INTER_WAVE_LDS = 2688;                 /* max(21 * IW_STRIDE + 2 * 9 * IC_STRIDE = 1452, 4 * 13 * QW_STRIDE + 4 * 2 * 5 * QC_STRIDE = 2672), rounded */

#ifndef INTER_OCC
#define INTER_OCC 8      /* macroblock-tile layout: 8 waves per SIMD (64 VGPRs, more spills) beat 7 / 6 / 5: 50.4 vs 54.4 / 58.9 / 59.2 ms per step — the kernel hides latency with wavefronts */
#endif
/* Three instantiations share the list: PATH 0 reconstructs the entries with one motion vector per macroblock (82 % of
 * them in the bundled 1080p stream), PATH 1 those with one per 8x8 quadrant (16x8, 8x16, 8x8 partitions: all the others
 * of that stream), PATH 2 the finer partitions.  Compiled separately, each gets the registers its own path needs — the
 * common cases do not pay (in spills at 8 waves per SIMD) for the per-lane window code of the rare one. */
#ifndef INTER_OCC_PART
#define INTER_OCC_PART 6     /* the partitioned paths: a hint of 6 lets the quadrant path take the 100 scalar registers it wants (60 VGPRs: it still runs
                                8 waves per SIMD); at a hint of 8 it spills 32 scalar registers into vector lanes */
#endif
#ifndef INTER_WG_WAVES
#define INTER_WG_WAVES 1     /* wavefronts (= macroblocks) per workgroup.  The wavefronts of this kernel share nothing, and a workgroup of four
                                needs a free slot on each of the four SIMDs of one CU at the same moment: 1 / 2 / 4 / 8 / 16 wavefronts per
                                workgroup take 34.0 / 35.8 / 38.8 / 42.9 / 48.9 ms per step (the average occupancy, not the instruction
                                count, was what held the kernel back: -10 % instructions had changed nothing) */
#endif
#ifndef INTER_PER_WAVE
#define INTER_PER_WAVE 2     /* list entries per wavefront, the one-vector path: 1 / 2 / 3 / 4 / 6 / 8 -> 34.4 / 32.7 / 33.1 / 33.1 / 33.6 / 34.2 ms per step (both paths' time) */
#endif
#ifndef INTER_PER_WAVE_QUAD
#define INTER_PER_WAVE_QUAD 2   /* ... the quadrant path (69 VGPRs: 7 wavefronts per SIMD instead of 8, and still 0.3 ms better); the finer partitions: always 1 */
#endif
#ifndef INTER_XCD
#define INTER_XCD 1         /* blockIdx -> list position: one XCD takes a contiguous eighth of a picture's list (k_recon_inter, below) */
#endif
template <int PATH> constexpr uint32_t inter_per_wave() { return PATH == 0 ? INTER_PER_WAVE : PATH == 1 ? INTER_PER_WAVE_QUAD : 1; }
#ifdef H264K_INTER_PROFILE
#define IPROF(k) do { if (PATH == 0) ipt[k] = __builtin_readcyclecounter(); } while (0)
#else
#define IPROF(k) do { } while (0)
#endif
template <int PATH>
__global__ __launch_bounds__(64 * INTER_WG_WAVES, PATH == 0 ? INTER_OCC : INTER_OCC_PART) void k_recon_inter(const FrameDesc *__restrict__ frames)
{
#ifdef H264K_INTER_PROFILE
    unsigned long long ipt[6] = { 0, 0, 0, 0, 0, 0 };      /* cycle accounting of one list entry (tools/inter_prof.py): begin | entry here | windows staged | predicted | before the store | end */
#endif
    __shared__ __attribute__((aligned(16))) uint8_t lds[INTER_WG_WAVES * INTER_WAVE_LDS];
    const FrameDesc &fd = FD_REF(frames, blockIdx.y);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   /* wave-uniform: the list entry, the record and
                                                                                 everything derived live in scalar registers */
    /* A wavefront reconstructs INTER_PER_WAVE consecutive list entries, one after the other (nothing of entry i + 1 is requested
     * before entry i is stored: no register is carried from one to the next).  What that amortises is the START of a wavefront:
     * with one macroblock per single-wavefront workgroup the kernel spends a third of its time launching workgroups that do
     * nothing yet (measured with the body cut out behind the first scalar loads: 8 of 22 ms per step), and neighbours in the list
     * are neighbours in the picture — their reference windows overlap, and the second one finds the first one's lines in the L1. */
    const uint32_t g_first = (PATH == 0 ? 0u : PATH == 1 ? fd.n_gen_uni : fd.n_gen_uni + fd.n_gen_quad);
    const uint32_t g_end = (PATH == 0 ? fd.n_gen_uni : PATH == 1 ? fd.n_gen_uni + fd.n_gen_quad : fd.n_gen);
#pragma unroll 1
  for (uint32_t it = 0; it < inter_per_wave<PATH>(); it++) {
    /* Workgroups are dealt to the eight XCDs round robin in dispatch order (x fastest), and every XCD has its own L2: with the plain
     * mapping two neighbouring macroblocks — whose reference windows overlap — never share an L2, and every 128-byte line a window
     * touches is fetched from HBM by up to four XCDs (FETCH 1.31 GB per tick for 0.48 GB of windows and coefficients).  INTER_XCD = 1
     * (default): the workgroups x = c (mod 8) of a picture — one XCD's — take the c-th contiguous eighth of its list, i.e. a band of the
     * picture: FETCH 0.88 GB (-33 %), time +0.3-0.4 ms per step (33.4 vs 33.0; HBM bytes are not what the kernel waits for — the request path
     * is).  INTER_XCD = n > 1: block-cyclic chunks of n workgroups per XCD (32: FETCH -18 %, time unchanged); 0: the plain mapping. */
#if INTER_XCD == 1
    const uint32_t cls8 = blockIdx.x & 7u, per8 = gridDim.x >> 3, rem8 = gridDim.x & 7u;
    const uint32_t bx_ = cls8 * per8 + (cls8 < rem8 ? cls8 : rem8) + (blockIdx.x >> 3);
#elif INTER_XCD > 1
    const uint32_t C_ = INTER_XCD, full_ = (gridDim.x / (8u * C_)) * (8u * C_);
    const uint32_t b_ = blockIdx.x, o_ = b_ % (8u * C_);
    const uint32_t bx_ = b_ >= full_ ? b_ : (b_ - o_) + (o_ & 7u) * C_ + (o_ >> 3);
#else
    const uint32_t bx_ = blockIdx.x;
#endif
    const uint32_t gi = g_first + (bx_ * INTER_WG_WAVES + wave) * inter_per_wave<PATH>() + it;
    if (gi >= g_end) return;
    IPROF(0);                                        /* (the first entry's count begins a few scalar loads into the wavefront's life) */
    /* list entry and record as whole dwords from a wave-uniform address in read-only memory: scalar loads (there is no scalar
     * byte load: a struct copy would fetch the byte-sized members with vector loads and wait for them) */
    FjGen ge;
    {
        const uint4 w = ld16c((const H264K_CONST FjGen *)fd.gen + gi);
        __builtin_memcpy(&ge, &w, 16);
    }
    const uint32_t mb = ge.mb;
#ifdef H264K_INTER_PROFILE
    if (PATH == 0) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
#endif
    IPROF(1);
    /* the QPs travel in the list entry (framejob.h); only the partitioned paths need the record, for their four references */
    const int qp_y = (int)FJ_GEN_QP_Y(ge.coef_idx), qp_c = (int)FJ_GEN_QP_C(ge.coef_idx);
    uint32_t rec_refs = 0u;                          /* FjMbRec.ref_slot[4]: one scalar dword instead of the 32-byte record */
    if (PATH != 0) rec_refs = *(const H264K_CONST uint32_t *)((const H264K_CONST uint8_t *)((const H264K_CONST FjMbRec *)fd.recs + mb) + offsetof(FjMbRec, ref_slot));
    int lane = threadIdx.x & 63;
    if (inter_per_wave<PATH>() > 1) asm volatile("" : "+v"(lane));                   /* (everything a lane derives from its number is worked out again for every entry: hoisted out of
                                                        the loop it would live in registers the kernel does not have at 8 wavefronts per SIMD) */
    uint8_t *lw = lds + wave * INTER_WAVE_LDS, *lc = lw + 21 * IW_STRIDE;
    const int wmb = fd.wmb, W = wmb * 16, H = fd.hmb * 16, CW = W >> 1, CH = H >> 1;
    const int mby = wmb == 1 ? (int)mb : (int)__umulhi(mb, fd.wmb_magic), mbx = (int)mb - mby * wmb;     /* scalar: FrameDesc.wmb_magic */
    /* (address spaces spelled out once: the loads below become global_load / s_load instead of flat_load) */
    /* partitioned macroblocks: the list entry's vector fields hold the index of their sixteen vectors in the sparse section */
    const uint32_t mvx_idx = PATH == 0 ? 0u : (uint32_t)(uint16_t)ge.mvx | ((uint32_t)(uint16_t)ge.mvy << 16);
    const int16_t *mvs = (const int16_t *)((const H264K_CONST int16_t *)fd.mvx + 32 * (size_t)mvx_idx);
    const int16_t *coef = (const int16_t *)((const H264K_CONST int16_t *)fd.coefs + 16 * (size_t)FJ_GEN_COEF_IDX(ge.coef_idx));
    H264K_GLOBAL uint8_t *cur = (H264K_GLOBAL uint8_t *)fd.cur;
    const int blk = lane >> 2, row = lane & 3, bx = blk & 3, by = blk >> 2;
    const bool uniform = PATH == 0, quadwise = PATH == 1;
    uint32_t refs = ge.slot * 0x01010101u, mv_mine = 0;
    const uint32_t mv0 = (uint32_t)(uint16_t)ge.mvx | ((uint32_t)(uint16_t)ge.mvy << 16);
    if (!uniform) {
        refs = rec_refs;
        if (!quadwise) mv_mine = *reinterpret_cast<const uint32_t *>(mvs + 2 * blk);
    }

    /* the coefficient rows are requested right behind the reference windows (whose loads come first: they are needed
     * first) and consumed after the prediction */
    ResidRows rrows;
    if (!uniform) rrows = mb_residual_fetch(ge.coded, coef, lane);
    s2 pl01 = pk(0), pl23 = pk(0);               /* the lane's four luma prediction samples, two packed pairs */
    s2 pc01 = pk(0), pc23 = pk(0);               /* lanes 0..31: the four chroma prediction samples of the lane's row */
    if (uniform) {
        const int mvx = (int16_t)(mv0 & 0xFFFFu), mvy = (int32_t)mv0 >> 16;
        const H264K_GLOBAL uint8_t *ref = (const H264K_GLOBAL uint8_t *)slot_ptr(fd, refs & 255u);
        const int xi = mbx * 16 + (mvx >> 2) - 2, yi = mby * 16 + (mvy >> 2) - 2;
        const int xs = (xi >> 4) << 4;                           /* first column of the window's first tile */
        const int cxi = mbx * 8 + (mvx >> 3), cyi = mby * 8 + (mvy >> 3);
        const int cxs = (cxi >> 3) << 3;
        /* ---- stage: luma rows yi..yi+20 x the tiles at xs, xs+16, xs+32 (the window needs columns xi..xi+20); chroma
         * rows cyi..cyi+8 x the tiles at cxs, cxs+8 (columns cxi..cxi+8).  One aligned 16-byte load per lane for luma
         * (lane = 3 * row + tile: 63 lanes), one 8-byte load for chroma (lane = 18 * plane + 2 * row + tile: 36 lanes). */
        const bool lfast = xi >= 0 && xi + 21 <= W && yi >= 0 && yi + 21 <= H;
        const bool cfast = cxi >= 0 && cxi + 9 <= CW && cyi >= 0 && cyi + 9 <= CH;
        /* All global loads of the macroblock are issued back to back — the window pieces here, the coefficient rows above —
         * and only then consumed: one memory round trip per macroblock.  (A load and the LDS store of its result inside one
         * `if` make the wavefront wait for that load before it issues the next one.)  Lanes without a piece load the first
         * bytes of the reference frame and drop them. */
        const int lr = (lane * 43) >> 7, lk = lane - 3 * lr, lx = xs + 16 * lk;          /* lane / 3, lane % 3 */
        /* (the window's 21 columns reach into the third tile only when they start in the last four columns of the first:
         * in three cases out of four that tile is not requested at all — nothing reads the bytes it would have filled) */
        const bool l_on = lfast && lane < 63 && lx < W && (lk < 2 || xi - xs >= 12);
        const uint4 vl = ld16g(ref + (l_on ? luma_at(wmb, lx, yi + lr) : (size_t)0));
        const int cp = lane >= 18, rem = cp ? lane - 18 : lane, cr = rem >> 1, ck = rem & 1, cx = cxs + 8 * ck;
        const bool c_on = cfast && lane < 36 && cx < CW;
        const uint2 vc = ld8g(ref + (c_on ? chroma_at(wmb, cp, cx, cyi + cr) : (size_t)0));
        rrows = mb_residual_fetch(ge.coded, coef, lane);
        if (l_on) {
            uint32_t *d32 = reinterpret_cast<uint32_t *>(lw + lr * IW_STRIDE + 16 * lk);
            d32[0] = vl.x; d32[1] = vl.y; d32[2] = vl.z; d32[3] = vl.w;
        }
        if (c_on) {
            uint32_t *d32 = reinterpret_cast<uint32_t *>(lc + cp * 9 * IC_STRIDE + cr * IC_STRIDE + 8 * ck);
            d32[0] = vc.x; d32[1] = vc.y;
        }
        if (!lfast) {
            for (int d = lane; d < 21 * 48; d += 64) {
                const int r = d / 48, c = d % 48;
                lw[r * IW_STRIDE + c] = ref[luma_at(wmb, clip3(0, W - 1, xs + c), clip3(0, H - 1, yi + r))];
            }
        }
        if (!cfast) {
            for (int d = lane; d < 2 * 9 * 16; d += 64) {
                const int pp = d / 144, r = (d % 144) / 16, c = d % 16;
                lc[pp * 9 * IC_STRIDE + r * IC_STRIDE + c] = ref[chroma_at(wmb, pp, clip3(0, CW - 1, cxs + c), clip3(0, CH - 1, cyi + r))];
            }
        }
        wave_sync();
#ifdef H264K_INTER_PROFILE
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
        IPROF(2);
        /* ---- luma: window rows (4*by+row)..+5, bytes o..o+11 with o = (xi-xs) + 4*bx ---- */
        {
            const int o = (xi - xs) + 4 * bx, sh = 8 * (o & 3);
            luma_pred_lds(lw + (4 * by + row) * IW_STRIDE + (o & ~3), IW_STRIDE, sh, mvx & 3, mvy & 3, pl01, pl23);
        }
        /* ---- chroma: lanes 0..31, 4 samples of one row = two pairs ---- */
        {
            const int k = (lane & 31) >> 2, plane = k >> 2, cbx = k & 1, cby = (k >> 1) & 1;
            const int cy = cby * 4 + row, cx0 = cbx * 4 + 2 * (lane >> 5);
            const uint8_t *s0 = lc + plane * 9 * IC_STRIDE + cy * IC_STRIDE + (cxi - cxs) + cx0;
            pc01 = chroma_pair_lds(s0, s0 + IC_STRIDE, mvx & 7, mvy & 7);
            pc23 = pair_from_upper_half(pc01);
        }
    } else if (quadwise) {
        /* ---- one motion vector per 8x8 quadrant (16x8, 8x16, 8x8 partitions): four 13x13 luma and four 5x5 (x2 planes)
         * chroma windows staged in LDS, then the same window arithmetic per lane.  Staging: per quadrant 13 luma rows x 2
         * tiles (26 aligned 16-byte pieces: lanes 0..25) and 2 planes x 5 chroma rows x 2 tiles (20 aligned 8-byte pieces:
         * lanes 26..45) — which piece a lane fetches is the same in every quadrant, what differs between the quadrants (motion
         * vector, reference, window origin, whether the window lies inside the picture) is wave-uniform: scalar registers.
         * All four quadrants are requested before the first is consumed (one memory round trip).  (Round 3 let every lane
         * derive quadrant, row and tile of TWO pieces from its lane number with divisions, and load its quadrant's motion
         * vector from memory: 627 vector instructions per macroblock against 290 on the one-vector path.) ---- */
        uint8_t *lq = lw, *cq = lw + 4 * 13 * QW_STRIDE;
        const H264K_CONST uint32_t *mvc = (const H264K_CONST uint32_t *)fd.mvx + 16 * (size_t)mvx_idx;   /* (x | y << 16) per 4x4 block, raster */
        const bool is_l = lane < 26, is_c = lane >= 26 && lane < 46;
        const int e = lane - 26, pp = e >= 10, e2 = pp ? e - 10 : e;
        const int pr = is_l ? lane >> 1 : e2 >> 1, pk2 = (is_l ? lane : e2) & 1;          /* the piece's row in the window, its tile (0 / 1) */
        uint32_t mvq[4];
        uint4 pv[4];
        bool pon[4], lfast[4], cfast[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            mvq[q] = mvc[(q >> 1) * 8 + (q & 1) * 2];
            const int mvx = (int16_t)(mvq[q] & 0xFFFFu), mvy = (int32_t)mvq[q] >> 16;
            const H264K_GLOBAL uint8_t *ref = (const H264K_GLOBAL uint8_t *)slot_ptr(fd, (refs >> (8 * q)) & 255u);
            const int xi = mbx * 16 + 8 * (q & 1) + (mvx >> 2) - 2, yi = mby * 16 + 8 * (q >> 1) + (mvy >> 2) - 2;
            const int cxi = mbx * 8 + 4 * (q & 1) + (mvx >> 3), cyi = mby * 8 + 4 * (q >> 1) + (mvy >> 3);
            lfast[q] = xi >= 0 && xi + 13 <= W && yi >= 0 && yi + 13 <= H;
            cfast[q] = cxi >= 0 && cxi + 5 <= CW && cyi >= 0 && cyi + 5 <= CH;
            const int xs = ((xi >> 4) << 4) + 16 * pk2, cxs = ((cxi >> 3) << 3) + 8 * pk2;
            pon[q] = is_l ? (lfast[q] && xs < W) : (is_c && cfast[q] && cxs < CW);
            const size_t off = !pon[q] ? (size_t)0 : is_l ? luma_at(wmb, xs, yi + pr) : chroma_at(wmb, pp, cxs, cyi + pr);
            pv[q] = ld16g(ref + off);        /* (no branch around a load: its end would wait for it.  Chroma lanes use the first 8 of the 16 bytes;
                                                 the rest is the plane's next row, or the first bytes of what follows the tile) */
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            if (pon[q]) {
                uint32_t *d32 = reinterpret_cast<uint32_t *>(is_l ? lq + q * 13 * QW_STRIDE + pr * QW_STRIDE + 16 * pk2
                                                                  : cq + (q * 2 + pp) * 5 * QC_STRIDE + pr * QC_STRIDE + 8 * pk2);
                d32[0] = pv[q].x; d32[1] = pv[q].y;
                if (is_l) { d32[2] = pv[q].z; d32[3] = pv[q].w; }
            }
            /* windows that leave the picture (h264bsdFillBlock, reconstruct.c:2244): gathered sample by sample, clamped */
            if (!lfast[q] || !cfast[q]) {                            /* wave-uniform */
                const int mvx = (int16_t)(mvq[q] & 0xFFFFu), mvy = (int32_t)mvq[q] >> 16;
                const uint8_t *ref = slot_ptr(fd, (refs >> (8 * q)) & 255u);
                const int xi = mbx * 16 + 8 * (q & 1) + (mvx >> 2) - 2, yi = mby * 16 + 8 * (q >> 1) + (mvy >> 2) - 2;
                const int cxi = mbx * 8 + 4 * (q & 1) + (mvx >> 3), cyi = mby * 8 + 4 * (q >> 1) + (mvy >> 3);
                if (!lfast[q] && is_l) {
                    const int xs = ((xi >> 4) << 4) + 16 * pk2, yy = clip3(0, H - 1, yi + pr);
                    uint32_t *d32 = reinterpret_cast<uint32_t *>(lq + q * 13 * QW_STRIDE + pr * QW_STRIDE + 16 * pk2);
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        uint32_t v = 0;
#pragma unroll
                        for (int i = 0; i < 4; i++) v |= (uint32_t)ref[luma_at(wmb, clip3(0, W - 1, xs + 4 * c + i), yy)] << (8 * i);
                        d32[c] = v;
                    }
                }
                if (!cfast[q] && is_c) {
                    const int cxs = ((cxi >> 3) << 3) + 8 * pk2, yy = clip3(0, CH - 1, cyi + pr);
                    uint32_t *d32 = reinterpret_cast<uint32_t *>(cq + (q * 2 + pp) * 5 * QC_STRIDE + pr * QC_STRIDE + 8 * pk2);
#pragma unroll
                    for (int c = 0; c < 2; c++) {
                        uint32_t v = 0;
#pragma unroll
                        for (int i = 0; i < 4; i++) v |= (uint32_t)ref[chroma_at(wmb, pp, clip3(0, CW - 1, cxs + 4 * c + i), yy)] << (8 * i);
                        d32[c] = v;
                    }
                }
            }
        }
        wave_sync();
        {
            const int q = (by >> 1) * 2 + (bx >> 1);
            const uint32_t mvm = q == 0 ? mvq[0] : q == 1 ? mvq[1] : q == 2 ? mvq[2] : mvq[3];          /* the lane's quadrant's vector: three selects */
            const int mvx = (int16_t)(mvm & 0xFFFFu), mvy = (int32_t)mvm >> 16;
            const int xi = mbx * 16 + 8 * (q & 1) + (mvx >> 2) - 2;
            const int o = (xi & 15) + 4 * (bx & 1), sh = 8 * (o & 3);
            luma_pred_lds(lq + q * 13 * QW_STRIDE + (4 * (by & 1) + row) * QW_STRIDE + (o & ~3), QW_STRIDE, sh, mvx & 3, mvy & 3, pl01, pl23);
        }
        {
            const int k = (lane & 31) >> 2, plane = k >> 2, cbx = k & 1, cby = (k >> 1) & 1;
            const int q = cby * 2 + cbx;                             /* a 4x4 chroma block = one luma quadrant */
            const uint32_t mvm = q == 0 ? mvq[0] : q == 1 ? mvq[1] : q == 2 ? mvq[2] : mvq[3];
            const int mvx = (int16_t)(mvm & 0xFFFFu), mvy = (int32_t)mvm >> 16;
            const int cxi = mbx * 8 + 4 * (q & 1) + (mvx >> 3);
            const uint8_t *s0 = cq + (q * 2 + plane) * 5 * QC_STRIDE + row * QC_STRIDE + (cxi & 7) + 2 * (lane >> 5);
            pc01 = chroma_pair_lds(s0, s0 + QC_STRIDE, mvx & 7, mvy & 7);
            pc23 = pair_from_upper_half(pc01);
        }
    } else {
        /* ---- per-lane windows straight from global memory ---- */
        {
            const int mvx = (int16_t)(mv_mine & 0xFFFFu), mvy = (int32_t)mv_mine >> 16;
            const uint8_t *ref = slot_ptr(fd, (refs >> (8 * ((by >> 1) * 2 + (bx >> 1)))) & 255u);
            const int x = mbx * 16 + bx * 4 + (mvx >> 2), y = mby * 16 + by * 4 + row + (mvy >> 2);
            int pl[4];
            if (((mvx | mvy) & 3) == 0 && x >= 0 && x + 3 < W && y >= 0 && y < H) {
                const uint32_t v = luma4_at(ref, wmb, x, y);
                pl[0] = v & 255; pl[1] = (v >> 8) & 255; pl[2] = (v >> 16) & 255; pl[3] = v >> 24;
            } else {
                uint32_t rw[6][3];
                luma_window_global(ref, wmb, W, H, x, y, rw);
                luma_from_window(rw, mvx & 3, mvy & 3, pl);
            }
            pl01 = as_s2((uint32_t)pl[0] | ((uint32_t)pl[1] << 16)); pl23 = as_s2((uint32_t)pl[2] | ((uint32_t)pl[3] << 16));
        }
        if (lane < 32) {
            int pc[4];
            const int k = lane >> 2, plane = k >> 2, cbx = k & 1, cby = (k >> 1) & 1;
            const int cy = cby * 4 + row, cx0 = cbx * 4;
#pragma unroll
            for (int pair = 0; pair < 2; pair++) {
                const int cx = cx0 + 2 * pair;
                const int lb = (cy >> 1) * 4 + (cx >> 1);             /* owning 4x4 luma block */
                const int mvx = mvs[2 * lb], mvy = mvs[2 * lb + 1];
                const uint8_t *ref = slot_ptr(fd, (refs >> (8 * ((cy >> 2) * 2 + (cx >> 2)))) & 255u);
                chroma_pred2(ref, wmb, plane, CW, CH, mbx * 8 + cx + (mvx >> 3), mby * 8 + cy + (mvy >> 3), mvx & 7, mvy & 7, pc + 2 * pair);
            }
            pc01 = as_s2((uint32_t)pc[0] | ((uint32_t)pc[1] << 16)); pc23 = as_s2((uint32_t)pc[2] | ((uint32_t)pc[3] << 16));
        }
    }

    /* (an unconditional use of the coefficient rows here — they arrived long ago — keeps the compiler from sinking their loads
     * into the residual code, where every coded macroblock would wait for a second memory round trip) */
    asm volatile("" :: "v"(rrows.y.x), "v"(rrows.y.y), "v"(rrows.c.x), "v"(rrows.c.y), "v"(rrows.cdc.x), "v"(rrows.cdc.y));
#ifdef H264K_INTER_PROFILE
    asm volatile("" :: "v"(pl01), "v"(pl23), "v"(pc01), "v"(pc23));
#endif
    IPROF(3);
    /* ---- residual add, clip, store.  Lane (block, row) holds 4 samples of row 4*by+row at column 4*bx: the 64 dwords of the
     * wavefront ARE the 256 luma bytes of the tile (each group of 16 lanes one 64-byte piece), the 32 chroma dwords its third
     * line — two coalesced stores, no detour through LDS.  A macroblock without coefficients (55 % of this list in the bundled
     * stream) stores its prediction as it is: no unpacking, no residual, no clipping ---- */
    H264K_GLOBAL uint8_t *T = cur + (size_t)mb * TILE;
    uint32_t luma_dw, chroma_dw;
    if ((ge.coded & 0x03FFFFFFu) == 0u) {                        /* wave-uniform */
        luma_dw = perm(as_u32(pl23), as_u32(pl01), 0x06040200u);
        chroma_dw = perm(as_u32(pc23), as_u32(pc01), 0x06040200u);
    } else if (!(ge.coded & FJ_CODED_WIDE)) {                    /* wave-uniform: the host proved that 16 bits hold every intermediate */
        s2 y01, y23, c01, c23;
        mb_residual_pk(ge.coded, qp_y, qp_c, lane, rrows, y01, y23, c01, c23);
        const s2 lo = pk(0), hi = pk(255);
        const s2 l01 = pk_clip(lo, hi, pl01 + y01), l23 = pk_clip(lo, hi, pl23 + y23);
        const s2 k01 = pk_clip(lo, hi, pc01 + c01), k23 = pk_clip(lo, hi, pc23 + c23);
        luma_dw = perm(as_u32(l23), as_u32(l01), 0x06040200u);
        chroma_dw = perm(as_u32(k23), as_u32(k01), 0x06040200u);
    } else {
        int ry[4], rc[4];
        report_residual_range(fd, mb_residual_compute<false>(ge.coded, qp_y, qp_c, false, coef, lane, rrows, ry, rc), lane);
        luma_dw = pack4(clip255(pl01.x + ry[0]), clip255(pl01.y + ry[1]), clip255(pl23.x + ry[2]), clip255(pl23.y + ry[3]));
        chroma_dw = pack4(clip255(pc01.x + rc[0]), clip255(pc01.y + rc[1]), clip255(pc23.x + rc[2]), clip255(pc23.y + rc[3]));
    }
#ifdef H264K_INTER_PROFILE
    asm volatile("" :: "v"(luma_dw), "v"(chroma_dw));
#endif
    IPROF(4);
    *reinterpret_cast<H264K_GLOBAL uint32_t *>(T + (by * 4 + row) * 16 + bx * 4) = luma_dw;
    if (lane < 32) {
        const int k = lane >> 2, plane = k >> 2, cbx = k & 1, cby = (k >> 1) & 1;
        *reinterpret_cast<H264K_GLOBAL uint32_t *>(T + T_CB + plane * 64 + (cby * 4 + row) * 8 + cbx * 4) = chroma_dw;
    }
    IPROF(5);
#ifdef H264K_INTER_PROFILE
    if (PATH == 0 && lane == 0 && (blockIdx.x & 63u) == 0u) {
        unsigned long long *pc64 = reinterpret_cast<unsigned long long *>(fd.err) + 8;
        for (int k = 0; k < 5; k++) atomicAdd(pc64 + k, ipt[k + 1] - ipt[k]);
        atomicAdd(pc64 + 5, 1ull);
        atomicAdd(pc64 + 6, (ge.coded & 0x03FFFFFFu) ? 1ull : 0ull);
    }
#endif
    wave_sync();          /* the staged windows are overwritten by the next entry's */
  }
}


} // namespace h264k
