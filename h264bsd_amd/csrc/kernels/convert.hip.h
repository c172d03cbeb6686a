/* kernels/convert.hip.h — tiles -> 32-bit pixels (RGBA / BGRA / YCbCrA) in packed 16-bit arithmetic: the device function behind
 * k_convert_tiles and the conversion the per-picture kernels host (k_frame_dbk).  Part of kernels.hip.h (which see); not a stand-alone header. */
#pragma once
namespace h264k {
/* The reference's conversion (src/h264bsd_decoder.c:1163-1370: integer BT.601, limited range, nearest chroma) is, with c = Y - 16,
 * d = Cb - 128, e = Cr - 128:   R = clip((298 c + 409 e + 128) >> 8),  G = clip((298 c - 100 d - 208 e + 128) >> 8),
 * B = clip((298 c + 516 d + 128) >> 8) — products of up to 17 bits.  Multiples of 256 pass through an arithmetic shift by 8
 * unchanged, so with 298 = 256 + 42, 409 = 256 + 153, 516 = 512 + 4, -208 = -256 + 48:
 *     R = clip(Y + (Cr - 144)     + ((42 Y + 153 Cr - 20128) >> 8))
 *     G = clip(Y + (112 - Cr)     + ((42 Y - 100 Cb + 48 Cr + 6112) >> 8))
 *     B = clip(Y + (2 Cb - 272)   + ((42 Y + 4 Cb - 1056) >> 8))
 * where every shifted sum lies in [-20128, 29597]: signed 16-bit halves hold it (intermediates may wrap: the arithmetic is
 * mod 2^16 until the shift).  tests/test_convert_algebra.py checks the identity for all 2^24 (Y, Cb, Cr).  Two samples per
 * register: the two ROWS of a 2 x 2 block that share one chroma sample, so the chroma terms are computed once per four pixels;
 * v_sat_pk_u8_i16 clips and narrows a pair in one instruction; three v_perm assemble two pixels.  12 vector instructions per
 * pixel where the 32-bit form took 25. */
struct ConvChroma { s2 tR, tG, tB, oR, oG, oB; uint32_t raw; };
__device__ __forceinline__ uint32_t sat_pk_u8(s2 v)
{
    uint32_t r;
    asm("v_sat_pk_u8_i16 %0, %1" : "=v"(r) : "v"(as_u32(v)));
    return r;
}
/* cc: bytes (Cb[2q], Cb[2q+1], Cr[2q], Cr[2q+1]) of one chroma row; j: which of the two samples */
__device__ __forceinline__ ConvChroma conv_chroma(uint32_t cc, int j)
{
    ConvChroma k;
    const s2 cb = as_s2(perm(0u, cc, 0x0C000C00u | (uint32_t)j | ((uint32_t)j << 16)));
    const s2 cr = as_s2(perm(0u, cc, 0x0C000C00u | (uint32_t)(2 + j) | ((uint32_t)(2 + j) << 16)));
    k.tR = cr * pk(153) + pk(-20128);
    k.tG = cb * pk(-100) + (cr * pk(48) + pk(6112));
    k.tB = cb * pk(4) + pk(-1056);
    k.oR = cr + pk(-144);
    k.oG = pk(112) - cr;
    k.oB = cb + cb + pk(-272);
    k.raw = cc;
    return k;
}
/* the pixels of column x (0..3) of the lane's two rows: y0 / y1 = the four luma bytes of the upper / lower row */
template <int FMT, int X>
__device__ __forceinline__ void conv_column(uint32_t y0, uint32_t y1, const ConvChroma &k, uint32_t &p0, uint32_t &p1)
{
    constexpr int j = X >> 1;
    if (FMT == 2) {                                  /* YCbCrA: bytes (Y, Cb, Cr, 0xFF) */
        p0 = perm(k.raw, y0, (uint32_t)X | ((uint32_t)(4 + j) << 8) | ((uint32_t)(6 + j) << 16) | 0x0D000000u);
        p1 = perm(k.raw, y1, (uint32_t)X | ((uint32_t)(4 + j) << 8) | ((uint32_t)(6 + j) << 16) | 0x0D000000u);
        return;
    }
    const s2 Y = as_s2(perm(y1, y0, 0x0C000C00u | (uint32_t)X | ((uint32_t)(4 + X) << 16)));
    const s2 m = Y * pk(42);
    const uint32_t r = sat_pk_u8(((m + k.tR) >> pk(8)) + k.oR + Y);
    const uint32_t g = sat_pk_u8(((m + k.tG) >> pk(8)) + k.oG + Y);
    const uint32_t b = sat_pk_u8(((m + k.tB) >> pk(8)) + k.oB + Y);
    /* fmt 0 RGBA: bytes (R, G, B, FF); fmt 1 BGRA: bytes (B, G, R, FF) */
    const uint32_t first = FMT == 0 ? r : b, third = FMT == 0 ? b : r;
    const uint32_t fg = perm(g, first, 0x05010400u);             /* (first.row0, G.row0, first.row1, G.row1) */
    p0 = perm(third, fg, 0x0D040100u);
    p1 = perm(third, fg, 0x0D050302u);
}

/* One wavefront converts TWO horizontally adjacent macroblock tiles: lanes 0..31 the tile at (mbx, mby), lanes 32..63 the one to its
 * right; lane (rp, q) of a half = rows 2 rp, 2 rp + 1, columns 4 q .. 4 q + 3: two dwords of luma, two bytes of each chroma plane,
 * two 16-byte stores — a store instruction of the wavefront covers 128 contiguous bytes of eight picture rows.
 * tile0: the left tile; dst0: its first pixel in the picture of W = 16 wmb pixels per row (both wave-uniform: scalar bases, the
 * lane's part of every address is a 32-bit offset that does not change from pair to pair); right_on: the pair has a right tile. */
struct ConvLane { uint32_t in_off, c_off, out_off; };
__device__ __forceinline__ ConvLane conv_lane(uint32_t W, uint32_t lane)
{
    const uint32_t half = lane >> 5, l = lane & 31u, rp = l >> 2, q = l & 3u;
    ConvLane c;
    c.in_off = half * TILE + rp * 32u + q * 4u;
    c.c_off = half * TILE + T_CB + rp * 8u + q * 2u;
    c.out_off = 4u * (2u * rp * W + half * 16u + q * 4u);
    return c;
}
/* the lane's samples of one tile pair: requested (conv_pair_load), converted and stored later (conv_pair_store) — a wavefront keeps
 * several pairs in flight */
struct ConvRaw { uint32_t y0, y1, cc0, cc1; };
__device__ __forceinline__ ConvRaw conv_pair_load(const uint8_t *__restrict__ tile0, const ConvLane &cl)
{
    const H264K_GLOBAL uint8_t *T = (const H264K_GLOBAL uint8_t *)tile0;
    ConvRaw r;
    r.y0 = __builtin_nontemporal_load((const H264K_GLOBAL uint32_t *)(T + cl.in_off));
    r.y1 = __builtin_nontemporal_load((const H264K_GLOBAL uint32_t *)(T + cl.in_off + 16u));
    r.cc0 = __builtin_nontemporal_load((const H264K_GLOBAL uint16_t *)(T + cl.c_off));
    r.cc1 = __builtin_nontemporal_load((const H264K_GLOBAL uint16_t *)(T + cl.c_off + (T_CR - T_CB)));
    return r;
}
template <int FMT>
__device__ __forceinline__ void conv_pair_store(const ConvRaw &r, uint32_t *__restrict__ dst0, uint32_t W, const ConvLane &cl)
{
    const uint32_t cc = r.cc0 | (r.cc1 << 16);
    u32x4 a, b;
    {
        const ConvChroma k = conv_chroma(cc, 0);
        uint32_t p0, p1;
        conv_column<FMT, 0>(r.y0, r.y1, k, p0, p1); a.x = p0; b.x = p1;
        conv_column<FMT, 1>(r.y0, r.y1, k, p0, p1); a.y = p0; b.y = p1;
    }
    {
        const ConvChroma k = conv_chroma(cc, 1);
        uint32_t p0, p1;
        conv_column<FMT, 2>(r.y0, r.y1, k, p0, p1); a.z = p0; b.z = p1;
        conv_column<FMT, 3>(r.y0, r.y1, k, p0, p1); a.w = p0; b.w = p1;
    }
    H264K_GLOBAL uint8_t *o = (H264K_GLOBAL uint8_t *)dst0;
    __builtin_nontemporal_store(a, (H264K_GLOBAL u32x4 *)(o + cl.out_off));
    __builtin_nontemporal_store(b, (H264K_GLOBAL u32x4 *)(o + cl.out_off + 4u * W));
}
/* Batches of CONV_BATCH consecutive pairs, software-pipelined: the loads of batch k + 1 are issued BEFORE batch k is converted and
 * stored.  Vector memory operations complete in issue order as far as s_waitcnt vmcnt can tell, so a wavefront that alternates
 * "load a batch" and "store a batch" waits, at every load, for the previous batch's stores to be acknowledged by the memory
 * system; with the order load(k+1), store(k) the wait for load(k+1) leaves exactly the stores of batch k outstanding.  (Measured
 * with 12 wavefronts per CU inside k_frame_dbk, conversion of 256 pictures: 740 us per tick before, 560 us after.)
 * next(): first pair of the wavefront's next batch, >= end when there is none.  fmt: 0 RGBA, 1 BGRA, 2 YCbCrA (bytes in memory
 * order); everything but the lane number is wave-uniform. */
#ifndef CONV_BATCH_N
#define CONV_BATCH_N 8
#endif
constexpr uint32_t CONV_BATCH = CONV_BATCH_N;
struct ConvPic { const uint8_t *src; uint32_t *dst; uint32_t wmb, W, ppr, magic; };
__device__ __forceinline__ ConvPic conv_pic(const uint8_t *src, uint32_t *dst, uint32_t wmb)
{
    ConvPic c;
    c.src = src; c.dst = dst; c.wmb = wmb; c.W = wmb * 16u; c.ppr = (wmb + 1u) >> 1;
    c.magic = c.ppr == 1u ? 0u : 0xFFFFFFFFu / c.ppr + 1u;          /* p / ppr = mulhi(p, magic): exact for p * ppr < 2^32 */
    return c;
}
struct ConvBatch { ConvRaw raw[CONV_BATCH]; };
/* (no branch around a load: pairs behind the end of the share read the share's last pair, the right-hand half of a pair behind
 * the last column reads the left tile — what they load is dropped by conv_batch_store) */
__device__ __forceinline__ void conv_batch_load(const ConvPic &c, uint32_t first, uint32_t end, const ConvLane &cl, uint32_t lane, ConvBatch &b)
{
#pragma unroll
    for (uint32_t j = 0; j < CONV_BATCH; j++) {
        const uint32_t p = min(first + j, end - 1u);
        const uint32_t mby = c.ppr == 1u ? p : __umulhi(p, c.magic), mbx = 2u * (p - mby * c.ppr);
        ConvLane l2 = cl;
        if (mbx + 1u >= c.wmb && lane >= 32u) { l2.in_off -= TILE; l2.c_off -= TILE; }
        b.raw[j] = conv_pair_load(c.src + ((size_t)mby * c.wmb + mbx) * TILE, l2);
    }
}
template <int FMT>
__device__ __forceinline__ void conv_batch_store(const ConvPic &c, uint32_t first, uint32_t end, const ConvLane &cl, uint32_t lane, const ConvBatch &b)
{
#pragma unroll
    for (uint32_t j = 0; j < CONV_BATCH; j++) {
        const uint32_t p = first + j;
        const uint32_t mby = c.ppr == 1u ? p : __umulhi(p, c.magic), mbx = 2u * (p - mby * c.ppr);
        if (p < end && (mbx + 1u < c.wmb || lane < 32u)) conv_pair_store<FMT>(b.raw[j], c.dst + (size_t)mby * 16u * c.W + mbx * 16u, c.W, cl);
    }
}
template <int FMT, class Next>
__device__ __forceinline__ void conv_pipeline_fmt(const ConvPic &c, uint32_t end, const ConvLane &cl, uint32_t lane, Next next)
{
    ConvBatch A, B;
    uint32_t pa = next(), pb;
    if (pa >= end) return;
    conv_batch_load(c, pa, end, cl, lane, A);
    for (;;) {
        pb = next();
        if (pb < end) conv_batch_load(c, pb, end, cl, lane, B);
        conv_batch_store<FMT>(c, pa, end, cl, lane, A);
        if (pb >= end) break;
        pa = next();
        if (pa < end) conv_batch_load(c, pa, end, cl, lane, A);
        conv_batch_store<FMT>(c, pb, end, cl, lane, B);
        if (pa >= end) break;
    }
}
template <class Next>
__device__ __forceinline__ void conv_pipeline(const ConvPic &c, uint32_t end, int fmt, const ConvLane &cl, uint32_t lane, Next next)
{
    if (fmt == 1) conv_pipeline_fmt<1>(c, end, cl, lane, next);
    else if (fmt == 0) conv_pipeline_fmt<0>(c, end, cl, lane, next);
    else conv_pipeline_fmt<2>(c, end, cl, lane, next);
}

/* ---- hosted conversion: the tile pairs [*, end) of FrameDesc.conv_src, handed out in batches through a counter in LDS ----
 * Whoever has issue slots to spare converts: the conversion wavefronts of a k_frame_dbk workgroup while its other wavefronts walk
 * the picture's dependency chains (which leave a third of the CU's issue slots and most of the HBM bandwidth unused), and every
 * wavefront of the workgroup once its graphs are exhausted.  (The counter was a device-scope atomic in HBM at first: 4 us per
 * claim under load.)  next_lds: LDS address of the workgroup's counter (an address-space-3 atomic: through a generic pointer it
 * would be a flat atomic, and waiting for one waits for every outstanding vector memory operation), set to the first pair of the
 * workgroup's share before any wavefront calls this. */
__device__ __noinline__ void conv_drain(const FrameDesc &fd, uint32_t next_lds, uint32_t end, uint32_t lane)
{
    /* (arguments and what is loaded through the descriptor's address arrive in vector registers: back into scalar ones, so that the
     * loop's branches are scalar and its loads' addresses are scalar base + lane offset) */
    auto uni = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
    auto uni_ptr = [&](const void *q) { const uint64_t a = (uint64_t)(uintptr_t)q; return (uintptr_t)((uint64_t)uni((uint32_t)a) | ((uint64_t)uni((uint32_t)(a >> 32)) << 32)); };
    end = uni(end);
    H264K_LDS uint32_t *ctr = (H264K_LDS uint32_t *)(uintptr_t)uni(next_lds);
    const ConvPic c = conv_pic((const uint8_t *)uni_ptr(fd.conv_src), (uint32_t *)uni_ptr(fd.conv_dst), uni(fd.wmb));
    const ConvLane cl = conv_lane(c.W, lane);
    conv_pipeline(c, end, (int)uni(fd.conv_fmt), cl, lane, [&]() -> uint32_t {
        uint32_t first = 0;
        if (lane == 0) first = __hip_atomic_fetch_add(ctr, CONV_BATCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return (uint32_t)__builtin_amdgcn_readfirstlane((int)first);
    });
}

} // namespace h264k
