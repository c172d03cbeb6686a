/* kernels/common.hip.h — shared types, helpers, residual transforms, frame layout (every kernel of kernels.hip.h includes this first).  Part of kernels.hip.h (which see); not a stand-alone header. */
#pragma once
/* Per-picture launch descriptor, built on the host from the frame-job header (saves the kernels one
 * dependent load: blob header -> section pointers). */
struct FrameDesc {
    const FjMbRec  *recs;
    const int16_t  *mvx;          /* the sparse vector section (FjHeader.mvx_off): 16 x (x, y) per macroblock that has more than one vector */
    const int16_t  *coefs;
    const uint32_t *lvl;          /* lvl_start[n_levels+1] */
    const uint16_t *idx;          /* intra MB addresses sorted by level */
    const FjCopy   *copy;         /* whole-sample copy macroblocks */
    const FjGen    *gen;          /* all other inter macroblocks */
    const uint16_t *dbki;         /* macroblocks whose boundary strengths are not trivially zero */
    uint8_t        *dbk;          /* per-stream scratch: n_mbs x 32-byte deblocking records, then n_mbs "any" bytes */
    uint8_t        *cur;          /* slot that receives the picture */
    uint32_t        n_mbs, n_levels, n_copy, n_gen, n_dbk;
    uint32_t        n_gen_uni;    /* the first n_gen_uni entries of gen have one motion vector for the whole macroblock */
    uint32_t        n_gen_quad;   /* the next n_gen_quad one motion vector per 8x8 quadrant, the rest finer partitions */
    uint16_t        wmb, hmb;
    uint32_t        wmb_magic;    /* floor(2^32 / wmb) + 1: mb / wmb = mulhi(mb, wmb_magic) for every macroblock address (< 2^16) — a scalar
                                     multiply where the compiler's division by a run-time value is a dozen vector instructions */
    uint32_t        any_deblock;
    uint16_t        dbk_bands, intra_bands;   /* row bands (= workgroups) the per-picture kernels may split this picture into (>= 1; the launch caps it) */
    uint16_t        heavy;                    /* 1: mostly intra coded — several times the work of the other pictures of its tick */
    uint32_t       *err;          /* device error word of the engine (DEVERR_* bits, atomicOr: must stay 0), err[1] = number of times a tripwire fired */
    uint8_t        *slot[FJ_MAX_SLOTS];
    /* Hosted colour conversion (kernels/convert.hip.h, conv_drain): while this picture is filtered, wavefronts of its k_frame_dbk
     * workgroup convert ANOTHER, finished picture of the stream (conv_src: its tiles) into conv_dst (32-bit pixels, conv_fmt 0 RGBA /
     * 1 BGRA / 2 YCbCrA).  conv_src == nullptr: nothing to convert.  A tick whose pictures need no filtering launches
     * k_convert_rest instead (engine.hip, launch_tick). */
    const uint8_t  *conv_src;
    uint32_t       *conv_dst;
    uint32_t        conv_fmt, conv_pad;
};

/* Bits of the device error word.  None of them can be set by a frame job the host parser built: they are tripwires. */
#define DEVERR_RESIDUAL_RANGE 1u  /* a residual sample left [-512,511]: the reference fails the macroblock there
                                     (src/h264bsd_transform.c:184-188); the host decides this error while it parses
                                     (hd_resid.c), so a job that reaches the kernels never contains one          */
#define DEVERR_INTRA_SCHED    2u  /* k_frame_intra gave up waiting for a ready macroblock (scheduling bug)          */
#define DEVERR_DBK_SCHED      4u  /* k_frame_dbk did                                                               */

/* Deblocking record of one macroblock (48 bytes), written by k_dbk, read by k_frame_dbk:
 *   bytes 0..15  boundary strengths, one nibble per (dir, edge e, segment k): n = 16*dir + 4*e + k (byte n >> 1, low nibble first)
 *   bytes 16..39 six dwords, one per threshold class c = luma{left,top,inner}, chroma{left,top,inner}:
 *                byte 0 alpha, byte 1 beta, byte 2 tc0 for bS 1, byte 3 tc0 for bS 2 — the VALUES of Tables 8-16 / 8-17, looked up once
 *                per macroblock here instead of once per edge and lane in the filter
 *   bytes 40..45 tc0 for bS 3 of the six classes
 *   byte 46 FJ_DBK_* flags (LEFT / TOP only where that neighbour exists), byte 47 "any strength non-zero"
 * followed (at dbk + 48*n_mbs) by one flag byte per MB (DBKF_*). */
#define DBK_REC_BYTES 48
/* the per-macroblock flag byte behind the records */
#define DBKF_ANY  1u   /* at least one non-zero strength: the macroblock is filtered                         */
#define DBKF_LEFT 2u   /* its left macroblock edge has a non-zero strength: it reads and rewrites the last columns of (x-1,y) */
#define DBKF_TOP  4u   /* its upper macroblock edge has one: it reads and rewrites the last rows of (x,y-1)                   */
#define DBKF_INNER 8u  /* an edge INSIDE the macroblock has one.  A filtered macroblock without it only touches columns -3..2 (left
                          edge) and / or rows -3..2 (upper edge): its right-hand neighbour has to wait for it only if its UPPER edge is
                          filtered, the macroblock below only if its LEFT edge is (k_frame_dbk, dependency rule) */
/* Per-stream deblocking scratch (FrameDesc.dbk): n_mbs records | n4 flag bytes (DBKF_*) | n4 "done" bytes of k_frame_dbk's row
 * bands for its luma graph | n4 for its chroma graph | n4 "done" bytes of k_frame_intra's row bands | exit counters of the two
 * kernels (u32 each) — n4 = n_mbs rounded up to a multiple of 4.  Flags, done bytes and counters are zero between pictures (the
 * last band to leave cleans up). */
#define DBK_SCRATCH_BYTES(n_mbs) ((size_t)(n_mbs) * (DBK_REC_BYTES + 4) + 64)
#define SCRATCH_DONE_DBK_LUMA 0
#define SCRATCH_DONE_DBK_CHROMA 1
#define SCRATCH_DONE_INTRA 2


namespace h264k {

/* LevelScale(qp % 6, class) of 8.5.9 — classes: both indices even (10,11,13,14,16,18), mixed (13,14,16,18,20,23), both odd
 * (16,18,20,23,25,29) — five bits per entry in an immediate: a lookup is a shift and a mask, not a (lane-indexed = global
 * memory) table read in the middle of every macroblock's residual */
__device__ __forceinline__ int level_scale(int m, int cls)
{
    const uint32_t c = cls == 0 ? 0x2507356Au : cls == 1 ? 0x2F4941CDu : 0x3B9BD250u;
    return (int)((c >> (5 * m)) & 31u);
}
__constant__ uint8_t c_alpha[52] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13,
    15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255 };
__constant__ uint8_t c_beta[52] = { 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6,
    7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18 };
__constant__ uint8_t c_tc0[52][4] = {
    { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 },
    { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 },
    { 0, 0, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 1, 0 }, { 0, 0, 1, 0 }, { 0, 0, 1, 0 }, { 0, 1, 1, 0 }, { 0, 1, 1, 0 }, { 1, 1, 1, 0 },
    { 1, 1, 1, 0 }, { 1, 1, 1, 0 }, { 1, 1, 1, 0 }, { 1, 1, 2, 0 }, { 1, 1, 2, 0 }, { 1, 1, 2, 0 }, { 1, 1, 2, 0 }, { 1, 2, 3, 0 },
    { 1, 2, 3, 0 }, { 2, 2, 3, 0 }, { 2, 2, 4, 0 }, { 2, 3, 4, 0 }, { 2, 3, 4, 0 }, { 3, 3, 5, 0 }, { 3, 4, 6, 0 }, { 3, 4, 6, 0 },
    { 4, 5, 7, 0 }, { 4, 5, 8, 0 }, { 4, 6, 9, 0 }, { 5, 7, 10, 0 }, { 6, 8, 11, 0 }, { 6, 8, 13, 0 }, { 7, 10, 14, 0 }, { 8, 11, 16, 0 },
    { 9, 12, 18, 0 }, { 10, 13, 20, 0 }, { 11, 15, 23, 0 }, { 13, 17, 25, 0 } };
__constant__ uint8_t c_qpc[52] = { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23,
    24, 25, 26, 27, 28, 29, 29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39 };

/* QPc of a chroma qp index (Table 8-15, the c_qpc table) without a lane-indexed (= global memory) lookup */
__device__ __forceinline__ int qpc_of(int qpi)
{
    const int i = qpi - 30;
    const uint32_t c = i < 8 ? 0x55433210u : i < 16 ? 0x98887766u : 0x00AAAA99u;     /* (QPc - 29) for qp index 30..51, a nibble each */
    return qpi < 30 ? qpi : 29 + (int)((c >> (4 * (i & 7))) & 15u);
}
__device__ __forceinline__ int clip255(int v) { return min(max(v, 0), 255); }
__device__ __forceinline__ int clip3(int lo, int hi, int v) { return min(max(v, lo), hi); }
__device__ __forceinline__ int z_of(int x, int y) { return ((y >> 1) << 3) | ((x >> 1) << 2) | ((y & 1) << 1) | (x & 1); }
__device__ __forceinline__ uint32_t pack4(int a, int b, int c, int d)
{
    return (uint32_t)a | ((uint32_t)b << 8) | ((uint32_t)c << 16) | ((uint32_t)d << 24);
}
__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t *p)
{
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

/* ---- packed 16-bit helpers (two samples per register, v_pk_*_i16) ---- */
typedef short s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ s2 pk(int x) { return (s2){ (short)x, (short)x }; }
__device__ __forceinline__ s2 pk_lt(s2 a, s2 b) { return (a - b) >> pk(15); }                  /* a < b ? -1 : 0 */
__device__ __forceinline__ s2 pk_abs(s2 a) { return __builtin_elementwise_max(a, -a); }
__device__ __forceinline__ s2 pk_clip(s2 lo, s2 hi, s2 v) { return __builtin_elementwise_min(__builtin_elementwise_max(v, lo), hi); }
__device__ __forceinline__ s2 pk_sel(s2 m, s2 a, s2 b) { return (a & m) | (b & ~m); }

__device__ __forceinline__ s2 as_s2(uint32_t x) { s2 r; __builtin_memcpy(&r, &x, 4); return r; }
__device__ __forceinline__ uint32_t as_u32(s2 x) { uint32_t r; __builtin_memcpy(&r, &x, 4); return r; }
/* v_perm_b32(hi, lo, sel): result byte i = byte sel[i] of the 8 bytes {lo = 0..3, hi = 4..7}; selector 12 = 0x00 */
__device__ __forceinline__ uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }



/* ---- hand-over between workgroups (row bands of one picture, k_frame_dbk / k_frame_intra) ----
 * Workgroups of one launch may sit on different XCDs, whose L2s are not coherent with each other, and a CU's vector L1 is
 * never refreshed by another CU's stores (MI355X_MICROARCH.md, "inter-workgroup visibility").  The samples a band hands to
 * the band below therefore travel write-through: relaxed agent-scope stores (global_store ... sc1: the line leaves the
 * producer's L2) and relaxed agent-scope loads (global_load ... sc1: past the L1) on the consumer's side, the "done" byte
 * stored after s_waitcnt vmcnt(0) the same way.  No fences: a release fence writes back the whole L2 of the XCD. */
/* ---- hand-over inside a workgroup (the per-picture schedulers) ----
 * A wavefront that has stored a macroblock tells its dependants through LDS.  All wavefronts of a workgroup run on one CU
 * and share its vector L1, and the CU's memory pipeline keeps vector memory instructions in issue order: a load issued by
 * another wavefront of the workgroup after it has seen the LDS release observes the stores issued before that release.  That
 * is the architecture's contract, not an observation: for a workgroup-scope release in front of global stores the compiler
 * emits no s_waitcnt vmcnt(0) on gfx950 (it does under -mtgsplit, where a workgroup may span CUs; this code is never built
 * that way).  So the release does not wait for the stores to be acknowledged by the L2 — several hundred cycles that used
 * to sit on every link of a dependency chain.  Only a macroblock that another WORKGROUP will read (the last row of a band)
 * still waits: its "done" byte must not pass its samples on the way to the other CU. */
#ifndef H264K_RELEASE_WAITS
#define H264K_RELEASE_WAITS 0                                /* 1: the conservative form (wait for every store) for A/B runs */
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#if !defined(__gfx950__) && !defined(__gfx942__)
#error "kernels.hip.h relies on gfx942 / gfx950 memory ordering inside a workgroup (release_stores): build with --offload-arch=gfx950"
#endif
#endif
/* (-mtgsplit, under which a workgroup may span compute units, defines no macro: the Makefile and tools/experiments/build_variant.sh
 * refuse the flag, and tests/test_abi.py checks the kernel descriptors of the built library for the threadgroup-split bit) */
__device__ __forceinline__ void release_stores(bool leaves_the_workgroup)
{
    if (H264K_RELEASE_WAITS || __ballot(leaves_the_workgroup) != 0ull) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}

#define H264K_GLOBAL __attribute__((address_space(1)))      /* HBM pointers: global_load / global_store instead of flat */
#define H264K_LDS    __attribute__((address_space(3)))
#define H264K_CONST  __attribute__((address_space(4)))      /* frame-job sections: nothing writes them while kernels run, so a load
                                                               from a wave-uniform address may be a scalar load (s_load) */
__device__ __forceinline__ uint32_t ld_agent_u32(const void *p)
{
    return __hip_atomic_load((const H264K_GLOBAL uint32_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t ld_agent_u8(const void *p)
{
    return __hip_atomic_load((const H264K_GLOBAL uint8_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent_u32(void *p, uint32_t v)
{
    __hip_atomic_store((H264K_GLOBAL uint32_t *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent_u8(void *p, uint32_t v)
{
    __hip_atomic_store((H264K_GLOBAL uint8_t *)p, (uint8_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
/* 4 / 8 / 16 bytes to a picture: plain, or write-through for samples another band will read */
__device__ __forceinline__ void put4(void *p, uint32_t v, bool wt) { if (wt) st_agent_u32(p, v); else *reinterpret_cast<uint32_t *>(p) = v; }
__device__ __forceinline__ void put8(void *p, uint2 v, bool wt)
{
    if (wt) { st_agent_u32(p, v.x); st_agent_u32(reinterpret_cast<uint8_t *>(p) + 4, v.y); }
    else *reinterpret_cast<uint2 *>(p) = v;
}
__device__ __forceinline__ void put16(void *p, uint4 v, bool wt)
{
    if (wt) {
        uint8_t *q = reinterpret_cast<uint8_t *>(p);
        st_agent_u32(q, v.x); st_agent_u32(q + 4, v.y); st_agent_u32(q + 8, v.z); st_agent_u32(q + 12, v.w);
    } else *reinterpret_cast<uint4 *>(p) = v;
}
/* The launch descriptors are read-only while kernels run: reached through the constant address space, a descriptor field is a
 * scalar load from the scalar cache wherever it is used — not a vector load from global memory that a wavefront waits for
 * in the middle of a macroblock (the reference is handed through the inlined helpers as an ordinary one; the address space is
 * inferred from this cast). */
#define FD_REF(frames, i) (*(const FrameDesc *)((const H264K_CONST FrameDesc *)(frames) + (i)))
/* row of a macroblock address without a division: FrameDesc.wmb_magic (a run-time division is a dozen or two vector instructions) */
__device__ __forceinline__ uint32_t mb_row(const FrameDesc &fd, uint32_t mb) { return fd.wmb == 1 ? mb : __umulhi(mb, fd.wmb_magic); }
/* plain loads / stores with the address space spelled out (global_load / global_store / s_load instead of flat); the HIP vector
 * classes cannot be copied out of a qualified address space, the native vector types can */
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 ld16g(const H264K_GLOBAL uint8_t *p) { const u32x4 v = *(const H264K_GLOBAL u32x4 *)p; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint2 ld8g(const H264K_GLOBAL uint8_t *p) { const u32x2 v = *(const H264K_GLOBAL u32x2 *)p; return make_uint2(v.x, v.y); }
__device__ __forceinline__ void st16g(H264K_GLOBAL uint8_t *p, uint4 v) { *(H264K_GLOBAL u32x4 *)p = (u32x4){ v.x, v.y, v.z, v.w }; }
__device__ __forceinline__ uint4 ld16c(const H264K_CONST void *p) { const u32x4 v = *(const H264K_CONST u32x4 *)p; return make_uint4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ uint8_t *scratch_flags(const FrameDesc &fd) { return fd.dbk + (size_t)fd.n_mbs * DBK_REC_BYTES; }
__device__ __forceinline__ uint8_t *scratch_done(const FrameDesc &fd, int which)      /* SCRATCH_DONE_* */
{
    return scratch_flags(fd) + (size_t)(1 + which) * ((fd.n_mbs + 3u) & ~3u);
}
__device__ __forceinline__ uint32_t *scratch_exits(const FrameDesc &fd, int which)
{
    return reinterpret_cast<uint32_t *>(scratch_flags(fd) + (size_t)4 * ((fd.n_mbs + 3u) & ~3u)) + which;
}

/* 4x4 transpose across the 4 lanes of a quad: lane q holds row q in v[0..3] -> holds column q */
/* lane ^ 1 / lane ^ 2 inside a quad: DPP quad_perm moves (one VALU cycle, no LDS crossbar round trip) */
__device__ __forceinline__ int quad_xor1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, false); }   /* quad_perm [1,0,3,2] */
__device__ __forceinline__ int quad_xor2(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, false); }   /* quad_perm [2,3,0,1] */

__device__ __forceinline__ void quad_transpose(int v[4], int q)
{
    const bool o1 = q & 1, o2 = q & 2;
    int t0 = quad_xor1(o1 ? v[0] : v[1]);
    int t1 = quad_xor1(o1 ? v[2] : v[3]);
    if (o1) { v[0] = t0; v[2] = t1; } else { v[1] = t0; v[3] = t1; }
    t0 = quad_xor2(o2 ? v[0] : v[2]);
    t1 = quad_xor2(o2 ? v[1] : v[3]);
    if (o2) { v[0] = t0; v[1] = t1; } else { v[2] = t0; v[3] = t1; }
}

/* Dequantise + inverse-transform one 4x4 block held one ROW per lane of a quad (H.264 8.5.12):
 * in: c[0..3] = raster row q of the level block (zeros when the block is not coded); dc_override
 * replaces element (0,0) after scaling (Intra16x16 / chroma DC paths).  out: residual row q. */
__device__ __forceinline__ void idct_quad(int c[4], int q, int qp, bool use_dc, int dc)
{
    const int m = qp % 6, sh = qp / 6;
    /* the QP is wave-uniform: three scalar table reads and a select, not a lane-indexed (= global-memory) lookup */
    const int ls0 = level_scale(m, 0), ls1 = level_scale(m, 1), ls2 = level_scale(m, 2);
    const int lsa = (q & 1) ? ls1 : ls0, lsb = (q & 1) ? ls2 : ls1;
    int d0 = (c[0] * lsa) << sh, d1 = (c[1] * lsb) << sh, d2 = (c[2] * lsa) << sh, d3 = (c[3] * lsb) << sh;
    if (use_dc && q == 0) d0 = dc;
    int e0 = d0 + d2, e1 = d0 - d2, e2 = (d1 >> 1) - d3, e3 = d1 + (d3 >> 1);
    int f[4] = { e0 + e3, e1 + e2, e1 - e2, e0 - e3 };
    quad_transpose(f, q);                    /* lane q: column q, f[k] = row k */
    e0 = f[0] + f[2]; e1 = f[0] - f[2]; e2 = (f[1] >> 1) - f[3]; e3 = f[1] + (f[3] >> 1);
    int r[4] = { (e0 + e3 + 32) >> 6, (e1 + e2 + 32) >> 6, (e1 - e2 + 32) >> 6, (e0 - e3 + 32) >> 6 };
    quad_transpose(r, q);                    /* back to row q */
    c[0] = r[0]; c[1] = r[1]; c[2] = r[2]; c[3] = r[3];
}

__device__ __forceinline__ void load_row4(const int16_t *p, bool valid, int c[4])
{
    int2 w = valid ? *reinterpret_cast<const int2 *>(p) : make_int2(0, 0);
    c[0] = (int16_t)(w.x & 0xFFFF); c[1] = w.x >> 16; c[2] = (int16_t)(w.y & 0xFFFF); c[3] = w.y >> 16;
}

/* Residual of the macroblock, distributed over the wave:
 *   ry[0..3]: luma, lane = 4*blk + row (blk raster 0..15): samples (row, 0..3) of block blk
 *   rc[0..3]: chroma, lanes 0..31: lane = 4*k + row, k = 4*plane + 2*by + bx
 * Must be called by all 64 lanes (quad shuffles).  coef = first coefficient block of the MB. */
/* The coefficient rows a lane needs, fetched ahead of their use (k_recon_inter requests them together with the
 * reference windows): luma row, chroma AC row, chroma DC quartet. */
struct ResidRows { int2 y, c, cdc; int ldc; };     /* ldc: level (lane & 15) of the Intra16x16 luma DC block */
__device__ __forceinline__ ResidRows mb_residual_fetch(uint32_t coded, const int16_t *coef, int lane)
{
    ResidRows r;
    r.y = r.c = r.cdc = make_int2(0, 0);
    r.ldc = 0;
    const int q = lane & 3;
    const int has_ldc = (coded >> 24) & 1, has_cdc = (coded >> 25) & 1;
    if (has_ldc) r.ldc = coef[lane & 15];                        /* wave-uniform; the first block of the macroblock */
    if (coded & 0x0100FFFFu) {                                   /* wave-uniform */
        const int blk = lane >> 2, bx = blk & 3, by = blk >> 2, z = z_of(bx, by);
        const int off = has_ldc + __popc(coded & ((1u << z) - 1u));
        if ((coded >> z) & 1) r.y = *reinterpret_cast<const int2 *>(coef + 16 * off + 4 * q);
    }
    if (coded & 0x02FF0000u) {                                   /* wave-uniform */
        const int k = (lane >> 2) & 7;
        const int base = has_ldc + __popc(coded & 0xFFFFu);
        if (has_cdc) r.cdc = *reinterpret_cast<const int2 *>(coef + 16 * base + 4 * (k >> 2));
        const int off = base + has_cdc + __popc((coded >> 16) & ((1u << k) - 1u));
        if ((coded >> (16 + k)) & 1) r.c = *reinterpret_cast<const int2 *>(coef + 16 * off + 4 * q);
    }
    return r;
}

__device__ __forceinline__ void unpack_row4(int2 w, int c[4])
{
    c[0] = (int16_t)(w.x & 0xFFFF); c[1] = w.x >> 16; c[2] = (int16_t)(w.y & 0xFFFF); c[3] = w.y >> 16;
}

/* returns true in the lanes that hold a residual sample outside [-512,511] (DEVERR_RESIDUAL_RANGE).  LDC = false: the caller
 * never sees an Intra16x16 luma DC block (inter macroblocks) and the code for it is left out. */
template <bool LDC = true>
__device__ __forceinline__ bool mb_residual_compute(uint32_t coded, int qp_y, int qp_c, bool is_i16, const int16_t *coef, int lane,
                                                    const ResidRows &rows, int ry[4], int rc[4])
{
    const int q = lane & 3;
    const int has_ldc = (coded >> 24) & 1, has_cdc = (coded >> 25) & 1;
    ry[0] = ry[1] = ry[2] = ry[3] = 0;
    rc[0] = rc[1] = rc[2] = rc[3] = 0;
    if (coded & 0x0100FFFFu) {                                   /* wave-uniform */
        const int blk = lane >> 2, bx = blk & 3, by = blk >> 2;
        int dc = 0;
        if (LDC && has_ldc) {
            /* 4x4 Hadamard element (by,bx) of the DC block, then the 8.5.10 scaling.  The 16 levels arrived with the other
             * coefficient rows (one per lane, mb_residual_fetch): they are read out of lanes 0..15 into scalar registers —
             * no memory access in the middle of the macroblock */
            const uint32_t neg = 0xA6C0u;                        /* sign rows: 0000 1100 0110 1010 (bit k of row i) */
            const uint32_t nr = (neg >> (4 * by)) & 15, ncl = (neg >> (4 * bx)) & 15;
            int acc = 0;
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int l = 0; l < 4; l++) {
                    const int v = __builtin_amdgcn_readlane(rows.ldc, 4 * k + l);
                    acc += (((nr >> k) ^ (ncl >> l)) & 1) ? -v : v;
                }
            const int ls = level_scale(qp_y % 6, 0), q6 = qp_y / 6;
            dc = q6 >= 2 ? (acc * ls) << (q6 - 2) : (acc * ls + (1 << (1 - q6))) >> (2 - q6);
            if (coded & FJ_CODED_LUMA_DC_RAW) dc = __shfl(rows.ldc, 4 * by + bx);   /* wave-uniform; damaged streams only (framejob.h) */
        }
        unpack_row4(rows.y, ry);
        idct_quad(ry, q, qp_y, is_i16, dc);
    }
    if (coded & 0x02FF0000u) {                                   /* wave-uniform */
        const int k = (lane >> 2) & 7;
        int dc = 0;
        if (has_cdc) {
            const int i = k & 3;
            int cc[4];
            unpack_row4(rows.cdc, cc);
            const int f = cc[0] + ((i & 1) ? -cc[1] : cc[1]) + ((i & 2) ? -cc[2] : cc[2]) + ((i == 1 || i == 2) ? -cc[3] : cc[3]);
            const int ls = level_scale(qp_c % 6, 0), q6 = qp_c / 6;
            dc = q6 >= 1 ? (f * ls) << (q6 - 1) : (f * ls) >> 1;
        }
        unpack_row4(rows.c, rc);
        idct_quad(rc, q, qp_c, true, dc);
    }
    /* un-processed blocks are all zero, so testing every value is exactly the reference's per-block test */
    uint32_t over = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) over |= (uint32_t)(ry[i] + 512) | (uint32_t)(rc[i] + 512);
    return over > 1023u;
}

/* ---- the same residual in PACKED 16-bit arithmetic, luma and chroma in ONE pass (inter macroblocks whose FJ_CODED_WIDE is clear) ----
 * The host's magnitude bound (hd_resid.c: sum of the level magnitudes x the largest scale <= 32735 per plane) proves that every
 * dequantised level, every intermediate of the two butterflies and every "+ 32" sum fits a signed 16-bit half and that the
 * residual lies in [-512, 511]: nothing wraps, no tripwire is needed.  Every register holds the luma value in its low half and —
 * in lanes 0..31, lane = 4 * (chroma block k) + row like the luma lanes' 4 * block + row — the chroma value in its high half, so
 * the chroma transform costs nothing on top of the luma one: dequantisation by one v_pk_mul_lo_u16 with (scale << qp/6) per half,
 * two butterflies of 10 + 16 packed instructions, two quad transposes of 12 (select fused with the DPP move).  The 32-bit form
 * above spends 135 instructions per coded macroblock on two transforms, this one ~80. */
__device__ __forceinline__ void quad_transpose4(uint32_t &v0, uint32_t &v1, uint32_t &v2, uint32_t &v3, int lane)
{
    /* lane q of a quad holds row q in v0..v3 -> holds column q.  Exchange with lane ^ 1, then with lane ^ 2.  Selects are BITWISE
     * with a lane mask in a vector register (v_bitop3_b32: 2.3 cycles per wave64 instruction when two wavefronts share a SIMD),
     * not v_cndmask: its VOP2 form — the only one that can carry a DPP move — takes ~19 cycles when two of them follow each
     * other (tools/probes/op_cost_probe.hip), its VOP3 form 4.3.  Per exchanged pair: one select of what is given away, one
     * v_mov_b32_dpp, two selects of what is kept: 12 + 4 instructions per transpose. */
    uint32_t m1 = (uint32_t)-(lane & 1), m2 = (uint32_t)-((lane >> 1) & 1);
    asm("" : "+v"(m1), "+v"(m2));                        /* (masks of unknown origin: left to itself the compiler turns the bitwise selects back into v_cmp + v_cndmask) */
    auto sel = [](uint32_t m, uint32_t a, uint32_t b) { return (uint32_t)__builtin_amdgcn_bitop3_b32(a, b, m, 0xE4); };      /* (a & m) | (b & ~m) */
    auto x1 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true); };   /* quad_perm [1,0,3,2] */
    auto x2 = [](uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true); };   /* quad_perm [2,3,0,1] */
    {
        const uint32_t r01 = x1(sel(m1, v0, v1)), r23 = x1(sel(m1, v2, v3));      /* odd lanes give v0 / v2 and get the neighbour's v1 / v3 */
        v1 = sel(m1, v1, r01); v0 = sel(m1, r01, v0);
        v3 = sel(m1, v3, r23); v2 = sel(m1, r23, v2);
    }
    {
        const uint32_t r02 = x2(sel(m2, v0, v2)), r13 = x2(sel(m2, v1, v3));
        v2 = sel(m2, v2, r02); v0 = sel(m2, r02, v0);
        v3 = sel(m2, v3, r13); v1 = sel(m2, r13, v1);
    }
}
/* out: the lane's four residual samples as packed pairs — luma (y01, y23), chroma (c01, c23; lanes 0..31).  Must be called by
 * all 64 lanes. */
__device__ __forceinline__ void mb_residual_pk(uint32_t coded, int qp_y, int qp_c, int lane, const ResidRows &rows, s2 &y01, s2 &y23, s2 &c01, s2 &c23)
{
    const int q = lane & 3;
    const bool odd = q & 1;
    /* (scale << qp / 6) per coefficient class, luma | chroma << 16: wave-uniform, scalar registers */
    const int my = qp_y % 6, sy = qp_y / 6, mc = qp_c % 6, sc = qp_c / 6;
    const uint32_t L0 = (uint32_t)(level_scale(my, 0) << sy) | ((uint32_t)(level_scale(mc, 0) << sc) << 16);
    const uint32_t L1 = (uint32_t)(level_scale(my, 1) << sy) | ((uint32_t)(level_scale(mc, 1) << sc) << 16);
    const uint32_t L2 = (uint32_t)(level_scale(my, 2) << sy) | ((uint32_t)(level_scale(mc, 2) << sc) << 16);
    const s2 A = as_s2(odd ? L1 : L0), B = as_s2(odd ? L2 : L1);         /* columns 0, 2 / columns 1, 3 of row q */
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    auto mul = [](uint32_t a, s2 b) { us2 x, y; __builtin_memcpy(&x, &a, 4); __builtin_memcpy(&y, &b, 4); x = x * y; s2 r; __builtin_memcpy(&r, &x, 4); return r; };
    const uint32_t yx = (uint32_t)rows.y.x, yy = (uint32_t)rows.y.y, cx = (uint32_t)rows.c.x, cy = (uint32_t)rows.c.y;
    s2 d0 = mul(perm(cx, yx, 0x05040100u), A), d1 = mul(perm(cx, yx, 0x07060302u), B);
    s2 d2 = mul(perm(cy, yy, 0x05040100u), A), d3 = mul(perm(cy, yy, 0x07060302u), B);
    if (coded & 0x02FF0000u) {                                   /* wave-uniform: chroma has coefficients */
        /* the chroma block's DC replaces element (0, 0) after scaling (8.5.11), as in mb_residual_compute */
        int dc = 0;
        if (coded & FJ_CODED_CHROMA_DC) {
            const int i = (lane >> 2) & 3;
            int cc[4];
            unpack_row4(rows.cdc, cc);
            const int f = cc[0] + ((i & 1) ? -cc[1] : cc[1]) + ((i & 2) ? -cc[2] : cc[2]) + ((i == 1 || i == 2) ? -cc[3] : cc[3]);
            const int ls = level_scale(mc, 0);
            dc = sc >= 1 ? (f * ls) << (sc - 1) : (f * ls) >> 1;
        }
        if (q == 0) d0 = as_s2(perm((uint32_t)dc, as_u32(d0), 0x05040100u));
    }
    {
        const s2 e0 = d0 + d2, e1 = d0 - d2, e2 = (d1 >> pk(1)) - d3, e3 = d1 + (d3 >> pk(1));
        uint32_t f0 = as_u32(e0 + e3), f1 = as_u32(e1 + e2), f2 = as_u32(e1 - e2), f3 = as_u32(e0 - e3);
        quad_transpose4(f0, f1, f2, f3, lane);                   /* lane q: column q, f_k = row k */
        const s2 g0 = as_s2(f0) + as_s2(f2) + pk(32), g1 = as_s2(f0) - as_s2(f2) + pk(32);
        const s2 g2 = (as_s2(f1) >> pk(1)) - as_s2(f3), g3 = as_s2(f1) + (as_s2(f3) >> pk(1));
        uint32_t r0 = as_u32((g0 + g3) >> pk(6)), r1 = as_u32((g1 + g2) >> pk(6)), r2 = as_u32((g1 - g2) >> pk(6)), r3 = as_u32((g0 - g3) >> pk(6));
        quad_transpose4(r0, r1, r2, r3, lane);                   /* back to row q: r_k = sample k, luma | chroma << 16 */
        y01 = as_s2(perm(r1, r0, 0x05040100u)); y23 = as_s2(perm(r3, r2, 0x05040100u));
        c01 = as_s2(perm(r1, r0, 0x07060302u)); c23 = as_s2(perm(r3, r2, 0x07060302u));
    }
}

__device__ __forceinline__ bool mb_residual(uint32_t coded, int qp_y, int qp_c, bool is_i16, const int16_t *coef, int lane, int ry[4], int rc[4])
{
    const ResidRows rows = mb_residual_fetch(coded, coef, lane);
    return mb_residual_compute(coded, qp_y, qp_c, is_i16, coef, lane, rows, ry, rc);
}
/* a tripwire fired: its bit in the sticky error word, and one more EVENT in the counter next to it (the word cannot say that a bit
 * which is already set fired again: the counter can — tests and the per-decoder copy-elision guard look at its delta) */
__device__ __forceinline__ void report_device_error(const FrameDesc &fd, uint32_t bit)
{
    atomicOr(fd.err, bit);
    atomicAdd(fd.err + 1, 1u);
}
__device__ __forceinline__ void report_residual_range(const FrameDesc &fd, bool bad, int lane)
{
    const unsigned long long m = __ballot(bad);
    if (m != 0ull && lane == (int)__ffsll((long long)m) - 1) report_device_error(fd, DEVERR_RESIDUAL_RANGE);
}

/* DPB slot k of the picture's stream.  The slots of a stream are contiguous (engine.hip make_desc), so the address is
 * arithmetic: no lane-indexed table lookup (= a dependent global-memory round trip) in front of the sample loads. */
__device__ __forceinline__ uint8_t *slot_ptr(const FrameDesc &fd, uint32_t k)
{
    return fd.slot[0] + (size_t)k * ((size_t)fd.wmb * fd.hmb * 384u);
}

/* ------------------------------------------------------------------ frame layout in HBM: macroblock tiles
 * A frame is its macroblocks in address order, 384 contiguous bytes each: Y[16][16] | Cb[8][8] | Cr[8][8] — three
 * 128-byte lines per macroblock (the slots are 128-byte aligned).  Every kernel of the path works macroblock by
 * macroblock, and what they pay for is the number of cache LINES a wavefront touches, not bytes: in the reference's
 * planar frame (image.h:46-55) the 16 + 16 row pieces of a macroblock lie in 32 different lines, here in 3; a
 * neighbour's last columns are 2 lines instead of 16, its last rows 1 line.  (Round 1 measured the per-CU line-request
 * rate as the limit of k_frame_dbk — its time did not change between 4 and 16 wavefronts per picture nor with a 21 %
 * shorter dependency chain — and 2.9 -> 4.6 TB/s for k_copy's pattern in tools/probes/layout_probe.hip.)
 * The reference's planar I420 is produced where pictures leave the device (k_detile / k_output / k_convert). */
constexpr int TILE = 384, T_CB = 256, T_CR = 320;
__device__ __forceinline__ size_t luma_at(int wmb, int x, int y)
{
    return (size_t)((y >> 4) * wmb + (x >> 4)) * TILE + ((y & 15) << 4) + (x & 15);
}
__device__ __forceinline__ size_t chroma_at(int wmb, int plane, int x, int y)
{
    return (size_t)((y >> 3) * wmb + (x >> 3)) * TILE + T_CB + (plane << 6) + ((y & 7) << 3) + (x & 7);
}
/* 4 luma samples x..x+3 of row y (inside the picture): one load, or two when they straddle two tiles */
__device__ __forceinline__ uint32_t luma4_at(const uint8_t *__restrict__ f, int wmb, int x, int y)
{
    const int c = x & 15;
    const uint8_t *t = f + (size_t)((y >> 4) * wmb + (x >> 4)) * TILE + ((y & 15) << 4);
    if (c <= 12) return load_u32_unaligned(t + c);
    const unsigned long long v = (unsigned long long)load_u32_unaligned(t + 12) | ((unsigned long long)load_u32_unaligned(t + TILE) << 32);
    return (uint32_t)(v >> (8 * (c - 12)));
}


/* ------------------------------------------------------------------ deblocking records */
/* concealed macroblocks are filtered as Intra4x4 (reference src/h264bsd_conceal.c:309) */
__device__ __forceinline__ bool is_intra_kind(int k)
{
    return k == FJ_MB_I4x4 || k == FJ_MB_I16x16 || k == FJ_MB_IPCM || k == FJ_MB_CONCEAL_I || k == FJ_MB_CONCEAL_P || k == FJ_MB_STALE;
}

__device__ __forceinline__ void wave_sync()
{
    /* LDS operations of one wavefront execute in issue order; only the compiler has to be told */
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

} // namespace h264k
