/*
 * kernels.hip.h — hand-written gfx950 (CDNA4, wave64) kernels of the macroblock reconstruction path.
 *
 * Frames live in HBM as MACROBLOCK TILES: 384 contiguous bytes per macroblock (Y 16x16 | Cb 8x8 | Cr 8x8), tiles in
 * macroblock address order — see "frame layout in HBM" below.  Planar I420 exists only where pictures leave the device.
 *
 *   k_copy        : inter macroblocks that are whole-sample copies without residual (host-built list of runs of up to
 *                   8 tiles; 61 % of the macroblocks of the 1080p stream): a zero-displacement run is one contiguous
 *                   block of count x 384 bytes, 16 bytes per lane per access, every load before the first store.
 *                       reference: the P_Skip / integer-mv path of src/h264bsd_inter_prediction.c:361-482,
 *                                  h264bsdFillBlock src/h264bsd_reconstruct.c:2244, src/h264bsd_image.c:81
 *   k_recon_inter : every other inter macroblock, one wavefront each (= one workgroup: the wavefronts share nothing), list-driven.
 *                   Reference windows staged in LDS tile row by tile row (aligned 16-byte loads), 6-tap luma on packed
 *                   sample pairs / bilinear chroma, dequant + 4x4 inverse transform with DPP quad transposes, residual
 *                   add; the macroblock leaves as its tile (24 x 16 bytes).
 *                       reference: src/h264bsd_reconstruct.c, src/h264bsd_inter_prediction.c:361-482,
 *                                  src/h264bsd_transform.c, src/h264bsd_image.c:172
 *   k_dbk         : boundary strengths + alpha / beta / tc0 VALUES of every macroblock the host could not prove strength-free
 *                   (metadata only, four MBs per wavefront) -> 48-byte deblocking records + one flag byte per MB
 *                   (DBKF_*: filtered at all / touches the left / the upper neighbour / has an active inner edge).
 *                       reference: src/h264bsd_deblocking.c:1187-1541
 *   k_frame_intra : ONE workgroup per picture (a picture never leaves its CU): intra and concealed macroblocks,
 *                   dataflow-scheduled in LDS (dependency counters + ready queue, a free wavefront takes one ready MB).
 *                       reference: src/h264bsd_intra_prediction.c, src/h264bsd_conceal.c
 *   k_frame_dbk   : ONE workgroup per picture: the in-loop filter, dataflow-scheduled in LDS at macroblock-EDGE
 *                   granularity (a macroblock waits only for the neighbours whose samples it really shares), eight-lane
 *                   workers (up to eight macroblocks per wavefront step), two ready lists, packed 16-bit arithmetic.
 *                       reference: src/h264bsd_deblocking.c:575-1745
 *   k_convert / k_output / k_detile : YUV420 -> RGBA / BGRA / YCbCrA, cropped windows, tiles -> planar I420.
 *                       reference: src/h264bsd_decoder.c:1163-1370, :970-1001
 *   k_checksum    : position-weighted 64-bit checksum of a frame in planar order (on-device verification).
 *
 * Everything is integer arithmetic on u8 samples / i16 levels / i32 intermediates.  The one contraction on the path — the 6-tap
 * filter, a Toeplitz product — is a fifth of k_recon_inter's vector instructions, and an i8 MFMA would take over half of THAT
 * only for the one-dimensional classes ("the matrix pipe"): no MFMA.
 */
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "framejob.h"

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
 * bands | n4 "done" bytes of k_frame_intra's row bands | exit counters of the two kernels (u32 each) — n4 = n_mbs rounded up
 * to a multiple of 4.  Flags, done bytes and counters are zero between pictures (the last band to leave cleans up). */
#define DBK_SCRATCH_BYTES(n_mbs) ((size_t)(n_mbs) * (DBK_REC_BYTES + 3) + 64)

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
__device__ __forceinline__ uint8_t *scratch_done(const FrameDesc &fd, int which)      /* 0: k_frame_dbk, 1: k_frame_intra */
{
    return scratch_flags(fd) + (size_t)(1 + which) * ((fd.n_mbs + 3u) & ~3u);
}
__device__ __forceinline__ uint32_t *scratch_exits(const FrameDesc &fd, int which)
{
    return reinterpret_cast<uint32_t *>(scratch_flags(fd) + (size_t)3 * ((fd.n_mbs + 3u) & ~3u)) + which;
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

/* Boundary strengths (8.7.2.1) + threshold indices from metadata only; one macroblock per 32 lanes.
 * reference: GetBoundaryStrengths / GetLumaEdgeThresholds / GetChromaEdgeThresholds,
 * src/h264bsd_deblocking.c:1187-1541 */
#ifndef DBK_WGS
#define DBK_WGS 32           /* workgroups per picture: each walks the picture's index list with stride 8 * DBK_WGS */
#endif
#ifndef DBK_WG_WAVES
#define DBK_WG_WAVES 4       /* wavefronts per workgroup of k_dbk (four macroblocks each) */
#endif
__global__ __launch_bounds__(64 * DBK_WG_WAVES) void k_dbk(const FrameDesc *__restrict__ frames)
{
    /* Tables 8-16 / 8-17 in LDS (alpha[64] | beta[64] | tc0[64] as dwords {bS 1, bS 2, bS 3, 0}): a lane-indexed __constant__
     * lookup is a global load */
    __shared__ uint32_t s_tab[32 + 64];
    if (threadIdx.x < 64) {
        const uint32_t t = threadIdx.x, ok = t < 52;
        reinterpret_cast<uint8_t *>(s_tab)[t] = ok ? c_alpha[t] : 0;
        reinterpret_cast<uint8_t *>(s_tab)[64 + t] = ok ? c_beta[t] : 0;
        s_tab[32 + t] = ok ? (uint32_t)c_tc0[t][0] | ((uint32_t)c_tc0[t][1] << 8) | ((uint32_t)c_tc0[t][2] << 16) : 0u;
    }
    __syncthreads();
    const FrameDesc &fd = FD_REF(frames, blockIdx.y);
    /* One macroblock per 16 lanes, four per wavefront: lane m of a group owns byte m of the 16-byte strength array, i.e. the
     * two segments k = 2*kh, 2*kh+1 of edge (dir, e) — m = 8*dir + 2*e + kh.  (Two macroblocks per wavefront with one segment
     * per lane cost the wavefront 1.7 times the instructions per macroblock: the index walk, the record decode, the threshold
     * indices and the stores are per wavefront, not per segment.) */
    const int m = threadIdx.x & 15;
    const int wmb = fd.wmb;
    const uint32_t n_dbk = fd.n_dbk;
    /* A fixed number of workgroups per picture walks the index list with a stride; the next index is requested while
     * the current macroblock is worked on, and everything a macroblock needs — its record, the records of its left and
     * upper neighbours — is requested TOGETHER, whether the flags in the record (still in flight) will want it or not: two
     * dependent memory round trips per macroblock, a third for the vectors of partitioned macroblocks. */
    uint32_t di = blockIdx.x * (4 * DBK_WG_WAVES) + (threadIdx.x >> 4);
    const bool live0 = di < n_dbk;
    if (__ballot(live0) == 0ull) return;
    uint32_t mb = live0 ? fd.dbki[di] : 0u;
    bool live = live0;
  for (;;) {
    const uint32_t ndi = di + (4u * DBK_WG_WAVES) * gridDim.x;
    uint32_t nmb = mb;
    const bool nlive = live && ndi < n_dbk;
    if (nlive) nmb = fd.dbki[ndi];
    const uint32_t mbx = mb - mb_row(fd, mb) * (uint32_t)wmb;
    const uint32_t mbl = mbx ? mb - 1 : mb, mbt = mb >= (uint32_t)wmb ? mb - wmb : mb;    /* in-picture stand-ins */
    FjMbRec q, pl, pt;
    const int dir = m >> 3, e = (m >> 1) & 3, kh = m & 1;
    int qx[2], qy[2], px[2], py[2];
    uint32_t mva[2], mvb[2];
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
        const int k = 2 * kh + kk;
        qx[kk] = dir ? k : e; qy[kk] = dir ? e : k;
        px[kk] = dir ? k : (e ? e - 1 : 3); py[kk] = dir ? (e ? e - 1 : 3) : k;
    }
    {
        /* the three records as whole 16-byte pieces, the motion vectors of both sides — all requested before anything is
         * looked at (a struct copy lets the compiler fetch member by member where each is used: five dependent round trips) */
        const H264K_GLOBAL uint8_t *rq = (const H264K_GLOBAL uint8_t *)(fd.recs + mb), *rl = (const H264K_GLOBAL uint8_t *)(fd.recs + mbl),
                                   *rt = (const H264K_GLOBAL uint8_t *)(fd.recs + mbt);
        uint4 w[6] = { ld16g(rq), ld16g(rq + 16), ld16g(rl), ld16g(rl + 16), ld16g(rt), ld16g(rt + 16) };
#pragma unroll
        for (int i = 0; i < 6; i++) asm volatile("" : "+v"(w[i].x), "+v"(w[i].y), "+v"(w[i].z), "+v"(w[i].w));
        __builtin_memcpy(&q, &w[0], 32); __builtin_memcpy(&pl, &w[2], 32); __builtin_memcpy(&pt, &w[4], 32);
    }
    {
        /* motion vectors on both sides of the lane's two segments.  A macroblock with ONE vector carries it in its record
         * (FJ_PRED_UNIFORM_MV: five of six — nothing more to fetch); the others have their sixteen in the sparse section, one
         * more dependent round trip for the lanes that look at such a macroblock */
        const FjMbRec &pr = e ? q : (dir ? pt : pl);
        const bool q_one = (q.pred & FJ_PRED_UNIFORM_MV) || q.kind != FJ_MB_INTER, p_one = (pr.pred & FJ_PRED_UNIFORM_MV) || pr.kind != FJ_MB_INTER;
        const uint32_t q_mv = q.kind == FJ_MB_INTER ? (uint32_t)(uint16_t)q.mv[0] | ((uint32_t)(uint16_t)q.mv[1] << 16) : 0u;
        const uint32_t p_mv = pr.kind == FJ_MB_INTER ? (uint32_t)(uint16_t)pr.mv[0] | ((uint32_t)(uint16_t)pr.mv[1] << 16) : 0u;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
            mva[kk] = q_mv; mvb[kk] = p_mv;
            if (!q_one) mva[kk] = *(const H264K_GLOBAL uint32_t *)(fd.mvx + 32 * (size_t)q.mvx + 2 * (4 * qy[kk] + qx[kk]));
            if (!p_one) mvb[kk] = *(const H264K_GLOBAL uint32_t *)(fd.mvx + 32 * (size_t)pr.mvx + 2 * (4 * py[kk] + px[kk]));
        }
    }
    uint8_t *out = fd.dbk + (size_t)mb * DBK_REC_BYTES;
    uint8_t *any_out = fd.dbk + (size_t)fd.n_mbs * DBK_REC_BYTES + mb;
    const bool filtered = live && q.dbk && q.kind != FJ_MB_ABSENT;
    if (live && !filtered && m == 0) { *reinterpret_cast<uint16_t *>(out + 46) = 0; *any_out = 0; }
    uint32_t bs2 = 0;                                              /* the lane's two strengths: low and high nibble of byte m */
    /* k_frame_dbk relies on it for its addresses: a left / upper macroblock edge is only ever active where that neighbour exists
     * (the host never says otherwise: GetMbFilteringFlags, deblocking.c:289-320 — enforced here for hand-built jobs) */
    const bool f_left = (q.dbk & FJ_DBK_LEFT) && mbx, f_top = (q.dbk & FJ_DBK_TOP) && mb >= (uint32_t)wmb;
    if (filtered) {
        const bool edge_on = e ? true : (dir ? f_top : f_left);
        if (edge_on) {
            const int p_kind = e ? q.kind : (dir ? pt.kind : pl.kind);
            const int parts = (q.pred >> FJ_PRED_PARTS_SHIFT) & 3;
            const uint32_t p_coded = e ? q.coded : (dir ? pt.coded : pl.coded);
            uint32_t qrefs, prefs, t0, t1;
            __builtin_memcpy(&qrefs, q.ref_slot, 4);
            __builtin_memcpy(&t0, pl.ref_slot, 4);
            __builtin_memcpy(&t1, pt.ref_slot, 4);
            prefs = e ? qrefs : (dir ? t1 : t0);
            const bool intra_edge = is_intra_kind(q.kind) || is_intra_kind(p_kind);
            /* inside a macroblock motion is compared only across the partition boundaries its type has (FJ_PARTS_*,
             * reference deblocking.c:1266-1345) */
            const bool no_motion_edge = e && (parts == FJ_PARTS_16x16 || (parts == FJ_PARTS_16x8 && !(dir == 1 && e == 2)) || (parts == FJ_PARTS_8x16 && !(dir == 0 && e == 2)));
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                int my_bs;
                if (intra_edge) my_bs = e ? 3 : 4;
                else if (((q.coded >> z_of(qx[kk], qy[kk])) & 1) || ((p_coded >> z_of(px[kk], py[kk])) & 1)) my_bs = 2;
                else if (no_motion_edge) my_bs = 0;
                else if (((qrefs >> (8 * ((qy[kk] >> 1) * 2 + (qx[kk] >> 1)))) & 255u) != ((prefs >> (8 * ((py[kk] >> 1) * 2 + (px[kk] >> 1)))) & 255u)) my_bs = 1;
                else {
                    const int ax = (int16_t)(mva[kk] & 0xFFFFu), ay = (int32_t)mva[kk] >> 16, bx2 = (int16_t)(mvb[kk] & 0xFFFFu), by2 = (int32_t)mvb[kk] >> 16;
                    my_bs = (abs(ax - bx2) >= 4 || abs(ay - by2) >= 4) ? 1 : 0;
                }
                bs2 |= (uint32_t)my_bs << (4 * kk);
            }
        }
    }
    /* bytes -> dwords: lanes m = 0,4,8,12 of a group end up with one dword each (DPP row_shl:1/2: lane i reads lane i+1/2) */
    uint32_t v = bs2;
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x101, 0xF, 0xF, true) << 8;
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x102, 0xF, 0xF, true) << 16;
    const unsigned long long bal = __ballot(bs2 != 0u);
    const uint32_t bal16 = (uint32_t)(bal >> (threadIdx.x & 48)) & 0xFFFFu;      /* bit m: byte m of this macroblock is non-zero */
    const bool any = bal16 != 0u;
    /* scheduling flags of k_frame_dbk: does this macroblock touch its left / upper neighbour at all? (bytes 0,1 = left edge, 8,9 = upper) */
    const uint32_t sched = (any ? DBKF_ANY : 0u) | ((bal16 & 0x0003u) ? DBKF_LEFT : 0u) | ((bal16 & 0x0300u) ? DBKF_TOP : 0u) |
                           ((bal16 & 0xFCFCu) ? DBKF_INNER : 0u);
    if (filtered) {
        if ((m & 3) == 0) *reinterpret_cast<uint32_t *>(out + m) = v;
        /* thresholds: lane m < 6 computes indexA and indexB of class m (luma left / top / inner, chroma left / top / inner), looks
         * alpha, beta and the three tc0 up and stores the class's dword and its bS-3 byte */
        if (m < 6) {
            const int c = m;                                           /* class */
            const int side = c % 3;                                    /* 0: across the left edge, 1: across the upper edge, 2: inside */
            const int pqp = side == 0 ? (int)pl.qp_y : side == 1 ? (int)pt.qp_y : (int)q.qp_y;
            int a = (int)q.qp_y, b = pqp;
            if (c >= 3) {                                              /* chroma: QPc of both sides with the CURRENT macroblock's offset (deblocking.c:1501,1523) */
                a = qpc_of(clip3(0, 51, a + q.cqp_off));
                b = qpc_of(clip3(0, 51, b + q.cqp_off));
            }
            const int qpav = (a + b + 1) >> 1;
            const int ia = clip3(0, 51, qpav + q.alpha_off), ib = clip3(0, 51, qpav + q.beta_off);
            const uint32_t t = s_tab[32 + ia];
            *reinterpret_cast<uint32_t *>(out + 16 + 4 * c) = (uint32_t)reinterpret_cast<const uint8_t *>(s_tab)[ia] |
                ((uint32_t)reinterpret_cast<const uint8_t *>(s_tab)[64 + ib] << 8) | ((t & 0xFFFFu) << 16);
            out[40 + c] = (uint8_t)(t >> 16);
        }
        if (m == 12) {
            *reinterpret_cast<uint16_t *>(out + 46) = (uint16_t)((f_left ? FJ_DBK_LEFT : 0u) | (f_top ? FJ_DBK_TOP : 0u) | (q.dbk & FJ_DBK_INNER) | (any ? 0x100u : 0u));
            *any_out = (uint8_t)sched;
        }
    }
    if (__ballot(nlive) == 0ull) return;
    di = ndi; mb = nmb; live = nlive;
  }
}

/* ------------------------------------------------------------------ whole-sample copy macroblocks */
/* List entries are runs of up to 8 horizontally adjacent MBs with one displacement.  With macroblock tiles a run whose
 * displacement is zero (P_Skip with zero motion: almost all of them) is ONE contiguous block of count x 384 bytes in
 * the reference frame and in the current one: 24 x count 16-byte pieces, up to six per lane, every load issued before
 * the first store.  Displaced (and clamped) runs gather their samples 4 at a time. */
#ifndef COPY_WGS
#define COPY_WGS 16          /* workgroups per picture: each walks the picture's run list with stride 4 * COPY_WGS.  k_dbk runs next to
                                k_copy and k_recon_inter, and the three together are bound by instruction issue: with 8 / 16 / 24 / 32 / 48
                                workgroups k_copy takes 14.3 / 18.4 / 21.3 / 25.7 / 28.1 ms per step (5.0 TB/s with 8) and k_recon_inter
                                52.3 / 48.3 / 45.3 / 41.8 / 40.7: the sum stays at 66.6-68.8, the step at 139.6-141.8 ms.  Two runs per
                                loop trip (all loads of both before the first store) lost: 37.4 ms */
#endif
__global__ __launch_bounds__(256) void k_copy(const FrameDesc *__restrict__ frames)
{
    const FrameDesc &fd = FD_REF(frames, blockIdx.y);
    const int lane = threadIdx.x & 63;
    const int wmb = fd.wmb;
    const uint32_t n_copy = fd.n_copy;
    /* a fixed number of workgroups per picture, every wavefront walks the run list with a stride: no workgroup is
     * launched for nothing (the grid used to be sized by the longest list of the tick), and the next list entry is
     * requested while the current run moves.  The list position is wave-uniform and the list read-only: entries come
     * through the scalar cache (one s_load_dwordx2), not through the vector memory pipeline the samples use. */
    uint32_t ci = blockIdx.x * 4 + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (ci >= n_copy) return;
    const H264K_CONST u32x2 *list = (const H264K_CONST u32x2 *)fd.copy;
    u32x2 ew = list[ci];
  for (;;) {
    const uint32_t nci = ci + 4u * gridDim.x;
    u32x2 nw = ew;
    if (nci < n_copy) nw = list[nci];
    FjCopy e;
    __builtin_memcpy(&e, &ew, 8);
    const int cnt = e.count;
    const uint8_t *ref = slot_ptr(fd, e.slot);
    if ((e.dx | e.dy) == 0) {
        /* every load of the run before its first store, and none of them inside a branch (a load whose result leaves an
         * `if` is waited for at the end of that `if`: the three pieces used to take three memory round trips): lanes behind
         * the end of the run load its last piece again and drop it */
        const H264K_GLOBAL uint8_t *src = (const H264K_GLOBAL uint8_t *)ref + (size_t)e.mb * TILE;
        H264K_GLOBAL uint8_t *dst = (H264K_GLOBAL uint8_t *)fd.cur + (size_t)e.mb * TILE;
        const int n16 = cnt * (TILE / 16);
        constexpr int PIECES = (FJ_COPY_RUN * (TILE / 16) + 63) / 64;
        uint4 v[PIECES];
#pragma unroll
        for (int j = 0; j < PIECES; j++) v[j] = ld16g(src + 16 * min(lane + 64 * j, n16 - 1));
#pragma unroll
        for (int j = 0; j < PIECES; j++) if (lane + 64 * j < n16) st16g(dst + 16 * (lane + 64 * j), v[j]);
    } else {
    /* displaced: clamp-to-edge sample gather (h264bsdFillBlock, reconstruct.c:2244), lane = (row, 4-sample piece) */
    const int W = wmb * 16, H = fd.hmb * 16, CW = W >> 1, CH = H >> 1;
    const int mby = (int)mb_row(fd, e.mb), mbx = (int)e.mb - mby * wmb;
    for (int m = 0; m < cnt; m++) {
        const int x0 = (mbx + m) * 16 + e.dx, y0 = mby * 16 + e.dy;
        uint8_t *dt = fd.cur + (size_t)(e.mb + m) * TILE;
        {
            const int r = lane >> 2, q = lane & 3, yy = clip3(0, H - 1, y0 + r);
            uint32_t a = 0;
            if (x0 >= 0 && x0 + 16 <= W) a = luma4_at(ref, wmb, x0 + 4 * q, yy);
            else {
#pragma unroll
                for (int i = 0; i < 4; i++) a |= (uint32_t)ref[luma_at(wmb, clip3(0, W - 1, x0 + 4 * q + i), yy)] << (8 * i);
            }
            *reinterpret_cast<uint32_t *>(dt + r * 16 + 4 * q) = a;
        }
        if (lane < 32) {
            const int plane = lane >> 4, r = (lane >> 1) & 7, half = lane & 1, cy = clip3(0, CH - 1, (y0 >> 1) + r);
            uint32_t b2 = 0;
#pragma unroll
            for (int i = 0; i < 4; i++) b2 |= (uint32_t)ref[chroma_at(wmb, plane, clip3(0, CW - 1, (x0 >> 1) + 4 * half + i), cy)] << (8 * i);
            *reinterpret_cast<uint32_t *>(dt + T_CB + plane * 64 + r * 8 + 4 * half) = b2;
        }
    }
    }
    if (nci >= n_copy) return;
    ci = nci; ew = nw;
  }
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
    FjMbRec rec;                                     /* only the QPs (and, partitioned, the references) are needed from it */
    {
        const uint4 w = ld16c((const H264K_CONST FjMbRec *)fd.recs + mb), w2 = ld16c((const H264K_CONST uint8_t *)((const H264K_CONST FjMbRec *)fd.recs + mb) + 16);
        __builtin_memcpy(&rec, &w, 16);
        __builtin_memcpy(reinterpret_cast<uint8_t *>(&rec) + 16, &w2, 16);
    }
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
    const int16_t *coef = (const int16_t *)((const H264K_CONST int16_t *)fd.coefs + 16 * (size_t)ge.coef_idx);
    H264K_GLOBAL uint8_t *cur = (H264K_GLOBAL uint8_t *)fd.cur;
    const int blk = lane >> 2, row = lane & 3, bx = blk & 3, by = blk >> 2;
    const bool uniform = PATH == 0, quadwise = PATH == 1;
    uint32_t refs = ge.slot * 0x01010101u, mv_mine = 0;
    const uint32_t mv0 = (uint32_t)(uint16_t)ge.mvx | ((uint32_t)(uint16_t)ge.mvy << 16);
    if (!uniform) {
        __builtin_memcpy(&refs, rec.ref_slot, 4);
        if (!quadwise) mv_mine = *reinterpret_cast<const uint32_t *>(mvs + 2 * blk);
    }

    /* the coefficient rows are requested right behind the reference windows (whose loads come first: they are needed
     * first) and consumed after the prediction */
    ResidRows rrows;
    if (!uniform) rrows = mb_residual_fetch(ge.coded, coef, lane);
    s2 pl01 = pk(0), pl23 = pk(0);               /* the lane's four luma prediction samples, two packed pairs */
    int pc[4] = { 0, 0, 0, 0 };
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
#if defined(INTER_WHATIF) && (INTER_WHATIF & 4)
        if (ge.coded != 0xFFFFFFFFu) continue;
#endif
#if defined(INTER_WHATIF) && (INTER_WHATIF & 1)
        const uint4 vl = make_uint4(lane, mb, ge.coded, lx);
#else
        const uint4 vl = ld16g(ref + (l_on ? luma_at(wmb, lx, yi + lr) : (size_t)0));
#endif
        const int cp = lane >= 18, rem = cp ? lane - 18 : lane, cr = rem >> 1, ck = rem & 1, cx = cxs + 8 * ck;
        const bool c_on = cfast && lane < 36 && cx < CW;
#if defined(INTER_WHATIF) && (INTER_WHATIF & 1)
        const uint2 vc = make_uint2(lane, cx);
        rrows.y = rrows.c = rrows.cdc = make_int2(lane, mb); rrows.ldc = 0;
#else
        const uint2 vc = ld8g(ref + (c_on ? chroma_at(wmb, cp, cx, cyi + cr) : (size_t)0));
        rrows = mb_residual_fetch(ge.coded, coef, lane);
#endif
#if defined(INTER_WHATIF) && (INTER_WHATIF & 2)
        {
            H264K_GLOBAL uint8_t *T2 = cur + (size_t)mb * TILE;
            *reinterpret_cast<H264K_GLOBAL uint32_t *>(T2 + (by * 4 + row) * 16 + bx * 4) = vl.x ^ vl.y ^ vl.z ^ vl.w ^ (uint32_t)rrows.y.x ^ (uint32_t)rrows.c.x ^ (uint32_t)rrows.cdc.x;
            if (lane < 32) *reinterpret_cast<H264K_GLOBAL uint32_t *>(T2 + T_CB + 4 * lane) = vc.x ^ vc.y;
            continue;
        }
#endif
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
        if (lane < 32) {
            const int k = lane >> 2, plane = k >> 2, cbx = k & 1, cby = (k >> 1) & 1;
            const int cy = cby * 4 + row, cx0 = cbx * 4;
            const uint8_t *s0 = lc + plane * 9 * IC_STRIDE + cy * IC_STRIDE + (cxi - cxs) + cx0, *s1 = s0 + IC_STRIDE;
            int a[5], b[5];
#pragma unroll
            for (int i = 0; i < 5; i++) { a[i] = s0[i]; b[i] = s1[i]; }
            chroma_from_rows(a, b, mvx & 7, mvy & 7, pc);
            chroma_from_rows(a + 2, b + 2, mvx & 7, mvy & 7, pc + 2);
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
        if (lane < 32) {
            const int k = lane >> 2, plane = k >> 2, cbx = k & 1, cby = (k >> 1) & 1;
            const int q = cby * 2 + cbx;                             /* a 4x4 chroma block = one luma quadrant */
            const uint32_t mvm = q == 0 ? mvq[0] : q == 1 ? mvq[1] : q == 2 ? mvq[2] : mvq[3];
            const int mvx = (int16_t)(mvm & 0xFFFFu), mvy = (int32_t)mvm >> 16;
            const int cxi = mbx * 8 + 4 * (q & 1) + (mvx >> 3);
            const uint8_t *s0 = cq + (q * 2 + plane) * 5 * QC_STRIDE + row * QC_STRIDE + (cxi & 7), *s1 = s0 + QC_STRIDE;
            int a[5], b[5];
#pragma unroll
            for (int i = 0; i < 5; i++) { a[i] = s0[i]; b[i] = s1[i]; }
            chroma_from_rows(a, b, mvx & 7, mvy & 7, pc);
            chroma_from_rows(a + 2, b + 2, mvx & 7, mvy & 7, pc + 2);
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
        }
    }

    /* (an unconditional use of the coefficient rows here — they arrived long ago — keeps the compiler from sinking their loads
     * into the residual code, where every coded macroblock would wait for a second memory round trip) */
    asm volatile("" :: "v"(rrows.y.x), "v"(rrows.y.y), "v"(rrows.c.x), "v"(rrows.c.y), "v"(rrows.cdc.x), "v"(rrows.cdc.y));
#ifdef H264K_INTER_PROFILE
    asm volatile("" :: "v"(pl01), "v"(pl23), "v"(pc[0]), "v"(pc[1]), "v"(pc[2]), "v"(pc[3]));
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
        chroma_dw = pack4(pc[0], pc[1], pc[2], pc[3]);
    } else if (!(ge.coded & FJ_CODED_WIDE)) {                    /* wave-uniform: the host proved that 16 bits hold every intermediate */
        s2 y01, y23, c01, c23;
        mb_residual_pk(ge.coded, rec.qp_y, rec.qp_c, lane, rrows, y01, y23, c01, c23);
        const s2 lo = pk(0), hi = pk(255);
        const s2 l01 = pk_clip(lo, hi, pl01 + y01), l23 = pk_clip(lo, hi, pl23 + y23);
        const s2 k01 = pk_clip(lo, hi, as_s2((uint32_t)pc[0] | ((uint32_t)pc[1] << 16)) + c01), k23 = pk_clip(lo, hi, as_s2((uint32_t)pc[2] | ((uint32_t)pc[3] << 16)) + c23);
        luma_dw = perm(as_u32(l23), as_u32(l01), 0x06040200u);
        chroma_dw = perm(as_u32(k23), as_u32(k01), 0x06040200u);
    } else {
        int ry[4], rc[4];
        report_residual_range(fd, mb_residual_compute<false>(ge.coded, rec.qp_y, rec.qp_c, false, coef, lane, rrows, ry, rc), lane);
        luma_dw = pack4(clip255(pl01.x + ry[0]), clip255(pl01.y + ry[1]), clip255(pl23.x + ry[2]), clip255(pl23.y + ry[3]));
        chroma_dw = pack4(clip255(pc[0] + rc[0]), clip255(pc[1] + rc[1]), clip255(pc[2] + rc[2]), clip255(pc[3] + rc[3]));
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
#if defined(INTRA_WHATIF) && (INTRA_WHATIF & 1)      /* timing experiment: no neighbour / coefficient loads */
    L.nb_y = lane; L.nb_c = lane + 1; L.rows.y = L.rows.c = L.rows.cdc = make_int2(lane, 1); (void)Y; (void)cross;
    return;
#endif
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
#if defined(INTRA_WHATIF) && (INTRA_WHATIF & 2)      /* timing experiment: no residual arithmetic */
    ry[0] = ry[1] = ry[2] = ry[3] = L.rows.y.x & 7; rc[0] = rc[1] = rc[2] = rc[3] = L.rows.c.x & 7;
#else
    report_residual_range(fd, mb_residual_compute(rec.coded, rec.qp_y, rec.qp_c, rec.kind == FJ_MB_I16x16, coef, lane, L.rows, ry, rc), lane);
#endif

    if (nb_y_at >= 0) tile[nb_y_at] = (uint8_t)nb_y;
    if (nb_c_at >= 0) ctile0[nb_c_at] = (uint8_t)nb_c;
    wave_sync();
    const unsigned long long i1 = ITICK();

#if defined(INTRA_WHATIF) && (INTRA_WHATIF & 4)      /* timing experiment: no luma prediction */
    if (true) {
        res_defer = nullptr;
        put4(Y + (by * 4 + row) * 16 + bx * 4, pack4(ry[0] & 255, ry[1] & 255, ry[2] & 255, (ry[3] + tile[4 + lane]) & 255), wt);
    } else
#endif
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

/* ------------------------------------------------------------------ deblocking */
/* A deblocking WORKER is an eighth of a wavefront: 8 lanes per macroblock, up to eight macroblocks per wavefront step.
 * Lane l of a worker owns, in the vertical-edge pass, luma rows 2l, 2l+1 and then chroma rows 2(l&3), 2(l&3)+1 of plane l>>2;
 * in the horizontal-edge pass luma columns 2l, 2l+1 and then chroma columns 2(l&3), 2(l&3)+1 of plane l>>2 — two sample
 * lines per register (packed 16-bit), the four luma edges and then the two chroma edges of a direction one after the
 * other.  (Round 3 gave a macroblock 16 lanes, half of them chroma lanes that idled through two of the four edge slots, and
 * paid the per-step overhead — claim, addresses, record decode, release — once per FOUR macroblocks; the picture's compute
 * unit is bound by VALU issue, so what counts is wave instructions per macroblock.)
 * The worker's LDS tile is only the transposition medium between the two passes: the vertical pass takes its rows from the
 * registers the macroblock was loaded into and writes single bytes (ds_write_b8 / _d16_hi: no VALU packing), the horizontal
 * pass reads single bytes into register halves (ds_read_u8_d16 / _d16_hi: no VALU unpacking) and writes back what its
 * active edges changed. */
constexpr int LS = 48, LX = 16;                      /* deblock luma tile: 20 rows (4 above + 16) of LS bytes; the macroblock's columns at bytes LX .. LX+15 (16-byte
                                                        aligned: a row is one ds_read / ds_write_b128), the four columns to its left at LX-4 .. LX-1 */
constexpr int CS = 16, CX = 8;                       /* deblock chroma tiles: 2 planes x 10 rows (2 above + 8) of CS bytes; columns at CX .. CX+7, left strip at CX-4 .. CX-1 */
constexpr int WORKER_LDS = 20 * LS + 2 * 10 * CS + 16;   /* 1296 bytes = 324 dwords: the eight workers of a wavefront start four banks apart */
constexpr int DBK_LANES = 8;                         /* lanes per worker */

/* ---- single bytes from the HALVES of a register to LDS (ds_write_b8 / ds_write_b8_d16_hi): a packed pair of samples that
 * belong to different rows (or columns) of the tile leaves as two LDS instructions and no VALU work.  Left to itself the
 * compiler fuses neighbouring byte stores into 16-bit ones and spends three or four VALU instructions per pair building them
 * — on the pipe this kernel is bound by.  (The other direction does not exist here: with SRAM ECC a d16 LOAD clears the other
 * half of its register instead of keeping it — tried, every even column came back 0 — so the horizontal pass reads 16-bit
 * pairs and spreads them with one v_perm_b32 each.)  The compiler does not see the LDS traffic of an asm statement: the
 * statements are volatile and clobber "memory", which keeps them in order with its own LDS accesses; LDS instructions of one
 * wavefront execute in order. */
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)(const H264K_LDS uint8_t *)p; }
template <int OFF_LO, int OFF_HI>
__device__ __forceinline__ void lds_st_pair(uint32_t addr, s2 v)
{
    asm volatile("ds_write_b8 %0, %1 offset:%2\n\tds_write_b8_d16_hi %0, %1 offset:%3" :: "v"(addr), "v"(v), "n"(OFF_LO), "n"(OFF_HI) : "memory");
}
/* lds[addr + FIRST + i * STEP] = low byte of px[i].x, lds[addr + FIRST + i * STEP + PAIR] = low byte of px[i].y, i = 0 .. N-1 */
template <int N, int FIRST, int STEP, int PAIR, int I = 0>
__device__ __forceinline__ void lds_st_pairs(uint32_t addr, const s2 *px)
{
    if constexpr (I < N) { lds_st_pair<FIRST + I * STEP, FIRST + I * STEP + PAIR>(addr, px[I]); lds_st_pairs<N, FIRST, STEP, PAIR, I + 1>(addr, px); }
}

__device__ __forceinline__ s2 pk_splat_byte(uint32_t w, int byte)   /* (byte, byte) as two 16-bit halves: one v_perm_b32 */
{
    return as_s2(perm(w, w, byte == 0 ? 0x0C000C00u : byte == 1 ? 0x0C010C01u : byte == 2 ? 0x0C020C02u : 0x0C030C03u));
}

/* ---- packed edge filters: TWO lines per lane (v_pk_*_i16), every register holds the same sample position of both lines.
 * The two lines lie in one 4-sample segment of the edge: they share bS, alpha, beta and tc0.  Conditions are sign bits
 * (x - threshold < 0), combined with AND and spread by one arithmetic shift; selection is bitwise.  8.7.2.3 / 8.7.2.4,
 * reference FilterVerLumaEdge / FilterHorLuma / FilterVerChromaEdge ... src/h264bsd_deblocking.c:643-1180.
 * bs = 0 switches the lane off (alpha 0: |p0 - q0| < 0 never holds). */
__device__ __forceinline__ s2 pk_absdiff(s2 a, s2 b) { return __builtin_elementwise_max(a - b, b - a); }

__device__ __forceinline__ void filter_luma_pk(s2 v[8], int bs, s2 A, s2 B, int tc0, s2 one)
{
    const s2 p3 = v[0], p2 = v[1], p1 = v[2], p0 = v[3], q0 = v[4], q1 = v[5], q2 = v[6], q3 = v[7];
    const s2 zero = pk(0);
    const s2 Aon = bs != 0 ? A : zero;
    const s2 d0 = pk_absdiff(p0, q0);
    s2 fs = ((d0 - Aon) & (pk_absdiff(p1, p0) - B) & (pk_absdiff(q1, q0) - B)) >> pk(15);
    s2 ap = (pk_absdiff(p2, p0) - B) >> pk(15), aq = (pk_absdiff(q2, q0) - B) >> pk(15);      /* -1 where true */
    /* (masks of unknown origin: a select on a spread sign bit is turned into a 16-bit compare and a v_cndmask per HALF, nine
     * instructions for one v_bfi) */
    asm("" : "+v"(fs), "+v"(ap), "+v"(aq));
    /* bS < 4 */
    const s2 t0 = pk(tc0);
    const s2 tc = t0 - ap - aq;
    const s2 d = pk_clip(-tc, tc, (((q0 - p0) << pk(2)) + (p1 - q1) + pk(4)) >> pk(3));
    const s2 avg = (p0 + q0 + one) >> pk(1);            /* (`one` comes in a register: written as + 1 the compiler matches a rounding
                                                           average, which it then takes apart into seven 16-bit instructions) */
    s2 r_p0 = pk_clip(zero, pk(255), p0 + d), r_q0 = pk_clip(zero, pk(255), q0 - d);
    s2 r_p1 = p1 + pk_clip(-t0, t0, (p2 + avg - (p1 << pk(1))) >> pk(1));
    s2 r_q1 = q1 + pk_clip(-t0, t0, (q2 + avg - (q1 << pk(1))) >> pk(1));
    s2 r_p2 = p2, r_q2 = q2;
    s2 m_p1 = ap, m_q1 = aq, m_p2 = zero, m_q2 = zero;
    const bool strong = bs == 4;
    if (__ballot(strong)) {                              /* wave-uniform: intra edges only */
        s2 sm = (d0 - ((A >> pk(2)) + pk(2))) >> pk(15);                                    /* |p0 - q0| < (alpha >> 2) + 2 */
        asm("" : "+v"(sm));
        const s2 sp = sm & ap, sq = sm & aq;
        const s2 p0q0 = p0 + q0;
        const s2 s_p0 = pk_sel(sp, (p2 + ((p1 + p0q0) << pk(1)) + q1 + pk(4)) >> pk(3), ((p1 << pk(1)) + p0 + q1 + pk(2)) >> pk(2));
        const s2 s_q0 = pk_sel(sq, (p1 + ((p0q0 + q1) << pk(1)) + q2 + pk(4)) >> pk(3), ((q1 << pk(1)) + q0 + p1 + pk(2)) >> pk(2));
        if (strong) {
            r_p0 = s_p0; r_q0 = s_q0;
            r_p1 = (p2 + p1 + p0q0 + pk(2)) >> pk(2); r_q1 = (p0q0 + q1 + q2 + pk(2)) >> pk(2);
            r_p2 = ((p3 << pk(1)) + p2 + (p2 << pk(1)) + p1 + p0q0 + pk(4)) >> pk(3);
            r_q2 = ((q3 << pk(1)) + q2 + (q2 << pk(1)) + q1 + p0q0 + pk(4)) >> pk(3);
            m_p1 = sp; m_q1 = sq; m_p2 = sp; m_q2 = sq;
        }
    }
    v[3] = pk_sel(fs, r_p0, p0);
    v[4] = pk_sel(fs, r_q0, q0);
    v[2] = pk_sel(fs & m_p1, r_p1, p1);
    v[5] = pk_sel(fs & m_q1, r_q1, q1);
    v[1] = pk_sel(fs & m_p2, r_p2, p2);
    v[6] = pk_sel(fs & m_q2, r_q2, q2);
}

/* chroma (chromaEdgeFlag = 1): only p0 and q0 change; v = p1, p0, q0, q1 */
__device__ __forceinline__ void filter_chroma_pk(s2 v[4], int bs, s2 A, s2 B, int tc0)
{
    const s2 p1 = v[0], p0 = v[1], q0 = v[2], q1 = v[3];
    const s2 zero = pk(0);
    const s2 Aon = bs != 0 ? A : zero;
    s2 fs = ((pk_absdiff(p0, q0) - Aon) & (pk_absdiff(p1, p0) - B) & (pk_absdiff(q1, q0) - B)) >> pk(15);
    asm("" : "+v"(fs));
    const s2 tc = pk(tc0 + 1);
    const s2 d = pk_clip(-tc, tc, (((q0 - p0) << pk(2)) + (p1 - q1) + pk(4)) >> pk(3));
    s2 r_p0 = pk_clip(zero, pk(255), p0 + d), r_q0 = pk_clip(zero, pk(255), q0 - d);
    const bool strong = bs == 4;
    if (__ballot(strong)) {
        const s2 s_p0 = ((p1 << pk(1)) + p0 + q1 + pk(2)) >> pk(2), s_q0 = ((q1 << pk(1)) + q0 + p1 + pk(2)) >> pk(2);
        if (strong) { r_p0 = s_p0; r_q0 = s_q0; }
    }
    v[1] = pk_sel(fs, r_p0, p0);
    v[2] = pk_sel(fs, r_q0, q0);
}

/* Everything a macroblock's worker loads, all of it requested before the first use (one memory round trip per step): the
 * macroblock's own samples, the strips of the left and upper neighbour that its two macroblock edges work on (whether they
 * are needed is in the record that is still in flight) and its 48-byte record.  Addresses are 32-bit offsets from wave-uniform
 * bases (global_load with an SGPR base). */
struct DbkLoads {
    uint4 y0, y1, c;               /* luma rows 2l, 2l+1 (32 contiguous bytes of the tile); chroma rows 2(l&3), 2(l&3)+1 of plane l>>2 (16 contiguous bytes) */
    uint32_t ly0, ly1, lc0, lc1;   /* the last four columns of the tile to the left, same rows                           */
    uint2 ty; uint32_t tc;         /* this lane's share of the last four luma rows (dwords 2l, 2l+1 of 16) and of the last two
                                      chroma rows of both planes (dword l of 8) of the tile above                         */
    uint4 r0, r1, r2;              /* the record                                                                          */
};

/* cross: the macroblock lies in the first row of a row band: the tile above belongs to another workgroup, its last rows are
 * read past the L1 (ld_agent_u32) — and only when this macroblock's upper edge is filtered at all (want_top), because an
 * unconditional load could run ahead of the other band's stores. */
__device__ __forceinline__ void dbk_load(const FrameDesc &fd, int mb, int l, DbkLoads &p, bool cross, bool want_top)
{
    if (mb < 0) return;
    const H264K_GLOBAL uint8_t *cur = (const H264K_GLOBAL uint8_t *)fd.cur;
    const H264K_GLOBAL uint8_t *recs = (const H264K_GLOBAL uint8_t *)fd.dbk;
    const uint32_t umb = (uint32_t)mb, wmb = fd.wmb;
    const uint32_t t = umb * TILE, tl = (umb ? umb - 1u : 0u) * TILE, tu = (umb >= wmb ? umb - wmb : umb) * TILE;     /* stand-ins where there is no neighbour: never used (k_dbk: LEFT / TOP only where it exists) */
    const uint32_t ro = umb * DBK_REC_BYTES;
    p.r0 = ld16g(recs + ro);
    p.r1 = ld16g(recs + ro + 16u);
    p.r2 = ld16g(recs + ro + 32u);
#if defined(DBK_WHATIF) && (DBK_WHATIF & 4)      /* timing experiment: no sample loads (only the record travels) */
    p.y0 = p.y1 = p.c = make_uint4(umb, t, tl, tu); p.ly0 = p.ly1 = p.lc0 = p.lc1 = umb; p.ty = make_uint2(t, tl); p.tc = tu;
    return;
#endif
    p.y0 = ld16g(cur + t + 32u * l);
    p.y1 = ld16g(cur + t + 32u * l + 16u);
    p.c = ld16g(cur + t + T_CB + 16u * l);
    p.ly0 = *(const H264K_GLOBAL uint32_t *)(cur + tl + 32u * l + 12u);
    p.ly1 = *(const H264K_GLOBAL uint32_t *)(cur + tl + 32u * l + 28u);
    p.lc0 = *(const H264K_GLOBAL uint32_t *)(cur + tl + T_CB + 16u * l + 4u);
    p.lc1 = *(const H264K_GLOBAL uint32_t *)(cur + tl + T_CB + 16u * l + 12u);
    const uint32_t uy = tu + 192u + 8u * l, uc = tu + T_CB + 64u * (l >> 2) + 48u + 4u * (l & 3);
    p.ty = make_uint2(0u, 0u); p.tc = 0u;
    if (!cross) {
        p.ty = ld8g(cur + uy);
        p.tc = *(const H264K_GLOBAL uint32_t *)(cur + uc);
    } else if (want_top) {
        p.ty = make_uint2(ld_agent_u32(fd.cur + uy), ld_agent_u32(fd.cur + uy + 4));
        p.tc = ld_agent_u32(fd.cur + uc);
    }
}

/* the lane's strength at edge slot e (luma edges 0..3; chroma slots 0, 1 = luma edges 0, 2) of a direction whose eight strength
 * bytes are w0 | w1, already shifted right by 4 * (segment of the lane): nibble n = 4 * e + k sits at bit 16 * (e & 1) of dword e >> 1 */
__device__ __forceinline__ int bs_of(uint32_t w0s, uint32_t w1s, int e) { return (int)(((e & 2 ? w1s : w0s) >> (16 * (e & 1))) & 15u); }
/* tc0 of a class for strength bs: t4 = { 0, tc0(1), tc0(2), tc0(3) } as bytes; bs 0 and 4 give 0 */
__device__ __forceinline__ int tc0_of(uint32_t t4, int bs) { return (int)((t4 >> (8 * (bs & 3))) & 255u); }

/* In-loop filter of one macroblock by one worker = 8 lanes: vertical edges, then horizontal edges (8.7).
 * mb < 0: this eighth of the wavefront idles.  w = worker-private LDS.
 * inner: the macroblock has an active inner edge (DBKF_INNER).  Without one it STORES only what its two macroblock edges can
 * have changed — rows 0..2 (upper edge) and columns 0..3 (left edge) of its own tile — because the macroblocks to its right
 * and below it no longer wait for it unless those very samples concern them (k_frame_dbk, dependency rule) and may be
 * rewriting the rest of its tile at the same time.
 * wt: the macroblock lies in the last row of a row band, the band below reads what it writes: everything goes write-through. */
/* SLOTS = 4: every edge.  SLOTS = 1: none of the wavefront's macroblocks has an active inner edge (DBKF_INNER clear: the step was
 * claimed from the second ready list, k_frame_dbk) — only the left and the upper macroblock edge exist: a third of the work. */
#if defined(DBK_WHATIF) && (DBK_WHATIF & 2)
#define DBK_BS(x) 0
#else
#define DBK_BS(x) (x)
#endif
template <bool BANDED, int SLOTS>
__device__ __forceinline__ void deblock_mb(const FrameDesc &fd, int mb, int l, const DbkLoads &p, uint8_t *w, bool wt, bool inner, unsigned long long *tp = nullptr)
{
    constexpr int NG = SLOTS == 4 ? 5 : 2;                         /* groups of four sample positions a luma pass touches */
#define DTICK() (tp ? __builtin_readcyclecounter() : 0ull)
    const unsigned long long d0 = DTICK();
    w = static_cast<uint8_t *>(__builtin_assume_aligned(w, 16));
    uint8_t *lt = w, *ct = w + 20 * LS + (l >> 2) * 10 * CS;      /* luma tile; this lane's chroma plane */
    const bool act = mb >= 0;
    const int c4 = l & 3;                                          /* chroma line pair of this lane */
    /* the record: strengths (dir 0 = vertical edges: r0.x, r0.y; dir 1: r0.z, r0.w), class dwords r1.x .. r2.y, bS-3 bytes and flags r2.z, r2.w */
    const uint32_t flags = p.r2.w >> 16;                          /* byte 46: FJ_DBK_*, byte 47: any */
    const bool f_left = act && (flags & FJ_DBK_LEFT) && (p.r0.x & 0xFFFFu), f_top = act && (flags & FJ_DBK_TOP) && (p.r0.z & 0xFFFFu);
    const bool any_v = __ballot(act && (SLOTS == 4 ? (p.r0.x | p.r0.y) : (p.r0.x & 0xFFFFu))) != 0ull;   /* wave-wide phase skips */
    const bool any_h = __ballot(act && (SLOTS == 4 ? (p.r0.z | p.r0.w) : (p.r0.z & 0xFFFFu))) != 0ull;
    /* thresholds per class: A / B = alpha / beta in both halves, t4 = { 0, tc0(1), tc0(2), tc0(3) } */
    const uint32_t w_ll = p.r1.x, w_lt = p.r1.y, w_li = p.r1.z, w_cl = p.r1.w, w_ctp = p.r2.x, w_ci = p.r2.y;
    const uint32_t t3a = p.r2.z, t3b = p.r2.w;                    /* bytes 40..43, 44..47 */
    s2 one = pk(1);
    asm volatile("" : "+v"(one));

    /* ---- staging: what the horizontal pass needs and the vertical pass does not produce — the upper strips ---- */
    if (act) {
        *reinterpret_cast<uint2 *>(&lt[(l >> 1) * LS + LX + 8 * (l & 1)]) = p.ty;                          /* strip dwords 2l, 2l+1: row l >> 1, columns 8 * (l & 1) .. + 7 */
        *reinterpret_cast<uint32_t *>(&ct[(c4 >> 1) * CS + CX + 4 * (c4 & 1)]) = p.tc;                     /* strip dword l: plane l >> 2, row (l & 3) >> 1, columns 4 * (l & 1) .. + 3 */
        *reinterpret_cast<uint32_t *>(&ct[(2 + 2 * c4) * CS + CX - 4]) = p.lc0;                            /* the left chroma strip: the vertical pass only rewrites its last byte */
        *reinterpret_cast<uint32_t *>(&ct[(3 + 2 * c4) * CS + CX - 4]) = p.lc1;
    }

    if (tp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long d1 = DTICK();
    /* ---- vertical edges, luma: rows 2l (low halves) and 2l+1 (high halves) across all four edges ---- */
    {
        s2 px[4 * NG];
        px[0] = as_s2(perm(p.ly1, p.ly0, 0x0C040C00u)); px[1] = as_s2(perm(p.ly1, p.ly0, 0x0C050C01u));
        px[2] = as_s2(perm(p.ly1, p.ly0, 0x0C060C02u)); px[3] = as_s2(perm(p.ly1, p.ly0, 0x0C070C03u));
        const uint32_t ra[4] = { p.y0.x, p.y0.y, p.y0.z, p.y0.w }, rb[4] = { p.y1.x, p.y1.y, p.y1.z, p.y1.w };
#pragma unroll
        for (int w4 = 0; w4 < NG - 1; w4++) {
            px[4 + 4 * w4 + 0] = as_s2(perm(rb[w4], ra[w4], 0x0C040C00u));
            px[4 + 4 * w4 + 1] = as_s2(perm(rb[w4], ra[w4], 0x0C050C01u));
            px[4 + 4 * w4 + 2] = as_s2(perm(rb[w4], ra[w4], 0x0C060C02u));
            px[4 + 4 * w4 + 3] = as_s2(perm(rb[w4], ra[w4], 0x0C070C03u));
        }
        if (any_v) {
            const int k = l >> 1;                                  /* both rows lie in segment k of every vertical edge */
            const uint32_t w0s = p.r0.x >> (4 * k), w1s = p.r0.y >> (4 * k);
            const s2 A_l = pk_splat_byte(w_ll, 0), B_l = pk_splat_byte(w_ll, 1), A_i = pk_splat_byte(w_li, 0), B_i = pk_splat_byte(w_li, 1);
            const uint32_t t4_l = perm(t3a, w_ll, 0x0403020Cu), t4_i = perm(t3a, w_li, 0x0603020Cu);    /* { 0, tc0(1), tc0(2), tc0(3) } */
#pragma unroll
            for (int e = 0; e < SLOTS; e++) {
                const int bs = DBK_BS(act ? bs_of(w0s, w1s, e) : 0);
                if (__ballot(bs != 0)) filter_luma_pk(px + 4 * e, bs, e ? A_i : A_l, e ? B_i : B_l, tc0_of(e ? t4_i : t4_l, bs), one);
            }
        }
        if (act) {
            /* back to rows: four packed pairs -> one dword of row 2l and one of row 2l+1 (samples are < 256: two pairs merge with a
             * shift-or, the rows come apart with a byte permute each); the columns a pass did not touch go as they came */
            uint32_t na[5] = { 0u, ra[0], ra[1], ra[2], ra[3] }, nb[5] = { 0u, rb[0], rb[1], rb[2], rb[3] };
#pragma unroll
            for (int g = 0; g < NG; g++) {
                const uint32_t t01 = as_u32(px[4 * g]) | (as_u32(px[4 * g + 1]) << 8), t23 = as_u32(px[4 * g + 2]) | (as_u32(px[4 * g + 3]) << 8);
                na[g] = perm(t23, t01, 0x05040100u); nb[g] = perm(t23, t01, 0x07060302u);
            }
            uint8_t *rowa = &lt[(4 + 2 * l) * LS];
            *reinterpret_cast<uint32_t *>(rowa + LX - 4) = na[0]; *reinterpret_cast<uint32_t *>(rowa + LS + LX - 4) = nb[0];
            *reinterpret_cast<uint4 *>(rowa + LX) = make_uint4(na[1], na[2], na[3], na[4]);
            *reinterpret_cast<uint4 *>(rowa + LS + LX) = make_uint4(nb[1], nb[2], nb[3], nb[4]);
        }
    }
    /* ---- vertical edges, chroma: rows 2 c4, 2 c4 + 1 of plane l >> 2; edges at columns 0 and 4 = luma edges 0 and 2 ---- */
    {
        s2 px[12];
        px[2] = as_s2(perm(p.lc1, p.lc0, 0x0C060C02u)); px[3] = as_s2(perm(p.lc1, p.lc0, 0x0C070C03u));
        px[4] = as_s2(perm(p.c.z, p.c.x, 0x0C040C00u)); px[5] = as_s2(perm(p.c.z, p.c.x, 0x0C050C01u));
        px[6] = as_s2(perm(p.c.z, p.c.x, 0x0C060C02u)); px[7] = as_s2(perm(p.c.z, p.c.x, 0x0C070C03u));
        if (SLOTS == 4) {
            px[8] = as_s2(perm(p.c.w, p.c.y, 0x0C040C00u)); px[9] = as_s2(perm(p.c.w, p.c.y, 0x0C050C01u));
            px[10] = as_s2(perm(p.c.w, p.c.y, 0x0C060C02u)); px[11] = as_s2(perm(p.c.w, p.c.y, 0x0C070C03u));
        }
        if (any_v) {
            const uint32_t w0s = p.r0.x >> (4 * c4), w1s = p.r0.y >> (4 * c4);     /* chroma rows 2 c4, 2 c4 + 1 = luma rows 4 c4 .. 4 c4 + 3: segment c4 */
            const int bs0 = DBK_BS(act ? (int)(w0s & 15u) : 0), bs1 = DBK_BS(act ? (int)(w1s & 15u) : 0);
            if (__ballot(bs0 != 0))
                filter_chroma_pk(px + 2, bs0, pk_splat_byte(w_cl, 0), pk_splat_byte(w_cl, 1), tc0_of(perm(t3a, w_cl, 0x0703020Cu), bs0));
            if constexpr (SLOTS == 4) if (__ballot(bs1 != 0))
                filter_chroma_pk(px + 6, bs1, pk_splat_byte(w_ci, 0), pk_splat_byte(w_ci, 1), tc0_of(perm(t3b, w_ci, 0x0503020Cu), bs1));
        }
        if (act) {
            uint8_t *rowa = &ct[(2 + 2 * c4) * CS];
            uint32_t ra[2] = { 0u, p.c.y }, rb[2] = { 0u, p.c.w };
#pragma unroll
            for (int g = 0; g < (SLOTS == 4 ? 2 : 1); g++) {
                const uint32_t t01 = as_u32(px[4 + 4 * g]) | (as_u32(px[5 + 4 * g]) << 8), t23 = as_u32(px[6 + 4 * g]) | (as_u32(px[7 + 4 * g]) << 8);
                ra[g] = perm(t23, t01, 0x05040100u); rb[g] = perm(t23, t01, 0x07060302u);
            }
            lds_st_pair<CX - 1, CS + CX - 1>(lds_addr(rowa), px[3]);                   /* p0 of the left edge: the last byte of the strip */
            *reinterpret_cast<uint2 *>(rowa + CX) = make_uint2(ra[0], ra[1]);
            *reinterpret_cast<uint2 *>(rowa + CS + CX) = make_uint2(rb[0], rb[1]);
        }
    }
    wave_sync();
    const unsigned long long d2 = DTICK();

    /* ---- horizontal edges, luma: columns 2l (low halves), 2l+1 (high halves); tile rows 0..3 = the strip above ---- */
    if (any_h) {
        const uint8_t *colp = &lt[LX + 2 * l];
        s2 px[4 * NG];
#pragma unroll
        for (int r = 0; r < 4 * NG; r++) px[r] = as_s2(perm(0u, *reinterpret_cast<const uint16_t *>(colp + r * LS), 0x0C010C00u));
        const int k = l >> 1;
        const uint32_t w0s = p.r0.z >> (4 * k), w1s = p.r0.w >> (4 * k);
        const s2 A_t = pk_splat_byte(w_lt, 0), B_t = pk_splat_byte(w_lt, 1), A_i = pk_splat_byte(w_li, 0), B_i = pk_splat_byte(w_li, 1);
        const uint32_t t4_t = perm(t3a, w_lt, 0x0503020Cu), t4_i = perm(t3a, w_li, 0x0603020Cu);
#pragma unroll
        for (int e = 0; e < SLOTS; e++) {
            const int bs = DBK_BS(act ? bs_of(w0s, w1s, e) : 0);
            if (__ballot(bs != 0)) {
                filter_luma_pk(px + 4 * e, bs, e ? A_i : A_t, e ? B_i : B_t, tc0_of(e ? t4_i : t4_t, bs), one);
                if (act) {
#pragma unroll
                    for (int r = 4 * e + 1; r < 4 * e + 7; r++) *reinterpret_cast<uint16_t *>(const_cast<uint8_t *>(colp) + r * LS) = (uint16_t)perm(0u, as_u32(px[r]), 0x0C0C0200u);
                }
            }
        }
    }
    /* ---- horizontal edges, chroma: columns 2 c4, 2 c4 + 1 of plane l >> 2; tile rows 0, 1 = the strip above; edges at rows 2 and 6 ---- */
    if (any_h) {
        const uint8_t *colp = &ct[CX + 2 * c4];
        s2 px[SLOTS == 4 ? 10 : 4];
#pragma unroll
        for (int r = 0; r < (SLOTS == 4 ? 10 : 4); r++) px[r] = as_s2(perm(0u, *reinterpret_cast<const uint16_t *>(colp + r * CS), 0x0C010C00u));
        const uint32_t w0s = p.r0.z >> (4 * c4), w1s = p.r0.w >> (4 * c4);
        const int bs0 = DBK_BS(act ? (int)(w0s & 15u) : 0), bs1 = DBK_BS(act ? (int)(w1s & 15u) : 0);
        if (__ballot(bs0 != 0)) {
            filter_chroma_pk(px + 0, bs0, pk_splat_byte(w_ctp, 0), pk_splat_byte(w_ctp, 1), tc0_of(perm(t3b, w_ctp, 0x0403020Cu), bs0));
            if (act) { *reinterpret_cast<uint16_t *>(const_cast<uint8_t *>(colp) + 1 * CS) = (uint16_t)perm(0u, as_u32(px[1]), 0x0C0C0200u); *reinterpret_cast<uint16_t *>(const_cast<uint8_t *>(colp) + 2 * CS) = (uint16_t)perm(0u, as_u32(px[2]), 0x0C0C0200u); }
        }
        if constexpr (SLOTS == 4) if (__ballot(bs1 != 0)) {
            filter_chroma_pk(px + 4, bs1, pk_splat_byte(w_ci, 0), pk_splat_byte(w_ci, 1), tc0_of(perm(t3b, w_ci, 0x0503020Cu), bs1));
            if (act) { *reinterpret_cast<uint16_t *>(const_cast<uint8_t *>(colp) + 5 * CS) = (uint16_t)perm(0u, as_u32(px[5]), 0x0C0C0200u); *reinterpret_cast<uint16_t *>(const_cast<uint8_t *>(colp) + 6 * CS) = (uint16_t)perm(0u, as_u32(px[6]), 0x0C0C0200u); }
        }
    }
    wave_sync();
    const unsigned long long d3 = DTICK();

    /* ---- store: own macroblock (whole, or what its two macroblock edges can have changed), the last 3 (1) columns of the left
     * and rows of the upper neighbour.  Everything that may be stored is read from the tile FIRST, unconditionally, in one burst
     * of LDS reads; the stores that follow are predicated but wait for nothing.  (Written the natural way — every condition
     * reads what it stores — the compiler emits ten read -> wait -> store sequences one after the other behind their exec-mask
     * branches: 4-5.6 k cycles per step, more than a filter pass, measured with tools/prof_tail.py.) ---- */
#ifdef DBK_WHATIF
    if (act && !(DBK_WHATIF & 1)) {          /* (timing experiment: bit 0 = no store phase, bit 1 = no filter arithmetic; results are wrong) */
#else
    if (act) {
#endif
        const uint32_t t = (uint32_t)mb * TILE;
        uint8_t *cur = fd.cur;
        H264K_GLOBAL uint8_t *curg = (H264K_GLOBAL uint8_t *)fd.cur;
        uint4 yr[2];
        uint2 cr[2];
        uint32_t lyv[2], lcv[2];
#pragma unroll
        for (int h = 0; h < 2; h++) {
            yr[h] = *reinterpret_cast<const uint4 *>(&lt[(4 + 2 * l + h) * LS + LX]);
            cr[h] = *reinterpret_cast<const uint2 *>(&ct[(2 + 2 * c4 + h) * CS + CX]);
            lyv[h] = *reinterpret_cast<const uint32_t *>(&lt[(4 + 2 * l + h) * LS + LX - 4]);
            lcv[h] = *reinterpret_cast<const uint32_t *>(&ct[(2 + 2 * c4 + h) * CS + CX - 4]);
        }
        uint2 tyv = *reinterpret_cast<const uint2 *>(&lt[(l >> 1) * LS + LX + 8 * (l & 1)]);
        uint32_t tcv = *reinterpret_cast<const uint32_t *>(&ct[1 * CS + CX + 4 * (c4 & 1)]);
#pragma unroll
        for (int h = 0; h < 2; h++)
            asm volatile("" : "+v"(yr[h].x), "+v"(yr[h].y), "+v"(yr[h].z), "+v"(yr[h].w), "+v"(cr[h].x), "+v"(cr[h].y), "+v"(lyv[h]), "+v"(lcv[h]));
        asm volatile("" : "+v"(tyv.x), "+v"(tyv.y), "+v"(tcv));
        if (tp) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tp[12] += DTICK() - d3; }
#pragma unroll
        for (int h = 0; h < 2; h++) {                              /* luma rows 2l, 2l+1 */
            const int row = 2 * l + h;
            const uint32_t o = t + 16u * row;
            if (inner || (f_top && row < 3)) {
                if (BANDED && wt) put16(cur + o, yr[h], true); else st16g(curg + o, yr[h]);
            } else if (f_left) {
                if (BANDED && wt) put4(cur + o, yr[h].x, true); else *(H264K_GLOBAL uint32_t *)(curg + o) = yr[h].x;
            }
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {                              /* chroma rows 2 c4, 2 c4 + 1 of plane l >> 2 */
            const int row = 2 * c4 + h;
            const uint32_t o = t + T_CB + 64u * (l >> 2) + 8u * row;
            if (inner || (f_top && row == 0)) {
                if (BANDED && wt) put8(cur + o, cr[h], true); else *(H264K_GLOBAL u32x2 *)(curg + o) = (u32x2){ cr[h].x, cr[h].y };
            } else if (f_left) {
                if (BANDED && wt) put4(cur + o, cr[h].x, true); else *(H264K_GLOBAL uint32_t *)(curg + o) = cr[h].x;
            }
        }
        if (tp) tp[13] += DTICK() - d3;
        if (f_left) {
            const uint32_t tl = t - TILE;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const uint32_t oy = tl + 16u * (2 * l + h) + 12u, oc = tl + T_CB + 64u * (l >> 2) + 8u * (2 * c4 + h) + 4u;
                if (BANDED && wt) { put4(cur + oy, lyv[h], true); put4(cur + oc, lcv[h], true); }
                else { *(H264K_GLOBAL uint32_t *)(curg + oy) = lyv[h]; *(H264K_GLOBAL uint32_t *)(curg + oc) = lcv[h]; }
            }
        }
        if (f_top) {
            const uint32_t tu = t - (uint32_t)fd.wmb * TILE;
            if (l >= 2) {                                          /* luma strip rows 1..3 (row 0 = p3 never changes): dwords 2l, 2l+1 of the strip */
                const uint32_t o = tu + 192u + 8u * l;
                if (BANDED && wt) put8(cur + o, tyv, true); else *(H264K_GLOBAL u32x2 *)(curg + o) = (u32x2){ tyv.x, tyv.y };
            }
            if (c4 >= 2) {                                         /* chroma strip row 1 (row 0 = p1 never changes) */
                const uint32_t o = tu + T_CB + 64u * (l >> 2) + 56u + 4u * (c4 & 1);
                if (BANDED && wt) put4(cur + o, tcv, true); else *(H264K_GLOBAL uint32_t *)(curg + o) = tcv;
            }
        }
    }
    if (tp) tp[14] += DTICK() - d3;
    wave_sync();          /* tiles are reused by this worker's next macroblock */
    if (tp) { const unsigned long long d4 = DTICK(); tp[8] += d1 - d0; tp[9] += d2 - d1; tp[10] += d3 - d2; tp[11] += d4 - d3; }
#undef DTICK
}

/* ONE workgroup per picture in the two kernels below (a picture never leaves its CU): dependencies inside a picture
 * are tracked in LDS, no kernel boundary and no inter-workgroup traffic inside a picture.  Occupancy comes from
 * batching streams (256 pictures = one workgroup per CU).  k_frame_intra: 12 wavefronts = 162 VGPRs without spills
 * (16 wavefronts cap the kernel at 128 VGPRs: 33 spilled, 20.8 vs 19.3 ms per step; 8: 22.2 ms). */
#ifndef TAIL_WAVES_N
#define TAIL_WAVES_N 12
#endif
constexpr int TAIL_WAVES = TAIL_WAVES_N;
#ifndef DBK_WAVES_N
#define DBK_WAVES_N 12
#endif
constexpr int DBK_WAVES = DBK_WAVES_N;              /* most wavefronts of k_frame_dbk (the launch chooses: TailConfig.dbk_waves); a wavefront
                                                        is eight workers (deblock_mb) */

/* Which picture and which row band a workgroup of the two per-picture kernels works on.  Workgroups take a ticket when they
 * start (one device-scope atomic): ticket t = band t % max_bands of picture t / max_bands.  A band only ever waits for the
 * band above it, which holds a smaller ticket and has therefore STARTED — whatever order the dispatcher chose — so a waiting
 * workgroup can never keep the one it waits for off the machine.  tickets[0] = tickets taken, tickets[1] = workgroups that
 * left: the last one to leave zeroes both for the next launch on this HIP stream (one pair per stream, engine.hip).
 * tickets == nullptr: blockIdx.x is the ticket (single-band launches). */
__device__ __forceinline__ uint32_t take_ticket(uint32_t *tickets, uint32_t *slot)
{
    if (threadIdx.x == 0) *slot = tickets ? atomicAdd(&tickets[0], 1u) : blockIdx.x;
    __syncthreads();
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)*slot);
}
__device__ __forceinline__ void return_ticket(uint32_t *tickets)
{
    if (tickets && threadIdx.x == 0 && atomicAdd(&tickets[1], 1u) == gridDim.x - 1u) { atomicExch(&tickets[0], 0u); atomicExch(&tickets[1], 0u); }
}
/* rows per band and number of bands for a picture of hmb macroblock rows that wants `want` bands, in a launch with
 * max_bands workgroups per picture whose LDS holds the state of at most rows_cap rows */
__device__ __forceinline__ void band_split(int hmb, uint32_t want, uint32_t heavy, uint32_t max_bands, uint32_t light_cap, uint32_t rows_cap, int &rows, int &bands)
{
    if (!heavy && want > light_cap) want = light_cap;         /* (the launch may grant bands to the heavy pictures of a tick only) */
    int w = (int)(want < max_bands ? want : max_bands);
    if (w < 1) w = 1;
    rows = (hmb + w - 1) / w;
    if (rows > (int)rows_cap) rows = (int)rows_cap;
    bands = (hmb + rows - 1) / rows;
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
 * a "done" byte (scratch_done(fd, 1)), an idle wavefront of the band below polls, the consumers read the row above past
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
    uint8_t *done_g = scratch_done(fd, 1);

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
#if defined(INTRA_WHATIF) && (INTRA_WHATIF & 4)
            else if (false) {
#else
            else if (kind == FJ_MB_I4x4 && k > 1) {
#endif
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

/* In-loop deblocking of one picture.  The filter of macroblock (x,y) touches its own samples, the last
 * 4 columns of (x-1,y) and the last 4 rows of (x,y-1); in the standard's raster order that makes it
 * depend on exactly three earlier steps: (x-1,y), (x,y-1) and (x+1,y-1) — and only if those macroblocks
 * are filtered at all (most P-picture macroblocks have all-zero strengths and are never touched).
 *
 * ROW BANDS.  A picture is split into up to max_bands bands of consecutive macroblock rows, one workgroup each
 * (grid = max_bands x pictures; a picture that wants fewer bands leaves the surplus workgroups idle).  Inside a band
 * the dependencies are tracked in LDS as before; the only dependencies that cross a band boundary are those of a
 * band's FIRST row on the LAST row of the band above — (x,y-1) and (x+1,y-1) — and they are handed over through HBM:
 *   producer: a macroblock of a band's last row writes everything write-through (put4/8/16 with wt), waits for its stores
 *             (s_waitcnt vmcnt(0)) and then sets its "done" byte (scratch_done, agent scope);
 *   consumer: a wavefront of the band below that finds nothing ready polls the done bytes of the producers its first row
 *             still waits for (one poller per band at a time, relaxed agent-scope loads, s_sleep between passes), marks
 *             each seen producer once (LDS bit) and releases its dependants into the band's ready queue; the first-row
 *             macroblock then reads the last rows of the tile above past the L1 (dbk_prefetch, cross).
 * Every pair of macroblocks that touches a common sample is ordered as in the reference's raster scan
 * (src/h264bsd_deblocking.c:604-638) whether both lie in one band or not.  The same spin limit that guards the LDS scheduler
 * ends a wait that never finishes in DEVERR_DBK_SCHED.  Small workgroups (4 wavefronts by default) leave most of a
 * compute unit's registers to other workgroups — bands of other pictures, the list-driven kernels of other stream groups.
 *
 * Dataflow scheduling inside a band, all state in LDS (indices are band-local: row r0-1 .. r1-1):
 *   anyf[]    DBKF_* flags of the band's rows and of the row above
 *   dep[mb]   number of filtered macroblocks among those three that are not finished yet
 *   queue[]   ready list: every filtered macroblock is pushed exactly once, when its dep reaches 0
 *   head/tail claim / publish cursors (LDS atomics)
 * A worker is an EIGHTH of a wavefront (8 lanes, two sample lines per lane, packed 16-bit arithmetic, luma and then chroma:
 * deblock_mb).  A free wavefront pulls up to eight READY macroblocks of ONE of the two ready lists at once (compare-and-swap on
 * that list's head: macroblocks with an active inner edge / with macroblock edges only), one per worker, fetches their
 * samples, records and neighbour strips in one memory round trip, filters, stores, then releases the
 * three dependants (x+1,y), (x,y+1), (x-1,y+1).  No level barriers.  What a P picture costs is the LATENCY of its ~100
 * dependent steps (a step is ~9.5 k cycles: claim 0.5, one memory round trip 1.5, the two passes 2.4 + 2.4, stores 1.2, their
 * completion and the release 0.8; the wavefronts find nothing ready 40-60 % of the time, the vector pipe is 45 % busy), which is
 * why edge-only macroblocks have their own list and their own short instruction stream.  Same-CU visibility of the stores needs
 * no wait at all (release_stores above); rounds 1-3 waited for the stores' acknowledgement on every step.
 * Dynamic LDS: workers x WORKER_LDS tiles | anyf | dep | queue u16 | counters | seen bits (dbk_lds_bytes). */
__host__ __device__ inline size_t dbk_lds_bytes(uint32_t waves, uint32_t wmb, uint32_t band_rows)
{
    const size_t n_loc16 = (((size_t)band_rows + 1) * wmb + 15) & ~(size_t)15, nq8 = ((size_t)band_rows * wmb + 7) & ~(size_t)7;
    return (((size_t)waves * (64 / DBK_LANES) * WORKER_LDS + 15) & ~(size_t)15) + 2 * n_loc16 + 2 * nq8 + 32 + 4 * ((((size_t)wmb + 31) / 32 + 3) & ~(size_t)3);
}
#ifndef DBK_OCC
#define DBK_OCC 4            /* 127 VGPRs, nothing spilled (cycle accounting compiled out): a SIMD could hold four of these wavefronts */
#endif
template <bool BANDED>
__global__ __launch_bounds__(64 * DBK_WAVES, BANDED ? 3 : DBK_OCC) void k_frame_dbk(const FrameDesc *__restrict__ frames, unsigned long long *prof,
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
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, grp = lane / DBK_LANES, l = lane % DBK_LANES;
    const int wmb = fd.wmb, hmb = fd.hmb, n_mbs = (int)fd.n_mbs;
    int R = hmb, nb = 1;
    if (BANDED) band_split(hmb, fd.dbk_bands, fd.heavy, max_bands, light_cap, rows_cap, R, nb);
    if (!fd.any_deblock || (int)band >= nb) { if (BANDED) return_ticket(tickets); return; }
    const int r0 = (int)band * R, r1 = min(hmb, r0 + R);
    const int base = (r0 - 1) * wmb;                        /* band-local index of macroblock mb: mb - base (row r0-1 first) */
    const int n_loc = (R + 1) * wmb, n_loc16 = (n_loc + 15) & ~15, nq8 = (R * wmb + 7) & ~7;
    const bool has_up = BANDED && band > 0, has_down = BANDED && r1 < hmb;
    uint8_t *anyf = lds + (((blockDim.x / DBK_LANES) * WORKER_LDS + 15) & ~15);   /* 8 workers per launched wavefront */
    uint8_t *dep = anyf + n_loc16;
    uint16_t *queue = reinterpret_cast<uint16_t *>(dep + n_loc16);
    uint32_t *ctr = reinterpret_cast<uint32_t *>(queue + nq8);   /* [0] head, [1] tail, [2] total, [3] producers awaited, [4] producers seen, [5] poll lock */
    uint32_t *seen = ctr + 8;                                    /* one bit per column: the done byte of (x, r0-1) has been seen */
    uint8_t *wlds = lds + (wave * (64 / DBK_LANES) + grp) * WORKER_LDS;
    uint8_t *flags_g = scratch_flags(fd), *done_g = scratch_done(fd, 0);
    /* debug accounting (h264bsdmiDebugTailProfile): band 0 of picture 0 only, per wavefront: [0] cycles with nothing ready,
     * [1] cycles filtering, [2] cycles waiting for own stores, [3] macroblocks filtered (both halves), [4] total */
#ifdef H264K_TAIL_PROFILE
    unsigned long long *tp = (prof && ticket == 0) ? prof + wave * 16 : nullptr;
#else
    unsigned long long *const tp = nullptr;
    (void)prof;
#endif
    unsigned long long t_idle = 0, t_work = 0, t_store = 0, n_done = 0, n_steps = 0;
    const unsigned long long t_begin = tp ? __builtin_readcyclecounter() : 0ull;
    unsigned long long t_mark = t_begin;

    {
        /* flags of rows r0-1 .. r1-1 (row -1 of band 0: zeros) */
        const int src0 = base < 0 ? 0 : base, n_src = r1 * wmb - src0;
        for (int i = tid; i < src0 - base; i += blockDim.x) anyf[i] = 0;
        if (((src0 | (src0 - base)) & 3) == 0) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(flags_g + src0);
            uint32_t *dst = reinterpret_cast<uint32_t *>(anyf + (src0 - base));
            for (int i = tid; i < (n_src + 3) / 4; i += blockDim.x) dst[i] = src[i];       /* (the scratch area is padded) */
        } else {
            for (int i = tid; i < n_src; i += blockDim.x) anyf[src0 - base + i] = flags_g[src0 + i];
        }
        for (int i = tid; i < nq8 / 2; i += blockDim.x) reinterpret_cast<uint32_t *>(queue)[i] = 0xFFFFFFFFu;
        if (tid < 8) ctr[tid] = 0;
        for (int i = tid; i < (wmb + 31) >> 5; i += blockDim.x) seen[i] = 0;
    }
    __syncthreads();
    /* Dependencies at edge granularity.  A filtered macroblock waits for
     *   (x-1,y)    only if its own left edge is active (DBKF_LEFT): otherwise it neither reads nor writes that neighbour;
     *   (x,y-1)    only if its own upper edge is active (DBKF_TOP);
     *   (x+1,y-1)  only if its upper edge is active AND that macroblock's left edge is: only then does (x+1,y-1)
     *              rewrite the columns of (x,y-1) whose last rows this macroblock reads and rewrites;
     * and not even then if the neighbour cannot have touched the samples in question: a macroblock without an active INNER edge
     * (DBKF_INNER clear: 48 % of the filtered macroblocks of the bundled 1080p stream, the neighbours of coded ones) only
     * touches columns -3..2 through its left edge and rows -3..2 through its upper edge, so (x-1,y) matters to the last four
     * columns this macroblock's left edge works on only if its UPPER edge was filtered (rows 0..2 of those columns), and
     * (x,y-1) to the last four rows only if its LEFT edge was.
     * Every pair of macroblocks that touches a common sample is still ordered as in the reference's raster scan
     * (deblocking.c:604-638); the longest chain of the bundled 1080p stream shrinks by 21 % (9562 -> 7512 steps).
     * For the band's first row the macroblocks above belong to the band above: they count like any other and are
     * released by the poller (below) instead of by the wavefront that filtered them. */
    /* TWO ready lists in one array: macroblocks with an active inner edge are published from the front (cursors ctr[0] / ctr[1]),
     * those without — only the left and / or upper macroblock edge: a third of the work — from the back (ctr[6] / ctr[7]).  A
     * wavefront claims from ONE list, so that a step of edge-only macroblocks runs the short instruction stream (deblock_mb,
     * SLOTS = 1): in a P picture a step is a link of a dependency chain and its length is what the picture's time is made of. */
    const int nq_last = nq8 - 1;
    auto push = [&](int mb, uint32_t flags) {
        if (flags & DBKF_INNER) queue[atomicAdd(&ctr[1], 1u)] = (uint16_t)mb;
        else queue[nq_last - (int)atomicAdd(&ctr[7], 1u)] = (uint16_t)mb;
    };
    /* (k_dbk sets DBKF_LEFT / DBKF_TOP only where that neighbour exists: a macroblock in column 0 never has LEFT — so the
     * macroblock "to the left" of it, the last one of the row above, is never counted, and neither is the first one of the next
     * row as the right-hand neighbour of the last column: no division by the picture width anywhere in this kernel) */
    for (int mb = r0 * wmb + tid; mb < r1 * wmb; mb += blockDim.x) {
        const int li = mb - base;
        const uint32_t f = anyf[li];
        if (!(f & DBKF_ANY)) continue;
        const int d = ((f & DBKF_LEFT) && (anyf[li - 1] & (DBKF_INNER | DBKF_TOP)) ? 1 : 0) +
                      ((f & DBKF_TOP) && (anyf[li - wmb] & (DBKF_INNER | DBKF_LEFT)) ? 1 : 0) +
                      ((f & DBKF_TOP) && (anyf[li - wmb + 1] & DBKF_LEFT) ? 1 : 0);
        dep[li] = (uint8_t)d;
        atomicAdd(&ctr[2], 1u);
        if (d == 0) push(mb, f);
    }
    if (has_up)
        for (int x = tid; x < wmb; x += blockDim.x)
            if (anyf[x] & DBKF_ANY) atomicAdd(&ctr[3], 1u);
    __syncthreads();
    const uint32_t total = ctr[2], n_await = ctr[3];
    volatile H264K_LDS uint16_t *vq = (volatile H264K_LDS uint16_t *)queue;      /* (a generic volatile pointer would read LDS through flat_load) */

    /* one dependency of band-local macroblock li is gone: publish it when it was the last */
    auto release = [&](int li) {
        /* byte-wide counters: decrement through a 32-bit LDS atomic on the containing word */
        uint32_t *w = reinterpret_cast<uint32_t *>(dep + (li & ~3));
        const uint32_t sh = 8u * (li & 3);
        const uint32_t old = atomicSub(w, 1u << sh);
        if (((old >> sh) & 255u) == 1u) push(li + base, anyf[li]);
    };

    /* Pull model: a free wavefront takes up to four READY macroblocks at once (one per quarter).  Ready macroblocks
     * are therefore packed into as few wavefronts as possible — the loop is instruction-issue bound, so a step that
     * runs with one busy quarter costs as much as a full one — and a wavefront with nothing to do issues nothing. */
    uint32_t spins = 0;                  /* safety net: a scheduling bug must end in a reported error (DEVERR_*), never in a hung GPU */
    volatile H264K_LDS uint32_t *vctr = (volatile H264K_LDS uint32_t *)ctr;
    for (;;) {
        uint32_t cbase = 0, k = 0, cls = 0;
        if (lane == 0) {
            const uint32_t h0 = vctr[0], t0 = vctr[1], h1 = vctr[6], t1 = vctr[7];
            const uint32_t a0 = t0 - h0, a1 = t1 - h1;
            if (a0 | a1) {
                cls = a1 >= a0 ? 1u : 0u;                            /* the longer list; the cheaper one when they tie */
                const uint32_t h = cls ? h1 : h0, a = cls ? a1 : a0;
                k = a < 8u ? a : 8u;
                if (atomicCAS(&ctr[cls ? 6 : 0], h, h + k) != h) k = 0;       /* lost the race: look again */
                cbase = h;
            } else if (h0 + h1 >= total) {
                k = 0xFFFFFFFFu;                                    /* everything has been claimed */
            }
        }
        cbase = __shfl(cbase, 0); k = __shfl(k, 0); cls = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(cls, 0));
        if (k == 0xFFFFFFFFu) break;
        if (++spins > (1u << 24)) { if (lane == 0) report_device_error(fd, DEVERR_DBK_SCHED); break; }
        if (k == 0) {
            /* nothing ready.  If the first row still waits for macroblocks of the band above, look whether they are done:
             * one wavefront of the band at a time, lane -> column */
            bool polled = false;
            if (has_up && vctr[4] < n_await) {
                uint32_t got = 0;
                if (lane == 0) got = atomicCAS(&ctr[5], 0u, 1u) == 0u;
                got = __shfl(got, 0);
                if (got) {
                    polled = true;
                    for (int x = lane; x < wmb; x += 64) {
                        const uint32_t fu = anyf[x];
                        const uint32_t bit = 1u << (x & 31);
                        if (!(fu & DBKF_ANY) || (seen[x >> 5] & bit)) continue;
                        if (!ld_agent_u8(done_g + base + x)) continue;
                        if (atomicOr(&seen[x >> 5], bit) & bit) continue;
                        atomicAdd(&ctr[4], 1u);
                        /* the mirror image of the dependency rule: (x, r0) waits for it through its upper edge, (x-1, r0)
                         * if this producer's left edge was filtered */
                        const uint32_t fc = anyf[wmb + x];
                        if ((fc & DBKF_ANY) && (fc & DBKF_TOP) && (fu & (DBKF_INNER | DBKF_LEFT))) release(wmb + x);
                        if (x > 0 && (fu & DBKF_LEFT)) {
                            const uint32_t fl = anyf[wmb + x - 1];
                            if ((fl & DBKF_ANY) && (fl & DBKF_TOP)) release(wmb + x - 1);
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) atomicExch(&ctr[5], 0u);
                }
            }
            if (polled) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(1);
            continue;
        }
        if (tp) { const unsigned long long t = __builtin_readcyclecounter(); t_idle += t - t_mark; t_mark = t; }
        int run = -1;
        if ((uint32_t)grp < k) {
            int v;
            const int slot = cls ? nq_last - (int)(cbase + grp) : (int)(cbase + grp);
            do { v = vq[slot]; } while (v == 0xFFFF);                /* the publisher bumps the cursor, then writes the slot */
            run = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int lo_mb = r0 * wmb, hi_mb = r1 * wmb;                /* the band's own macroblocks */
        const bool cross = has_up && run >= 0 && run < lo_mb + wmb;  /* first row: the tile above belongs to the band above */
        const bool wt = has_down && run >= hi_mb - wmb;              /* last row: the band below reads what this macroblock writes */
        const uint32_t fm = run >= 0 ? anyf[run - base] : 0u;
        bool want_top = true;
        if (BANDED && __ballot(cross) != 0ull) want_top = !cross || (fm & DBKF_TOP);
        DbkLoads cp;
        dbk_load(fd, run, l, cp, BANDED && cross, want_top);
        if (cls) deblock_mb<BANDED, 1>(fd, run, l, cp, wlds, wt, false, (tp && lane == 0) ? tp : nullptr);
        else deblock_mb<BANDED, 4>(fd, run, l, cp, wlds, wt, (fm & DBKF_INNER) != 0u, (tp && lane == 0) ? tp : nullptr);
        if (tp) { const unsigned long long t = __builtin_readcyclecounter(); t_work += t - t_mark; t_mark = t; n_done += __popcll(__ballot(run >= 0 && l == 0)); n_steps++; }
        /* release: stores done -> dependants */
        release_stores(BANDED && wt && run >= 0);
        if (BANDED && wt && l == 3) st_agent_u8(done_g + run, 1u);   /* hand-over to the band below */
        if (run >= 0 && l < 3) {
            /* dependants: l = 0: (x+1, y), l = 1: (x, y+1), l = 2: (x-1, y+1) — the mirror image of the dependency rule above.  The
             * "neighbours" of the first / last column that lie in another row never qualify: a macroblock of column 0 has no LEFT */
            const int dmb = l == 0 ? run + 1 : l == 1 ? run + wmb : run + wmb - 1;
            bool waits = false;
            if (dmb < hi_mb) {
                const uint32_t fd_ = anyf[dmb - base];
                waits = (fd_ & DBKF_ANY) && (l == 0 ? ((fd_ & DBKF_LEFT) != 0u && (fm & (DBKF_INNER | DBKF_TOP)) != 0u)
                                                  : l == 1 ? ((fd_ & DBKF_TOP) != 0u && (fm & (DBKF_INNER | DBKF_LEFT)) != 0u)
                                                           : ((fd_ & DBKF_TOP) != 0u && (fm & DBKF_LEFT) != 0u));
            }
            if (waits) release(dmb - base);
        }
        if (tp) { const unsigned long long t = __builtin_readcyclecounter(); t_store += t - t_mark; t_mark = t; }
    }
    if (tp && lane == 0) {
        tp[0] += t_idle; tp[1] += t_work; tp[2] += t_store; tp[3] += n_done; tp[4] += __builtin_readcyclecounter() - t_begin; tp[5] += n_steps;
    }
    /* the last band of the picture to leave zeroes the flags (k_dbk only visits non-trivial macroblocks) and the done bytes
     * for the next picture of this stream */
    bool last = true;
    if (BANDED && nb > 1) {
        __syncthreads();
        if (tid == 0) s_misc[1] = atomicAdd(scratch_exits(fd, 0), 1u);
        __syncthreads();
        last = s_misc[1] == (uint32_t)nb - 1u;
    }
    if (last) {
        uint32_t *z = reinterpret_cast<uint32_t *>(flags_g);
        const int words = (BANDED && nb > 1 ? 2 : 1) * (int)((fd.n_mbs + 3u) >> 2);   /* flags and this kernel's done bytes are adjacent */
        for (int i = tid; i < words; i += blockDim.x) z[i] = 0;
        if (BANDED && nb > 1 && tid == 0) atomicExch(scratch_exits(fd, 0), 0u);
    }
    (void)n_mbs;
    if (BANDED) return_ticket(tickets);
}

/* ------------------------------------------------------------------ frame jobs entering the device */
/* The frame jobs of one tick, fetched from the parser's pinned staging buffers by ONE launch: item i = one job (source in
 * host memory, mapped into the device's address space; destination in the lane's arena).  The host used to enqueue one
 * hipMemcpyAsync per job — 256 runtime calls per tick, 0.7 of the 0.9 ms the enqueueing thread spends between two rounds of
 * parsing (the parser threads wait for it).  H2D_CHUNKS workgroups per job, 16 bytes per lane and trip; sizes are multiples
 * of 32 (FjHeader.total_bytes). */
struct H2dItem { const uint8_t *src; uint8_t *dst; uint32_t bytes, pad; };
constexpr int H2D_CHUNKS = 8;
__global__ __launch_bounds__(256) void k_h2d(const H2dItem *__restrict__ items)
{
    const H2dItem it = items[blockIdx.y];
    const u32x4 *src = reinterpret_cast<const u32x4 *>(it.src);
    u32x4 *dst = reinterpret_cast<u32x4 *>(it.dst);
    const uint32_t n = it.bytes >> 4, stride = gridDim.x * 256u;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += 4u * stride) {
        /* four loads in flight per lane: the link's latency is microseconds */
        const uint32_t i1 = i + stride, i2 = i1 + stride, i3 = i2 + stride;
        const u32x4 a = __builtin_nontemporal_load(src + i);
        u32x4 b = a, c = a, d = a;
        if (i1 < n) b = __builtin_nontemporal_load(src + i1);
        if (i2 < n) c = __builtin_nontemporal_load(src + i2);
        if (i3 < n) d = __builtin_nontemporal_load(src + i3);
        dst[i] = a;
        if (i1 < n) dst[i1] = b;
        if (i2 < n) dst[i2] = c;
        if (i3 < n) dst[i3] = d;
    }
}

/* ------------------------------------------------------------------ pictures leaving the device */
/* The reference's output format is planar I420, uncropped (image.h:46-55).  Frames live in HBM as macroblock tiles,
 * so every path that hands a picture out reads tiles: k_detile (whole frame -> planar), k_output (cropped window ->
 * planar or converted), k_convert with tiled != 0 (whole frame -> RGBA / BGRA / YCbCrA).  k_convert with tiled == 0 is
 * the stateless h264bsdConvertTo*(), whose input is the caller's planar picture. */
__device__ __forceinline__ uint32_t yuv_luma4(const uint8_t *__restrict__ src, int tiled, uint32_t width, uint32_t x, uint32_t y)
{
    return tiled ? *reinterpret_cast<const uint32_t *>(src + luma_at((int)(width >> 4), (int)x, (int)y))
                 : *reinterpret_cast<const uint32_t *>(src + (size_t)y * width + x);
}
__device__ __forceinline__ uint32_t yuv_chroma2(const uint8_t *__restrict__ src, int tiled, uint32_t width, uint32_t height, int plane, uint32_t cx, uint32_t cy)
{
    const uint8_t *p = tiled ? src + chroma_at((int)(width >> 4), plane, (int)cx, (int)cy)
                             : src + (size_t)width * height + (plane ? (size_t)(width >> 1) * (height >> 1) : 0) + (size_t)cy * (width >> 1) + cx;
    return *reinterpret_cast<const uint16_t *>(p);
}
__device__ __forceinline__ uint32_t yuv_pixel(int fmt, int Yv, int cb, int cr)
{
    if (fmt == 2) return 0xFF000000u | ((uint32_t)cr << 16) | ((uint32_t)cb << 8) | (uint32_t)Yv;
    const int c = Yv - 16, d = cb - 128, e = cr - 128;
    const uint32_t r = clip255((298 * c + 409 * e + 128) >> 8);
    const uint32_t g = clip255((298 * c - 100 * d - 208 * e + 128) >> 8);
    const uint32_t b = clip255((298 * c + 516 * d + 128) >> 8);
    return fmt == 0 ? 0xFF000000u | (b << 16) | (g << 8) | r : 0xFF000000u | (r << 16) | (g << 8) | b;
}

/* 4 horizontally adjacent pixels per thread, one 16-byte store; fmt 0 RGBA, 1 BGRA, 2 YCbCrA (bytes in memory order);
 * integer BT.601 limited range, nearest chroma (reference src/h264bsd_decoder.c:1163-1370) */
__global__ __launch_bounds__(256) void k_convert(const uint8_t *__restrict__ yuv, uint32_t *__restrict__ out,
                                                 uint32_t width, uint32_t height, int fmt, size_t in_stride, size_t out_stride, int tiled)
{
    const uint8_t *src = yuv + blockIdx.y * in_stride;
    uint32_t *dst = out + blockIdx.y * out_stride;
    const uint32_t quads_per_row = width >> 2;
    const uint32_t total = quads_per_row * height;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t y = i / quads_per_row, x = (i % quads_per_row) * 4;
        const uint32_t yy = yuv_luma4(src, tiled, width, x, y);
        const uint32_t cb2 = yuv_chroma2(src, tiled, width, height, 0, x >> 1, y >> 1), cr2 = yuv_chroma2(src, tiled, width, height, 1, x >> 1, y >> 1);
        uint32_t px[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            px[k] = yuv_pixel(fmt, (yy >> (8 * k)) & 255, (cb2 >> (8 * (k >> 1))) & 255, (cr2 >> (8 * (k >> 1))) & 255);
        *reinterpret_cast<uint4 *>(dst + (size_t)y * width + x) = make_uint4(px[0], px[1], px[2], px[3]);
    }
}

/* Whole frames, tiles -> 32-bit pixels: ONE WAVEFRONT PER MACROBLOCK, lane = (row, quad of 4 pixels).  The 256 luma bytes
 * of the tile are one contiguous 4-byte-per-lane load, the 2 x 64 chroma bytes one 4-byte load of lanes 0..31 that is
 * handed round with two shuffles, and the four wavefronts of a workgroup take four neighbouring macroblocks, so that a
 * store instruction of the workgroup covers 256 contiguous bytes of 16 picture rows.  The chroma terms of the conversion
 * are computed once per pixel pair.  Same arithmetic as k_convert / yuv_pixel (reference decoder.c:1163-1370). */
__global__ __launch_bounds__(256) void k_convert_tiles(const uint8_t *__restrict__ yuv, uint32_t *__restrict__ out, uint32_t wmb, uint32_t hmb,
                                                       int fmt, size_t in_stride, size_t out_stride)
{
    const uint8_t *src = yuv + blockIdx.y * in_stride;
    uint32_t *dst = out + blockIdx.y * out_stride;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_mbs = wmb * hmb, W = wmb * 16;
    const uint32_t r = lane >> 2, q = lane & 3u;
    for (uint32_t mb = (blockIdx.x * 4u + wave); mb < n_mbs; mb += gridDim.x * 4u) {
        const uint8_t *T = src + (size_t)mb * TILE;
        const uint32_t yy = reinterpret_cast<const uint32_t *>(T)[lane];
        uint32_t cw = 0;
        if (lane < 32u) cw = reinterpret_cast<const uint32_t *>(T + T_CB)[lane];          /* lanes 0-15 Cb, 16-31 Cr */
        const int ci = (int)((r >> 1) * 2u + (q >> 1));                                    /* dword of chroma row r/2 holding samples 2q, 2q+1 */
        const uint32_t cbw = (uint32_t)__shfl((int)cw, ci) >> (16u * (q & 1u)), crw = (uint32_t)__shfl((int)cw, 16 + ci) >> (16u * (q & 1u));
        uint32_t px[4];
        if (fmt == 2) {
#pragma unroll
            for (int k = 0; k < 4; k++)
                px[k] = 0xFF000000u | (((crw >> (8 * (k >> 1))) & 255u) << 16) | (((cbw >> (8 * (k >> 1))) & 255u) << 8) | ((yy >> (8 * k)) & 255u);
        } else {
#pragma unroll
            for (int h2 = 0; h2 < 2; h2++) {
                const int d = (int)((cbw >> (8 * h2)) & 255u) - 128, e = (int)((crw >> (8 * h2)) & 255u) - 128;
                const int tr = 409 * e + 128, tg = -100 * d - 208 * e + 128, tb = 516 * d + 128;
#pragma unroll
                for (int k2 = 0; k2 < 2; k2++) {
                    const int k = 2 * h2 + k2, c = 298 * ((int)((yy >> (8 * k)) & 255u) - 16);
                    const uint32_t R = (uint32_t)clip255((c + tr) >> 8), G = (uint32_t)clip255((c + tg) >> 8), B = (uint32_t)clip255((c + tb) >> 8);
                    px[k] = fmt == 0 ? 0xFF000000u | (B << 16) | (G << 8) | R : 0xFF000000u | (R << 16) | (G << 8) | B;
                }
            }
        }
        const uint32_t mbx = mb % wmb, mby = mb / wmb;
        *reinterpret_cast<uint4 *>(dst + (size_t)(mby * 16u + r) * W + mbx * 16u + q * 4u) = make_uint4(px[0], px[1], px[2], px[3]);
    }
}

/* Whole frames, tiles -> planar I420 (what h264bsdNextOutputPicture() returns): 16 bytes per thread, a luma row piece
 * or two chroma row pieces of one tile; reads are contiguous per tile, writes 16-byte pieces of planar rows. */
__global__ __launch_bounds__(256) void k_detile(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint32_t wmb, uint32_t hmb,
                                                size_t in_stride, size_t out_stride)
{
    const uint8_t *s = src + blockIdx.y * in_stride;
    uint8_t *d = dst + blockIdx.y * out_stride;
    const uint32_t W = wmb * 16, CW = W >> 1;
    const size_t ysz = (size_t)W * hmb * 16, csz = ysz >> 2;
    const uint32_t total = wmb * hmb * (TILE / 16);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const uint32_t mb = i / (TILE / 16), pc = i % (TILE / 16), mbx = mb % wmb, mby = mb / wmb;
        const uint4 v = *reinterpret_cast<const uint4 *>(s + (size_t)i * 16);
        if (pc < 16) *reinterpret_cast<uint4 *>(d + (size_t)(mby * 16 + pc) * W + mbx * 16) = v;
        else {
            const uint32_t plane = (pc - 16) >> 2, r = ((pc - 16) & 3) * 2;     /* two 8-byte chroma rows */
            uint8_t *q = d + ysz + (plane ? csz : 0) + (size_t)(mby * 8 + r) * CW + mbx * 8;
            *reinterpret_cast<uint2 *>(q) = make_uint2(v.x, v.y);
            *reinterpret_cast<uint2 *>(q + CW) = make_uint2(v.z, v.w);
        }
    }
}

/* Device-resident output: the window (x0,y0,w,h) of a decoded frame (even offsets and sizes, multiples of 4 for the
 * window width) either converted (fmt 0..2, tightly packed w*h u32, 4 pixels per lane) or as a tight I420 picture
 * (fmt 3: w*h Y, then the two (w/2)*(h/2) chroma planes; 4 luma samples or 2 chroma samples per lane). */
__global__ __launch_bounds__(256) void k_output(const uint8_t *__restrict__ src, uint8_t *__restrict__ dst, uint32_t width,
                                                uint32_t height, int fmt, uint32_t x0, uint32_t y0, uint32_t w, uint32_t h)
{
    const uint32_t qw = w >> 2;
    if (w & 3u) {
        /* window width not a multiple of 4 (cropping is in units of 2 luma samples): one sample / pixel per lane */
        const int twmb = (int)(width >> 4);
        if (fmt == 3) {
            const uint32_t ny = w * h, nc = (w >> 1) * (h >> 1);
            for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ny + 2 * nc; i += gridDim.x * blockDim.x) {
                if (i < ny) dst[i] = src[luma_at(twmb, (int)(x0 + i % w), (int)(y0 + i / w))];
                else {
                    const uint32_t j = (i - ny) % nc, cw = w >> 1;
                    dst[i] = src[chroma_at(twmb, i - ny >= nc, (int)((x0 >> 1) + j % cw), (int)((y0 >> 1) + j / cw))];
                }
            }
        } else {
            uint32_t *o32 = reinterpret_cast<uint32_t *>(dst);
            for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < w * h; i += gridDim.x * blockDim.x) {
                const uint32_t y = y0 + i / w, x = x0 + i % w;
                o32[i] = yuv_pixel(fmt, src[luma_at(twmb, (int)x, (int)y)], src[chroma_at(twmb, 0, (int)(x >> 1), (int)(y >> 1))],
                                   src[chroma_at(twmb, 1, (int)(x >> 1), (int)(y >> 1))]);
            }
        }
        return;
    }
    if (fmt == 3) {
        const uint32_t ny4 = qw * h, cw = w >> 1, nc2 = (cw >> 1) * (h >> 1);
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < ny4 + 2 * nc2; i += gridDim.x * blockDim.x) {
            if (i < ny4) {
                const uint32_t y = i / qw, x = (i % qw) * 4;
                uint32_t v = 0;
#pragma unroll
                for (int k = 0; k < 4; k++) v |= (uint32_t)src[luma_at((int)(width >> 4), (int)(x0 + x + k), (int)(y0 + y))] << (8 * k);
                *reinterpret_cast<uint32_t *>(dst + (size_t)y * w + x) = v;
            } else {
                const uint32_t j = i - ny4, plane = j >= nc2, jj = plane ? j - nc2 : j, y = jj / (cw >> 1), x = (jj % (cw >> 1)) * 2;
                uint32_t v = 0;
#pragma unroll
                for (int k = 0; k < 2; k++) v |= (uint32_t)src[chroma_at((int)(width >> 4), (int)plane, (int)((x0 >> 1) + x + k), (int)((y0 >> 1) + y))] << (8 * k);
                *reinterpret_cast<uint16_t *>(dst + (size_t)w * h + (plane ? (size_t)cw * (h >> 1) : 0) + (size_t)y * cw + x) = (uint16_t)v;
            }
        }
        return;
    }
    (void)height;
    uint32_t *out = reinterpret_cast<uint32_t *>(dst);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < qw * h; i += gridDim.x * blockDim.x) {
        const uint32_t y = y0 + i / qw, x = x0 + (i % qw) * 4;
        uint32_t px[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int Yv = src[luma_at((int)(width >> 4), (int)(x + k), (int)y)];
            const int cb = src[chroma_at((int)(width >> 4), 0, (int)((x + k) >> 1), (int)(y >> 1))];
            const int cr = src[chroma_at((int)(width >> 4), 1, (int)((x + k) >> 1), (int)(y >> 1))];
            px[k] = yuv_pixel(fmt, Yv, cb, cr);
        }
        *reinterpret_cast<uint4 *>(out + (size_t)(i / qw) * w + (i % qw) * 4) = make_uint4(px[0], px[1], px[2], px[3]);
    }
}

/* ------------------------------------------------------------------ on-device verification */
/* sum over the 32-bit words w[i] of the PLANAR picture of (w[i] ^ i*0x9E3779B1) * (2i+1)  (mod 2^64); one block per
 * frame.  The frame is stored as tiles: every 4-byte piece of a tile is one word of the planar picture, whose index i
 * follows from the macroblock position — the value is the one the golden files hold for the reference's output. */
__global__ __launch_bounds__(256) void k_checksum(const uint8_t *__restrict__ base, size_t stride, uint32_t wmb, uint32_t hmb,
                                                  unsigned long long *__restrict__ out)
{
    __shared__ unsigned long long part[256];
    const uint32_t *w = reinterpret_cast<const uint32_t *>(base + blockIdx.x * stride);
    const uint32_t words = wmb * hmb * (TILE / 4), W4 = wmb * 4, CW4 = wmb * 2;
    const uint32_t ywords = W4 * hmb * 16, cwords = ywords >> 2;
    unsigned long long acc = 0;
    for (uint32_t t = threadIdx.x; t < words; t += 256) {
        const uint32_t mb = t / (TILE / 4), k = t % (TILE / 4), mbx = mb % wmb, mby = mb / wmb;
        uint32_t i;
        if (k < 64) i = (mby * 16 + (k >> 2)) * W4 + mbx * 4 + (k & 3);
        else {
            const uint32_t kk = k - 64, plane = kk >> 4, r = (kk & 15) >> 1, half = kk & 1;
            i = ywords + plane * cwords + (mby * 8 + r) * CW4 + mbx * 2 + half;
        }
        acc += (unsigned long long)(w[t] ^ (i * 0x9E3779B1u)) * (unsigned long long)(2u * i + 1u);
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = part[0];
}

} // namespace h264k
