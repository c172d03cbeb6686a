/* kernels/k_frame_dbk.hip.h — k_frame_dbk: the in-loop filter, one workgroup per picture (or row band).  Part of kernels.hip.h (which see); not a stand-alone header. */
#pragma once
namespace h264k {
/* ------------------------------------------------------------------ deblocking */
/* Round 6 decomposition.  LUMA and CHROMA of a picture are two independent dependency graphs (same shape, same flags) walked by
 * different wavefronts of the picture's workgroup: a picture's time is the longest chain of dependent steps, and a chroma
 * filter that runs inside the luma step lengthens every link of it.  A WORKER is an eighth of a wavefront (8 lanes per
 * macroblock, up to eight macroblocks per wavefront step); samples are two lines per register (packed 16-bit):
 *   luma   vertical edges: lane l holds rows 2l, 2l+1 — straight from the 32 bytes it loaded (tile rows are contiguous);
 *          horizontal edges: lane l holds columns 2l, 2l+1.
 *   chroma lane l = plane l>>2, rows / columns 2(l&3), 2(l&3)+1.
 * Between the two passes the macroblock is transposed as an 8 x 8 matrix of DWORDS — a dword is the 2 x 2 sample block
 * (row pair, column pair) — through a worker-private LDS buffer: two ds_write_b128 + four ds_read2_b32 per lane, free of
 * bank conflicts (layout below), eight v_perm to pack and sixteen to unpack; the same trip brings the filtered columns
 * back to rows, so that a macroblock leaves as two 16-byte row stores per lane.  What the neighbours own comes and goes in
 * the layout of the pass that needs it: the left strip as row dwords, the upper strip as 16-bit column pairs straight
 * from / to global memory.  (Rounds 3-5 kept a 20 x 48 byte tile per worker in LDS and moved single bytes and 16-bit pairs
 * between it and the passes: 840 of the 1500 vector instructions of a full step.)
 * reference: h264bsdFilterPicture and below, src/h264bsd_deblocking.c:575-1745 (FilterLuma order :1551-1623). */
constexpr int DBK_LANES = 8;                         /* lanes per worker */
/* worker-private exchange buffers.  Luma: two 128-byte halves (block columns 0..3 | 4..7 of every lane's row) 144 bytes
 * apart, then the 64-byte strip area (four rows of the upper neighbour); a worker's buffer 352 bytes.  For one ds_read_b32 phase
 * the eight lanes of a worker then read eight consecutive banks and the four workers of a 32-lane group start 24, 16 and 8
 * banks apart (352 / 4 = 88 = 24 mod 32).  Chroma: 128 bytes, 16 free, a 32-byte strip area (two rows per plane); 176 bytes. */
constexpr int DBK_LW = 352, DBK_LW_HI = 144, DBK_LW_STRIP = 288, DBK_CW = 176, DBK_CW_STRIP = 144;
constexpr int DBK_WAVE_LDS = 8 * DBK_LW;             /* 2816 bytes per wavefront, whichever role it plays */

__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)(const H264K_LDS uint8_t *)p; }
/* LDS accesses by 32-bit address (ds_write_b128 / ds_read_b32 / ds_read_u16; the compiler pairs neighbouring reads into ds_read2_b32) */
__device__ __forceinline__ void lds_st128(uint32_t a, uint32_t x, uint32_t y, uint32_t z, uint32_t w) { *(H264K_LDS u32x4 *)(uintptr_t)a = (u32x4){ x, y, z, w }; }
__device__ __forceinline__ uint32_t lds_ld32(uint32_t a) { return *(const H264K_LDS uint32_t *)(uintptr_t)a; }
__device__ __forceinline__ uint32_t lds_ld16(uint32_t a) { return *(const H264K_LDS uint16_t *)(uintptr_t)a; }
__device__ __forceinline__ uint2 lds_ld64(uint32_t a) { const u32x2 v = *(const H264K_LDS u32x2 *)(uintptr_t)a; return make_uint2(v.x, v.y); }
__device__ __forceinline__ void lds_st64(uint32_t a, uint32_t x, uint32_t y) { *(H264K_LDS u32x2 *)(uintptr_t)a = (u32x2){ x, y }; }
__device__ __forceinline__ void lds_st32(uint32_t a, uint32_t x) { *(H264K_LDS uint32_t *)(uintptr_t)a = x; }
__device__ __forceinline__ void lds_st16(uint32_t a, uint32_t x) { *(H264K_LDS uint16_t *)(uintptr_t)a = (uint16_t)x; }

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
/* (m & a) | (~m & b) in ONE instruction.  (Written in C the compiler shares sub-terms between the six selects of an edge and ends
 * up with two instructions for most of them.) */
__device__ __forceinline__ uint32_t bfi(uint32_t m, uint32_t a, uint32_t b)
{
    uint32_t r;
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(r) : "v"(m), "v"(a), "v"(b));
    return r;
}
/* The sample registers of a pass are plain 32-bit values that the filters read and write as packed pairs: kept as <2 x i16>
 * across the branches that skip inactive edges, every register that passes a branch is taken apart into its halves and put
 * together again (v_lshrrev_b32 + v_perm_b32 per register and edge: a quarter of a full step's vector instructions). */
__device__ __forceinline__ void filter_luma_pk(uint32_t v[8], int bs, s2 A, s2 B, int tc0, s2 one)
{
    const s2 p3 = as_s2(v[0]), p2 = as_s2(v[1]), p1 = as_s2(v[2]), p0 = as_s2(v[3]), q0 = as_s2(v[4]), q1 = as_s2(v[5]), q2 = as_s2(v[6]), q3 = as_s2(v[7]);
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
        v[1] = bfi(as_u32(fs & m_p2), as_u32(r_p2), v[1]);
        v[6] = bfi(as_u32(fs & m_q2), as_u32(r_q2), v[6]);
    }
    v[3] = bfi(as_u32(fs), as_u32(r_p0), v[3]);
    v[4] = bfi(as_u32(fs), as_u32(r_q0), v[4]);
    v[2] = bfi(as_u32(fs & m_p1), as_u32(r_p1), v[2]);
    v[5] = bfi(as_u32(fs & m_q1), as_u32(r_q1), v[5]);
}

/* chroma (chromaEdgeFlag = 1): only p0 and q0 change; v = p1, p0, q0, q1 */
__device__ __forceinline__ void filter_chroma_pk(uint32_t v[4], int bs, s2 A, s2 B, int tc0)
{
    const s2 p1 = as_s2(v[0]), p0 = as_s2(v[1]), q0 = as_s2(v[2]), q1 = as_s2(v[3]);
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
    v[1] = bfi(as_u32(fs), as_u32(r_p0), v[1]);
    v[2] = bfi(as_u32(fs), as_u32(r_q0), v[2]);
}

/* ---- stores into the picture: plain, or write-through (agent scope) for samples another row band will read ---- */
__device__ __forceinline__ void stg_u8(uint8_t *cur, uint32_t o, uint32_t v, bool wt)
{
    if (wt) st_agent_u8(cur + o, v); else *((H264K_GLOBAL uint8_t *)cur + o) = (uint8_t)v;
}
__device__ __forceinline__ void stg_u16(uint8_t *cur, uint32_t o, uint32_t v, bool wt)
{
    if (wt) __hip_atomic_store((H264K_GLOBAL uint16_t *)(cur + o), (uint16_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *(H264K_GLOBAL uint16_t *)((H264K_GLOBAL uint8_t *)cur + o) = (uint16_t)v;
}
__device__ __forceinline__ void stg_u32(uint8_t *cur, uint32_t o, uint32_t v, bool wt)
{
    if (wt) st_agent_u32(cur + o, v); else *(H264K_GLOBAL uint32_t *)((H264K_GLOBAL uint8_t *)cur + o) = v;
}
__device__ __forceinline__ void stg_b64(uint8_t *cur, uint32_t o, uint2 v, bool wt)
{
    if (wt) put8(cur + o, v, true); else *(H264K_GLOBAL u32x2 *)((H264K_GLOBAL uint8_t *)cur + o) = (u32x2){ v.x, v.y };
}
__device__ __forceinline__ void stg_b128(uint8_t *cur, uint32_t o, uint4 v, bool wt)
{
    if (wt) put16(cur + o, v, true); else st16g((H264K_GLOBAL uint8_t *)cur + o, v);
}
__device__ __forceinline__ uint32_t ldg_u16(const uint8_t *cur, uint32_t o) { return *(const H264K_GLOBAL uint16_t *)((const H264K_GLOBAL uint8_t *)cur + o); }
__device__ __forceinline__ uint32_t ldg_u32(const uint8_t *cur, uint32_t o) { return *(const H264K_GLOBAL uint32_t *)((const H264K_GLOBAL uint8_t *)cur + o); }
__device__ __forceinline__ uint32_t ld_agent_u16(const void *p)
{
    return __hip_atomic_load((const H264K_GLOBAL uint16_t *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

/* the lane's strength at edge slot e (luma edges 0..3) of a direction whose eight strength bytes are w0 | w1, already shifted
 * right by 4 * (segment of the lane): nibble n = 4 * e + k sits at bit 16 * (e & 1) of dword e >> 1 */
__device__ __forceinline__ int bs_of(uint32_t w0s, uint32_t w1s, int e) { return (int)(((e & 2 ? w1s : w0s) >> (16 * (e & 1))) & 15u); }
/* tc0 of a class for strength bs: t4 = { 0, tc0(1), tc0(2), tc0(3) } as bytes; bs 0 and 4 give 0 */
/* (one v_perm_b32 with the strength as the selector: byte bs of { t4, 0 }; the selector's upper bytes pick byte 0 = 0) */
__device__ __forceinline__ int tc0_of(uint32_t t4, int bs) { return (int)perm(0u, t4, (uint32_t)bs); }

/* the 2 x 2 block dword (a.lo, a.hi, b.lo, b.hi) of two packed pairs a, b (samples are < 256: byte 0 and byte 2 of each) */
__device__ __forceinline__ uint32_t blk_of(uint32_t a, uint32_t b) { return perm(b, a, 0x06040200u); }
/* ... and the other way round: (a.lo, b.lo, a.hi, b.hi) — the block of two COLUMN pairs a, b in (row-major pair, column) order */
__device__ __forceinline__ uint32_t blk_t_of(uint32_t a, uint32_t b) { return perm(b, a, 0x06020400u); }
__device__ __forceinline__ uint32_t pair_lo(uint32_t blk) { return perm(0u, blk, 0x0C020C00u); }     /* (byte 0, byte 2) */
__device__ __forceinline__ uint32_t pair_hi(uint32_t blk) { return perm(0u, blk, 0x0C030C01u); }     /* (byte 1, byte 3) */

#ifdef H264K_TAIL_PROFILE
#define DTICK() (tp ? (uint32_t)__builtin_readcyclecounter() : 0u)
#else
#define DTICK() 0u
#endif

/* cycle accounting of a wavefront's steps (debug builds with -DH264K_TAIL_PROFILE; members, not an array: registers, not scratch memory) */
struct DbkProf { uint32_t wait, v, h, st, q, ld; };       /* (32 bits: a tick is a million cycles) */

/* The descriptor fields the steps use, fetched ONCE per workgroup.  (Read through the FrameDesc reference, the picture's address is a
 * scalar load in front of every step's loads and another one in front of its stores — the compiler re-materialises loads from the
 * constant address space instead of keeping two scalar registers — and each waits for the scalar cache, which 256 pictures' descriptors
 * and twelve wavefronts' kernel arguments do not fit.) */
struct DbkCtx { uint8_t *cur; const uint8_t *recs; uint32_t wmb; };
__device__ __forceinline__ DbkCtx dbk_ctx(const FrameDesc &fd)
{
    DbkCtx c;
    unsigned long long a = (unsigned long long)(uintptr_t)fd.cur, b = (unsigned long long)(uintptr_t)fd.dbk;
    uint32_t w = fd.wmb;
    asm volatile("" : "+s"(a), "+s"(b), "+s"(w));                  /* opaque: values, not reloadable expressions */
    c.cur = (uint8_t *)(uintptr_t)a; c.recs = (const uint8_t *)(uintptr_t)b; c.wmb = w;
    return c;
}

/* ================================================================== luma */
/* What a vector memory instruction costs the compute unit (tools/probes/vmem_issue_probe.hip, 12 wavefronts per CU, cycles of the
 * CU's address path per wave-level instruction): it grows with the 32-byte SECTORS and 128-byte lines the 64 lanes touch, hardly
 * with the bytes — a dword per lane out of every second row of eight tiles (the left strips) 52 as a load and 62 as a store,
 * 16 bits per lane out of one row of eight tiles 17 / 20, a 16-byte row piece per lane 18 / 56, eight lanes reading the same 16
 * bytes 14.  Twelve wavefronts walking chains share that path, and a step used to open with 13 loads and end with 7-9 stores.
 * So the strips travel WIDE and in as few instructions as possible, and what a pass needs in another shape is reshaped in LDS:
 *   left strip   loaded as the 8-byte row halves that hold it; stored as dwords, lanes regrouped by DPP so that one instruction
 *                covers rows 0..7 (one 128-byte line per tile) and the other rows 8..15;
 *   upper strip  loaded as 8 bytes per lane (the four rows are 64 contiguous bytes), turned into column pairs through the
 *                worker's strip area in LDS; rows 13..15 return the same way and leave as 8 bytes from six lanes;
 *   own rows     the second transposition hands every lane the rows that make instruction A write rows 0..7 (a full line per
 *                tile, every sector whole) and instruction B rows 8..15. */
/* Everything a luma worker loads, all of it requested before the first use (one memory round trip per step). */
struct DbkLumaLoads {
    uint4 y0, y1;             /* rows 2l, 2l+1: 32 contiguous bytes of the tile                                          */
    uint2 l0, l1;             /* columns 8..15 of the same rows of the tile to the left (.y = its last four columns)       */
    uint2 t;                  /* dwords 2l, 2l+1 of the last four rows of the tile above (row 12 + l/2, columns 8 (l&1) ..) */
    uint4 r0, r1; uint2 r2;   /* the record: strengths | class dwords luma left / top / inner (, chroma left) | bytes 40..47 */
};

/* cross: the macroblock lies in the first row of a row band: the tile above belongs to another workgroup, its last rows are
 * read past the L1 — and only when this macroblock's upper edge is filtered at all (want_top), because an unconditional
 * load could run ahead of the other band's stores. */
__device__ __forceinline__ void dbk_luma_load(const DbkCtx &fd, int mb, int l, DbkLumaLoads &p, bool cross, bool want_top)
{
    if (mb < 0) return;
    const uint8_t *cur = fd.cur;
    const H264K_GLOBAL uint8_t *curg = (const H264K_GLOBAL uint8_t *)cur;
    const H264K_GLOBAL uint8_t *recs = (const H264K_GLOBAL uint8_t *)fd.recs;
    const uint32_t umb = (uint32_t)mb, wmb = fd.wmb;
    const uint32_t t = umb * TILE, tl = (umb ? umb - 1u : 0u) * TILE, tu = (umb >= wmb ? umb - wmb : umb) * TILE;     /* stand-ins where there is no neighbour: never used (k_dbk: LEFT / TOP only where it exists) */
    const uint32_t ro = umb * DBK_REC_BYTES;
    p.r0 = ld16g(recs + ro);
    p.r1 = ld16g(recs + ro + 16u);
    p.r2 = ld8g(recs + ro + 40u);
    p.y0 = ld16g(curg + t + 32u * l);
    p.y1 = ld16g(curg + t + 32u * l + 16u);
    p.l0 = ld8g(curg + tl + 32u * l + 8u);
    p.l1 = ld8g(curg + tl + 32u * l + 24u);
    const uint32_t uy = tu + 192u + 8u * l;
    p.t = make_uint2(0u, 0u);
    if (!cross) p.t = ld8g(curg + uy);
    else if (want_top) p.t = make_uint2(ld_agent_u32(cur + uy), ld_agent_u32(cur + uy + 4u));
}

/* In-loop filter of the luma of one macroblock by one worker = 8 lanes: vertical edges, then horizontal edges (8.7).
 * mb < 0: this eighth of the wavefront idles.  wb = LDS address of the worker's exchange buffer.
 * SLOTS = 4: every edge (the step was claimed from the list of macroblocks with an active inner edge: DBKF_INNER); the
 *   macroblock is stored whole.
 * SLOTS = 1: none of the wavefront's macroblocks has an active inner edge — only the left and the upper macroblock edge
 *   exist: a quarter of the arithmetic, no transposition (the four rows the upper edge works on pass through LDS as rows).
 *   Such a macroblock STORES only what its two macroblock edges can have changed — columns 0..3 (left edge) and rows 0..2
 *   (upper edge) of its own tile — because the macroblocks to its right and below it no longer wait for it unless those
 *   very samples concern them (k_frame_dbk, dependency rule) and may be rewriting the rest of its tile at the same time.
 * wt: the macroblock lies in the last row of a row band, the band below reads what it writes: everything goes write-through. */
template <bool BANDED, int SLOTS>
__device__ __forceinline__ void dbk_luma_step(const DbkCtx &fd, int mb, int l, const DbkLumaLoads &p, uint32_t wb, bool wt_, DbkProf *tp)
{
    constexpr int NPX = SLOTS == 4 ? 20 : 8;
    const bool wt = BANDED && wt_;
    const bool act = mb >= 0;
    const uint32_t d0 = DTICK();
    const uint32_t flags = p.r2.y >> 16;                          /* byte 46: FJ_DBK_*, byte 47: any */
    const bool f_left = act && (flags & FJ_DBK_LEFT) && (p.r0.x & 0xFFFFu), f_top = act && (flags & FJ_DBK_TOP) && (p.r0.z & 0xFFFFu);
    const bool any_v = __ballot(act && (SLOTS == 4 ? (p.r0.x | p.r0.y) : (p.r0.x & 0xFFFFu))) != 0ull;   /* wave-wide phase skips */
    const bool any_h = __ballot(act && (SLOTS == 4 ? (p.r0.z | p.r0.w) : (p.r0.z & 0xFFFFu))) != 0ull;
    /* thresholds per class: A / B = alpha / beta in both halves, t4 = { 0, tc0(1), tc0(2), tc0(3) } */
    const uint32_t w_ll = p.r1.x, w_lt = p.r1.y, w_li = p.r1.z, t3a = p.r2.x;      /* t3a: bytes 40..43 = tc0(3) of luma left / top / inner */
    const int k = l >> 1;                                          /* both lines of a lane lie in segment k of every edge */
    const uint32_t sb = wb + DBK_LW_STRIP;                         /* the strip area: four rows of 16 bytes */
    const bool lo4 = l < 4;
    s2 one = pk(1);
    asm volatile("" : "+v"(one));
#ifdef H264K_TAIL_PROFILE
    if (tp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    const uint32_t d1 = DTICK();

    /* ---- vertical edges: rows 2l (low halves) and 2l+1 (high halves); px[0..3] = the left neighbour's last columns ---- */
    asm volatile("" :: "v"(p.l0.x), "v"(p.l1.x));            /* (the unused halves keep the two loads 8 bytes wide: as dword loads of every second row's last
                                                                 columns they cost the CU's address path 52 cycles each instead of 18) */
    uint32_t px[NPX];
    px[0] = perm(p.l1.y, p.l0.y, 0x0C040C00u); px[1] = perm(p.l1.y, p.l0.y, 0x0C050C01u);
    px[2] = perm(p.l1.y, p.l0.y, 0x0C060C02u); px[3] = perm(p.l1.y, p.l0.y, 0x0C070C03u);
    {
        const uint32_t ra[4] = { p.y0.x, p.y0.y, p.y0.z, p.y0.w }, rb[4] = { p.y1.x, p.y1.y, p.y1.z, p.y1.w };
#pragma unroll
        for (int g = 0; g < NPX / 4 - 1; g++) {
            px[4 + 4 * g + 0] = perm(rb[g], ra[g], 0x0C040C00u);
            px[4 + 4 * g + 1] = perm(rb[g], ra[g], 0x0C050C01u);
            px[4 + 4 * g + 2] = perm(rb[g], ra[g], 0x0C060C02u);
            px[4 + 4 * g + 3] = perm(rb[g], ra[g], 0x0C070C03u);
        }
    }
    if (any_v) {
        const uint32_t w0s = p.r0.x >> (4 * k), w1s = p.r0.y >> (4 * k);
        const s2 A_l = pk_splat_byte(w_ll, 0), B_l = pk_splat_byte(w_ll, 1), A_i = pk_splat_byte(w_li, 0), B_i = pk_splat_byte(w_li, 1);
        const uint32_t t4_l = perm(t3a, w_ll, 0x0403020Cu), t4_i = perm(t3a, w_li, 0x0603020Cu);    /* { 0, tc0(1), tc0(2), tc0(3) } */
#pragma unroll
        for (int e = 0; e < SLOTS; e++) {
            const int bs = act ? bs_of(w0s, w1s, e) : 0;
            if (__ballot(bs != 0)) filter_luma_pk(px + 4 * e, bs, e ? A_i : A_l, e ? B_i : B_l, tc0_of(e ? t4_i : t4_l, bs), one);
        }
    }
    lds_st64(sb + 8u * l, p.t.x, p.t.y);                           /* the upper strip as rows; read back as column pairs below */
    /* the left neighbour's columns 12..15 as the two row dwords they are stored as */
    uint32_t lo0, lo1;
    {
        const uint32_t a = blk_of(px[0], px[1]), b = blk_of(px[2], px[3]);
        lo0 = perm(b, a, 0x06040200u); lo1 = perm(b, a, 0x07050301u);
    }
    const uint32_t d2 = DTICK();

    /* ---- to columns: hx[r + 4] = row r, columns 2l (low half), 2l+1 (high half); hx[0..3] = the upper neighbour's last rows ---- */
    uint32_t hx[NPX];
    uint32_t own0 = 0u, own1 = 0u;                                 /* SLOTS = 1: columns 0..3 of rows 2l, 2l+1 after the vertical pass */
    if constexpr (SLOTS == 4) {
        /* block (row pair l, column pair j) = (row 2l col 2j, row 2l+1 col 2j, row 2l col 2j+1, row 2l+1 col 2j+1) */
        uint32_t D[8];
#pragma unroll
        for (int j = 0; j < 8; j++) D[j] = blk_of(px[4 + 2 * j], px[5 + 2 * j]);
        lds_st128(wb + 16u * l, D[0], D[1], D[2], D[3]);
        lds_st128(wb + DBK_LW_HI + 16u * l, D[4], D[5], D[6], D[7]);
        wave_sync();
        const uint32_t ra_ = wb + (lo4 ? 4u * l : (uint32_t)DBK_LW_HI + 4u * (l - 4));     /* this lane's column of blocks */
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t e = lds_ld32(ra_ + 16u * r);            /* block (row pair r, column pair l) */
            hx[4 + 2 * r] = pair_lo(e); hx[5 + 2 * r] = pair_hi(e);
        }
    } else {
        {
            const uint32_t a = blk_of(px[4], px[5]), b = blk_of(px[6], px[7]);
            own0 = perm(b, a, 0x06040200u); own1 = perm(b, a, 0x07050301u);
        }
        /* rows as rows: row r at wb + 16 r; the upper edge reads rows 0..3 */
        lds_st128(wb + 32u * l, own0, p.y0.y, p.y0.z, p.y0.w);
        lds_st128(wb + 32u * l + 16u, own1, p.y1.y, p.y1.z, p.y1.w);
        wave_sync();
#pragma unroll
        for (int r = 0; r < 4; r++) hx[4 + r] = perm(0u, lds_ld16(wb + 16u * r + 2u * l), 0x0C010C00u);
    }
#pragma unroll
    for (int r = 0; r < 4; r++) hx[r] = perm(0u, lds_ld16(sb + 16u * r + 2u * l), 0x0C010C00u);

    /* ---- horizontal edges ---- */
    if (any_h) {
        const uint32_t w0s = p.r0.z >> (4 * k), w1s = p.r0.w >> (4 * k);
        const s2 A_t = pk_splat_byte(w_lt, 0), B_t = pk_splat_byte(w_lt, 1), A_i = pk_splat_byte(w_li, 0), B_i = pk_splat_byte(w_li, 1);
        const uint32_t t4_t = perm(t3a, w_lt, 0x0503020Cu), t4_i = perm(t3a, w_li, 0x0603020Cu);
#pragma unroll
        for (int e = 0; e < SLOTS; e++) {
            const int bs = act ? bs_of(w0s, w1s, e) : 0;
            if (__ballot(bs != 0)) filter_luma_pk(hx + 4 * e, bs, e ? A_i : A_t, e ? B_i : B_t, tc0_of(e ? t4_i : t4_t, bs), one);
        }
    }
    const uint32_t d3 = DTICK();

    /* ---- store ---- */
    uint8_t *cur = fd.cur;
    const uint32_t t = (uint32_t)mb * TILE;
    /* rows 13..15 of the upper neighbour (row 12 = p3 never changes) go back through the strip area ... */
#pragma unroll
    for (int r = 1; r < 4; r++) lds_st16(sb + 16u * r + 2u * l, perm(0u, hx[r], 0x0C0C0200u));
    if constexpr (SLOTS == 4) {
        /* back to rows: block (row pair r, column pair l) = (row 2r col 2l, row 2r+1 col 2l, row 2r col 2l+1, row 2r+1 col 2l+1) */
        uint32_t F[8];
#pragma unroll
        for (int r = 0; r < 8; r++) F[r] = blk_t_of(hx[4 + 2 * r], hx[5 + 2 * r]);
        lds_st128(wb + 16u * l, F[0], F[1], F[2], F[3]);           /* (every lane has read the first trip: LDS operations of a wavefront execute in order) */
        lds_st128(wb + DBK_LW_HI + 16u * l, F[4], F[5], F[6], F[7]);
        wave_sync();
        /* instruction A stores rows 0..7, instruction B rows 8..15 — one whole 128-byte line per tile each: lanes 0..3 take the even
         * row of row pair l & 3 (+ 4 for B), lanes 4..7 the odd one (two lanes read the same blocks: an LDS broadcast) */
        const uint32_t rsel = lo4 ? 0x06040200u : 0x07050301u;
        const uint32_t rq = wb + 4u * (l & 3);
        uint32_t Ga[8], Gb[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { Ga[j] = lds_ld32(rq + 16u * j); Gb[j] = lds_ld32(rq + DBK_LW_HI + 16u * j); }      /* blocks (row pair l & 3 (+ 4), column pair j) */
        if (act) {
            const uint32_t o = t + 32u * (l & 3) + (lo4 ? 0u : 16u);
            stg_b128(cur, o, make_uint4(perm(Ga[1], Ga[0], rsel), perm(Ga[3], Ga[2], rsel), perm(Ga[5], Ga[4], rsel), perm(Ga[7], Ga[6], rsel)), wt);
            stg_b128(cur, o + 128u, make_uint4(perm(Gb[1], Gb[0], rsel), perm(Gb[3], Gb[2], rsel), perm(Gb[5], Gb[4], rsel), perm(Gb[7], Gb[6], rsel)), wt);
        }
    } else {
        /* columns 0..3 of the rows the upper edge does not rewrite: lanes regrouped so that one instruction covers rows 0..7, the
         * other rows 8..15 (DPP row_shr:4 / row_shl:4 with a bank mask: the other half of the worker keeps its own dword) */
        const uint32_t va = (uint32_t)__builtin_amdgcn_update_dpp((int)own0, (int)own1, 0x114, 0xF, 0xA, false);   /* lanes 4..7: row 2(l-4)+1 of lane l-4 */
        const uint32_t vb = (uint32_t)__builtin_amdgcn_update_dpp((int)own1, (int)own0, 0x104, 0xF, 0x5, false);   /* lanes 0..3: row 2(l+4) of lane l+4 */
        if (f_left) {
            const int row_a = lo4 ? 2 * l : 2 * (l - 4) + 1;      /* rows 0..2 leave with the horizontal pass when the upper edge is filtered */
            if (!(f_top && row_a < 3)) stg_u32(cur, t + 16u * row_a, va, wt);
            stg_u32(cur, t + 128u + 32u * (l & 3) + (lo4 ? 0u : 16u), vb, wt);
        }
        /* rows 0..2 after both passes: as rows again through LDS, 8 bytes from each of six lanes */
#pragma unroll
        for (int r = 0; r < 3; r++) lds_st16(wb + 16u * r + 2u * l, perm(0u, hx[4 + r], 0x0C0C0200u));
        wave_sync();
        if (f_top && l < 6) {
            const uint2 v = lds_ld64(wb + 8u * l);
            stg_b64(cur, t + 8u * l, v, wt);
        }
    }
    if (f_left) {
        const uint32_t tl = t - TILE;
        const uint32_t va = (uint32_t)__builtin_amdgcn_update_dpp((int)lo0, (int)lo1, 0x114, 0xF, 0xA, false);
        const uint32_t vb = (uint32_t)__builtin_amdgcn_update_dpp((int)lo1, (int)lo0, 0x104, 0xF, 0x5, false);
        const uint32_t o = tl + 32u * (l & 3) + (lo4 ? 12u : 28u);
        stg_u32(cur, o, va, wt);
        stg_u32(cur, o + 128u, vb, wt);
    }
    wave_sync();
    if (f_top && l < 6) {
        const uint2 v = lds_ld64(sb + 16u + 8u * l);               /* ... and leave as 8 bytes from each of six lanes */
        stg_b64(cur, t - (uint32_t)fd.wmb * TILE + 208u + 8u * l, v, wt);
    }
    wave_sync();          /* the exchange buffer is reused by this worker's next macroblock */
#ifdef H264K_TAIL_PROFILE
    if (tp) { const uint32_t d4 = DTICK(); tp->wait += d1 - d0; tp->v += d2 - d1; tp->h += d3 - d2; tp->st += d4 - d3; }
#endif
    (void)d0; (void)d1; (void)d2; (void)d3; (void)tp;
}

/* ================================================================== chroma */
struct DbkChromaLoads {
    uint4 c;                  /* rows 2 c4, 2 c4 + 1 of plane l >> 2 (16 contiguous bytes: x, y = row 2 c4; z, w = row 2 c4 + 1) */
    uint4 lc;                 /* the same rows of the tile to the left (.y, .w = its last four columns)                  */
    uint32_t t;               /* dword l & 3 of the last two rows of the tile above, same plane (row 6 + (l & 3) / 2)     */
    uint4 r0; uint32_t w_cl; uint2 w_ci; uint2 r2;     /* strengths | chroma left | chroma top, inner | bytes 40..47     */
};

__device__ __forceinline__ void dbk_chroma_load(const DbkCtx &fd, int mb, int l, DbkChromaLoads &p, bool cross, bool want_top)
{
    if (mb < 0) return;
    const uint8_t *cur = fd.cur;
    const H264K_GLOBAL uint8_t *curg = (const H264K_GLOBAL uint8_t *)cur;
    const H264K_GLOBAL uint8_t *recs = (const H264K_GLOBAL uint8_t *)fd.recs;
    const uint32_t umb = (uint32_t)mb, wmb = fd.wmb;
    const uint32_t t = umb * TILE, tl = (umb ? umb - 1u : 0u) * TILE, tu = (umb >= wmb ? umb - wmb : umb) * TILE;
    const uint32_t ro = umb * DBK_REC_BYTES;
    p.r0 = ld16g(recs + ro);
    p.w_cl = *(const H264K_GLOBAL uint32_t *)(recs + ro + 28u);
    p.w_ci = ld8g(recs + ro + 32u);
    p.r2 = ld8g(recs + ro + 40u);
    p.c = ld16g(curg + t + T_CB + 16u * l);
    p.lc = ld16g(curg + tl + T_CB + 16u * l);
    const uint32_t uc = tu + T_CB + 64u * (l >> 2) + 48u + 4u * (l & 3);
    p.t = 0u;
    if (!cross) p.t = ldg_u32(cur, uc);
    else if (want_top) p.t = ld_agent_u32(cur + uc);
}

/* Chroma of one macroblock: both planes, edges at columns / rows 0 and 4 (= luma edges 0 and 2).  SLOTS = 2: both; SLOTS = 1:
 * the macroblock edges only (partial stores as in dbk_luma_step: column 0 and row 0 of the own planes). */
template <bool BANDED, int SLOTS>
__device__ __forceinline__ void dbk_chroma_step(const DbkCtx &fd, int mb, int l, const DbkChromaLoads &p, uint32_t wb, bool wt_, DbkProf *tp)
{
    constexpr int NPX = SLOTS == 2 ? 10 : 4;
    const bool wt = BANDED && wt_;
    const bool act = mb >= 0;
    const uint32_t d0 = DTICK();
    const int c4 = l & 3, pl = l >> 2;
    const uint32_t flags = p.r2.y >> 16;
    const bool f_left = act && (flags & FJ_DBK_LEFT) && (p.r0.x & 0xFFFFu), f_top = act && (flags & FJ_DBK_TOP) && (p.r0.z & 0xFFFFu);
    const bool any_v = __ballot(act && (SLOTS == 2 ? ((p.r0.x | p.r0.y) & 0xFFFFu) : (p.r0.x & 0xFFFFu))) != 0ull;
    const bool any_h = __ballot(act && (SLOTS == 2 ? ((p.r0.z | p.r0.w) & 0xFFFFu) : (p.r0.z & 0xFFFFu))) != 0ull;
    const uint32_t w_cl = p.w_cl, w_ctp = p.w_ci.x, w_ci = p.w_ci.y, t3a = p.r2.x, t3b = p.r2.y;    /* byte 43: tc0(3) chroma left; bytes 44, 45: top, inner */
    const uint32_t sb = wb + DBK_CW_STRIP + 16u * pl;              /* the strip area of this plane: two rows of 8 bytes */
#ifdef H264K_TAIL_PROFILE
    if (tp) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    const uint32_t d1 = DTICK();

    /* ---- vertical edges: px[0], px[1] = columns 6, 7 of the left neighbour; px[2 + c] = column c ---- */
    uint32_t px[NPX];
    px[0] = perm(p.lc.w, p.lc.y, 0x0C060C02u); px[1] = perm(p.lc.w, p.lc.y, 0x0C070C03u);
    px[2] = perm(p.c.z, p.c.x, 0x0C040C00u); px[3] = perm(p.c.z, p.c.x, 0x0C050C01u);
    if constexpr (SLOTS == 2) {
        px[4] = perm(p.c.z, p.c.x, 0x0C060C02u); px[5] = perm(p.c.z, p.c.x, 0x0C070C03u);
        px[6] = perm(p.c.w, p.c.y, 0x0C040C00u); px[7] = perm(p.c.w, p.c.y, 0x0C050C01u);
        px[8] = perm(p.c.w, p.c.y, 0x0C060C02u); px[9] = perm(p.c.w, p.c.y, 0x0C070C03u);
    }
    if (any_v) {
        const uint32_t w0s = p.r0.x >> (4 * c4), w1s = p.r0.y >> (4 * c4);     /* chroma rows 2 c4, 2 c4 + 1 = luma rows 4 c4 .. 4 c4 + 3: segment c4 */
        const int bs0 = act ? (int)(w0s & 15u) : 0, bs1 = act ? (int)(w1s & 15u) : 0;
        if (__ballot(bs0 != 0))
            filter_chroma_pk(px + 0, bs0, pk_splat_byte(w_cl, 0), pk_splat_byte(w_cl, 1), tc0_of(perm(t3a, w_cl, 0x0703020Cu), bs0));
        if constexpr (SLOTS == 2) if (__ballot(bs1 != 0))
            filter_chroma_pk(px + 4, bs1, pk_splat_byte(w_ci, 0), pk_splat_byte(w_ci, 1), tc0_of(perm(t3b, w_ci, 0x0503020Cu), bs1));
    }
    const uint32_t d2 = DTICK();

    /* ---- to columns: hx[2 + r] = row r, columns 2 c4, 2 c4 + 1; hx[0], hx[1] = rows 6, 7 of the upper neighbour ---- */
    uint32_t hx[NPX];
    const uint32_t ra_ = wb + 64u * pl + 4u * c4;
    lds_st32(sb + 4u * c4, p.t);
    if constexpr (SLOTS == 2) {
        lds_st128(wb + 16u * l, blk_of(px[2], px[3]), blk_of(px[4], px[5]), blk_of(px[6], px[7]), blk_of(px[8], px[9]));
        wave_sync();
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t e = lds_ld32(ra_ + 16u * r);
            hx[2 + 2 * r] = pair_lo(e); hx[3 + 2 * r] = pair_hi(e);
        }
    } else {
        /* rows as rows (8 bytes each, plane after plane): the upper edge reads rows 0, 1 */
        const uint32_t a = blk_of(px[2], px[3]);                   /* (row 2 c4 col 0, row 2 c4 + 1 col 0, row 2 c4 col 1, row 2 c4 + 1 col 1) */
        const uint32_t own0 = perm(p.c.x, a, 0x07060200u), own1 = perm(p.c.z, a, 0x07060301u);       /* columns 0, 1 filtered, 2, 3 as loaded */
        lds_st128(wb + 16u * l, own0, p.c.y, own1, p.c.w);
        wave_sync();
#pragma unroll
        for (int r = 0; r < 2; r++) hx[2 + r] = perm(0u, lds_ld16(wb + 64u * pl + 8u * r + 2u * c4), 0x0C010C00u);
    }
    hx[0] = perm(0u, lds_ld16(sb + 2u * c4), 0x0C010C00u); hx[1] = perm(0u, lds_ld16(sb + 8u + 2u * c4), 0x0C010C00u);

    /* ---- horizontal edges ---- */
    if (any_h) {
        const uint32_t w0s = p.r0.z >> (4 * c4), w1s = p.r0.w >> (4 * c4);
        const int bs0 = act ? (int)(w0s & 15u) : 0, bs1 = act ? (int)(w1s & 15u) : 0;
        if (__ballot(bs0 != 0))
            filter_chroma_pk(hx + 0, bs0, pk_splat_byte(w_ctp, 0), pk_splat_byte(w_ctp, 1), tc0_of(perm(t3b, w_ctp, 0x0403020Cu), bs0));
        if constexpr (SLOTS == 2) if (__ballot(bs1 != 0))
            filter_chroma_pk(hx + 4, bs1, pk_splat_byte(w_ci, 0), pk_splat_byte(w_ci, 1), tc0_of(perm(t3b, w_ci, 0x0503020Cu), bs1));
    }
    const uint32_t d3 = DTICK();

    /* ---- store ---- */
    uint8_t *cur = fd.cur;
    const uint32_t t = (uint32_t)mb * TILE, tp_ = t + T_CB + 64u * pl;
    if constexpr (SLOTS == 2) {
        uint32_t F[4];
#pragma unroll
        for (int r = 0; r < 4; r++) F[r] = blk_t_of(hx[2 + 2 * r], hx[3 + 2 * r]);
        lds_st128(wb + 16u * l, F[0], F[1], F[2], F[3]);
        wave_sync();
        uint32_t G[4];
#pragma unroll
        for (int j = 0; j < 4; j++) G[j] = lds_ld32(ra_ + 16u * j);
        if (act) {
            uint4 v;
            v.x = perm(G[1], G[0], 0x06040200u); v.y = perm(G[3], G[2], 0x06040200u);
            v.z = perm(G[1], G[0], 0x07050301u); v.w = perm(G[3], G[2], 0x07050301u);
            stg_b128(cur, t + T_CB + 16u * l, v, wt);
        }
    } else {
        /* column 0 of rows 2 c4, 2 c4 + 1 (q0 of the left edge), lanes regrouped (quad_perm [2,3,0,1]) so that one instruction covers
         * rows 0..3 of both planes and the other rows 4..7; row 0 leaves with the horizontal pass when the upper edge is filtered */
        const uint32_t q0 = px[2], q0x = (uint32_t)quad_xor2((int)q0);
        const uint32_t va = c4 < 2 ? (q0 & 255u) : (q0x >> 16), vb = c4 < 2 ? (q0x & 255u) : (q0 >> 16);
        const uint32_t row_a = c4 < 2 ? 2u * c4 : 2u * (c4 - 2) + 1u;
        if (f_left) {
            if (!(f_top && row_a == 0u)) stg_u8(cur, tp_ + 8u * row_a, va, wt);
            stg_u8(cur, tp_ + 8u * (row_a + 4u), vb, wt);
        }
        if (f_top) stg_u16(cur, tp_ + 2u * c4, perm(0u, hx[2], 0x0C0C0200u), wt);
    }
    if (f_left) {
        /* p0 of the left edge: column 7 of the left neighbour, regrouped the same way */
        const uint32_t p0 = px[1], p0x = (uint32_t)quad_xor2((int)p0);
        const uint32_t va = c4 < 2 ? (p0 & 255u) : (p0x >> 16), vb = c4 < 2 ? (p0x & 255u) : (p0 >> 16);
        const uint32_t row_a = c4 < 2 ? 2u * c4 : 2u * (c4 - 2) + 1u;
        stg_u8(cur, tp_ - TILE + 8u * row_a + 7u, va, wt);
        stg_u8(cur, tp_ - TILE + 8u * (row_a + 4u) + 7u, vb, wt);
    }
    if (f_top) stg_u16(cur, tp_ - (uint32_t)fd.wmb * TILE + 56u + 2u * c4, perm(0u, hx[1], 0x0C0C0200u), wt);    /* p0 of the upper edge: row 7 of the upper neighbour */
    wave_sync();
#ifdef H264K_TAIL_PROFILE
    if (tp) { const uint32_t d4 = DTICK(); tp->wait += d1 - d0; tp->v += d2 - d1; tp->h += d3 - d2; tp->st += d4 - d3; }
#endif
    (void)d0; (void)d1; (void)d2; (void)d3; (void)tp;
}

/* ================================================================== the per-picture scheduler */
/* In-loop deblocking of one picture.  The filter of macroblock (x,y) touches its own samples, the last
 * 4 columns of (x-1,y) and the last 4 rows of (x,y-1); in the standard's raster order that makes it
 * depend on exactly three earlier steps: (x-1,y), (x,y-1) and (x+1,y-1) — and only if those macroblocks
 * are filtered at all (most P-picture macroblocks have all-zero strengths and are never touched).
 *
 * TWO GRAPHS.  Luma and chroma share no sample: each has its own dependency counters, ready lists and cursors (DbkGraph)
 * over the same flags and the same rule.  The first waves of the workgroup walk the luma graph, the last `chroma_waves`
 * the chroma graph (at a lower issue priority: a luma step is the longer link of the longer chain); a wave whose own graph
 * has nothing left to claim goes on with the other one.
 *
 * ROW BANDS.  A picture is split into up to max_bands bands of consecutive macroblock rows, one workgroup each
 * (grid = max_bands x pictures; a picture that wants fewer bands leaves the surplus workgroups idle).  Inside a band
 * the dependencies are tracked in LDS; the only dependencies that cross a band boundary are those of a
 * band's FIRST row on the LAST row of the band above — (x,y-1) and (x+1,y-1) — and they are handed over through HBM:
 *   producer: a macroblock of a band's last row writes everything write-through (wt), waits for its stores
 *             (s_waitcnt vmcnt(0)) and then sets its "done" byte of its graph (scratch_done, agent scope);
 *   consumer: a wavefront of the band below that finds nothing ready polls the done bytes of the producers its first row
 *             still waits for (one poller per band and graph at a time, relaxed agent-scope loads, s_sleep between passes),
 *             marks each seen producer once (LDS bit) and releases its dependants into the band's ready queue; the first-row
 *             macroblock then reads the last rows of the tile above past the L1 (dbk_*_load, cross).
 * Every pair of macroblocks that touches a common sample is ordered as in the reference's raster scan
 * (src/h264bsd_deblocking.c:604-638) whether both lie in one band or not.  The same spin limit that guards the LDS scheduler
 * ends a wait that never finishes in DEVERR_DBK_SCHED.
 *
 * Dataflow scheduling inside a band, all state in LDS (indices are band-local: row r0-1 .. r1-1):
 *   anyf[]    DBKF_* flags of the band's rows and of the row above (shared by both graphs)
 *   dep[mb]   number of filtered macroblocks among those three that are not finished yet
 *   queue[]   ready lists: every filtered macroblock is pushed exactly once, when its dep reaches 0 — macroblocks with an
 *             active inner edge from the front, those with macroblock edges only from the back (a step takes ONE kind, so
 *             that edge-only macroblocks run their short instruction stream)
 * A free wavefront pulls up to eight READY macroblocks of one list at once (compare-and-swap on that list's head), one per
 * worker, fetches their samples, records and neighbour strips in one memory round trip, filters, stores, then releases the
 * three dependants (x+1,y), (x,y+1), (x-1,y+1).  No level barriers.  Same-CU visibility of the stores needs no wait
 * (release_stores, common.hip.h).
 * Dynamic LDS: waves x DBK_WAVE_LDS exchange buffers | anyf | dep x 2 | queue x 2 (u16) | counters x 2 | seen bits x 2 (dbk_lds_bytes). */
__host__ __device__ inline size_t dbk_lds_bytes(uint32_t waves, uint32_t wmb, uint32_t band_rows)
{
    const size_t n_loc16 = (((size_t)band_rows + 1) * wmb + 15) & ~(size_t)15, nq8 = ((size_t)band_rows * wmb + 7) & ~(size_t)7;
    return (size_t)waves * DBK_WAVE_LDS + 3 * n_loc16 + 2 * (2 * nq8 + 32 + 4 * ((((size_t)wmb + 31) / 32 + 3) & ~(size_t)3));
}
struct DbkGraph {
    uint8_t *dep; uint16_t *queue; uint32_t *ctr;    /* ctr: [0] head, [1] tail of the inner list, [2] total, [3] producers awaited, [4] producers seen, [5] poll lock, [6] head, [7] tail of the edge-only list */
    uint32_t *seen;                                   /* one bit per column: the done byte of (x, r0-1) has been seen */
    uint8_t *done_g;                                  /* this graph's "done" bytes in the stream's scratch area */
};
#ifndef DBK_OCC
#define DBK_OCC 3
#endif
#ifndef CONV_PRIO
#define CONV_PRIO 0            /* issue priority of the wavefronts that convert beside the chains */
#endif
#ifndef DBK_CHROMA_PRIO
#define DBK_CHROMA_PRIO 1
#endif
template <bool BANDED>
__global__ __launch_bounds__(64 * DBK_WAVES, BANDED ? 3 : DBK_OCC) void k_frame_dbk(const FrameDesc *__restrict__ frames, unsigned long long *prof,
                                                              uint32_t *tickets, uint32_t max_bands, uint32_t rows_cap, uint32_t light_cap, uint32_t chroma_waves,
                                                              uint32_t conv_waves)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    __shared__ uint32_t s_misc[4];
    const uint32_t ticket = BANDED ? take_ticket(tickets, &s_misc[0]) : blockIdx.x;
    const uint32_t pic = BANDED ? ticket / max_bands : ticket, band = BANDED ? ticket - pic * max_bands : 0u;
    const FrameDesc &fd = FD_REF(frames, pic);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, grp = lane / DBK_LANES, l = lane % DBK_LANES;
    const int n_waves = (int)(blockDim.x >> 6);
    const DbkCtx cx = dbk_ctx(fd);
    const int wmb = (int)cx.wmb, hmb = fd.hmb;
    int R = hmb, nb = 1;
    if (BANDED) band_split(hmb, fd.dbk_bands, fd.heavy, max_bands, light_cap, rows_cap, R, nb);
    /* hosted colour conversion (FrameDesc.conv_src, kernels/convert.hip.h): a workgroup with nothing to filter converts */
    const bool conv_on = fd.conv_src != nullptr;
    if (!fd.any_deblock || (int)band >= nb) {
        if (conv_on && band == 0) {                              /* (a picture without filtering: ONE workgroup, its first, converts all of it) */
            if (tid == 0) s_misc[2] = 0u;
            __syncthreads();
            conv_drain(fd, lds_addr(&s_misc[2]), ((uint32_t)wmb + 1u) / 2u * (uint32_t)hmb, (uint32_t)lane);
        }
        if (BANDED) return_ticket(tickets);
        return;
    }
    const int r0 = (int)band * R, r1 = min(hmb, r0 + R);
    /* the workgroup's share of the other picture: the tile pairs of its own rows */
    const uint32_t conv_ppr = ((uint32_t)wmb + 1u) / 2u, conv_end = (uint32_t)r1 * conv_ppr;
    if (conv_on && tid == 0) s_misc[2] = (uint32_t)r0 * conv_ppr;             /* (the barriers below publish it) */
    const int base = (r0 - 1) * wmb;                        /* band-local index of macroblock mb: mb - base (row r0-1 first) */
    const int n_loc = (R + 1) * wmb, n_loc16 = (n_loc + 15) & ~15, nq8 = (R * wmb + 7) & ~7;
    const int seen_words = (((wmb + 31) >> 5) + 3) & ~3;
    const bool has_up = BANDED && band > 0, has_down = BANDED && r1 < hmb;
    uint8_t *anyf = lds + (size_t)n_waves * DBK_WAVE_LDS;
    uint8_t *flags_g = scratch_flags(fd);
    const int n4 = (int)((fd.n_mbs + 3u) & ~3u);
    auto graph = [&](int which) {                            /* 0: luma, 1: chroma (arithmetic, not a table: no scratch memory) */
        DbkGraph g;
        uint8_t *p = anyf + n_loc16;
        g.dep = p + which * n_loc16; p += 2 * n_loc16;
        g.queue = reinterpret_cast<uint16_t *>(p) + which * nq8; p += 4 * (size_t)nq8;
        g.ctr = reinterpret_cast<uint32_t *>(p) + which * 8; p += 64;
        g.seen = reinterpret_cast<uint32_t *>(p) + which * seen_words;
        g.done_g = scratch_done(fd, SCRATCH_DONE_DBK_LUMA) + which * n4;      /* (SCRATCH_DONE_DBK_CHROMA follows it) */
        return g;
    };
    const DbkGraph G0 = graph(0), G1 = graph(1);
    const uint32_t wb = lds_addr(lds + (size_t)wave * DBK_WAVE_LDS);
    /* debug accounting (h264bsdmiDebugTailProfile): band 0 of picture 0 only, per wavefront: [0] cycles with nothing ready,
     * [1] cycles filtering, [2] cycles waiting for own stores, [3] macroblocks filtered, [4] total, [5] steps */
#ifdef H264K_TAIL_PROFILE
    unsigned long long *tp = (prof && ticket == 0) ? prof + wave * 16 : nullptr;
#else
    unsigned long long *const tp = nullptr;
    (void)prof;
#endif
    uint32_t t_idle = 0, t_work = 0, t_store = 0, n_done = 0, n_steps = 0;
    DbkProf acc = { 0, 0, 0, 0, 0, 0 };       /* inside the steps: load wait, vertical, horizontal, store; claim won -> queue slot read, -> loads issued */
    const uint32_t t_begin = tp ? (uint32_t)__builtin_readcyclecounter() : 0u;
    uint32_t t_mark = t_begin;

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
        for (int i = tid; i < nq8; i += blockDim.x) reinterpret_cast<uint32_t *>(G0.queue)[i] = 0xFFFFFFFFu;     /* both queues */
        if (tid < 16) G0.ctr[tid] = 0;
        for (int i = tid; i < 2 * seen_words; i += blockDim.x) G0.seen[i] = 0;
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
     * (deblocking.c:604-638).  For the band's first row the macroblocks above belong to the band above: they count like any
     * other and are released by the poller (below) instead of by the wavefront that filtered them.
     * (The chroma graph uses the luma flags: chroma has fewer edges, so the rule orders more pairs than it must, never fewer.) */
    const int nq_last = nq8 - 1;
    auto push = [&](const DbkGraph &g, int mb, uint32_t flags) {
        if (flags & DBKF_INNER) g.queue[atomicAdd(&g.ctr[1], 1u)] = (uint16_t)mb;
        else g.queue[nq_last - (int)atomicAdd(&g.ctr[7], 1u)] = (uint16_t)mb;
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
        G0.dep[li] = (uint8_t)d; G1.dep[li] = (uint8_t)d;
        atomicAdd(&G0.ctr[2], 1u);
        if (d == 0) { push(G0, mb, f); push(G1, mb, f); }
    }
    if (has_up)
        for (int x = tid; x < wmb; x += blockDim.x)
            if (anyf[x] & DBKF_ANY) atomicAdd(&G0.ctr[3], 1u);
    __syncthreads();
    const uint32_t total = G0.ctr[2], n_await = G0.ctr[3];

    /* one dependency of band-local macroblock li is gone: publish it when it was the last */
    auto release = [&](const DbkGraph &g, int li) {
        /* byte-wide counters: decrement through a 32-bit LDS atomic on the containing word */
        uint32_t *w = reinterpret_cast<uint32_t *>(g.dep + (li & ~3));
        const uint32_t sh = 8u * (li & 3);
        const uint32_t old = atomicSub(w, 1u << sh);
        if (((old >> sh) & 255u) == 1u) push(g, li + base, anyf[li]);
    };

    /* the first waves walk the luma graph, the last chroma_waves the chroma graph; then the other one */
    const int n_chroma = n_waves >= 2 ? min(max((int)chroma_waves, 1), n_waves - 1) : 0;
    int role = wave >= n_waves - n_chroma ? 1 : 0;
    const int lo_mb = r0 * wmb, hi_mb = r1 * wmb;                /* the band's own macroblocks */
    /* the last conv_waves of the luma wavefronts convert first (at the lowest issue priority: they take the slots the chains leave)
     * and join the luma graph when the other picture is done */
    if (conv_on && wave >= n_waves - n_chroma - (int)conv_waves && wave < n_waves - n_chroma) {
        __builtin_amdgcn_s_setprio(CONV_PRIO);
        conv_drain(fd, lds_addr(&s_misc[2]), conv_end, (uint32_t)lane);
    }
    for (int pass = 0; pass < 2; pass++, role ^= 1) {
        const DbkGraph g = graph(role);
        /* these wavefronts walk dependency chains: whatever shares their SIMDs takes the issue slots they leave, not the ones they need */
        if (role == 0) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(DBK_CHROMA_PRIO);
        volatile H264K_LDS uint16_t *vq = (volatile H264K_LDS uint16_t *)g.queue;      /* (a generic volatile pointer would read LDS through flat_load) */
        volatile H264K_LDS uint32_t *vctr = (volatile H264K_LDS uint32_t *)g.ctr;
        uint32_t spins = 0;                  /* safety net: a scheduling bug must end in a reported error (DEVERR_*), never in a hung GPU */
        for (;;) {
            uint32_t cbase = 0, k = 0, cls = 0;
            if (lane == 0) {
                const uint32_t h0 = vctr[0], t0 = vctr[1], h1 = vctr[6], t1 = vctr[7];
                const uint32_t a0 = t0 - h0, a1 = t1 - h1;
                if (a0 | a1) {
                    cls = a1 >= a0 ? 1u : 0u;                            /* the longer list; the cheaper one when they tie */
                    const uint32_t h = cls ? h1 : h0, a = cls ? a1 : a0;
                    k = a < 8u ? a : 8u;
                    if (atomicCAS(&g.ctr[cls ? 6 : 0], h, h + k) != h) k = 0;       /* lost the race: look again */
                    cbase = h;
                } else if (h0 + h1 >= total) {
                    k = 0xFFFFFFFFu;                                    /* everything has been claimed */
                }
            }
            /* (through the LDS crossbar on purpose: handing lane 0's values over with v_readfirstlane — one LDS round trip less —
             * costs the kernel 12 %, 31.7 -> 35.6 ms per step, with or without a pause of the same length in its place) */
            cbase = __shfl(cbase, 0); k = __shfl(k, 0); cls = (uint32_t)__builtin_amdgcn_readfirstlane((int)__shfl(cls, 0));
            if (k == 0xFFFFFFFFu) break;
            if (++spins > (1u << 24)) { if (lane == 0) report_device_error(fd, DEVERR_DBK_SCHED); break; }
            if (k == 0) {
                /* nothing ready.  If the first row still waits for macroblocks of the band above, look whether they are done:
                 * one wavefront of the band and graph at a time, lane -> column */
                bool polled = false;
                if (has_up && vctr[4] < n_await) {
                    uint32_t got = 0;
                    if (lane == 0) got = atomicCAS(&g.ctr[5], 0u, 1u) == 0u;
                    got = __shfl(got, 0);
                    if (got) {
                        polled = true;
                        for (int x = lane; x < wmb; x += 64) {
                            const uint32_t fu = anyf[x];
                            const uint32_t bit = 1u << (x & 31);
                            if (!(fu & DBKF_ANY) || (g.seen[x >> 5] & bit)) continue;
                            if (!ld_agent_u8(g.done_g + base + x)) continue;
                            if (atomicOr(&g.seen[x >> 5], bit) & bit) continue;
                            atomicAdd(&g.ctr[4], 1u);
                            /* the mirror image of the dependency rule: (x, r0) waits for it through its upper edge, (x-1, r0)
                             * if this producer's left edge was filtered */
                            const uint32_t fc = anyf[wmb + x];
                            if ((fc & DBKF_ANY) && (fc & DBKF_TOP) && (fu & (DBKF_INNER | DBKF_LEFT))) release(g, wmb + x);
                            if (x > 0 && (fu & DBKF_LEFT)) {
                                const uint32_t fl = anyf[wmb + x - 1];
                                if ((fl & DBKF_ANY) && (fl & DBKF_TOP)) release(g, wmb + x - 1);
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        if (lane == 0) atomicExch(&g.ctr[5], 0u);
                    }
                }
                if (polled) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(1);
                continue;
            }
            if (tp) { const uint32_t t = (uint32_t)__builtin_readcyclecounter(); t_idle += t - t_mark; t_mark = t; }
            int run = -1;
            if ((uint32_t)grp < k) {
                int v;
                const int slot = cls ? nq_last - (int)(cbase + grp) : (int)(cbase + grp);
                do { v = vq[slot]; } while (v == 0xFFFF);                /* the publisher bumps the cursor, then writes the slot */
                run = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef H264K_TAIL_PROFILE
            if (tp) acc.q += (uint32_t)__builtin_readcyclecounter() - t_mark;      /* claim won -> queue slot read */
#endif
            const bool cross = has_up && run >= 0 && run < lo_mb + wmb;  /* first row: the tile above belongs to the band above */
            const bool wt = has_down && run >= hi_mb - wmb;              /* last row: the band below reads what this macroblock writes */
            const uint32_t fm = run >= 0 ? anyf[run - base] : 0u;
            bool want_top = true;
            if (BANDED && __ballot(cross) != 0ull) want_top = !cross || (fm & DBKF_TOP);
            /* the dependants' flags (release, below), read while the loads are in flight: l = 0: (x+1, y), l = 1: (x, y+1), l = 2: (x-1, y+1) */
            const int dmb = l == 0 ? run + 1 : l == 1 ? run + wmb : run + wmb - 1;
            uint32_t fdep = 0u;
            if (run >= 0 && l < 3 && dmb < hi_mb) fdep = anyf[dmb - base];
            DbkProf *stp = tp ? &acc : nullptr;
            if (role == 0) {
                DbkLumaLoads cp;
                dbk_luma_load(cx, run, l, cp, BANDED && cross, want_top);
#ifdef H264K_TAIL_PROFILE
                if (tp) acc.ld += (uint32_t)__builtin_readcyclecounter() - t_mark;  /* ... -> loads issued */
#endif
                if (cls) dbk_luma_step<BANDED, 1>(cx, run, l, cp, wb + (uint32_t)grp * DBK_LW, wt, stp);
                else dbk_luma_step<BANDED, 4>(cx, run, l, cp, wb + (uint32_t)grp * DBK_LW, wt, stp);
            } else {
                DbkChromaLoads cp;
                dbk_chroma_load(cx, run, l, cp, BANDED && cross, want_top);
#ifdef H264K_TAIL_PROFILE
                if (tp) acc.ld += (uint32_t)__builtin_readcyclecounter() - t_mark;
#endif
                if (cls) dbk_chroma_step<BANDED, 1>(cx, run, l, cp, wb + (uint32_t)grp * DBK_CW, wt, stp);
                else dbk_chroma_step<BANDED, 2>(cx, run, l, cp, wb + (uint32_t)grp * DBK_CW, wt, stp);
            }
            if (tp) { const uint32_t t = (uint32_t)__builtin_readcyclecounter(); t_work += t - t_mark; t_mark = t; n_done += __popcll(__ballot(run >= 0 && l == 0)); n_steps++; }
            /* release: stores done -> dependants */
            release_stores(BANDED && wt && run >= 0);
            if (BANDED && wt && l == 3) st_agent_u8(g.done_g + run, 1u);   /* hand-over to the band below */
            if (run >= 0 && l < 3) {
                /* dependants: l = 0: (x+1, y), l = 1: (x, y+1), l = 2: (x-1, y+1) — the mirror image of the dependency rule above.  The
                 * "neighbours" of the first / last column that lie in another row never qualify: a macroblock of column 0 has no LEFT */
                bool waits = false;
                if (dmb < hi_mb) {
                    const uint32_t fd_ = fdep;
                    waits = (fd_ & DBKF_ANY) && (l == 0 ? ((fd_ & DBKF_LEFT) != 0u && (fm & (DBKF_INNER | DBKF_TOP)) != 0u)
                                                      : l == 1 ? ((fd_ & DBKF_TOP) != 0u && (fm & (DBKF_INNER | DBKF_LEFT)) != 0u)
                                                               : ((fd_ & DBKF_TOP) != 0u && (fm & DBKF_LEFT) != 0u));
                }
                if (waits) release(g, dmb - base);
            }
            if (tp) { const uint32_t t = (uint32_t)__builtin_readcyclecounter(); t_store += t - t_mark; t_mark = t; }
        }
    }
    if (conv_on) {                                               /* both graphs exhausted: what is left of the other picture */
        __builtin_amdgcn_s_setprio(0);
        conv_drain(fd, lds_addr(&s_misc[2]), conv_end, (uint32_t)lane);
    }
    if (tp && lane == 0) {
        tp[0] += t_idle; tp[1] += t_work; tp[2] += t_store; tp[3] += n_done; tp[4] += (uint32_t)__builtin_readcyclecounter() - t_begin; tp[5] += n_steps;
        tp[8] += acc.wait; tp[9] += acc.v; tp[10] += acc.h; tp[11] += acc.st; tp[12] += acc.q; tp[13] += acc.ld;
    }
    /* the last band of the picture to leave zeroes the flags (k_dbk only visits non-trivial macroblocks) and the done bytes
     * of both graphs for the next picture of this stream */
    bool last = true;
    if (BANDED && nb > 1) {
        __syncthreads();
        if (tid == 0) s_misc[1] = atomicAdd(scratch_exits(fd, 0), 1u);
        __syncthreads();
        last = s_misc[1] == (uint32_t)nb - 1u;
    }
    if (last) {
        uint32_t *z = reinterpret_cast<uint32_t *>(flags_g);
        const int words = (BANDED && nb > 1 ? 3 : 1) * (int)((fd.n_mbs + 3u) >> 2);   /* flags and this kernel's done bytes (luma, chroma) are adjacent */
        for (int i = tid; i < words; i += blockDim.x) z[i] = 0;
        if (BANDED && nb > 1 && tid == 0) atomicExch(scratch_exits(fd, 0), 0u);
    }
    if (BANDED) return_ticket(tickets);
}

} // namespace h264k
