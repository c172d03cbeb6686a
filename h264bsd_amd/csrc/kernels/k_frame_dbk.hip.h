/* kernels/k_frame_dbk.hip.h — k_frame_dbk: the in-loop filter, one workgroup per picture (or row band).  Part of kernels.hip.h (which see); not a stand-alone header. */
#pragma once
namespace h264k {
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

} // namespace h264k
