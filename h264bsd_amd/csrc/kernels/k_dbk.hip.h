/* kernels/k_dbk.hip.h — k_dbk: boundary strengths and threshold values -> 48-byte deblocking records.  Part of kernels.hip.h (which see); not a stand-alone header. */
#pragma once
namespace h264k {
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

} // namespace h264k
