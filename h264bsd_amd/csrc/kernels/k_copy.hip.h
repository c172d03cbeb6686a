/* kernels/k_copy.hip.h — k_copy: whole-sample copy macroblocks.  Part of kernels.hip.h (which see); not a stand-alone header. */
#pragma once
namespace h264k {
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


} // namespace h264k
