// Layout probe (round-2 planning): how fast can 16x16+2x8x8 macroblocks be copied between two frames when a macroblock's
// rows are (a) 16-byte pieces of a planar 1920-wide frame (today's DPB), (b) 16-byte pieces of a row-interleaved strip of
// T bytes, (c) 384 contiguous bytes (macroblock tiles)?  Same bytes, same number of macroblocks, same run structure as
// k_copy (runs of R horizontally adjacent macroblocks, one run per wavefront, 16 B per lane per access).
// build: hipcc -O3 --offload-arch=gfx950 layout_probe.hip -o layout_probe ; run: ./layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int WMB = 120, HMB = 68, W = WMB * 16, H = HMB * 16;
constexpr size_t FRAME = (size_t)WMB * HMB * 384;

// byte offset of luma sample (x, y) / chroma sample (plane, x, y) in a frame laid out with strip width T (T == W: planar;
// T == 16: macroblock tiles with the chroma behind the luma of the same macroblock)
__device__ __host__ inline size_t luma_off(int T, int x, int y)
{
    if (T == 16) return ((size_t)(y >> 4) * WMB + (x >> 4)) * 384 + (y & 15) * 16 + (x & 15);
    const int strips = W / T;
    return (((size_t)(y >> 4) * strips + x / T) * 16 + (y & 15)) * T + x % T;
}
__device__ __host__ inline size_t chroma_off(int T, int plane, int x, int y)
{
    if (T == 16) return ((size_t)(y >> 3) * WMB + (x >> 3)) * 384 + 256 + plane * 64 + (y & 7) * 8 + (x & 7);
    const int Tc = T / 2, strips = (W / 2) / Tc;
    return (size_t)W * H + (size_t)plane * (W / 2) * (H / 2) + (((size_t)(y >> 3) * strips + x / Tc) * 8 + (y & 7)) * Tc + x % Tc;
}

struct Run { uint16_t mb; uint8_t count, pad; };

template <int R>
__global__ __launch_bounds__(256) void k_probe(const uint8_t *src, uint8_t *dst, size_t stream_stride, const Run *runs, int n_runs, int T)
{
    const int ri = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ri >= n_runs) return;
    const Run r = runs[ri];
    const uint8_t *s = src + blockIdx.y * stream_stride;
    uint8_t *d = dst + blockIdx.y * stream_stride;
    const int lane = threadIdx.x & 63;
    const int mbx = r.mb % WMB, mby = r.mb / WMB;
    // luma: R MBs x 16 rows x 16 B = 16R pieces; chroma: 2 planes x 8 rows x R MBs x 8 B = 16R pieces of 8 B -> 8R of 16 B
    uint4 v[(16 * R + 63) / 64];
    uint2 c[(16 * R + 63) / 64];
#pragma unroll
    for (int j = 0; j < (16 * R + 63) / 64; j++) {
        const int i = lane + 64 * j, row = i / R, seg = i % R;
        if (i < 16 * R && seg < r.count) v[j] = *reinterpret_cast<const uint4 *>(s + luma_off(T, (mbx + seg) * 16, mby * 16 + row));
        const int plane = i / (8 * R), crow = (i / R) % 8;
        if (i < 16 * R && seg < r.count) c[j] = *reinterpret_cast<const uint2 *>(s + chroma_off(T, plane, (mbx + seg) * 8, mby * 8 + crow));
    }
#pragma unroll
    for (int j = 0; j < (16 * R + 63) / 64; j++) {
        const int i = lane + 64 * j, row = i / R, seg = i % R;
        if (i < 16 * R && seg < r.count) *reinterpret_cast<uint4 *>(d + luma_off(T, (mbx + seg) * 16, mby * 16 + row)) = v[j];
        const int plane = i / (8 * R), crow = (i / R) % 8;
        if (i < 16 * R && seg < r.count) *reinterpret_cast<uint2 *>(d + chroma_off(T, plane, (mbx + seg) * 8, mby * 8 + crow)) = c[j];
    }
}

int main()
{
    const int S = 256;                                  // streams, as in the bench
    std::mt19937 rng(1);
    // runs like the bundled stream's copy list: 61 % of the macroblocks, average run 5 of at most 8
    std::vector<Run> runs;
    size_t n_mbs = 0;
    for (int y = 0; y < HMB; y++)
        for (int x = 0; x < WMB;) {
            if (rng() % 100 < 39) { x++; continue; }
            int len = 1 + (int)(rng() % 8);
            len = std::min(len, WMB - x);
            runs.push_back(Run{ (uint16_t)(y * WMB + x), (uint8_t)len, 0 });
            n_mbs += (size_t)len; x += len;
        }
    uint8_t *a, *b; Run *d_runs;
    CK(hipMalloc((void **)&a, FRAME * S + 4096)); CK(hipMalloc((void **)&b, FRAME * S + 4096));
    CK(hipMemset(a, 1, FRAME * S + 4096)); CK(hipMemset(b, 2, FRAME * S + 4096));
    CK(hipMalloc((void **)&d_runs, runs.size() * sizeof(Run)));
    CK(hipMemcpy(d_runs, runs.data(), runs.size() * sizeof(Run), hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%zu runs, %zu macroblocks per picture (%.0f %%), %d pictures per launch\n", runs.size(), n_mbs, 100.0 * n_mbs / (WMB * HMB), S);
    for (int T : { W, 128, 64, 32, 16 }) {
        const dim3 grid((runs.size() + 3) / 4, S);
        for (int it = 0; it < 3; it++) hipLaunchKernelGGL(k_probe<8>, grid, dim3(256), 0, 0, a, b, FRAME, d_runs, (int)runs.size(), T);
        CK(hipEventRecord(e0));
        const int reps = 10;
        for (int it = 0; it < reps; it++) hipLaunchKernelGGL(k_probe<8>, grid, dim3(256), 0, 0, a, b, FRAME, d_runs, (int)runs.size(), T);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bytes = 2.0 * 384 * n_mbs * S * reps;
        printf("strip width %4d%s: %.3f ms per launch, %.2f TB/s (read + write)\n", T, T == W ? " (planar)" : T == 16 ? " (MB tiles)" : "", ms / reps, bytes / (ms * 1e-3) / 1e12);
    }
    return 0;
}
