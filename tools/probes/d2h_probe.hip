// How fast do decoded pictures reach pinned host memory?  256 pictures of 3.1 MB (1080p I420) per round, as in bench.py's
// end_to_end_host_output leg:
//   (a) hipMemcpyAsync device -> pinned host, one call per picture on one stream (what sink_fetch did through round 5: SDMA)
//   (b) a kernel that WRITES the host mirror directly through its device pointer, with 16 / 64 / 128 / 256 / 1024 contiguous bytes
//       per row piece (a macroblock tile row is 16 bytes: how many neighbouring macroblocks must a store instruction cover?)
//   (c) the same kernels with 1 / 4 / 16 pictures per launch
// build: hipcc --offload-arch=gfx950 -O2 -o d2h_probe d2h_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// every lane moves 16 bytes; SEG = contiguous bytes a group of SEG/16 consecutive lanes writes; consecutive groups write rows W bytes apart
// (like tile rows of neighbouring macroblocks laid side by side in a planar picture)
template <int SEG>
__global__ __launch_bounds__(256) void k_rows(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16, uint32_t row_bytes)
{
    constexpr uint32_t L = SEG / 16;                 // lanes per contiguous piece
    const size_t pic = blockIdx.y;
    src += pic * n16; dst += pic * n16;
    const uint32_t pieces_per_row = row_bytes / SEG;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        // piece p = i / L; a block of 16 pieces with the same column goes to 16 consecutive rows (a "macroblock row" of pieces)
        const size_t p = i / L; const uint32_t l = (uint32_t)(i % L);
        const size_t blk = p / 16; const uint32_t r = (uint32_t)(p % 16);
        const size_t col = blk % pieces_per_row, band = blk / pieces_per_row;
        const size_t o = ((band * 16 + r) * (size_t)row_bytes + col * SEG) / 16 + l;
        if (o < n16) dst[o] = src[i];
    }
}

int main()
{
    const size_t pic = 1920 * 1088 * 3 / 2, n16 = pic / 16;
    const int N = 256;
    uint8_t *d, *h, *hd;
    CK(hipMalloc(&d, pic * N));
    CK(hipHostMalloc((void **)&h, pic * N, hipHostMallocDefault));
    CK(hipHostGetDevicePointer((void **)&hd, h, 0));
    CK(hipMemset(d, 1, pic * N));
    hipStream_t st; CK(hipStreamCreate(&st));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](auto a, auto b) { return std::chrono::duration<double>(b - a).count(); };
    for (int rep = 0; rep < 2; rep++) {
        auto t0 = now();
        for (int i = 0; i < N; i++) CK(hipMemcpyAsync(h + i * pic, d + i * pic, pic, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        double s = secs(t0, now());
        printf("hipMemcpyAsync x %d on one stream: %.1f ms, %.1f GB/s\n", N, s * 1e3, pic * N / s / 1e9);
    }
    {
        hipStream_t st2; CK(hipStreamCreate(&st2));
        auto t0 = now();
        for (int i = 0; i < N; i++) CK(hipMemcpyAsync(h + i * pic, d + i * pic, pic, hipMemcpyDeviceToHost, (i & 1) ? st2 : st));
        CK(hipStreamSynchronize(st)); CK(hipStreamSynchronize(st2));
        double s = secs(t0, now());
        printf("hipMemcpyAsync x %d on two streams: %.1f ms, %.1f GB/s\n", N, s * 1e3, pic * N / s / 1e9);
    }
#define RUN(SEG, PER, WGS) do { \
        for (int rep = 0; rep < 2; rep++) { \
            auto t0 = now(); \
            for (int i = 0; i < N; i += PER) hipLaunchKernelGGL(k_rows<SEG>, dim3(WGS, PER), dim3(256), 0, st, (const uint4 *)(d + i * pic), (uint4 *)(hd + i * pic), n16, 1920u); \
            CK(hipStreamSynchronize(st)); \
            double s = secs(t0, now()); \
            if (rep) printf("kernel writes, %4d contiguous bytes per row piece, %2d pictures per launch, %4d workgroups per picture: %.1f ms, %.1f GB/s\n", SEG, PER, WGS, s * 1e3, pic * N / s / 1e9); \
        } } while (0)
    RUN(16, 1, 256); RUN(64, 1, 256); RUN(128, 1, 256); RUN(256, 1, 256); RUN(1920, 1, 256);
    RUN(64, 1, 64); RUN(128, 1, 64); RUN(128, 1, 1024);
    RUN(64, 4, 256); RUN(128, 4, 256); RUN(128, 16, 64); RUN(1920, 16, 64);
    // is what arrived what was sent?
    size_t bad = 0; for (size_t i = 0; i < pic * N; i += 4099) bad += h[i] != 1;
    printf("spot check: %zu bad bytes\n", bad);
    return 0;
}
