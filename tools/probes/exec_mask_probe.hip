// Does the COST of a VALU instruction depend on which lanes are active?  (follow-up of lane_util_probe.hip, which found 14.5
// instead of 4.5 cycles per instruction with 8 or fewer active lanes.)  One kernel, the EXEC mask of the instruction stream is
// a launch argument; every wavefront times itself (one wave per SIMD on every CU).
// build: hipcc --offload-arch=gfx950 -O2 -o exec_mask_probe exec_mask_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define ITERS 2048
__global__ __launch_bounds__(256) void k_mask(unsigned long long *out, unsigned long long mask, int waves_on)
{
    int a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 7 + i;
    const int b = 0x00010003 + (threadIdx.x & 1), c = 0x05040100;
    const int lane = threadIdx.x & 63;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (((mask >> lane) & 1ull) && (int)(threadIdx.x >> 6) < waves_on) {
        for (int it = 0; it < ITERS; it++) {
            asm volatile(
                "v_pk_add_i16 %0, %0, %8\n\tv_pk_add_i16 %1, %1, %8\n\tv_pk_add_i16 %2, %2, %8\n\tv_pk_add_i16 %3, %3, %8\n\t"
                "v_pk_add_i16 %4, %4, %8\n\tv_pk_add_i16 %5, %5, %8\n\tv_pk_add_i16 %6, %6, %8\n\tv_pk_add_i16 %7, %7, %8\n\t"
                "v_perm_b32 %0, %0, %8, %9\n\tv_perm_b32 %1, %1, %8, %9\n\tv_perm_b32 %2, %2, %8, %9\n\tv_perm_b32 %3, %3, %8, %9\n\t"
                "v_perm_b32 %4, %4, %8, %9\n\tv_perm_b32 %5, %5, %8, %9\n\tv_perm_b32 %6, %6, %8, %9\n\tv_perm_b32 %7, %7, %8, %9\n\t"
                "v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\t"
                "v_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8\n\t"
                "v_mad_i32_i24 %0, %0, %8, %9\n\tv_mad_i32_i24 %1, %1, %8, %9\n\tv_mad_i32_i24 %2, %2, %8, %9\n\tv_mad_i32_i24 %3, %3, %8, %9\n\t"
                "v_mad_i32_i24 %4, %4, %8, %9\n\tv_mad_i32_i24 %5, %5, %8, %9\n\tv_mad_i32_i24 %6, %6, %8, %9\n\tv_mad_i32_i24 %7, %7, %8, %9"
                : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    int sink = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) sink += a[i];
    if (sink == 0x7fffffff) out[0] = sink;
    if (lane == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
int main()
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    unsigned long long *d;
    (void)hipMalloc(&d, (size_t)cus * 4 * sizeof(unsigned long long));
    struct { const char *name; unsigned long long mask; } cases[] = {
        { "all 64", ~0ull }, { "lanes 0-15", 0xFFFFull }, { "lanes 0-11", 0xFFFull }, { "lanes 0-8 (9)", 0x1FFull }, { "lanes 0-7", 0xFFull },
        { "lanes 8-15", 0xFF00ull }, { "lanes 0-3", 0xFull }, { "lane 0", 1ull }, { "lanes 0,16,32,48", 0x0001000100010001ull },
        { "lanes 0-7 + 32-39", 0xFF000000FFull }, { "lanes 0-7 + 16-23", 0xFF00FFull }, { "even lanes 0-14 (8)", 0x5555ull },
        { "lanes 0-3 of every 16 (16)", 0x000F000F000F000Full }, { "lane 0 of every 8 (8)", 0x0101010101010101ull }, { "lanes 0-2 of every 8 (24)", 0x0707070707070707ull },
    };
    for (int waves_on = 4; waves_on >= 1; waves_on -= 3)
        for (auto &c : cases) {
            std::vector<unsigned long long> h((size_t)cus * 4);
            hipLaunchKernelGGL(k_mask, dim3(cus), dim3(256), 0, 0, d, c.mask, waves_on);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h.data(), d, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
            std::vector<unsigned long long> busy;
            for (int b = 0; b < cus; b++) for (int w = 0; w < waves_on; w++) busy.push_back(h[b * 4 + w]);
            std::sort(busy.begin(), busy.end());
            printf("%d busy wave(s) per CU, %-28s: cycles per instruction median %.2f max %.2f\n", waves_on, c.name, busy[busy.size() / 2] / (double)(ITERS * 32), busy.back() / (double)(ITERS * 32));
        }
    return 0;
}
