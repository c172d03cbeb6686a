// What do SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU / SQ_ACTIVE_INST_VALU say about ACTIVE LANES on gfx950?  
//
// One kernel per number of active lanes (64, 32, 16, 8, 1): every wavefront runs a long unrolled stream of v_pk_add_i16 /
// v_perm_b32 / v_add_u32 with EXEC restricted to the first ACTIVE lanes.  Run under
//     rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU -- ./lane_util_probe
// (k_lanes<N, false>: eight independent register chains; <N, true>: dependent back-to-back instructions)
// and divide: the ratio THREAD_CYCLES / INSTS of k_lanes<64> calibrates tools/sq_counters.py ("active lanes per VALU
// instruction" = ratio x 64 / ratio of the full wavefront).  The kernel also times itself: does a half-empty wavefront issue faster?
//
// build: hipcc --offload-arch=gfx950 -O2 -o lane_util_probe lane_util_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define ITERS 4096

template <int ACTIVE, bool DEP>
__global__ __launch_bounds__(256) void k_lanes(unsigned long long *out)
{
    int a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 7 + i;
    const int b = 0x00010003 + (threadIdx.x & 1), c = 0x05040100;
    const unsigned long long t0 = __builtin_readcyclecounter();
    if ((int)(threadIdx.x & 63) < ACTIVE) {
        for (int it = 0; it < ITERS; it++) {
            if (DEP) {
                /* ONE dependent chain per register, the four instructions of a register back to back (no s_nop between asm statements: one block) */
#pragma unroll
                for (int i = 0; i < 8; i++)
                    asm volatile("v_pk_add_i16 %0, %0, %1\n\tv_perm_b32 %0, %0, %1, %2\n\tv_add_u32 %0, %0, %1\n\tv_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            } else {
                /* eight independent chains interleaved: a register is touched again seven instructions later */
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
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    int sink = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) sink += a[i];
    if (sink == 0x7fffffff) out[0] = sink;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int ACTIVE, bool DEP>
static void run(unsigned long long *d, int cus)
{
    std::vector<unsigned long long> h((size_t)cus * 4);
    hipLaunchKernelGGL((k_lanes<ACTIVE, DEP>), dim3(cus), dim3(256), 0, 0, d);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double n = (double)ITERS * 32;
    printf("active lanes %2d, %s: %8.0f wave instructions, cycles per instruction (one wave per SIMD): median %.2f, max %.2f\n", ACTIVE, DEP ? "dependent (4-instruction chains back to back)" : "8 independent chains", n,
           h[h.size() / 2] / n, h.back() / n);
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    unsigned long long *d;
    hipMalloc(&d, (size_t)cus * 4 * sizeof(unsigned long long));
    printf("%s, %d CUs; every kernel: %d workgroups x 4 wavefronts x %d VALU instructions\n", p.name, cus, cus, ITERS * 32);
    run<64, false>(d, cus); run<32, false>(d, cus); run<16, false>(d, cus); run<8, false>(d, cus); run<1, false>(d, cus);
    run<64, true>(d, cus); run<16, true>(d, cus); run<8, true>(d, cus); run<1, true>(d, cus);
    return 0;
}
