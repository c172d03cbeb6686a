// What a vector memory instruction costs a compute unit when its data is cache resident: 256 workgroups (one per CU) of 4 / 8 / 12
// wavefronts, every wavefront issues blocks of 8 independent loads (or stores) of one width and one address pattern and waits for
// them; prints shader cycles per wave-level instruction AS THE CU SEES IT (a wave's cycles per instruction / waves per CU).
// Patterns (w = lane >> 3, l = lane & 7: the eight-lane workers of k_frame_dbk):
//   lin    lane * size                      (fully coalesced)
//   rows   w * 384 + 32 * l                 (a worker's lanes read two-row pieces of ITS macroblock tile)
//   same   w * 48                           (the eight lanes of a worker read the same bytes: the 48-byte record)
//   strip  w * 384 + 32 * l + 12            (the last four columns of a tile's rows)
//   col    w * 384 + 192 + 2 * l            (16 bits per lane out of one row)
//   one8   like rows, but only lane 0 of every worker is active
// build: hipcc --offload-arch=gfx950 -O2 -o vmem_issue_probe vmem_issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define ITERS 256
enum { LD8, LD16, LD32, LD64, LD128, ST8, ST16, ST32, ST128, NOPS };
static const char *op_names[] = { "global_load_ubyte", "global_load_ushort", "global_load_dword", "global_load_dwordx2", "global_load_dwordx4",
                                  "global_store_byte", "global_store_short", "global_store_dword", "global_store_dwordx4" };
static const char *pat_names[] = { "lin", "rows", "same", "strip", "col", "one8" };
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <int OP> __device__ __forceinline__ void blk(const unsigned char *p, unsigned &acc)
{
    typedef __attribute__((address_space(1))) unsigned char *gp;
    gp g = (gp)p;
    if constexpr (OP <= LD128) {
        unsigned r[8]; u32x2 r2[8]; u32x4 r4[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if constexpr (OP == LD8) asm volatile("global_load_ubyte %0, %1, off offset:%2" : "=v"(r[i]) : "v"(g), "n"(i * 1024 - 4096) : "memory");
            if constexpr (OP == LD16) asm volatile("global_load_ushort %0, %1, off offset:%2" : "=v"(r[i]) : "v"(g), "n"(i * 1024 - 4096) : "memory");
            if constexpr (OP == LD32) asm volatile("global_load_dword %0, %1, off offset:%2" : "=v"(r[i]) : "v"(g), "n"(i * 1024 - 4096) : "memory");
            if constexpr (OP == LD64) asm volatile("global_load_dwordx2 %0, %1, off offset:%2" : "=v"(r2[i]) : "v"(g), "n"(i * 1024 - 4096) : "memory");
            if constexpr (OP == LD128) asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(r4[i]) : "v"(g), "n"(i * 1024 - 4096) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 8; i++) { if constexpr (OP <= LD32) acc += r[i]; else if constexpr (OP == LD64) acc += r2[i].x; else acc += r4[i].x; }
    } else {
        u32x4 v = { acc, acc, acc, acc };
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if constexpr (OP == ST8) asm volatile("global_store_byte %0, %1, off offset:%2" :: "v"(g), "v"(acc), "n"(i * 1024 - 4096) : "memory");
            if constexpr (OP == ST16) asm volatile("global_store_short %0, %1, off offset:%2" :: "v"(g), "v"(acc), "n"(i * 1024 - 4096) : "memory");
            if constexpr (OP == ST32) asm volatile("global_store_dword %0, %1, off offset:%2" :: "v"(g), "v"(acc), "n"(i * 1024 - 4096) : "memory");
            if constexpr (OP == ST128) asm volatile("global_store_dwordx4 %0, %1, off offset:%2" :: "v"(g), "v"(v), "n"(i * 1024 - 4096) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}
template <int OP> __global__ void k(unsigned char *buf, unsigned long long *out, int pat)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, w = lane >> 3, l = lane & 7;
    // 32 KB per wavefront: 8 instructions x 1024 bytes apart
    unsigned char *base = buf + ((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * 32768;
    const int size = OP == LD8 || OP == ST8 ? 1 : OP == LD16 || OP == ST16 ? 2 : OP == LD32 || OP == ST32 ? 4 : OP == LD64 ? 8 : 16;
    int off = pat == 0 ? lane * size : pat == 1 || pat == 5 ? w * 384 + 32 * l : pat == 2 ? w * 48 : pat == 3 ? w * 384 + 32 * l + 12 : w * 384 + 192 + 2 * l;
    off &= ~(size - 1);
    const unsigned char *p = base + 4096 + off;
    unsigned acc = lane;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_readcyclecounter();
    if (pat != 5 || l == 0)
        for (int it = 0; it < ITERS; it++) blk<OP>(p, acc);
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (acc == 0x7fffffffu) out[0] = acc;
    if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + wave] = t1 - t0;
}
template <int OP> static void run(unsigned char *buf, unsigned long long *d, int cus)
{
    for (int pat = 0; pat < 6; pat++) {
        double res[3];
        for (int wi = 0; wi < 3; wi++) {
            const int waves = 4 + 4 * wi;
            std::vector<unsigned long long> h((size_t)cus * waves);
            hipLaunchKernelGGL(k<OP>, dim3(cus), dim3(64 * waves), 0, 0, buf, d, pat);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h.data(), d, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            res[wi] = h[h.size() / 2] / (double)(ITERS * 8) / waves;
        }
        printf("%-22s %-6s cycles per instruction and CU with 4 / 8 / 12 wavefronts: %6.1f %6.1f %6.1f\n", op_names[OP], pat_names[pat], res[0], res[1], res[2]);
    }
}
int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    unsigned char *buf; (void)hipMalloc(&buf, (size_t)cus * 12 * 32768 + 65536); (void)hipMemset(buf, 1, (size_t)cus * 12 * 32768 + 65536);
    unsigned long long *d; (void)hipMalloc(&d, (size_t)cus * 12 * sizeof(unsigned long long));
    run<LD8>(buf, d, cus); run<LD16>(buf, d, cus); run<LD32>(buf, d, cus); run<LD64>(buf, d, cus); run<LD128>(buf, d, cus);
    run<ST8>(buf, d, cus); run<ST16>(buf, d, cus); run<ST32>(buf, d, cus); run<ST128>(buf, d, cus);
    return 0;
}
