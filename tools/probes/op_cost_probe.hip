// Cost of single VALU instructions that the r4 probe could not time cleanly (its per-statement asm got s_nop padding or VCC
// spills between statements): one asm block of 32 instructions on 8 independent registers per loop trip, 4 waves per CU (one per
// SIMD) and 8 waves per CU... prints cycles per wave64 instruction as one wave sees it and per SIMD with 2 waves on it.
// build: hipcc --offload-arch=gfx950 -O2 -o op_cost_probe op_cost_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define ITERS 2048
#define R8(T) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
template <int OP> __device__ __forceinline__ void body(int (&a)[8], int b, int c)
{
#define BLK(fmt) asm volatile(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7) \
        : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(b), "v"(c), "s"(0xaaaaaaaa) : "vcc", "s22", "s23")
#define F0(i) "v_add_u32 %" #i ", %" #i ", %8\n\t"
#define F1(i) "v_sub_u32 %" #i ", %" #i ", %8\n\t"
#define F2(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t"
#define F3(i) "v_cndmask_b32_dpp %" #i ", %" #i ", %8, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define F4(i) "v_mov_b32_dpp %" #i ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
#define F5(i) "v_pk_add_u16 %" #i ", %" #i ", %8\n\t"
#define F6(i) "v_pk_mul_lo_u16 %" #i ", %" #i ", %8\n\t"
#define F7(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n\t"
#define F8(i) "v_ashrrev_i32 %" #i ", 1, %" #i "\n\t"
#define F9(i) "v_mul_i32_i24 %" #i ", %" #i ", %8\n\t"
#define F10(i) "v_bitop3_b32 %" #i ", %" #i ", %8, %9 bitop3:0xe4\n\t"
#define F11(i) "v_med3_i32 %" #i ", %" #i ", %8, %9\n\t"
#define F12(i) "v_add_u32_e64 %" #i ", %" #i ", %8\n\t"
#define F13(i) "v_and_b32 %" #i ", %" #i ", %8\n\t"
#define F14(i) "v_mov_b32 %" #i ", %8\n\t"
#define F15(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]\n\t"
#define F16(i) "v_dot4_u32_u8 %" #i ", %8, %9, %" #i "\n\t"
#define F17(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 8\n\t"
#define F18(i) "v_mul_i32_i24_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n\t"
#define F19(i) "v_lshl_add_u32 %" #i ", %" #i ", 2, %8\n\t"
#define F20(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc\n\t"
#define F21(i) "v_cmp_lt_i32_e64 s[22:23], %" #i ", %8\n\tv_cndmask_b32_e64 %" #i ", %" #i ", %8, s[22:23]\n\t"
#define F22(i) "v_cmp_lt_i32 vcc, %" #i ", %8\n\tv_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\t"
#define F23(i) "v_bfi_b32 %" #i ", %9, %" #i ", %8\n\t"
#define F24(i) "v_cndmask_b32_sdwa %" #i ", %" #i ", %8, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n\t"
#define F25(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %8, vcc\n\t"
#define F26(i) "v_max_i32 %" #i ", %" #i ", %8\n\t"
#define F27(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n\t"
#define F28(i) "v_or_b32 %" #i ", %" #i ", %8\n\t"
#define F29(i) "v_pk_max_i16 %" #i ", %" #i ", %8\n\t"
#define F30(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\tv_add_u32 %" #i ", %" #i ", %8\n\t"
#define F31(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\tv_pk_add_u16 %" #i ", %" #i ", %8\n\t"
#define F32(i) "v_cmp_lt_i32 vcc, %" #i ", %8\n\tv_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\tv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n\tv_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\tv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n\t"
#define F33(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n\ts_nop 0\n\t"
#define F34(i) "v_cndmask_b32_dpp %" #i ", %" #i ", %8, vcc quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n\tv_pk_add_u16 %" #i ", %" #i ", %8\n\t"
    if constexpr (OP == 0) BLK(F0); else if constexpr (OP == 1) BLK(F1); else if constexpr (OP == 2) BLK(F2); else if constexpr (OP == 3) BLK(F3);
    else if constexpr (OP == 4) BLK(F4); else if constexpr (OP == 5) BLK(F5); else if constexpr (OP == 6) BLK(F6); else if constexpr (OP == 7) BLK(F7);
    else if constexpr (OP == 8) BLK(F8); else if constexpr (OP == 9) BLK(F9); else if constexpr (OP == 10) BLK(F10); else if constexpr (OP == 11) BLK(F11);
    else if constexpr (OP == 12) BLK(F12); else if constexpr (OP == 13) BLK(F13); else if constexpr (OP == 14) BLK(F14); else if constexpr (OP == 15) BLK(F15);
    else if constexpr (OP == 16) BLK(F16); else if constexpr (OP == 17) BLK(F17); else if constexpr (OP == 18) BLK(F18); else if constexpr (OP == 19) BLK(F19); else if constexpr (OP == 20) BLK(F20); else if constexpr (OP == 21) BLK(F21); else if constexpr (OP == 22) BLK(F22);
    else if constexpr (OP == 23) BLK(F23); else if constexpr (OP == 24) BLK(F24); else if constexpr (OP == 25) BLK(F25); else if constexpr (OP == 26) BLK(F26); else if constexpr (OP == 27) BLK(F27); else if constexpr (OP == 28) BLK(F28); else if constexpr (OP == 29) BLK(F29); else if constexpr (OP == 30) BLK(F30); else if constexpr (OP == 31) BLK(F31); else if constexpr (OP == 32) BLK(F32); else if constexpr (OP == 33) BLK(F33); else BLK(F34);
}
static const char *names[] = { "v_add_u32", "v_sub_u32", "v_cndmask_b32 (vcc)", "v_cndmask_b32_dpp quad_perm", "v_mov_b32_dpp quad_perm", "v_pk_add_u16", "v_pk_mul_lo_u16", "v_perm_b32",
    "v_ashrrev_i32", "v_mul_i32_i24", "v_bitop3_b32", "v_med3_i32", "v_add_u32_e64", "v_and_b32", "v_mov_b32", "v_cndmask_b32_e64 (sgpr pair)", "v_dot4_u32_u8", "v_alignbit_b32", "v_mul_i32_i24_sdwa", "v_lshl_add_u32", "v_cndmask_b32_e64 (vcc)", "v_cmp_e64 + v_cndmask_e64 (2 instr, sgpr pair)", "v_cmp + v_cndmask (2 instr, vcc)", "v_bfi_b32", "v_cndmask_b32_sdwa (vcc)", "v_addc_co_u32 (vcc in and out)", "v_max_i32", "v_lshlrev_b32", "v_or_b32", "v_pk_max_i16", "v_cndmask(vcc) + v_add_u32 (PAIR)", "v_cndmask(vcc) + v_pk_add_u16 (PAIR)", "v_cmp + 4 x v_cndmask(vcc) (FIVE)", "v_cndmask(vcc) + s_nop 0 (PAIR)", "v_cndmask_dpp(vcc) + v_pk_add (PAIR)" };
template <int OP> __global__ void k(unsigned long long *out)
{
    int a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 7 + i;
    const int b = 0x00010003 + (threadIdx.x & 1), c = 0x05040100;
    asm volatile("s_mov_b32 s20, 0x55555555\n\ts_mov_b32 s21, 0x55555555\n\ts_mov_b32 vcc_lo, 0xaaaaaaaa\n\ts_mov_b32 vcc_hi, 0xaaaaaaaa" ::: "s20", "s21", "s22", "s23", "vcc");
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++) body<OP>(a, b, c);
    const unsigned long long t1 = __builtin_readcyclecounter();
    int sink = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) sink += a[i];
    if (sink == 0x7fffffff) out[0] = sink;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}
template <int OP> static void run(unsigned long long *d, int cus)
{
    double res[2];
    for (int w = 0; w < 2; w++) {
        const int waves = w ? 8 : 4;
        std::vector<unsigned long long> h((size_t)cus * waves);
        hipLaunchKernelGGL(k<OP>, dim3(cus), dim3(64 * waves), 0, 0, d);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), d, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        res[w] = h[h.size() / 2] / (double)(ITERS * 32);
    }
    printf("%-32s one wave per SIMD: %5.2f cycles per instruction; two per SIMD: %5.2f as a wave sees it = %5.2f per SIMD\n", names[OP], res[0], res[1], res[1] / 2);
}
int main()
{
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    unsigned long long *d; (void)hipMalloc(&d, (size_t)cus * 8 * sizeof(unsigned long long));
    run<0>(d, cus); run<1>(d, cus); run<2>(d, cus); run<3>(d, cus); run<4>(d, cus); run<5>(d, cus); run<6>(d, cus); run<7>(d, cus); run<8>(d, cus); run<9>(d, cus);
    run<10>(d, cus); run<11>(d, cus); run<12>(d, cus); run<13>(d, cus); run<14>(d, cus); run<15>(d, cus); run<16>(d, cus); run<17>(d, cus); run<18>(d, cus); run<19>(d, cus); run<20>(d, cus); run<21>(d, cus); run<22>(d, cus); run<23>(d, cus); run<24>(d, cus); run<25>(d, cus); run<26>(d, cus); run<27>(d, cus); run<28>(d, cus); run<29>(d, cus); run<30>(d, cus); run<31>(d, cus); run<32>(d, cus); run<33>(d, cus); run<34>(d, cus);
    return 0;
}
