// What does one wave64 instruction cost on a gfx950 SIMD?  
//
// Every kernel below is a long unrolled run of ONE instruction on 8 independent register chains (so that neither
// dependent-issue latency nor the register file is the limit) inside a loop; a launch puts W waves on every SIMD of
// every CU (grid = CUs, block = 256 * W) and each wave times itself with s_memtime (shader clock).  Printed per
// instruction and W:
//     cyc/instr (1 wave)  — the issue interval a single wave sees, back to back
//     cyc/instr (SIMD)    — W waves share the SIMD: wall cycles of the slowest wave / (W x instructions per wave),
//                           i.e. the SIMD's throughput cost of the instruction
// plus pairs of DIFFERENT instruction classes on the waves of one SIMD (VALU next to SALU, VALU next to LDS, VALU next
// to MFMA): do they overlap?
//
// build: hipcc --offload-arch=gfx950 -O2 -o valu_rate_probe valu_rate_probe.hip ; run: ./valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
#include <string>

#define ITERS 2048
#define PER_ITER 64            // instructions per loop trip (8 chains x 8)

typedef int v4i __attribute__((ext_vector_type(4)));
typedef long v2l __attribute__((ext_vector_type(2)));

// second table: single-instruction kernels generated from a list (name, asm template with %0 = chain register, %1 = b, %2 = c)
#define EXTRA_OPS(X) \
    X(E_AND, "v_and_b32 %0, %0, %1") \
    X(E_OR, "v_or_b32 %0, %0, %1") \
    X(E_XOR, "v_xor_b32 %0, %0, %1") \
    X(E_SUB, "v_sub_u32 %0, %0, %1") \
    X(E_LSHL, "v_lshlrev_b32 %0, 3, %0") \
    X(E_LSHR, "v_lshrrev_b32 %0, 3, %0") \
    X(E_ASHR, "v_ashrrev_i32 %0, 3, %0") \
    X(E_MAX_I32, "v_max_i32 %0, %0, %1") \
    X(E_MIN_U32, "v_min_u32 %0, %0, %1") \
    X(E_MOV, "v_mov_b32 %0, %1") \
    X(E_ADD_E64, "v_add_u32_e64 %0, %0, %1") \
    X(E_ADD_LIT, "v_add_u32 %0, 0x12345, %0") \
    X(E_MUL_U24, "v_mul_u32_u24 %0, %0, %1") \
    X(E_MUL_I24, "v_mul_i32_i24 %0, %0, %1") \
    X(E_MAD_U24, "v_mad_u32_u24 %0, %0, %1, %2") \
    X(E_ADD3, "v_add3_u32 %0, %0, %1, %2") \
    X(E_MED3, "v_med3_i32 %0, %0, %1, %2") \
    X(E_BFI, "v_bfi_b32 %0, %0, %1, %2") \
    X(E_AND_OR, "v_and_or_b32 %0, %0, %1, %2") \
    X(E_LSHL_OR, "v_lshl_or_b32 %0, %0, 8, %1") \
    X(E_ADD_U16, "v_add_u16 %0, %0, %1") \
    X(E_MAX_I16, "v_max_i16 %0, %0, %1") \
    X(E_MUL_LO_U16, "v_mul_lo_u16 %0, %0, %1") \
    X(E_MAD_I16, "v_mad_i16 %0, %0, %1, %2") \
    X(E_FMA_F32, "v_fma_f32 %0, %0, %1, %2") \
    X(E_FMAC_F32, "v_fmac_f32 %0, %1, %2") \
    X(E_ADD_F32, "v_add_f32 %0, %0, %1") \
    X(E_PK_ADD_F16, "v_pk_add_f16 %0, %0, %1") \
    X(E_PK_ADD_U16, "v_pk_add_u16 %0, %0, %1") \
    X(E_PK_SUB_I16, "v_pk_sub_i16 %0, %0, %1") \
    X(E_PK_ASHR, "v_pk_ashrrev_i16 %0, 1, %0") \
    X(E_PK_LSHL, "v_pk_lshlrev_b16 %0, 1, %0") \
    X(E_PK_MIN, "v_pk_min_i16 %0, %0, %1") \
    X(E_ADD_SDWA, "v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD") \
    X(E_MOV_SDWA, "v_mov_b32_sdwa %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2") \
    X(E_MOV_DPP, "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
    X(E_ADD_DPP, "v_add_u32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf") \
    X(E_CVT_PK_U8, "v_cvt_pk_u8_f32 %0, %1, 0, %0") \
    X(E_MSAD, "v_msad_u8 %0, %0, %1, %2") \
    X(E_SAD_U16, "v_sad_u16 %0, %0, %1, %2") \
    X(E_DOT2_I16, "v_dot2_i32_i16 %0, %1, %2, %0") \
    X(E_DOT8_I4, "v_dot8_i32_i4 %0, %1, %2, %0") \
    X(E_DOT4_U8, "v_dot4_u32_u8 %0, %1, %2, %0") \
    X(E_ALIGNBYTE, "v_alignbyte_b32 %0, %0, %1, 1") \
    X(E_BCNT, "v_bcnt_u32_b32 %0, %1, %0") \
    X(E_READLANE, "v_readlane_b32 s20, %0, 3") \
    X(E_CMP, "v_cmp_lt_i32 vcc, %0, %1") \
    X(E_CMP_E64, "v_cmp_lt_i32_e64 s[20:21], %0, %1")
enum ExtraOp {
#define X(n, t) n,
    EXTRA_OPS(X)
#undef X
    N_EXTRA };
static const char *extra_name[N_EXTRA] = {
#define X(n, t) t,
    EXTRA_OPS(X)
#undef X
};
template <int OP> __device__ __forceinline__ void extra_one(int &a, int b, int c);
#define X(n, t) template <> __device__ __forceinline__ void extra_one<n>(int &a, int b, int c) { asm volatile(t : "+v"(a) : "v"(b), "v"(c) : "vcc", "s20", "s21"); }
EXTRA_OPS(X)
#undef X

enum Op { PK_ADD, PK_MAD, PK_MUL, PK_MAX, PERM, MAD24, ALIGNBIT, DOT4, ADD_U32, CNDMASK, LSHL_ADD, BFE, MUL_LO, SAD, DS_READ, DS_READ128, SALU, MFMA_I8, MIX_PK_PERM, N_OPS };
static const char *op_name[N_OPS] = { "v_pk_add_i16", "v_pk_mad_i16", "v_pk_mul_lo_u16", "v_pk_max_i16", "v_perm_b32", "v_mad_i32_i24", "v_alignbit_b32",
    "v_dot4_i32_i8", "v_add_u32", "v_cndmask_b32", "v_lshl_add_u32", "v_bfe_u32", "v_mul_lo_u32", "v_sad_u8", "ds_read_b32", "ds_read_b128", "s_add_u32 (SALU)",
    "v_mfma_i32_16x16x64_i8", "v_pk_add_i16 + v_perm_b32 alternating" };

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__device__ __forceinline__ void body(int (&a)[8], int b, int c, v4i (&acc)[2], v4i (&wide)[8], const __attribute__((address_space(3))) int *lp, int (&s)[8])
{
#define ONE_PK_ADD(i) asm volatile("v_pk_add_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ONE_PK_MAD(i) asm volatile("v_pk_mad_i16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define ONE_PK_MUL(i) asm volatile("v_pk_mul_lo_u16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ONE_PK_MAX(i) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ONE_PERM(i) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define ONE_MAD24(i) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define ONE_ALIGNBIT(i) asm volatile("v_alignbit_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define ONE_DOT4(i) asm volatile("v_dot4_i32_i8 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
#define ONE_ADD(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ONE_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b) : "vcc");
#define ONE_LSHL_ADD(i) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(a[i]) : "v"(b));
#define ONE_BFE(i) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(a[i]));
#define ONE_MUL_LO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define ONE_SAD(i) asm volatile("v_sad_u8 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define ONE_DS(i) asm volatile("ds_read_b32 %0, %1 offset:" #i "*4" : "=v"(a[i]) : "v"(lp));
#define ONE_DS128(i) asm volatile("ds_read_b128 %0, %1 offset:" #i "*1024" : "=v"(wide[i]) : "v"(lp));
#define ONE_S(i) asm volatile("s_add_u32 %0, %0, %1" : "+s"(s[i]) : "s"(s[(i + 1) & 7]) : "scc");
    if constexpr (OP == PK_ADD) { REP8(ONE_PK_ADD) }
    else if constexpr (OP == PK_MAD) { REP8(ONE_PK_MAD) }
    else if constexpr (OP == PK_MUL) { REP8(ONE_PK_MUL) }
    else if constexpr (OP == PK_MAX) { REP8(ONE_PK_MAX) }
    else if constexpr (OP == PERM) { REP8(ONE_PERM) }
    else if constexpr (OP == MAD24) { REP8(ONE_MAD24) }
    else if constexpr (OP == ALIGNBIT) { REP8(ONE_ALIGNBIT) }
    else if constexpr (OP == DOT4) { REP8(ONE_DOT4) }
    else if constexpr (OP == ADD_U32) { REP8(ONE_ADD) }
    else if constexpr (OP == CNDMASK) { REP8(ONE_CND) }
    else if constexpr (OP == LSHL_ADD) { REP8(ONE_LSHL_ADD) }
    else if constexpr (OP == BFE) { REP8(ONE_BFE) }
    else if constexpr (OP == MUL_LO) { REP8(ONE_MUL_LO) }
    else if constexpr (OP == SAD) { REP8(ONE_SAD) }
    else if constexpr (OP == DS_READ) { REP8(ONE_DS) asm volatile("s_waitcnt lgkmcnt(0)"); }
    else if constexpr (OP == DS_READ128) { REP8(ONE_DS128) asm volatile("s_waitcnt lgkmcnt(0)"); }
    else if constexpr (OP == SALU) { REP8(ONE_S) }
    else if constexpr (OP == MFMA_I8) {
        // 8 MFMAs on two accumulators: A = 16x64 i8 (4 VGPRs per lane), B likewise
        v4i av = { a[0], a[1], a[2], a[3] }, bv = { a[4], a[5], a[6], a[7] };
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i & 1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, acc[i & 1], 0, 0, 0);
    }
    else if constexpr (OP == MIX_PK_PERM) {
        ONE_PK_ADD(0) ONE_PERM(1) ONE_PK_ADD(2) ONE_PERM(3) ONE_PK_ADD(4) ONE_PERM(5) ONE_PK_ADD(6) ONE_PERM(7)
    }
}

// out[wave_global] = {cycles, instructions}
template <int OP>
__device__ __forceinline__ void run_wave(unsigned long long *out, int iters)
{
    __shared__ int lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    __syncthreads();
    int a[8], s[8];
    v4i acc[2] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 } }, wide[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = threadIdx.x * 7 + i; s[i] = __builtin_amdgcn_readfirstlane(blockIdx.x + i); wide[i] = (v4i){ 0, 0, 0, 0 }; }
    const int b = 0x00010003 + (threadIdx.x & 1), c = 0x05040100;
    const __attribute__((address_space(3))) int *lp = (const __attribute__((address_space(3))) int *)lds + (threadIdx.x & 63) * (OP == DS_READ128 ? 4 : 1);
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < PER_ITER / 8; u++) body<OP>(a, b, c, acc, wide, lp, s);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    int sink = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) sink += a[i] + s[i] + wide[i].x + wide[i].w;
    sink += acc[0].x + acc[1].y;
    if (sink == 0x7fffffff) out[0] = sink;             // keep everything alive
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = (unsigned long long)iters * PER_ITER;
    }
}

template <int OP>
__global__ void k_one(unsigned long long *out, int iters) { run_wave<OP>(out, iters); }

template <int OP>
__global__ void k_extra(unsigned long long *out, int iters)
{
    int a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 7 + i;
    const int b = 0x00010003 + (threadIdx.x & 1), c = 0x05040100;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < PER_ITER / 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) extra_one<OP>(a[i], b, c);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    int sink = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) sink += a[i];
    if (sink == 0x7fffffff) out[0] = sink;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = (unsigned long long)iters * PER_ITER;
    }
}
template <typename K>
static void launch(K kern, int waves_per_simd, int n_cu, unsigned long long *d, std::vector<unsigned long long> &h);
template <int WHICH>
__global__ void k_wide(unsigned long long *out, int iters)
{
    long a[8];
#pragma unroll
    for (int i = 0; i < 8; i++) a[i] = threadIdx.x * 11 + i;
    const long b = 0x3f8000003f800000l + (threadIdx.x & 1), c = 0x0504010005040100l;
    __builtin_amdgcn_s_barrier();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < PER_ITER / 8; u++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if constexpr (WHICH == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                else if constexpr (WHICH == 1) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(a[i]));
                else if constexpr (WHICH == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
                else asm volatile("v_pk_mov_b32 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    long sink = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) sink += a[i];
    if (sink == 0x7fffffff) out[0] = sink;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        out[2 * w] = t1 - t0;
        out[2 * w + 1] = (unsigned long long)iters * PER_ITER;
    }
}
template <int WHICH>
static void measure_wide(const char *name, int n_cu, unsigned long long *d, std::vector<unsigned long long> &h)
{
    printf("%-100s", name);
    for (int w : { 1, 2, 4 }) {
        launch(k_wide<WHICH>, w, n_cu, d, h);
        launch(k_wide<WHICH>, w, n_cu, d, h);
        const int n_waves = n_cu * 4 * w;
        unsigned long long worst = 0; double sum = 0;
        for (int i = 0; i < n_waves; i++) { worst = std::max(worst, h[2 * i]); sum += (double)h[2 * i]; }
        const double instr = (double)h[1];
        printf("  W=%d: wave %5.2f simd %5.2f", w, sum / n_waves / instr, (double)worst / (instr * w));
    }
    printf("\n");
}

template <int OP>
static void measure_extra(int n_cu, unsigned long long *d, std::vector<unsigned long long> &h);

// waves 0..3 of a workgroup land on SIMDs 0..3, waves 4..7 again on 0..3 (round robin): odd "rounds" run OPB
template <int OPA, int OPB>
__global__ void k_pair(unsigned long long *out, int iters)
{
    if (((threadIdx.x >> 6) >> 2) & 1) run_wave<OPB>(out, iters); else run_wave<OPA>(out, iters);
}

struct Res { double one_wave, simd; };
template <typename K>
static void launch(K kern, int waves_per_simd, int n_cu, unsigned long long *d, std::vector<unsigned long long> &h)
{
    const int block = 256 * waves_per_simd;
    hipLaunchKernelGGL(kern, dim3(n_cu), dim3(block), 0, 0, d, ITERS);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
}

template <int OP>
static void measure(int n_cu, unsigned long long *d, std::vector<unsigned long long> &h)
{
    printf("%-40s", op_name[OP]);
    for (int w : { 1, 2, 4 }) {
        launch(k_one<OP>, w, n_cu, d, h);        // warm
        launch(k_one<OP>, w, n_cu, d, h);
        const int n_waves = n_cu * 4 * w;
        unsigned long long worst = 0; double sum = 0;
        for (int i = 0; i < n_waves; i++) { worst = std::max(worst, h[2 * i]); sum += (double)h[2 * i]; }
        const double instr = (double)h[1];
        printf("  W=%d: wave %.2f  simd %.2f", w, sum / n_waves / instr, (double)worst / (instr * w));
    }
    printf("   cycles per wave64 instruction\n");
}

template <int OP>
static void measure_extra(int n_cu, unsigned long long *d, std::vector<unsigned long long> &h)
{
    printf("%-100s", extra_name[OP]);
    for (int w : { 1, 2, 4 }) {
        launch(k_extra<OP>, w, n_cu, d, h);
        launch(k_extra<OP>, w, n_cu, d, h);
        const int n_waves = n_cu * 4 * w;
        unsigned long long worst = 0; double sum = 0;
        for (int i = 0; i < n_waves; i++) { worst = std::max(worst, h[2 * i]); sum += (double)h[2 * i]; }
        const double instr = (double)h[1];
        printf("  W=%d: wave %5.2f simd %5.2f", w, sum / n_waves / instr, (double)worst / (instr * w));
    }
    printf("\n");
}
template <int OP> static void measure_extra_all(int n_cu, unsigned long long *d, std::vector<unsigned long long> &h)
{
    if constexpr (OP < N_EXTRA) { measure_extra<OP>(n_cu, d, h); measure_extra_all<OP + 1>(n_cu, d, h); }
}

template <int OPA, int OPB>
static void measure_pair(int n_cu, unsigned long long *d, std::vector<unsigned long long> &h)
{
    // 2 waves per SIMD: one runs OPA, the other OPB; alone each would take (cycles per instruction) x instructions
    launch(k_pair<OPA, OPB>, 2, n_cu, d, h);
    launch(k_pair<OPA, OPB>, 2, n_cu, d, h);
    double sa = 0, sb = 0; int na = 0, nb = 0;
    for (int blk = 0; blk < n_cu; blk++)
        for (int w = 0; w < 8; w++) {
            const double cyc = (double)h[2 * (blk * 8 + w)] / (double)h[2 * (blk * 8 + w) + 1];
            if (w >= 4) { sb += cyc; nb++; } else { sa += cyc; na++; }
        }
    printf("pair on one SIMD: %-28s %.2f cyc/instr   next to   %-28s %.2f cyc/instr\n", op_name[OPA], sa / na, op_name[OPB], sb / nb);
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    printf("%s: %d CUs, clock %d kHz\n", prop.name, n_cu, prop.clockRate);
    unsigned long long *d;
    const size_t n = (size_t)n_cu * 4 * 8 * 2;
    hipMalloc(&d, n * 8);
    std::vector<unsigned long long> h(n);
    printf("W = waves per SIMD (one workgroup of 256*W threads per CU); 'wave' = what one wave sees, 'simd' = SIMD throughput cost\n");
    measure<PK_ADD>(n_cu, d, h);
    measure<PK_MAD>(n_cu, d, h);
    measure<PK_MUL>(n_cu, d, h);
    measure<PK_MAX>(n_cu, d, h);
    measure<PERM>(n_cu, d, h);
    measure<MIX_PK_PERM>(n_cu, d, h);
    measure<MAD24>(n_cu, d, h);
    measure<ALIGNBIT>(n_cu, d, h);
    measure<DOT4>(n_cu, d, h);
    measure<ADD_U32>(n_cu, d, h);
    measure<CNDMASK>(n_cu, d, h);
    measure<LSHL_ADD>(n_cu, d, h);
    measure<BFE>(n_cu, d, h);
    measure<MUL_LO>(n_cu, d, h);
    measure<SAD>(n_cu, d, h);
    measure<DS_READ>(n_cu, d, h);
    measure<DS_READ128>(n_cu, d, h);
    measure<SALU>(n_cu, d, h);
    measure<MFMA_I8>(n_cu, d, h);
    measure_extra_all<0>(n_cu, d, h);
    measure_wide<0>("v_pk_fma_f32 (64-bit operands)", n_cu, d, h);
    measure_wide<1>("v_lshlrev_b64", n_cu, d, h);
    measure_wide<2>("v_pk_add_f32", n_cu, d, h);
    measure_wide<3>("v_pk_mov_b32", n_cu, d, h);
    measure_pair<PK_ADD, PK_ADD>(n_cu, d, h);
    measure_pair<PK_ADD, SALU>(n_cu, d, h);
    measure_pair<PK_ADD, DS_READ>(n_cu, d, h);
    measure_pair<PK_ADD, MFMA_I8>(n_cu, d, h);
    measure_pair<PERM, ADD_U32>(n_cu, d, h);
    measure_pair<ADD_U32, SALU>(n_cu, d, h);
    // wall-clock cross-check of one case: all CUs, 4 waves per SIMD of v_pk_add_i16
    {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_one<PK_ADD>, dim3(n_cu), dim3(1024), 0, 0, d, ITERS);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)n_cu * 16 * ITERS * PER_ITER;
        printf("wall clock: %.3f ms for %.3g wave instructions on %d SIMDs = %.3g wave-instr/s/SIMD = %.2f cycles at %.2f GHz\n", ms, instr, n_cu * 4,
               instr / (n_cu * 4) / (ms * 1e-3), (ms * 1e-3) * (prop.clockRate * 1e3) / (instr / (n_cu * 4)), prop.clockRate * 1e-6);
    }
    hipFree(d);
    return 0;
}
