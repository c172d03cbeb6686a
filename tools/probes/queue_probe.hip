// How many HIP streams really run concurrently on this runtime?  S streams, each a chain of N short kernels
// (8 workgroups spinning for ~100 us); optional fork/join with a private side stream per lane, like the tick launcher.
// build: hipcc --offload-arch=gfx950 -O2 -o queue_probe queue_probe.hip ; run: ./queue_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <chrono>
__global__ void spin(long long cycles, int *sink)
{
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) { }
    if (sink && threadIdx.x == 9999) *sink = 1;
}
// the same with a private array that cannot live in registers: the kernel needs scratch memory
__global__ void spin_scratch(long long cycles, int *sink)
{
    volatile int a[256];
    for (int i = 0; i < 256; i++) a[i] = i * threadIdx.x;
    const long long t0 = wall_clock64();
    int acc = 0;
    while (wall_clock64() - t0 < cycles) acc += a[(acc + threadIdx.x) & 255];
    if (sink && acc == 123456789) *sink = acc;
}
static double run(int S, int N, bool side, int big)
{
    std::vector<hipStream_t> st(S), sd(S);
    std::vector<hipEvent_t> fork(S), join(S);
    for (int i = 0; i < S; i++) {
        hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
        hipStreamCreateWithFlags(&sd[i], hipStreamNonBlocking);
        hipEventCreateWithFlags(&fork[i], hipEventDisableTiming);
        hipEventCreateWithFlags(&join[i], hipEventDisableTiming);
    }
    const long long cyc = 10000;   // wall_clock64 ticks at 100 MHz: 100 us
    hipDeviceSynchronize();
    const auto t0 = std::chrono::steady_clock::now();
    for (int n = 0; n < N; n++)
        for (int i = 0; i < S; i++) {
            if (side) {
                hipEventRecord(fork[i], st[i]);
                hipStreamWaitEvent(sd[i], fork[i], 0);
                hipLaunchKernelGGL(spin, dim3(8), dim3(256), 0, sd[i], cyc, nullptr);
                hipEventRecord(join[i], sd[i]);
            }
            hipLaunchKernelGGL(big == 2 ? spin_scratch : spin, dim3(8), dim3(256), 0, st[i], cyc, nullptr);
            if (side) hipStreamWaitEvent(st[i], join[i], 0);
            hipLaunchKernelGGL(spin, dim3(8), dim3(big == 1 ? 768 : 256), big == 1 ? 96 * 1024 : 0, st[i], cyc, nullptr);
        }
    hipDeviceSynchronize();
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    for (int i = 0; i < S; i++) { hipStreamDestroy(st[i]); hipStreamDestroy(sd[i]); hipEventDestroy(fork[i]); hipEventDestroy(join[i]); }
    return ms;
}
int main()
{
    hipFuncSetAttribute((const void *)spin, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    run(1, 4, false, 0);
    const int N = 50;
    for (int side = 0; side < 2; side++)
        for (int big = 0; big < 3; big++)
            for (int S : {4, 8, 12, 16, 24, 32}) {
                const double ms = run(S, N, side, big);
                printf("side %d big %d streams %2d: %.2f ms (serial would be %.1f, perfectly concurrent %.1f)\n", side, big, S, ms, S * N * 0.2, N * 0.2);
            }
    return 0;
}
