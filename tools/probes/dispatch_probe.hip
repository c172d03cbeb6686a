// Does a launch with far more workgroups than the GPU holds keep OTHER streams' kernels from starting?
// Stream A: 20,000 workgroups x 256 threads, each spinning ~20 us (the whole launch ~0.2-0.3 ms).  Stream B, started at
// the same time: 8 workgroups spinning ~50 us.  B's completion time tells: ~50 us = dispatched next to A, ~A's duration =
// queued behind A's dispatch.  build: hipcc --offload-arch=gfx950 -O2 -o dispatch_probe dispatch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(long long ticks) { const long long t0 = wall_clock64(); while (wall_clock64() - t0 < ticks) { } }
int main()
{
    hipStream_t a, b, c;
    hipStreamCreateWithFlags(&a, hipStreamNonBlocking); hipStreamCreateWithFlags(&b, hipStreamNonBlocking); hipStreamCreateWithFlags(&c, hipStreamNonBlocking);
    hipEvent_t e0, ea, eb;
    hipEventCreate(&e0); hipEventCreate(&ea); hipEventCreate(&eb);
    for (hipStream_t s : { a, b, c }) { hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, 1); hipStreamSynchronize(s); }
    for (int big : { 2000, 20000, 100000 })
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0, c);
            hipStreamWaitEvent(a, e0, 0); hipStreamWaitEvent(b, e0, 0);
            hipLaunchKernelGGL(spin, dim3(big), dim3(256), 0, a, 2000);      // 20 us per workgroup
            hipLaunchKernelGGL(spin, dim3(8), dim3(256), 0, b, 5000);        // 50 us
            hipEventRecord(ea, a); hipEventRecord(eb, b);
            hipEventSynchronize(ea); hipEventSynchronize(eb);
            float ta = 0, tb = 0;
            hipEventElapsedTime(&ta, e0, ea); hipEventElapsedTime(&tb, e0, eb);
            printf("A: %6d workgroups done after %.3f ms; B (8 workgroups, 50 us) done after %.3f ms\n", big, ta, tb);
        }
    return 0;
}
