// Report-ready reproducer (reduced from close_hang.hip): two hipStreamDestroy calls a few microseconds apart; the second one does not
// return (thread in ioctl AMDKFD_IOC_WAIT_EVENTS).  ROCm 7.2.0, gfx950 (MI355X): 9 of 9 runs of the long form (NOTES_r05 D.5); a 50 ms
// pause between the destroys, or destroying the CU-masked stream first, avoids it.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/sdh tools/microbench/stream_destroy_hang.hip && timeout 60 /tmp/sdh ; echo rc=$?   (124 = hung)
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void touch(unsigned* p) { p[blockIdx.x * blockDim.x + threadIdx.x] = 1u; }
int main() {
    int v = 0; hipRuntimeGetVersion(&v); std::printf("HIP runtime %d\n", v);
    for (int round = 0; round < 8; ++round) {
        hipStream_t a, b; unsigned mask[8] = {0xfffffffeu, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}; unsigned* p;
        int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
        hipStreamCreateWithPriority(&a, hipStreamNonBlocking, hi);      // a priority stream
        hipExtStreamCreateWithCUMask(&b, 8, mask);                      // a stream kept off CU 0
        hipMalloc(&p, 1 << 22);
        for (int k = 0; k < 40; ++k) { hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, a, p); hipLaunchKernelGGL(touch, dim3(1024), dim3(256), 0, b, p); }
        hipStreamSynchronize(a); hipStreamSynchronize(b);               // both idle
        std::printf("round %d: destroy(priority) ...", round); std::fflush(stdout);
        hipStreamDestroy(a);
        std::printf(" destroy(masked) ..."); std::fflush(stdout);
        hipStreamDestroy(b);                                            // <- waits for ever
        std::printf(" ok\n"); hipFree(p);
    }
    std::printf("no hang\n");
}
