// Stand-alone form of NOTES C.8 / D.5: hipStreamDestroy of a stream created with hipExtStreamCreateWithCUMask, issued right after
// multi-GB hipFree calls, does not return (the thread sits in AMDKFD_IOC_WAIT_EVENTS).  The ingredients of lt_destroy in round 4:
//   streams: one plain, one with a CU mask (all CUs but the first), one with the first CU only; kernels have run on all of them and
//   every stream has been synchronised; then hipFree of several GB; then hipStreamDestroy of the masked streams.
//   hipcc -O2 --offload-arch=gfx950 tools/microbench/close_hang.hip -o /tmp/close_hang && timeout 120 /tmp/close_hang [rounds] [GB per block] [blocks] [order]
// order 0: free, then destroy (round 4);  1: destroy, then free (round 5).  Prints one line per step; a hang shows as the last line.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

__global__ void touch(uint4* p, size_t n, int rounds) {
    for (int r = 0; r < rounds; ++r)
        for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(r, 2, 3, 4);
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define STEP(what, expr)                                                         \
    do {                                                                         \
        const double t0 = now();                                                 \
        std::printf("%s ...", what);                                             \
        std::fflush(stdout);                                                     \
        hipError_t e = (expr);                                                   \
        std::printf(" %s in %.3f ms\n", hipGetErrorName(e), (now() - t0) * 1e3); \
        std::fflush(stdout);                                                     \
    } while (0)

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 6;
    const double gb = argc > 2 ? atof(argv[2]) : 1.4;
    const int blocks = argc > 3 ? atoi(argv[3]) : 16;
    const int order = argc > 4 ? atoi(argv[4]) : 0;
    // variations (bit flags): 1 = no priority stream at all; 2 = the masked streams are destroyed BEFORE the priority stream;
    // 4 = hipDeviceSynchronize() in front of the destroys; 8 = the plain stream is destroyed first; 16 = 50 ms pause in front
    const int mode = argc > 5 ? atoi(argv[5]) : 0;
    const size_t bytes = (size_t)(gb * (1u << 30));
    for (int r = 0; r < rounds; ++r) {
        std::printf("== round %d\n", r);
        hipStream_t plain, masked, one_cu, prio;
        uint32_t m_all[8], m_one[8] = {1u, 0, 0, 0, 0, 0, 0, 0};
        for (auto& w : m_all) w = 0xffffffffu;
        m_all[0] &= ~1u;
        STEP("hipStreamCreate(plain)", hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
        STEP("hipExtStreamCreateWithCUMask(all but CU 0)", hipExtStreamCreateWithCUMask(&masked, 8, m_all));
        STEP("hipExtStreamCreateWithCUMask(CU 0)", hipExtStreamCreateWithCUMask(&one_cu, 8, m_one));
        int lo = 0, hi = 0;
        hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (mode & 1) STEP("hipStreamCreate(instead of the priority stream)", hipStreamCreateWithFlags(&prio, hipStreamNonBlocking));
        else STEP("hipStreamCreateWithPriority", hipStreamCreateWithPriority(&prio, hipStreamNonBlocking, hi));
        std::vector<void*> p((size_t)blocks, nullptr);
        for (int i = 0; i < blocks; ++i)
            if (hipMalloc(&p[i], bytes) != hipSuccess) { std::printf("hipMalloc %d failed\n", i); return 1; }
        void* host = nullptr;
        hipHostMalloc(&host, 64 << 20, hipHostMallocDefault);
        hipStream_t st[4] = {plain, masked, one_cu, prio};
        for (int k = 0; k < 40; ++k)
            for (int i = 0; i < blocks; ++i) hipLaunchKernelGGL(touch, dim3(i % 4 == 2 ? 8 : 1024), dim3(256), 0, st[i % 4], (uint4*)p[i], (size_t)(8 << 20), 2);
        hipMemcpyAsync(host, p[0], 64 << 20, hipMemcpyDeviceToHost, prio);
        for (auto s : st) hipStreamSynchronize(s);
        std::printf("all streams idle\n");
        auto free_all = [&] {
            for (int i = 0; i < blocks; ++i) {
                char what[48];
                std::snprintf(what, sizeof what, "hipFree(block %d, %.1f GB)", i, gb);
                STEP(what, hipFree(p[i]));
            }
        };
        auto destroy_all = [&] {
            if (mode & 4) STEP("hipDeviceSynchronize", hipDeviceSynchronize());
            if (mode & 16) std::this_thread::sleep_for(std::chrono::milliseconds(50));
            if (mode & 8) STEP("hipStreamDestroy(plain)", hipStreamDestroy(plain));
            if (!(mode & 2)) STEP("hipStreamDestroy(prio)", hipStreamDestroy(prio));
            STEP("hipStreamDestroy(masked: all but CU 0)", hipStreamDestroy(masked));
            STEP("hipStreamDestroy(masked: CU 0)", hipStreamDestroy(one_cu));
            if (mode & 2) STEP("hipStreamDestroy(prio)", hipStreamDestroy(prio));
            if (!(mode & 8)) STEP("hipStreamDestroy(plain)", hipStreamDestroy(plain));
        };
        if (order == 0) { free_all(); destroy_all(); } else { destroy_all(); free_all(); }
        STEP("hipHostFree", hipHostFree(host));
    }
    std::printf("done: no hang\n");
    return 0;
}
