// Device -> page-locked host memory by a kernel, beside hipMemcpyAsync uploads on another stream: GB/s per grid size and store kind.
// hipcc -O3 --offload-arch=gfx950 d2h_kernel.hip -o /tmp/d2h_kernel && /tmp/d2h_kernel
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef unsigned int vec4u __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void k_copy(vec4u* __restrict__ dst, const vec4u* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
        else dst[i] = src[i];
    }
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t bytes = (size_t)354 << 20;   // 128 frames of 1280x720x3
    void *h_up, *h_down, *d_a, *d_b;
    const bool pageable = std::getenv("PAGEABLE_UP") != nullptr;    // uploads from ordinary memory (the runtime stages them), as NumPy frames are
    if (pageable) { h_up = std::malloc(bytes); if (!h_up) return 1; std::memset(h_up, 1, bytes); }
    else CK(hipHostMalloc(&h_up, bytes, hipHostMallocDefault));
    CK(hipHostMalloc(&h_down, bytes, hipHostMallocDefault));
    CK(hipMalloc(&d_a, bytes));
    CK(hipMalloc(&d_b, bytes));
    hipStream_t up, down;
    CK(hipStreamCreateWithFlags(&up, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&down, hipStreamNonBlocking));
    void* h_down_dev = nullptr;
    CK(hipHostGetDevicePointer(&h_down_dev, h_down, 0));
    auto run = [&](const char* name, int blocks, int nt, bool with_upload, bool engine) {
        const int reps = 6;
        CK(hipDeviceSynchronize());
        const double t0 = now();
        for (int r = 0; r < reps; ++r) {
            if (with_upload) CK(hipMemcpyAsync(d_a, h_up, bytes, hipMemcpyHostToDevice, up));
            if (engine) CK(hipMemcpyAsync(h_down, d_b, bytes, hipMemcpyDeviceToHost, down));
            else if (nt) hipLaunchKernelGGL(k_copy<true>, dim3(blocks), dim3(256), 0, down, (vec4u*)h_down_dev, (const vec4u*)d_b, bytes >> 4);
            else hipLaunchKernelGGL(k_copy<false>, dim3(blocks), dim3(256), 0, down, (vec4u*)h_down_dev, (const vec4u*)d_b, bytes >> 4);
        }
        CK(hipStreamSynchronize(down));
        const double t_down = now() - t0;
        CK(hipStreamSynchronize(up));
        const double t_all = now() - t0;
        std::printf("%-34s down %6.1f GB/s%s\n", name, reps * bytes / t_down / 1e9, with_upload ? "" : "  (alone)");
        if (with_upload) std::printf("%-34s   both directions done after %.1f ms (%.1f GB/s each way if equal)\n", "", t_all * 1e3, reps * bytes / t_all / 1e9);
        return 0;
    };
    char name[64];
    if (run("engine", 0, 0, false, true)) return 1;
    if (run("engine + upload", 0, 0, true, true)) return 1;
    for (int nt = 0; nt < 2; ++nt)
        for (int blocks : {4, 64, 1024}) {
            std::snprintf(name, sizeof name, "kernel %4d blocks%s", blocks, nt ? " nontemporal" : "");
            if (run(name, blocks, nt, false, false)) return 1;
            std::snprintf(name, sizeof name, "kernel %4d blocks%s + upload", blocks, nt ? " nontemporal" : "");
            if (run(name, blocks, nt, true, false)) return 1;
        }
    return 0;
}
