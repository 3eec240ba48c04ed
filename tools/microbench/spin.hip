// One workgroup that does nothing but s_sleep for a given time, on a stream of its own: does the mere presence of a second
// active compute queue slow the kernels of the first?  (tools/stream_interference2.py loads this through ctypes.)
//   hipcc -O3 --offload-arch=gfx950 -shared -fPIC tools/microbench/spin.hip -o /tmp/libspin.so
#include <hip/hip_runtime.h>
static hipStream_t g_stream = nullptr;
static int* g_buf = nullptr;
// touch_lds: 0 sleep only; 1 + LDS traffic; 2 busy f64 arithmetic (no sleep); 3 busy global loads + stores over a 1 MB buffer;
// 4 busy barriers; 5 busy 32-bit integer arithmetic
__global__ void k_spin(long long cycles, int touch_lds, int* out) {
    __shared__ int lds[64];
    const long long t0 = __builtin_readcyclecounter();
    int acc = 0;
    double d = 1.0 + threadIdx.x;
    while (__builtin_readcyclecounter() - t0 < cycles) {
        if (touch_lds <= 1) __builtin_amdgcn_s_sleep(32);
        if (touch_lds == 1) { lds[threadIdx.x & 63] = acc; acc += lds[(threadIdx.x + 1) & 63]; }
        if (touch_lds == 2) { for (int i = 0; i < 64; ++i) d = d * 1.0000001 + 1e-9 / d; }
        if (touch_lds == 6) { for (int i = 0; i < 8; ++i) d = d * 1.0000001 + 1e-9 / d; for (int i = 0; i < 40; ++i) __builtin_amdgcn_s_sleep(32); }   // f64 at a low duty
        if (touch_lds == 7) { for (int i = 0; i < 64; ++i) d = d * 1.0000001 + 1e-9; }                                                           // f64 mul/add only, no division
        if (touch_lds == 8 && threadIdx.x == 0) { for (int i = 0; i < 64; ++i) d = d * 1.0000001 + 1e-9 / d; }                                  // one lane only
        if (touch_lds == 9) { float f = (float)d; for (int i = 0; i < 64; ++i) f = f * 1.0000001f + 1e-9f / f; d = f; }                          // the same in f32
        if (touch_lds == 3 && out) { for (int i = 0; i < 16; ++i) { const int j = (threadIdx.x * 16 + i * 8192 + acc) & 0x3ffff; acc += out[j]; out[(j + 64) & 0x3ffff] = acc; } }
        if (touch_lds == 4) { for (int i = 0; i < 16; ++i) __syncthreads(); }
        if (touch_lds == 5) { for (int i = 0; i < 256; ++i) acc = acc * 1664525 + 1013904223; }
    }
    if (out && threadIdx.x == 0) out[0x40000] = acc + (int)d;
}
extern "C" int spin_start_lds(double ms, int threads, int blocks, int touch_lds, int dyn_lds_bytes) {
    if (!g_stream && hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return -1;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_spin), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    if (!g_buf && hipMalloc(reinterpret_cast<void**>(&g_buf), (0x40000 + 64) * sizeof(int)) != hipSuccess) return -3;
    hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(threads), dyn_lds_bytes, g_stream, (long long)(ms * 1e-3 * 100e6), touch_lds, g_buf);
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int spin_start(double ms, int threads, int blocks, int touch_lds) {
    if (!g_stream && hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking) != hipSuccess) return -1;
    hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(threads), 0, g_stream, (long long)(ms * 1e-3 * 100e6), touch_lds, (int*)nullptr);   // s_memtime / cycle counter at 100 MHz
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
extern "C" int spin_sync() { return g_stream ? (int)hipStreamSynchronize(g_stream) : 0; }
