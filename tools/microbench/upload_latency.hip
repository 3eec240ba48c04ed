// One frame's camera rows (238 rows x 3840 B = 914 KB at 1280x720) from ORDINARY host memory to the device, and a dependent kernel
// behind them: microseconds from the call to the kernel's end, per way of getting the bytes there.
//   engine2d / engine1d   hipMemcpy2DAsync / hipMemcpyAsync from pageable memory, then the kernel (what lt_upload_frame_rows_enqueue does)
//   bar T                 T host threads write the rows straight into device memory through the PCIe aperture (needs a large BAR;
//                         the mode runs in a process of its own because a box without one answers with SIGSEGV)
//   staged T              T host threads copy the rows into page-locked memory, a copy kernel reads them over the bus
//   pinned                the rows lie in page-locked memory already: the copy kernel alone
// hipcc -O3 --offload-arch=gfx950 upload_latency.hip -o /tmp/upload_latency -lpthread && /tmp/upload_latency <mode> [threads] [bytes]
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <algorithm>
#include <immintrin.h>
typedef unsigned int vec4u __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy(vec4u* __restrict__ dst, const vec4u* __restrict__ src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_sum(const unsigned* __restrict__ src, size_t n, unsigned long long* out) {
    unsigned long long s = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += src[i];
    atomicAdd(out, s);
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Pool {                                  // T - 1 spinning helpers + the caller: piece t of a copy each
    int T; std::vector<std::thread> th; std::atomic<int> gen{0}, done{0}; std::atomic<bool> stop{false};
    char* dst = nullptr; const char* src = nullptr; size_t bytes = 0; bool stream_stores = false;
    void piece(int t) {
        const size_t per = ((bytes / T) + 63) & ~(size_t)63, lo = std::min(bytes, per * t), hi = std::min(bytes, lo + per);
        if (hi <= lo) return;
        if (stream_stores) {
            const __m256i* s = (const __m256i*)(src + lo); __m256i* d = (__m256i*)(dst + lo);
            for (size_t i = 0; i < (hi - lo) / 32; ++i) _mm256_stream_si256(d + i, _mm256_loadu_si256(s + i));
            _mm_sfence();
        } else { std::memcpy(dst + lo, src + lo, hi - lo); _mm_sfence(); }
    }
    explicit Pool(int T_) : T(T_) {
        for (int t = 1; t < T; ++t) th.emplace_back([this, t] {
            int seen = 0;
            while (!stop.load(std::memory_order_relaxed)) {
                if (gen.load(std::memory_order_acquire) == seen) { _mm_pause(); continue; }
                ++seen; piece(t); done.fetch_add(1, std::memory_order_release);
            }
        });
    }
    void copy(void* d, const void* s, size_t n, bool nt) {
        dst = (char*)d; src = (const char*)s; bytes = n; stream_stores = nt; done.store(0);
        gen.fetch_add(1, std::memory_order_release);
        piece(0);
        while (done.load(std::memory_order_acquire) < T - 1) _mm_pause();
    }
    ~Pool() { stop = true; for (auto& t : th) t.join(); }
};

int main(int argc, char** argv) {
    const char* mode = argc > 1 ? argv[1] : "engine2d";
    const int T = argc > 2 ? std::atoi(argv[2]) : 1;
    const size_t rows = 238, rb = 3840, bytes = argc > 3 ? (size_t)std::atol(argv[3]) : rows * rb;
    int large_bar = -1;
    (void)hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
    // COLD=1: 256 frames 2.76 MB apart (a window of a video: nothing of a frame is in any cache when its turn comes); else four
    const bool cold = std::getenv("COLD") != nullptr;
    const size_t nsrc = cold ? 256 : 4, stride = cold ? std::max<size_t>(bytes * 3, 2764800) : bytes;
    char* h_page = (char*)std::malloc(stride * nsrc);
    for (size_t i = 0; i < stride * nsrc; ++i) h_page[i] = (char)(i * 2654435761u >> 13);
    void *h_pin, *d_rows, *h_pin_dev; unsigned long long *d_sum, *h_sum;
    CK(hipHostMalloc(&h_pin, bytes, hipHostMallocDefault));
    CK(hipHostGetDevicePointer(&h_pin_dev, h_pin, 0));
    CK(hipMalloc(&d_rows, bytes));
    CK(hipMalloc(&d_sum, 8));
    CK(hipHostMalloc((void**)&h_sum, 8, hipHostMallocDefault));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    Pool pool(std::max(T, 1));
    const bool nt = std::getenv("NT") != nullptr;
    std::vector<double> us; int bad = 0;
    std::vector<unsigned long long> wants(nsrc, 0);          // (summed up front: the loop must not read a frame just before its copy)
    for (size_t k = 0; k < nsrc; ++k)
        for (size_t i = 0; i < bytes / 4; ++i) wants[k] += ((const unsigned*)(h_page + k * stride))[i];
    for (int r = 0; r < 60; ++r) {
        const char* src = h_page + (size_t)((r * 37) % nsrc) * stride;
        unsigned long long want = wants[(r * 37) % nsrc];
        CK(hipMemsetAsync(d_sum, 0, 8, st)); CK(hipStreamSynchronize(st));
        const double t0 = now();
        if (!std::strcmp(mode, "engine2d")) CK(hipMemcpy2DAsync(d_rows, rb, src, rb, rb, bytes / rb, hipMemcpyHostToDevice, st));
        else if (!std::strcmp(mode, "engine2d1")) CK(hipMemcpy2DAsync(d_rows, stride, src, stride, bytes, 1, hipMemcpyHostToDevice, st));   // the library's form: one "row" of all the bytes
        else if (!std::strcmp(mode, "engine1d")) CK(hipMemcpyAsync(d_rows, src, bytes, hipMemcpyHostToDevice, st));
        else if (!std::strcmp(mode, "bar")) pool.copy(d_rows, src, bytes, nt);
        else if (!std::strcmp(mode, "staged")) { pool.copy(h_pin, src, bytes, false); hipLaunchKernelGGL(k_copy, dim3(64), dim3(256), 0, st, (vec4u*)d_rows, (const vec4u*)h_pin_dev, bytes >> 4); }
        else if (!std::strcmp(mode, "pinned")) { if (r < 4) std::memcpy(h_pin, src, bytes); else src = (const char*)h_pin; hipLaunchKernelGGL(k_copy, dim3(64), dim3(256), 0, st, (vec4u*)d_rows, (const vec4u*)h_pin_dev, bytes >> 4); }
        else { std::printf("unknown mode %s\n", mode); return 2; }
        const double t1 = now();
        hipLaunchKernelGGL(k_sum, dim3(64), dim3(256), 0, st, (const unsigned*)d_rows, bytes >> 2, d_sum);
        CK(hipMemcpyAsync(h_sum, d_sum, 8, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        const double t2 = now();
        if (!std::strcmp(mode, "pinned") && r >= 4) { want = 0; for (size_t i = 0; i < bytes / 4; ++i) want += ((const unsigned*)h_pin)[i]; }
        if (*h_sum != want) ++bad;
        if (r >= 10) us.push_back((t2 - t0) * 1e6);
        if (r == 59) std::printf("{\"mode\": \"%s\", \"cold\": %d, \"threads\": %d, \"nt\": %d, \"bytes\": %zu, \"large_bar\": %d, \"last_call_us\": %.1f, ", mode, (int)cold, T, (int)nt, bytes, large_bar, (t1 - t0) * 1e6);
    }
    std::sort(us.begin(), us.end());
    std::printf("\"us_median\": %.1f, \"us_p10\": %.1f, \"us_p90\": %.1f, \"wrong_sums\": %d}\n", us[us.size() / 2], us[us.size() / 10], us[us.size() * 9 / 10], bad);
    return bad ? 3 : 0;
}
