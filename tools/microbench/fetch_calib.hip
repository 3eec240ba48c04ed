// FETCH_SIZE / WRITE_SIZE calibration for gfx950 (rocprofv3 --pmc): every kernel below reads or writes a buffer of
// known size (1 GiB, four times the 256 MiB Infinity Cache) exactly once with one of the access shapes the mask-stage
// kernels use.  tools/microbench/run.sh profiles this program in separate FETCH_SIZE and WRITE_SIZE passes and
// tools/microbench/calib_summary.py divides the known byte counts by the counter values:
//     factor(pattern) = true bytes / reported bytes           (1.0 = the counter is right for that shape)
//
//   build: hipcc -O2 --offload-arch=gfx950 fetch_calib.hip -o fetch_calib
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr size_t BYTES = 1ull << 30;

// ---- reads: lane i of the grid-stride loop touches element i; one pass over the whole buffer ----
template <class T>
__global__ __launch_bounds__(256) void k_read(const T* __restrict__ p, size_t n, unsigned* sink, unsigned magic) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) {
        T v = p[i];
        const unsigned* w = reinterpret_cast<const unsigned*>(&v);
        if constexpr (sizeof(T) >= 4) {
            for (unsigned k = 0; k < sizeof(T) / 4; ++k) acc ^= w[k];
        } else {
            acc ^= (unsigned)v;
        }
    }
    if (acc == magic) sink[0] = acc;     // magic is a run-time value: the loads cannot be proven dead
}
// 8-byte loads at 3-byte pitch (the RGB bilinear tap pair of the undistortion: unaligned, overlapping)
extern "C" __global__ __launch_bounds__(256) void k_read_tap8(const uint8_t* __restrict__ p, size_t npx, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < npx; i += (size_t)gridDim.x * 256ull) {
        struct __attribute__((packed, aligned(1))) U { uint64_t v; };
        const uint64_t v = reinterpret_cast<const U*>(p + 3 * i)->v;
        acc ^= (unsigned)v ^ (unsigned)(v >> 32);
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// rows of a u8 plane read a dword per lane with a per-wave row change (the band staging of the threshold kernel)
extern "C" __global__ __launch_bounds__(256) void k_read_rows_dword(const uint32_t* __restrict__ p, int w4, int h, unsigned* sink) {
    unsigned acc = 0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int y = blockIdx.x * 4 + wv; y < h; y += gridDim.x * 4)
        for (int x = lane; x < w4; x += 64) acc ^= p[(size_t)y * w4 + x];
    if (acc == 0x12345678u) sink[0] = acc;
}

// ---- writes ----
template <class T>
__global__ __launch_bounds__(256) void k_write(T* __restrict__ p, size_t n, unsigned seed) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256ull) {
        T v;
        unsigned* w = reinterpret_cast<unsigned*>(&v);
        if constexpr (sizeof(T) >= 4) {
            for (unsigned k = 0; k < sizeof(T) / 4; ++k) w[k] = seed + (unsigned)i + k;
        } else {
            v = (T)(seed + i);
        }
        p[i] = v;
    }
}
// one 8-byte word per wave and row (the ballot stores of the bit planes): lane 0 of each wave stores
extern "C" __global__ __launch_bounds__(256) void k_write_lane0_u64(unsigned long long* __restrict__ p, size_t n, unsigned seed) {
    const size_t wave = (blockIdx.x * 256ull + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * 256ull) >> 6;
    for (size_t i = wave; i < n; i += nw)
        if ((threadIdx.x & 63) == 0) p[i] = seed + i;
}

struct u128 { unsigned a, b, c, d; };

int main() {
    CHECK(hipSetDevice(0));
    void *buf;
    unsigned* sink;
    CHECK(hipMalloc(&buf, BYTES + 64));
    CHECK(hipMalloc(&sink, 64));
    CHECK(hipMemset(buf, 1, BYTES + 64));
    CHECK(hipDeviceSynchronize());
    const int grid = 256 * 16;
    const unsigned magic = (unsigned)std::rand() | 0x100u;   // never equals an XOR of 0x01 bytes, but the compiler cannot know
    // every launch is followed by a sync so that the dispatches appear in this order in the counter CSV
    hipLaunchKernelGGL(k_read<uint8_t>, dim3(grid), dim3(256), 0, 0, (const uint8_t*)buf, BYTES, sink, magic);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_read<uint16_t>, dim3(grid), dim3(256), 0, 0, (const uint16_t*)buf, BYTES / 2, sink, magic);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_read<uint32_t>, dim3(grid), dim3(256), 0, 0, (const uint32_t*)buf, BYTES / 4, sink, magic);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_read<uint64_t>, dim3(grid), dim3(256), 0, 0, (const uint64_t*)buf, BYTES / 8, sink, magic);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_read<u128>, dim3(grid), dim3(256), 0, 0, (const u128*)buf, BYTES / 16, sink, magic);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_read_tap8, dim3(grid), dim3(256), 0, 0, (const uint8_t*)buf, BYTES / 3, sink);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_read_rows_dword, dim3(grid), dim3(256), 0, 0, (const uint32_t*)buf, 270, (int)(BYTES / 1080), sink);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_write<uint8_t>, dim3(grid), dim3(256), 0, 0, (uint8_t*)buf, BYTES, 3u);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_write<uint32_t>, dim3(grid), dim3(256), 0, 0, (uint32_t*)buf, BYTES / 4, 3u);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_write<uint64_t>, dim3(grid), dim3(256), 0, 0, (uint64_t*)buf, BYTES / 8, 3u);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_write<u128>, dim3(grid), dim3(256), 0, 0, (u128*)buf, BYTES / 16, 3u);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_write_lane0_u64, dim3(grid), dim3(256), 0, 0, (unsigned long long*)buf, BYTES / 8 / 8, 3u);
    CHECK(hipDeviceSynchronize());
    std::printf("fetch_calib: done, %zu bytes per pass\n", BYTES);
    return 0;
}
