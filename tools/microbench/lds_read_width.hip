// LDS read cost by width and alignment on gfx950, in the access pattern of the top-hat window reads: lane l reads
// entries l + i of a per-wave plane of 8-byte entries (i = compile-time offsets), i.e. lane stride 8 bytes.
//   b64          : 8 x ds_read_b64  (8 entries)                       -- what k_morph_runs2 does
//   b128 stride8 : 4 x ds_read_b128 at the same lane stride (half of the lanes start at 8 mod 16)
//   b128 aligned : 4 x ds_read_b128 with lane stride 16 bytes (every lane 16-byte aligned)
// 16 waves per CU (4 per SIMD), every CU busy; prints cycles per wave-instruction and bytes per clock per CU.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/lds_read_width.hip -o /tmp/lds_read_width && /tmp/lds_read_width
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned long long* out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned long long plane[4][512];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = lane; i < 512; i += 64) plane[wv][i] = (unsigned long long)i * 0x9E3779B97F4A7C15ull;
    __syncthreads();
    const unsigned base = (unsigned)(uintptr_t)&plane[wv][0] + (MODE == 2 ? lane * 16 : lane * 8);
    unsigned long long acc = 0;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            unsigned long long a0, a1, a2, a3, a4, a5, a6, a7;
            asm volatile("ds_read_b64 %0, %8 offset:0\n\tds_read_b64 %1, %8 offset:8\n\tds_read_b64 %2, %8 offset:16\n\tds_read_b64 %3, %8 offset:24\n\t"
                         "ds_read_b64 %4, %8 offset:32\n\tds_read_b64 %5, %8 offset:40\n\tds_read_b64 %6, %8 offset:48\n\tds_read_b64 %7, %8 offset:56\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3), "=&v"(a4), "=&v"(a5), "=&v"(a6), "=&v"(a7) : "v"(base) : "memory");
            acc += a0;   // one consumer only: the loop must be bound by the LDS pipe, not by VALU work on the results
        } else {
            typedef unsigned u4 __attribute__((ext_vector_type(4)));
            u4 b0, b1, b2, b3;
            asm volatile("ds_read_b128 %0, %4 offset:0\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\t"
                         "s_waitcnt lgkmcnt(0)"
                         : "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3) : "v"(base) : "memory");
            acc += b0.x;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE>
double run(unsigned long long* d, int blocks, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * 4, iters = 20000;   // 4 blocks x 4 waves = 16 waves per CU
    unsigned long long* d; hipMalloc(&d, (size_t)blocks * 256 * 8);
    const double clk = p.clockRate * 1e3;   // Hz
    const char* names[3] = {"8 x ds_read_b64, lane stride 8 B", "4 x ds_read_b128, lane stride 8 B (unaligned half)", "4 x ds_read_b128, lane stride 16 B (aligned)"};
    double ms[3] = {run<0>(d, blocks, iters), run<1>(d, blocks, iters), run<2>(d, blocks, iters)};
    for (int m = 0; m < 3; ++m) {
        const double insts_per_cu = 16.0 * iters * (m == 0 ? 8 : 4);            // wave instructions per CU
        const double cycles = ms[m] * 1e-3 * clk;
        const double bytes_per_cu = 16.0 * iters * 64 * 64;                     // 64 lanes x 64 bytes per iteration and wave
        printf("%-52s %8.3f ms  %6.2f cycles per wave-instruction per CU  %6.1f B/clk/CU\n", names[m], ms[m], cycles / insts_per_cu, bytes_per_cu / cycles);
    }
    return 0;
}
