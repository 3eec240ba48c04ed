// What a hipMalloc costs, depending on the history of the memory it is handed (round 5: the "slow mode" of a stream's first
// window).  Build and run on the GPU box:
//   hipcc -O2 --offload-arch=gfx950 tools/microbench/alloc_probe.hip -o /tmp/alloc_probe && /tmp/alloc_probe [GB per block] [blocks]
// Phases, one JSON line each (ms per hipMalloc / hipFree, ms per GB):
//   fresh      the first blocks of the process
//   reuse_now  the same sizes again right after hipFree of the fresh ones (memory this very process has just given back)
//   reuse_2s   ... and once more, two seconds after the frees
//   touched    the blocks are written by a kernel before they are freed (does "dirty" matter?)
// Run it twice in a row: the second process gets memory the first one has used.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void fill(uint4* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = make_uint4(1, 2, 3, 4);
}

int main(int argc, char** argv) {
    const double gb = argc > 1 ? atof(argv[1]) : 1.5;
    const int blocks = argc > 2 ? atoi(argv[2]) : 10;
    const size_t bytes = (size_t)(gb * (1u << 30));
    size_t free_b = 0, total_b = 0;
    hipMemGetInfo(&free_b, &total_b);
    printf("{\"phase\": \"start\", \"free_GB\": %.1f, \"total_GB\": %.1f, \"block_GB\": %.2f, \"blocks\": %d}\n", free_b / 1e9, total_b / 1e9, gb, blocks);
    std::vector<void*> p((size_t)blocks, nullptr);
    auto alloc_all = [&](const char* phase) {
        double worst = 0, sum = 0;
        for (int i = 0; i < blocks; ++i) {
            const double t0 = now_ms();
            if (hipMalloc(&p[i], bytes) != hipSuccess) { printf("{\"phase\": \"%s\", \"error\": \"hipMalloc %d failed\"}\n", phase, i); exit(1); }
            const double dt = now_ms() - t0;
            sum += dt;
            if (dt > worst) worst = dt;
        }
        printf("{\"phase\": \"%s\", \"malloc_ms_total\": %.3f, \"malloc_ms_worst\": %.3f, \"ms_per_GB\": %.3f}\n", phase, sum, worst, sum / (gb * blocks));
        fflush(stdout);
    };
    auto free_all = [&](const char* phase) {
        const double t0 = now_ms();
        for (int i = 0; i < blocks; ++i) hipFree(p[i]);
        printf("{\"phase\": \"%s\", \"free_ms_total\": %.3f}\n", phase, now_ms() - t0);
        fflush(stdout);
    };
    alloc_all("fresh");
    free_all("free_untouched");
    alloc_all("reuse_now_untouched");
    for (int i = 0; i < blocks; ++i) hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, (uint4*)p[i], bytes / 16);
    hipDeviceSynchronize();
    free_all("free_touched");
    alloc_all("reuse_now_touched");
    free_all("free_again");
    std::this_thread::sleep_for(std::chrono::seconds(2));
    alloc_all("reuse_2s");
    for (int i = 0; i < blocks; ++i) hipLaunchKernelGGL(fill, dim3(2048), dim3(256), 0, 0, (uint4*)p[i], bytes / 16);
    hipDeviceSynchronize();
    // leave without freeing: the driver reclaims at exit (what a process that ends does)
    printf("{\"phase\": \"exit_without_free\"}\n");
    return 0;
}
