// hipGraphLaunch against the same chain as individual launches: N dependent kernels of ~10 us each on one
// stream -- the shape of process()'s one-frame mask chain (6 launches since round 6, 11 before) -- issued (a) one hipLaunchKernel
// per kernel, (b) as ONE hipGraphLaunch of the captured chain.  Per variant: host time to issue a frame's chain, and the device
// time from the first kernel's start to the last one's end (hipEvents), median of 300 frames with the host waiting per frame.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/graph_launch tools/microbench/graph_launch.hip && /tmp/graph_launch [kernels=6] [spin_us=10]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_spin(unsigned long long ticks, unsigned* sink) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(sink, 1u);
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
int main(int argc, char** argv) {
    const int n = argc > 1 ? std::atoi(argv[1]) : 6;
    const double spin_us = argc > 2 ? std::atof(argv[2]) : 10.0;
    const unsigned long long ticks = (unsigned long long)(spin_us * 100.0);      // wall_clock64: 100 MHz
    hipStream_t s;
    CK(hipStreamCreate(&s));
    unsigned* sink;
    CK(hipMalloc(&sink, 4));
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto chain = [&]() { for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s, ticks, sink); };
    // (b): the chain captured once
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    chain();
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int variant = 0; variant < 2; ++variant) {
        std::vector<double> host, dev, wall;
        for (int f = 0; f < 320; ++f) {
            const double t0 = now();
            CK(hipEventRecord(a, s));
            if (variant == 0) chain(); else CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(b, s));
            const double t1 = now();
            CK(hipStreamSynchronize(s));
            const double t2 = now();
            float ms = 0;
            CK(hipEventElapsedTime(&ms, a, b));
            if (f >= 20) { host.push_back((t1 - t0) * 1e6); dev.push_back(ms * 1e3); wall.push_back((t2 - t0) * 1e6); }
        }
        std::printf("%-28s kernels %d x %.0f us: host issue %.1f us, device first-start to last-end %.1f us (ideal %.0f), call to idle %.1f us\n",
                    variant == 0 ? "one launch per kernel" : "one hipGraphLaunch", n, spin_us, med(host), med(dev), n * spin_us, med(wall));
    }
    return 0;
}
