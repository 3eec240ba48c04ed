// How long after a kernel has ended does the host know?  One short kernel per iteration (it busy-waits `work_us`, then stores a
// ticket into page-locked host memory); the host either polls that word or calls hipStreamSynchronize / hipEventSynchronize.
//   hipcc -O2 --offload-arch=gfx950 tools/microbench/sync_latency.hip -o /tmp/sync_latency && /tmp/sync_latency [work_us] [iterations]
// One JSON line: microseconds from the launch call to "the host has the result", median and 90th percentile, per way of
// waiting; `launch_call` is the hipLaunchKernel call itself.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

__global__ void work(volatile unsigned* flag, unsigned ticket, long long cycles) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) {}
    __threadfence_system();
    *flag = ticket;
}

int main(int argc, char** argv) {
    const double work_us = argc > 1 ? atof(argv[1]) : 20.0;
    const int iters = argc > 2 ? atoi(argv[2]) : 2000;
    unsigned* flag = nullptr;
    hipHostMalloc(reinterpret_cast<void**>(&flag), 64, hipHostMallocDefault);
    unsigned* dflag = nullptr;
    hipHostGetDevicePointer(reinterpret_cast<void**>(&dflag), flag, 0);
    hipStream_t st;
    hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    hipEvent_t ev;
    hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    int rate_khz = 100000;
    hipDeviceGetAttribute(&rate_khz, hipDeviceAttributeWallClockRate, 0);
    const long long cycles = (long long)(work_us * rate_khz / 1000.0);
    unsigned ticket = 0;
    auto stats = [](std::vector<double>& v, double& med, double& p90) {
        std::sort(v.begin(), v.end());
        med = v[v.size() / 2];
        p90 = v[v.size() * 9 / 10];
    };
    std::vector<double> launch, poll, sync, evs, syncspin;
    for (int mode = 0; mode < 3; ++mode)
        for (int i = 0; i < iters + 50; ++i) {
            *flag = 0;
            ++ticket;
            const double t0 = now_us();
            hipLaunchKernelGGL(work, dim3(1), dim3(64), 0, st, dflag, ticket, cycles);
            const double t1 = now_us();
            if (mode == 0) {
                while (*(volatile unsigned*)flag != ticket) __builtin_ia32_pause();
            } else if (mode == 1) {
                hipStreamSynchronize(st);
            } else {
                hipEventRecord(ev, st);
                hipEventSynchronize(ev);
            }
            const double t2 = now_us();
            if (i < 50) continue;
            if (mode == 0) { launch.push_back(t1 - t0); poll.push_back(t2 - t0); }
            else if (mode == 1) sync.push_back(t2 - t0);
            else evs.push_back(t2 - t0);
            if (mode == 0) hipStreamSynchronize(st);
        }
    double a, b, c, d, e, f, g, h;
    stats(launch, a, b);
    stats(poll, c, d);
    stats(sync, e, f);
    stats(evs, g, h);
    printf("{\"kernel_busy_us\": %.1f, \"launch_call\": [%.1f, %.1f], \"poll_pinned_word\": [%.1f, %.1f], \"hipStreamSynchronize\": [%.1f, %.1f], "
           "\"hipEventRecord+hipEventSynchronize\": [%.1f, %.1f]}\n", work_us, a, b, c, d, e, f, g, h);
    return 0;
}
