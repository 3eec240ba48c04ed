// Device-memory cache, page-locked host memory and the host copy threads of liblane_tracker_amd.so
// (see lt_ctx.h; public entry points: lt_device_cache_trim, lt_host_alloc / lt_host_free, lt_host_copy_*).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_ext.h>

#include "lt_ctx.h"

using namespace lt;

namespace lt {

// Device memory goes through a small cache instead of straight back to the driver.  Memory handed back with hipFree is wiped by
// the kernel driver in the background, on an SDMA engine -- and while that runs, the copy engine's device-to-host copies of
// THIS process drop from 50-56 to 28-30 GB/s (tools/copy_engine_probe.py: one lone 350 MB download takes 12.7 ms instead of
// 6.3 for the first third of a second after a 5 GB context is destroyed; a context growing twice -- freeing its 256- and
// 768-slot buffers -- does the same to the annotated stream that follows: 9.3 k instead of 15 k frames/s; uploads are not
// affected).  That is what rounds 2-3 described as "two states of the copy engine".  So freed blocks are kept, per device
// and exact size, and handed out again (a tracker closed and another of the same shape opened, a context growing back to a
// size it had); they go back to the driver only when more than LT_DEVICE_CACHE_GB (default: half of the device's memory,
// at most 128 GB) would be kept, largest first, or at lt_device_cache_trim / process exit.
struct DevCache {
    std::mutex m;
    std::multimap<std::pair<int, size_t>, void*> blocks;       // (device, bytes) -> free block
    std::map<void*, std::pair<int, size_t>> live;              // blocks handed out: their device and size
    size_t kept = 0;
    long long cap = -1;                                        // bytes; -1: not decided yet
};
DevCache& dev_cache() { static DevCache* c = new DevCache; return *c; }   // (never destroyed: no order problems at exit)

void* cached_alloc(size_t bytes) {
    DevCache& dc = dev_cache();
    int dev = 0;
    (void)hipGetDevice(&dev);
    {
        std::lock_guard<std::mutex> g(dc.m);
        auto it = dc.blocks.find({dev, bytes});
        if (it != dc.blocks.end()) {
            void* p = it->second;
            dc.blocks.erase(it);
            dc.kept -= bytes;
            dc.live[p] = {dev, bytes};
            return p;
        }
    }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) {                  // make room: everything kept goes back, then once more
        (void)hipGetLastError();
        (void)lt_device_cache_trim(0);
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    std::lock_guard<std::mutex> g(dc.m);
    dc.live[p] = {dev, bytes};
    return p;
}
void cached_free(void* p) {
    // hipFree waits for the device before it releases anything, and callers have always relied on that (a block freed while
    // another of the context's streams still works on it); a cached block can be handed out again at once, so the same wait
    // happens here.
    (void)hipDeviceSynchronize();
    DevCache& dc = dev_cache();
    std::unique_lock<std::mutex> g(dc.m);
    auto it = dc.live.find(p);
    if (it == dc.live.end()) { g.unlock(); (void)hipFree(p); return; }
    const std::pair<int, size_t> key = it->second;
    dc.live.erase(it);
    if (dc.cap < 0) {
        size_t free_b = 0, total_b = 0;
        const char* e = std::getenv("LT_DEVICE_CACHE_GB");
        if (e) dc.cap = (long long)(std::atof(e) * 1e9);
        else dc.cap = hipMemGetInfo(&free_b, &total_b) == hipSuccess ? (long long)std::min<size_t>(total_b / 2, (size_t)128 << 30) : 0;
    }
    if ((long long)key.second > dc.cap) { g.unlock(); (void)hipFree(p); return; }
    dc.blocks.insert({key, p});
    dc.kept += key.second;
    std::vector<void*> out;
    while ((long long)dc.kept > dc.cap && !dc.blocks.empty()) {       // over the cap: the largest blocks go back to the driver
        auto big = dc.blocks.begin();
        for (auto j = dc.blocks.begin(); j != dc.blocks.end(); ++j)
            if (j->first.second > big->first.second) big = j;
        dc.kept -= big->first.second;
        out.push_back(big->second);
        dc.blocks.erase(big);
    }
    g.unlock();
    static const bool trace = std::getenv("LT_TRACE_DESTROY") != nullptr;
    for (void* q : out) {
        if (trace) { std::fprintf(stderr, "device cache over its cap: hipFree(%p)\n", q); std::fflush(stderr); }
        (void)hipFree(q);
    }
}

}  // namespace lt

extern "C" {

// Page-locked host memory for the buffers a caller hands to the upload / download entry points: a copy from or to
// pageable memory is staged by the runtime at a fraction of the PCIe rate (2.8 MB annotated frame: ~0.3 ms against
// ~0.06 ms).  Plain allocation helpers: no context, usable as soon as a device exists.
int lt_host_alloc(size_t bytes, void** out) {
    if (!out || bytes == 0) return fail(LT_ERR_INVALID, "lt_host_alloc: null output or zero size");
    *out = nullptr;
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        return fail(LT_ERR_HIP, "hipHostMalloc(%zu) failed", bytes);
    }
    return LT_OK;
}

int lt_device_cache_trim(size_t keep_bytes) {
    DevCache& dc = dev_cache();
    std::vector<void*> out;
    {
        std::lock_guard<std::mutex> g(dc.m);
        while (dc.kept > keep_bytes && !dc.blocks.empty()) {
            auto it = dc.blocks.begin();
            dc.kept -= it->first.second;
            out.push_back(it->second);
            dc.blocks.erase(it);
        }
    }
    for (void* q : out) (void)hipFree(q);
    return LT_OK;
}

int lt_host_free(void* p) {
    if (!p) return LT_OK;
    if (hipHostFree(p) != hipSuccess) {
        (void)hipGetLastError();
        return fail(LT_ERR_HIP, "hipHostFree failed");
    }
    return LT_OK;
}

// ---- a second host thread for plain copies -------------------------------------------------------------------------------
// LaneTracker.process() fills the rows of its output array that no overlay can touch from the caller's frame (1.4 MB at
// 1280x720, 3.2 MB at 1920x1080: 60 / 130 us of memcpy).  The thread that feeds the device has launches to issue meanwhile;
// these have nothing else to do.  A few workers per process (LT_COPY_THREADS), started at the first request that can use them,
// joined when the library is unloaded.
extern "C++" {
namespace {
struct HostCopier {
    struct Job { uint8_t* dst; const uint8_t* src; size_t dpitch, spitch, width, height; };
    std::mutex m;
    std::condition_variable work, done;
    std::deque<Job> q;
    size_t pending = 0;          // pieces taken and not finished yet
    bool stop = false;
    std::vector<std::thread> th;
    void run() {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            work.wait(lk, [&] { return stop || !q.empty(); });
            if (q.empty()) return;               // stop
            const Job j = q.front();
            q.pop_front();
            lk.unlock();
            if (j.dpitch == j.width && j.spitch == j.width) std::memcpy(j.dst, j.src, j.width * j.height);
            else
                for (size_t r = 0; r < j.height; ++r) std::memcpy(j.dst + r * j.dpitch, j.src + r * j.spitch, j.width);
            lk.lock();
            if (--pending == 0) done.notify_all();
        }
    }
    int threads() {              // LT_COPY_THREADS (1 .. 16), default 4: a window of annotated frames is 0.36 GB of untouched rows
        static const int n = [] { const char* e = std::getenv("LT_COPY_THREADS"); const int v = e ? std::atoi(e) : 4; return std::min(std::max(v, 1), 16); }();
        return n;
    }
    void submit(const Job& whole) {
        // pieces of whole rows ("rows" of the 2-D copy: frames), a few per worker so that they finish together
        const size_t parts = whole.height <= 1 ? 1 : std::min<size_t>(whole.height, (size_t)threads() * 2);
        {
            std::lock_guard<std::mutex> lk(m);
            while ((int)th.size() < (whole.height <= 1 ? 1 : threads())) th.emplace_back([this] { run(); });
            for (size_t k = 0; k < parts; ++k) {
                const size_t r0 = whole.height * k / parts, r1 = whole.height * (k + 1) / parts;
                if (r1 > r0) { q.push_back({whole.dst + r0 * whole.dpitch, whole.src + r0 * whole.spitch, whole.dpitch, whole.spitch, whole.width, r1 - r0}); ++pending; }
            }
        }
        work.notify_all();
    }
    ~HostCopier() {
        { std::lock_guard<std::mutex> lk(m); stop = true; q.clear(); }
        work.notify_all();
        for (auto& t : th) if (t.joinable()) t.join();
    }
};
HostCopier& host_copier() { static HostCopier h; return h; }
}  // namespace
}  // extern "C++"

int lt_host_copy_async(void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return LT_OK;
    if (!dst || !src) return fail(LT_ERR_INVALID, "lt_host_copy_async: null pointer");
    host_copier().submit({static_cast<uint8_t*>(dst), static_cast<const uint8_t*>(src), bytes, bytes, bytes, 1});
    return LT_OK;
}

int lt_host_copy2d_async(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width, size_t height) {
    if (width == 0 || height == 0) return LT_OK;
    if (!dst || !src) return fail(LT_ERR_INVALID, "lt_host_copy2d_async: null pointer");
    if (dst_pitch < width || src_pitch < width) return fail(LT_ERR_INVALID, "lt_host_copy2d_async: a pitch below the width");
    host_copier().submit({static_cast<uint8_t*>(dst), static_cast<const uint8_t*>(src), dst_pitch, src_pitch, width, height});
    return LT_OK;
}

int lt_host_copy_wait(void) {
    HostCopier& h = host_copier();
    std::unique_lock<std::mutex> lk(h.m);
    h.done.wait(lk, [&] { return h.pending == 0; });
    return LT_OK;
}

}  // extern "C"
