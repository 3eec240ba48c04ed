// Device-memory cache, page-locked host memory and the host copy threads of liblane_tracker_amd.so
// (see lt_ctx.h; public entry points: lt_device_cache_trim, lt_host_alloc / lt_host_free, lt_host_copy_*).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <ctime>
#include <pthread.h>
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

bool trace_on() {
    static const bool on = [] { const char* e = std::getenv("LT_TRACE_START"); return e && e[0] && e[0] != '0'; }();
    return on;
}
double trace_now() {
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}
void trace_line(const char* what, double t0, size_t bytes) {
    const double t1 = trace_now();
    if (bytes) std::fprintf(stderr, "lt_start %.6f %s %.3f %zu\n", t0, what, (t1 - t0) * 1e3, bytes);
    else std::fprintf(stderr, "lt_start %.6f %s %.3f\n", t0, what, (t1 - t0) * 1e3);
}

// Device memory goes through a small cache instead of straight back to the driver.  Memory handed back with hipFree is wiped by
// the kernel driver in the background, on an SDMA engine -- and while that runs, the copy engine's device-to-host copies of
// THIS process drop from 50-56 to 28-30 GB/s (tools/copy_engine_probe.py: one lone 350 MB download takes 12.7 ms instead of
// 6.3 for the first third of a second after a 5 GB context is destroyed; a context growing twice -- freeing its 256- and
// 768-slot buffers -- does the same to the annotated stream that follows: 9.3 k instead of 15 k frames/s; uploads are not
// affected).  That is what rounds 2-3 described as "two states of the copy engine".  So freed blocks are kept, per device
// and exact size, and handed out again (a tracker closed and another of the same shape opened, a context growing back to a
// size it had).
//
// How much is kept (round 5): at most the high-water mark of what the process's live contexts have held at once, and at most
// 16 GB -- LT_DEVICE_CACHE_GB=<n> sets another limit, 0 turns the cache off.  (Round 4 kept up to half of the device memory:
// hostile to anything else in the process or on the GPU.)  Over the limit the blocks that have waited longest go back to the
// driver first; lt_device_cache_trim(keep) returns everything beyond `keep` at a moment of the caller's choosing, and a failed
// hipMalloc empties the cache and tries once more.
struct DevCache {
    struct Free { void* p; unsigned long long seq; };
    std::mutex m;
    std::multimap<std::pair<int, size_t>, Free> blocks;        // (device, bytes) -> free block
    std::map<void*, std::pair<int, size_t>> live;              // blocks handed out: their device and size
    size_t kept = 0, live_bytes = 0, high_water = 0;
    unsigned long long seq = 0;
    long long env_cap = -2;                                    // bytes from LT_DEVICE_CACHE_GB; -1: not set; -2: not read yet
    long long cap() {                                          // (under m)
        if (env_cap == -2) {
            const char* e = std::getenv("LT_DEVICE_CACHE_GB");
            env_cap = e ? (long long)(std::atof(e) * 1e9) : -1;
        }
        if (env_cap >= 0) return env_cap;
        return (long long)std::min<size_t>(high_water, (size_t)16 << 30);
    }
};
DevCache& dev_cache() { static DevCache* c = new DevCache; return *c; }   // (never destroyed: no order problems at exit)

// hipFree of blocks the cache gives up, each on its own device
static void release_blocks(const std::vector<std::pair<int, void*>>& out) {
    if (out.empty()) return;
    static const bool trace = std::getenv("LT_TRACE_DESTROY") != nullptr;
    int cur = 0;
    (void)hipGetDevice(&cur);
    int on = cur;
    for (const auto& q : out) {
        if (q.first != on) { (void)hipSetDevice(q.first); on = q.first; }
        if (trace) { std::fprintf(stderr, "device cache: hipFree(%p)\n", q.second); std::fflush(stderr); }
        (void)hipFree(q.second);
        if (trace) { std::fprintf(stderr, "device cache: hipFree returned\n"); std::fflush(stderr); }
    }
    if (on != cur) (void)hipSetDevice(cur);
}

void* cached_alloc(size_t bytes) {
    DevCache& dc = dev_cache();
    int dev = 0;
    (void)hipGetDevice(&dev);
    auto handed_out = [&](void* p) {                          // (under m)
        dc.live[p] = {dev, bytes};
        dc.live_bytes += bytes;
        dc.high_water = std::max(dc.high_water, dc.live_bytes);
    };
    {
        std::lock_guard<std::mutex> g(dc.m);
        auto it = dc.blocks.find({dev, bytes});
        if (it != dc.blocks.end()) {
            void* p = it->second.p;
            dc.blocks.erase(it);
            dc.kept -= bytes;
            handed_out(p);
            if (trace_on() && bytes >= (1u << 20)) trace_line("device_cache_hit", trace_now(), bytes);
            return p;
        }
    }
    void* p = nullptr;
    TraceScope ts_(bytes >= (1u << 20) ? "hipMalloc" : "hipMalloc_small", bytes);
    if (hipMalloc(&p, bytes) != hipSuccess) {                  // make room: everything kept goes back, then once more
        (void)hipGetLastError();
        (void)lt_device_cache_trim(0);
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    std::lock_guard<std::mutex> g(dc.m);
    handed_out(p);
    return p;
}
void cached_free(void* p) {
    DevCache& dc = dev_cache();
    std::pair<int, size_t> key;
    {
        std::lock_guard<std::mutex> g(dc.m);
        auto it = dc.live.find(p);
        if (it == dc.live.end()) { key = {-1, 0}; }
        else key = it->second;
    }
    if (key.first < 0) { (void)hipFree(p); return; }          // not ours (never happens: every dev_free pairs a dev_alloc)
    // hipFree waits for the device before it releases anything, and callers have always relied on that (a block freed while
    // another of the context's streams still works on it); a cached block can be handed out again at once, so the same wait
    // happens here -- for the device the BLOCK lives on, which need not be the calling thread's current one.
    int cur = 0;
    (void)hipGetDevice(&cur);
    if (cur != key.first) (void)hipSetDevice(key.first);
    (void)hipDeviceSynchronize();
    if (cur != key.first) (void)hipSetDevice(cur);
    std::vector<std::pair<int, void*>> out;
    {
        std::lock_guard<std::mutex> g(dc.m);
        dc.live.erase(p);
        dc.live_bytes -= key.second;
        const long long cap = dc.cap();
        if ((long long)key.second > cap) out.push_back({key.first, p});
        else {
            dc.blocks.insert({key, DevCache::Free{p, dc.seq++}});
            dc.kept += key.second;
            while ((long long)dc.kept > cap && !dc.blocks.empty()) {   // over the limit: the block that has waited longest goes
                auto old = dc.blocks.begin();
                for (auto j = dc.blocks.begin(); j != dc.blocks.end(); ++j)
                    if (j->second.seq < old->second.seq) old = j;
                dc.kept -= old->first.second;
                out.push_back({old->first.first, old->second.p});
                dc.blocks.erase(old);
            }
        }
    }
    release_blocks(out);
}

void cache_take(size_t keep_bytes, std::vector<std::pair<int, void*>>& out) {
    DevCache& dc = dev_cache();
    std::lock_guard<std::mutex> g(dc.m);
    while (dc.kept > keep_bytes && !dc.blocks.empty()) {
        auto old = dc.blocks.begin();
        for (auto j = dc.blocks.begin(); j != dc.blocks.end(); ++j)
            if (j->second.seq < old->second.seq) old = j;
        dc.kept -= old->first.second;
        out.push_back({old->first.first, old->second.p});
        dc.blocks.erase(old);
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
    TraceScope ts_("hipHostMalloc", bytes);
    if (hipHostMalloc(out, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        *out = nullptr;
        return fail(LT_ERR_HIP, "hipHostMalloc(%zu) failed", bytes);
    }
    return LT_OK;
}

int lt_device_cache_trim(size_t keep_bytes) {
    std::vector<std::pair<int, void*>> out;
    cache_take(keep_bytes, out);
    TraceScope ts_("device_cache_trim:hipFree", out.size());
    release_blocks(out);
    return LT_OK;
}

int lt_device_cache_stats(size_t* kept_bytes, size_t* live_bytes, size_t* limit_bytes, int* kept_blocks) {
    DevCache& dc = dev_cache();
    std::lock_guard<std::mutex> g(dc.m);
    if (kept_bytes) *kept_bytes = dc.kept;
    if (live_bytes) *live_bytes = dc.live_bytes;
    if (limit_bytes) *limit_bytes = (size_t)std::max<long long>(dc.cap(), 0);
    if (kept_blocks) *kept_blocks = (int)dc.blocks.size();
    return LT_OK;
}

int lt_host_free(void* p) {
    if (!p) return LT_OK;
    TraceScope ts_("hipHostFree");
    if (hipHostFree(p) != hipSuccess) {
        (void)hipGetLastError();
        return fail(LT_ERR_HIP, "hipHostFree failed");
    }
    return LT_OK;
}

// ---- host threads for plain copies ------------------------------------------------------------------------------------------
// LaneTracker.process() fills the rows of its output array that no overlay can touch from the caller's frame (1.4 MB at
// 1280x720, 3.2 MB at 1920x1080: 60 / 130 us of memcpy); a window of annotated frames has 0.36 GB of such rows.  The thread
// that feeds the device has launches to issue meanwhile; these have nothing else to do.  A few workers per process
// (LT_COPY_THREADS), started at the first request that can use them, joined by lt_shutdown() or when the library is unloaded.
//
// Completion is per GROUP: a copy belongs to the group it was submitted to and lt_host_copy_wait_group(g) waits for that group's
// copies only -- two trackers on two threads, or two windows of one stream, do not wait for each other's copies (round 4 had
// one pending counter for the process).  Group 0 is the process-wide default group of lt_host_copy_async / lt_host_copy2d_async;
// lt_host_copy_wait() keeps its meaning: every copy requested so far, by anybody, in any group.
//
// fork(): the child of a process that had started workers inherits the object but not the threads; a pthread_atfork handler
// gives the child a fresh copier (the old one is leaked, its threads never existed there), so a child's first request starts
// workers of its own instead of queueing for nobody.
extern "C++" {
namespace {
struct HostCopier {
    struct Job { uint8_t* dst; const uint8_t* src; size_t dpitch, spitch, width, height; int group; };
    std::mutex m;
    std::condition_variable work, done;
    std::deque<Job> q;
    std::map<int, size_t> pending;   // group -> pieces queued or being copied (absent: none)
    size_t pending_all = 0;
    int next_group = 1;
    std::map<int, bool> groups;      // live groups created by lt_host_copy_group_create
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
            --pending_all;
            auto it = pending.find(j.group);
            if (it != pending.end() && --it->second == 0) pending.erase(it);
            done.notify_all();
        }
    }
    int threads() {              // LT_COPY_THREADS (1 .. 16), default 4
        static const int n = [] { const char* e = std::getenv("LT_COPY_THREADS"); const int v = e ? std::atoi(e) : 4; return std::min(std::max(v, 1), 16); }();
        return n;
    }
    bool known(int group) { return group == 0 || groups.count(group) != 0; }      // (under m)
    int submit(const Job& whole) {
        // pieces of whole rows ("rows" of the 2-D copy: frames), a few per worker so that they finish together
        const size_t parts = whole.height <= 1 ? 1 : std::min<size_t>(whole.height, (size_t)threads() * 2);
        {
            std::unique_lock<std::mutex> lk(m);
            if (!known(whole.group)) return -1;
            done.wait(lk, [&] { return !stop; });          // an lt_shutdown() under way: workers start again once it is over
            while ((int)th.size() < (whole.height <= 1 ? 1 : threads())) th.emplace_back([this] { run(); });
            for (size_t k = 0; k < parts; ++k) {
                const size_t r0 = whole.height * k / parts, r1 = whole.height * (k + 1) / parts;
                if (r1 > r0) {
                    q.push_back({whole.dst + r0 * whole.dpitch, whole.src + r0 * whole.spitch, whole.dpitch, whole.spitch, whole.width, r1 - r0, whole.group});
                    ++pending[whole.group];
                    ++pending_all;
                }
            }
        }
        work.notify_all();
        return 0;
    }
    void wait_all() {
        std::unique_lock<std::mutex> lk(m);
        done.wait(lk, [&] { return pending_all == 0; });
    }
    int wait_group(int group) {
        std::unique_lock<std::mutex> lk(m);
        if (!known(group)) return -1;
        done.wait(lk, [&] { return pending.find(group) == pending.end(); });
        return 0;
    }
    void shutdown() {            // finish what is queued, then join the workers
        std::vector<std::thread> mine;
        {
            std::unique_lock<std::mutex> lk(m);
            done.wait(lk, [&] { return pending_all == 0; });
            stop = true;
            mine.swap(th);
        }
        work.notify_all();
        for (auto& t : mine) if (t.joinable()) t.join();
        { std::lock_guard<std::mutex> lk(m); stop = false; }
        done.notify_all();
    }
    ~HostCopier() {
        { std::lock_guard<std::mutex> lk(m); stop = true; q.clear(); }
        work.notify_all();
        for (auto& t : th) if (t.joinable()) t.join();
    }
};
HostCopier* g_copier = nullptr;
struct CopierOwner {             // joins the workers when the library is unloaded
    ~CopierOwner() { HostCopier* h = g_copier; g_copier = nullptr; delete h; }
};
HostCopier& host_copier() {
    static std::once_flag once;
    std::call_once(once, [] {
        g_copier = new HostCopier;
        static CopierOwner owner;
        // the child of a fork gets a copier of its own: the parent's workers do not exist there (the old object, its mutex
        // possibly held by a thread that is gone, is left alone)
        pthread_atfork(nullptr, nullptr, [] { g_copier = new HostCopier; });
    });
    return *g_copier;
}
}  // namespace
}  // extern "C++"

int lt_host_copy_group_create(int* group) {
    if (!group) return fail(LT_ERR_INVALID, "lt_host_copy_group_create: null output");
    HostCopier& h = host_copier();
    std::lock_guard<std::mutex> lk(h.m);
    *group = h.next_group++;
    h.groups[*group] = true;
    return LT_OK;
}

int lt_host_copy_group_destroy(int group) {
    if (group == 0) return LT_OK;
    HostCopier& h = host_copier();
    if (h.wait_group(group)) return fail(LT_ERR_INVALID, "lt_host_copy_group_destroy: unknown group %d", group);
    std::lock_guard<std::mutex> lk(h.m);
    h.groups.erase(group);
    return LT_OK;
}

int lt_host_copy_async_group(int group, void* dst, const void* src, size_t bytes) {
    if (bytes == 0) return LT_OK;
    if (!dst || !src) return fail(LT_ERR_INVALID, "lt_host_copy_async: null pointer");
    if (host_copier().submit({static_cast<uint8_t*>(dst), static_cast<const uint8_t*>(src), bytes, bytes, bytes, 1, group}))
        return fail(LT_ERR_INVALID, "lt_host_copy_async_group: unknown group %d", group);
    return LT_OK;
}

int lt_host_copy2d_async_group(int group, void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width, size_t height) {
    if (width == 0 || height == 0) return LT_OK;
    if (!dst || !src) return fail(LT_ERR_INVALID, "lt_host_copy2d_async: null pointer");
    if (dst_pitch < width || src_pitch < width) return fail(LT_ERR_INVALID, "lt_host_copy2d_async: a pitch below the width");
    if (host_copier().submit({static_cast<uint8_t*>(dst), static_cast<const uint8_t*>(src), dst_pitch, src_pitch, width, height, group}))
        return fail(LT_ERR_INVALID, "lt_host_copy2d_async_group: unknown group %d", group);
    return LT_OK;
}

int lt_host_copy_wait_group(int group) {
    if (host_copier().wait_group(group)) return fail(LT_ERR_INVALID, "lt_host_copy_wait_group: unknown group %d", group);
    return LT_OK;
}

int lt_host_copy_async(void* dst, const void* src, size_t bytes) { return lt_host_copy_async_group(0, dst, src, bytes); }

int lt_host_copy2d_async(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width, size_t height) {
    return lt_host_copy2d_async_group(0, dst, dst_pitch, src, src_pitch, width, height);
}

int lt_host_copy_wait(void) {
    host_copier().wait_all();
    return LT_OK;
}

int lt_shutdown(void) {
    if (g_copier) g_copier->shutdown();
    return LT_OK;
}

}  // extern "C"
