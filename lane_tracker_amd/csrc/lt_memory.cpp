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
#include <immintrin.h>
#include <sched.h>
#include <atomic>
#include <functional>
#include <memory>
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
// How much is kept (round 5; round 6: a quarter more): at most 1.25 x the high-water mark of what the process's live contexts have held at once + 256 MB, and at most
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
    unsigned long long hits = 0, misses = 0, evicted_blocks = 0, evicted_bytes = 0;   // lt_device_cache_counters
    long long env_cap = -2;                                    // bytes from LT_DEVICE_CACHE_GB; -1: not set; -2: not read yet
    long long cap() {                                          // (under m)
        if (env_cap == -2) {
            const char* e = std::getenv("LT_DEVICE_CACHE_GB");
            env_cap = e ? (long long)(std::atof(e) * 1e9) : -1;
        }
        if (env_cap >= 0) return env_cap;
        // a quarter above the live high-water mark (+ 256 MB): when the process's largest tracker closes, its blocks must fit BESIDE
        // the small ones already waiting (a closed two-slot context's, search buffers of other sizes) -- at exactly the high-water
        // mark every close evicted those, and the next tracker allocated them again (tests/test_gpu_soak.py: 482 MB per reopen)
        return (long long)std::min<size_t>(high_water + high_water / 4 + ((size_t)256 << 20), (size_t)16 << 30);
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

static void parked_to_driver();

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
            ++dc.hits;
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
        parked_to_driver();
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    }
    std::lock_guard<std::mutex> g(dc.m);
    ++dc.misses;
    handed_out(p);
    return p;
}
// A scope of frees on one thread (lt_reserve growing a context, lt_destroy): ONE wait for the device when it opens instead of one
// per block (a context is ~30 blocks; each device-wide wait also waits for every other tracker's work), and the blocks freed
// inside are PARKED until it closes -- they enter the cache together, behind whatever the scope allocated meanwhile.  That
// order is the point: a context that grows frees its old blocks and allocates larger ones; with the old blocks entering the
// cache first, the cache -- over its limit for a moment -- evicted its OLDEST blocks, which were exactly the larger ones a
// closed tracker had left and this context was about to ask for: 8-14 GB went back to the driver, were allocated again, and for
// the half second the driver took to wipe them every device-to-host copy of the process ran at half rate (round 5's "later
// annotated pass of a long-lived tracker": 19.9 k instead of 27 k frames/s, copy threads 49 % busy, in four of four bench runs;
// tools/annot_state_probe.py, profiles/NOTES_r06.md E.1).
namespace {
struct FreeScopeState { int depth = 0; int device = -1; std::vector<std::pair<void*, size_t>> parked; };
thread_local FreeScopeState t_free_scope;

void cache_insert_and_evict(DevCache& dc, int dev, const std::vector<std::pair<void*, size_t>>& in, std::vector<std::pair<int, void*>>& out) {   // (under dc.m)
    const long long cap = dc.cap();
    for (const auto& b : in) {
        if ((long long)b.second > cap) { out.push_back({dev, b.first}); ++dc.evicted_blocks; dc.evicted_bytes += b.second; continue; }
        dc.blocks.insert({{dev, b.second}, DevCache::Free{b.first, dc.seq++}});
        dc.kept += b.second;
    }
    while ((long long)dc.kept > cap && !dc.blocks.empty()) {   // over the limit: the block that has waited longest goes
        auto old = dc.blocks.begin();
        for (auto j = dc.blocks.begin(); j != dc.blocks.end(); ++j)
            if (j->second.seq < old->second.seq) old = j;
        dc.kept -= old->first.second;
        ++dc.evicted_blocks;
        dc.evicted_bytes += old->first.second;
        out.push_back({old->first.first, old->second.p});
        dc.blocks.erase(old);
    }
}
}  // namespace

FreeScope::FreeScope(int device) {
    FreeScopeState& fs = t_free_scope;
    if (fs.depth++ == 0) {
        fs.device = device;
        int cur = 0;
        (void)hipGetDevice(&cur);
        if (cur != device) (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();          // hipFree's own wait, once for the whole scope (see cached_free)
        if (cur != device) (void)hipSetDevice(cur);
    }
}
FreeScope::~FreeScope() {
    FreeScopeState& fs = t_free_scope;
    if (--fs.depth > 0) return;
    std::vector<std::pair<int, void*>> out;
    if (!fs.parked.empty()) {
        DevCache& dc = dev_cache();
        std::lock_guard<std::mutex> g(dc.m);
        cache_insert_and_evict(dc, fs.device, fs.parked, out);
    }
    fs.parked.clear();
    fs.device = -1;
    release_blocks(out);
}
// (cached_alloc's last resort: the parked blocks of this thread go back to the driver)
static void parked_to_driver() {
    FreeScopeState& fs = t_free_scope;
    std::vector<std::pair<int, void*>> out;
    for (const auto& b : fs.parked) out.push_back({fs.device, b.first});
    fs.parked.clear();
    release_blocks(out);
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
    FreeScopeState& fs = t_free_scope;
    if (fs.depth > 0 && fs.device == key.first) {             // inside a FreeScope of this device: waited for already, parked
        {
            std::lock_guard<std::mutex> g(dc.m);
            dc.live.erase(p);
            dc.live_bytes -= key.second;
        }
        fs.parked.push_back({p, key.second});
        return;
    }
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
        cache_insert_and_evict(dc, key.first, {{p, key.second}}, out);
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
        ++dc.evicted_blocks;
        dc.evicted_bytes += old->first.second;
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

int lt_device_cache_counters(unsigned long long* hits, unsigned long long* misses, unsigned long long* evicted_blocks, unsigned long long* evicted_bytes) {
    DevCache& dc = dev_cache();
    std::lock_guard<std::mutex> g(dc.m);
    if (hits) *hits = dc.hits;
    if (misses) *misses = dc.misses;
    if (evicted_blocks) *evicted_blocks = dc.evicted_blocks;
    if (evicted_bytes) *evicted_bytes = dc.evicted_bytes;
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
// Large copies with non-temporal stores (LT_COPY_NT=1; OFF by default): the idea -- the destination is not read again by this core,
// and a plain store first reads the destination's cache line -- does not pay on the GPU boxes (EPYC 9575F): measured SLOWER, an
// annotated 1920x1080 stream at 8.7-9.5 k frames/s against 10.7-11.9 k with glibc's memcpy (67-71 against 89-107 GB/s while
// the threads are busy), a window-sized raw copy on 12 threads at 59 against 135 GB/s.  Kept for other hosts.
__attribute__((target("avx2"))) static void copy_stream_avx2(uint8_t* dst, const uint8_t* src, size_t n) {
    size_t head = (32 - ((uintptr_t)dst & 31)) & 31;
    if (head > n) head = n;
    if (head) { std::memcpy(dst, src, head); dst += head; src += head; n -= head; }
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 64));
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i), a);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 96), d);
    }
    _mm_sfence();
    if (i < n) std::memcpy(dst + i, src + i, n - i);
}
static void copy_bytes(uint8_t* dst, const uint8_t* src, size_t n) {
    static const bool nt = [] {
        const char* e = LT_EXP_ENV("LT_COPY_NT");
        return e && e[0] == '1' && __builtin_cpu_supports("avx2");
    }();
    if (nt && n >= (64u << 10)) copy_stream_avx2(dst, src, n);
    else std::memcpy(dst, src, n);
}

struct HostCopier {
    // a piece of work for a worker: a 2-D copy, or (fn) anything else that runs on the host alone.  `hold` is released when the
    // last piece that carries it has run (a staging block going back to its pool).
    struct Job { uint8_t* dst; const uint8_t* src; size_t dpitch, spitch, width, height; int group; std::function<void()> fn; std::shared_ptr<void> hold; };
    struct Wait { hipEvent_t ev; int device; std::function<void()> then; };
    std::mutex m;
    std::condition_variable work, done, wwork;
    std::deque<Job> q;
    std::deque<Wait> wq;             // continuations behind device events, in submission order (the waiter thread)
    std::map<int, size_t> pending;   // group -> pieces queued, being copied or reserved (absent: none)
    size_t pending_all = 0;
    int next_group = 1;
    std::map<int, bool> groups;      // live groups created by lt_host_copy_group_create
    bool stop = false;
    std::vector<std::thread> th;
    std::thread waiter;
    void finished(int group) {       // (under m)
        --pending_all;
        auto it = pending.find(group);
        if (it != pending.end() && --it->second == 0) pending.erase(it);
        done.notify_all();
    }
    void run() {
        (void)pthread_setname_np(pthread_self(), "lt-copy");      // (visible in /proc/<pid>/task/*/comm, top -H, gdb)
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            if (q.empty() && !stop && spin_us() > 0 && spinning.load(std::memory_order_relaxed) < max_spinners()) {
                spinning.fetch_add(1, std::memory_order_relaxed);          // (under m: the bound holds)
                lk.unlock();
                const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us());
                while (queued.load(std::memory_order_acquire) == 0 && std::chrono::steady_clock::now() < until) __builtin_ia32_pause();
                lk.lock();
                spinning.fetch_sub(1, std::memory_order_relaxed);
            }
            ++sleeping;
            work.wait(lk, [&] { return stop || !q.empty(); });
            --sleeping;
            if (q.empty()) return;               // stop
            Job j = std::move(q.front());
            q.pop_front();
            queued.fetch_sub(1, std::memory_order_relaxed);
            // more pieces than this worker will take, and colleagues asleep: THIS thread wakes the next one (who wakes the next).  The
            // submitting thread -- process() between a frame's launches -- used to make those futex calls itself: 5-7 us each,
            // 37 us per frame in lt_host_copy2d_async_group (tools/process_trace.py, NOTES_r06 E.2)
            const bool pass_on = !q.empty() && sleeping > 0;
            lk.unlock();
            if (pass_on) work.notify_one();
            const auto t0 = std::chrono::steady_clock::now();
            if (j.fn) j.fn();
            else if (j.dpitch == j.width && j.spitch == j.width) copy_bytes(j.dst, j.src, j.width * j.height);
            else
                for (size_t r = 0; r < j.height; ++r) copy_bytes(j.dst + r * j.dpitch, j.src + r * j.spitch, j.width);
            busy_ns.fetch_add((unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(), std::memory_order_relaxed);
            bytes_done.fetch_add((unsigned long long)(j.width * j.height), std::memory_order_relaxed);
            jobs_done.fetch_add(1, std::memory_order_relaxed);
            j.fn = nullptr;
            j.hold.reset();                      // (outside the lock: the release may take a pool's lock)
            lk.lock();
            finished(j.group);
        }
    }
    void run_waiter() {
        (void)pthread_setname_np(pthread_self(), "lt-wait");
        std::unique_lock<std::mutex> lk(m);
        int on = -1;
        for (;;) {
            wwork.wait(lk, [&] { return stop || !wq.empty(); });
            if (wq.empty()) return;
            Wait w = std::move(wq.front());
            wq.pop_front();
            lk.unlock();
            if (w.device != on) { (void)hipSetDevice(w.device); on = w.device; }
            (void)hipEventSynchronize(w.ev);
            w.then();                            // submits the host side of the copy (and gives up its reservation)
            w.then = nullptr;
            lk.lock();
        }
    }
    // LT_COPY_THREADS (1 .. 16); default: half of the CPUs the process may use (affinity mask, cgroup quota), 2 .. 8 (with 12 of the 16 CPUs
    // the GPU boxes grant, the driving thread's Python work ran at half speed: the quota throttles it) -- a window of
    // annotated frames is 0.7 GB of rows to place (copies from the caller's window, strips from staging, text), and the GPU
    // boxes show 256 CPUs and grant 16
    int threads() {
        static const int n = [] {
            const char* e = std::getenv("LT_COPY_THREADS");
            if (e) return std::min(std::max(std::atoi(e), 1), 16);
            double cpus = (double)std::max(1u, std::thread::hardware_concurrency());
            cpu_set_t set;
            if (sched_getaffinity(0, sizeof set, &set) == 0) cpus = std::min(cpus, (double)CPU_COUNT(&set));
            if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
                char q[64];
                double per = 0.0;
                if (std::fscanf(f, "%63s %lf", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0.0) cpus = std::min(cpus, std::atof(q) / per);
                std::fclose(f);
            }
            return std::min(std::max((int)(cpus / 2.0), 2), 8);
        }();
        return n;
    }
    // Up to LT_COPY_SPINNERS workers (default 3) do not go to sleep at once when the queue runs empty: they poll for LT_COPY_SPIN_US
    // (default 400) first.  LaneTracker.process() sends a copy every 0.2-0.3 ms, and a worker woken from a futex on an idle
    // core starts 50-100 us late (deep C-states) -- the frame then waits for its own rows (1920x1080: _present 54 -> 240 us between
    // two runs on one box); and every wake-up is a system call of the SUBMITTING thread (10 us per submission on a quiet box,
    // 0.15-0.5 ms where futex wake-ups are slow: a VM, a host under load), so a submission wakes no more sleepers than it has
    // pieces the pollers cannot take.  Only while requests keep coming; LT_COPY_SPIN_US=0 turns the polling off.
    std::atomic<unsigned long long> busy_ns{0}, bytes_done{0}, jobs_done{0};   // lt_host_copy_stats (plain copies count their bytes, fn jobs 0)
    std::atomic<size_t> queued{0};
    std::atomic<int> spinning{0};
    int sleeping = 0;                // workers blocked in work.wait (under m)
    int max_spinners() {
        static const int v = [] { const char* e = std::getenv("LT_COPY_SPINNERS"); return e ? std::min(std::max(std::atoi(e), 0), 16) : 3; }();
        return std::min(v, threads());
    }
    void wake(size_t pieces) {      // after a push, outside the lock
        // A poller will take the first piece within a microsecond and wake a sleeper for the next one itself (run(): pass_on), and
        // so on down the queue: the submitting thread makes NO system call then.  With nobody polling, one sleeper is woken here
        // and passes it on the same way.  (Counted over everything that waits in the queue: a second submitter that finds the
        // only poller already claimed by another submission's piece wakes a sleeper of its own.)
        const size_t waiting = std::max(pieces, queued.load(std::memory_order_acquire));
        const size_t polling = (size_t)std::max(spinning.load(std::memory_order_relaxed), 0);
        if (waiting > polling) {
            // ... except that a large submission (a frame's 1.8-4 MB of rows: more pieces than pollers) gets up to two sleepers
            // from here as well: woken one after the other by their colleagues they would start 50-100 us apart, and the frame's
            // rows would not be there when its text is due (12 us of the submitter's time, which waits for the device anyway)
            const size_t direct = polling == 0 ? 1 : std::min<size_t>(2, waiting - polling);
            for (size_t i = 0; i < direct; ++i) work.notify_one();
        }
    }
    static int spin_us() {
        static const int v = [] { const char* e = std::getenv("LT_COPY_SPIN_US"); return e ? std::max(std::atoi(e), 0) : 400; }();
        return v;
    }
    int adopt_below = 0;             // the child of a fork(): ids below this were handed out by the parent's copier and are taken over on first use
    bool known(int group) {          // (under m)
        if (group == 0 || groups.count(group) != 0) return true;
        if (group > 0 && group < adopt_below) { groups[group] = true; return true; }
        return false;
    }
    void start_workers(std::unique_lock<std::mutex>& lk, int want) {               // (under m)
        done.wait(lk, [&] { return !stop; });          // an lt_shutdown() under way: workers start again once it is over
        while ((int)th.size() < want) th.emplace_back([this] { run(); });
    }
    int submit(const Job& whole) {
        // pieces of whole rows ("rows" of the 2-D copy: frames), a few per worker so that they finish together -- but none below
        // 256 KB: a piece costs a queue round trip (and, for a sleeper, a wake-up)
        const size_t by_size = std::max<size_t>(1, whole.width * whole.height / (256u << 10));
        const size_t parts = whole.height <= 1 ? 1 : std::min({whole.height, (size_t)threads() * 2, by_size});
        size_t pushed = 0;
        {
            std::unique_lock<std::mutex> lk(m);
            if (!known(whole.group)) return -1;
            start_workers(lk, whole.height <= 1 ? 1 : threads());
            for (size_t k = 0; k < parts; ++k) {
                const size_t r0 = whole.height * k / parts, r1 = whole.height * (k + 1) / parts;
                if (r1 > r0) {
                    q.push_back({whole.dst + r0 * whole.dpitch, whole.src + r0 * whole.spitch, whole.dpitch, whole.spitch, whole.width, r1 - r0, whole.group,
                                 nullptr, whole.hold});
                    queued.fetch_add(1, std::memory_order_release);
                    ++pending[whole.group];
                    ++pending_all;
                    ++pushed;
                }
            }
        }
        wake(pushed);
        return 0;
    }
    int submit_fn(int group, std::function<void()> fn, bool many) {
        {
            std::unique_lock<std::mutex> lk(m);
            if (!known(group)) return -1;
            start_workers(lk, many ? threads() : 1);
            q.push_back({nullptr, nullptr, 0, 0, 0, 0, group, std::move(fn), nullptr});
            queued.fetch_add(1, std::memory_order_release);
            ++pending[group];
            ++pending_all;
        }
        wake(1);
        return 0;
    }
    int reserve(int group) {         // a piece that will be submitted later (behind a device copy): the group is not complete without it
        std::lock_guard<std::mutex> lk(m);
        if (!known(group)) return -1;
        ++pending[group];
        ++pending_all;
        return 0;
    }
    void unreserve(int group) {
        std::lock_guard<std::mutex> lk(m);
        finished(group);
    }
    void after_event(hipEvent_t ev, int device, std::function<void()> then) {
        {
            std::unique_lock<std::mutex> lk(m);
            done.wait(lk, [&] { return !stop; });
            if (!waiter.joinable()) waiter = std::thread([this] { run_waiter(); });
            wq.push_back({ev, device, std::move(then)});
        }
        wwork.notify_one();
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
    void shutdown() {            // finish what is queued (continuations behind device events included), then join the threads
        std::vector<std::thread> mine;
        std::thread w;
        {
            std::unique_lock<std::mutex> lk(m);
            done.wait(lk, [&] { return pending_all == 0 && wq.empty(); });
            stop = true;
            mine.swap(th);
            w.swap(waiter);
        }
        work.notify_all();
        wwork.notify_all();
        for (auto& t : mine) if (t.joinable()) t.join();
        if (w.joinable()) w.join();
        { std::lock_guard<std::mutex> lk(m); stop = false; }
        done.notify_all();
    }
    ~HostCopier() {
        { std::lock_guard<std::mutex> lk(m); stop = true; q.clear(); wq.clear(); queued.store(0); }
        work.notify_all();
        wwork.notify_all();
        for (auto& t : th) if (t.joinable()) t.join();
        if (waiter.joinable()) waiter.join();
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
        // possibly held by a thread that is gone, is left alone).  Group ids the parent handed out keep their meaning: the child's
        // copier numbers its own groups behind them and takes an inherited id over, with nothing pending, when it is first used
        // (a LaneTracker created before the fork keeps working in the child).
        pthread_atfork(nullptr, nullptr, [] {
            HostCopier* old = g_copier;
            g_copier = new HostCopier;
            if (old) { g_copier->next_group = old->next_group; g_copier->adopt_below = old->next_group; }
        });
    });
    return *g_copier;
}

// Page-locked staging blocks, kept per size for the life of the process (lt_shutdown / lt_pinned_trim give them back): the
// strips of annotated frames land in these and are scattered into the caller's (pageable) frames by the copy threads.  A block is a
// few dozen MB -- page-locking costs ~0.2 ms per MB, so a window-sized page-locked output array (0.7 GB at 1280x720) was 0.13 s
// of the first window of every stream, three times over (NOTES D.1).
struct PinnedBlocks {
    std::mutex m;
    std::condition_variable freed;
    std::map<size_t, std::vector<void*>> free_;
    size_t allocated = 0;
    size_t limit() {
        static const size_t v = [] { const char* e = std::getenv("LT_STAGING_MB"); return (size_t)(e ? std::max(64, std::atoi(e)) : 1536) << 20; }();
        return v;
    }
    void* acquire(size_t bytes) {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            auto& v = free_[bytes];
            if (!v.empty()) { void* p = v.back(); v.pop_back(); return p; }
            if (allocated + bytes <= limit() || allocated == 0) break;
            // over the budget: blocks of other sizes that idle go first, then wait for one of ours
            bool dropped = false;
            for (auto& kv : free_)
                if (kv.first != bytes && !kv.second.empty()) {
                    void* p = kv.second.back();
                    kv.second.pop_back();
                    allocated -= kv.first;
                    lk.unlock();
                    (void)hipHostFree(p);
                    lk.lock();
                    dropped = true;
                    break;
                }
            if (!dropped) freed.wait(lk);
        }
        allocated += bytes;
        lk.unlock();
        void* p = nullptr;
        {
            lt::TraceScope ts_("hipHostMalloc(staging)", bytes);
            if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
        }
        if (!p) { lk.lock(); allocated -= bytes; }
        return p;
    }
    void release(void* p, size_t bytes) {
        { std::lock_guard<std::mutex> lk(m); free_[bytes].push_back(p); }
        freed.notify_all();
    }
    void trim() {
        std::vector<void*> out;
        {
            std::lock_guard<std::mutex> lk(m);
            for (auto& kv : free_) { for (void* p : kv.second) { out.push_back(p); allocated -= kv.first; } kv.second.clear(); }
        }
        for (void* p : out) (void)hipHostFree(p);
    }
};
PinnedBlocks& pinned_blocks() { static PinnedBlocks* b = new PinnedBlocks; return *b; }

struct EventPool {
    std::mutex m;
    std::vector<hipEvent_t> free_;
};
EventPool& event_pool() { static EventPool* e = new EventPool; return *e; }
}  // namespace
}  // extern "C++"

}  // extern "C"
namespace lt {
int host_submit_copy2d(int group, uint8_t* dst, size_t dpitch, const uint8_t* src, size_t spitch, size_t width, size_t height, std::shared_ptr<void> hold) {
    return host_copier().submit({dst, src, dpitch, spitch, width, height, group, nullptr, std::move(hold)});
}
int host_submit_fn(int group, std::function<void()> fn, bool many) { return host_copier().submit_fn(group, std::move(fn), many); }
int host_reserve(int group) { return host_copier().reserve(group); }
void host_unreserve(int group) { host_copier().unreserve(group); }
void host_after_event(hipEvent_t ev, int device, std::function<void()> then) { host_copier().after_event(ev, device, std::move(then)); }
int host_copy_threads() { return host_copier().threads(); }
int host_copy_pollers() { return std::max(host_copier().spinning.load(std::memory_order_relaxed), 0); }
void* pinned_block_acquire(size_t bytes) { return pinned_blocks().acquire(bytes); }
void pinned_block_release(void* p, size_t bytes) { pinned_blocks().release(p, bytes); }
hipEvent_t pooled_event() {
    EventPool& ep = event_pool();
    {
        std::lock_guard<std::mutex> lk(ep.m);
        if (!ep.free_.empty()) { hipEvent_t e = ep.free_.back(); ep.free_.pop_back(); return e; }
    }
    hipEvent_t e = nullptr;
    if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return e;
}
void pooled_event_release(hipEvent_t e) {
    if (!e) return;
    EventPool& ep = event_pool();
    std::lock_guard<std::mutex> lk(ep.m);
    ep.free_.push_back(e);
}

// One row of a character's cell, `n` pixels wide and wholly inside the frame, 16 pixels (48 bytes) per step:
//     out = v + ((255 - v) * alpha + 127) / 255,   t / 255 == (t + 1 + (t >> 8)) >> 8 for 0 <= t < 65536
// (tests/test_presentation_cpu.py holds this against the scalar form for every (v, alpha)).  Lanes right of the cell get alpha 0,
// which leaves their bytes as they are; the caller guarantees 48 readable / writable bytes from `p` and 16 readable alphas.
__attribute__((target("avx2"))) static void blend_row_avx2(uint8_t* p, const uint8_t* alpha, int n) {
    const __m128i idx = _mm_setr_epi8(0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15);
    const __m128i sh0 = _mm_setr_epi8(0, 0, 0, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5);
    const __m128i sh1 = _mm_setr_epi8(5, 5, 6, 6, 6, 7, 7, 7, 8, 8, 8, 9, 9, 9, 10, 10);
    const __m128i sh2 = _mm_setr_epi8(10, 11, 11, 11, 12, 12, 12, 13, 13, 13, 14, 14, 14, 15, 15, 15);
    const __m256i c255 = _mm256_set1_epi16(255), c127 = _mm256_set1_epi16(127), c1 = _mm256_set1_epi16(1);
    for (int g0 = 0; g0 < n; g0 += 16, p += 48, alpha += 16) {
        __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(alpha));
        a = _mm_and_si128(a, _mm_cmpgt_epi8(_mm_set1_epi8((char)std::min(n - g0, 16)), idx));       // alphas beyond the cell: 0
        if (_mm_testz_si128(a, a)) continue;
        const __m128i a3[3] = {_mm_shuffle_epi8(a, sh0), _mm_shuffle_epi8(a, sh1), _mm_shuffle_epi8(a, sh2)};
        for (int k = 0; k < 3; ++k) {
            __m128i* q = reinterpret_cast<__m128i*>(p + 16 * k);
            const __m256i v = _mm256_cvtepu8_epi16(_mm_loadu_si128(q)), al = _mm256_cvtepu8_epi16(a3[k]);
            const __m256i t = _mm256_add_epi16(_mm256_mullo_epi16(_mm256_sub_epi16(c255, v), al), c127);                 // < 65536: unsigned 16 bit
            const __m256i d = _mm256_srli_epi16(_mm256_add_epi16(_mm256_add_epi16(t, c1), _mm256_srli_epi16(t, 8)), 8);   // t / 255
            const __m256i o = _mm256_add_epi16(v, d);
            const __m256i pk = _mm256_permute4x64_epi64(_mm256_packus_epi16(o, o), 0x08);
            _mm_storeu_si128(q, _mm256_castsi256_si128(pk));
        }
    }
}

// k_overlay_text's arithmetic (csrc/k_overlay.hip) for one frame on the host: every character's cell up to its advance width,
// white over the frame, out = v + ((255 - v) * alpha + 127) / 255 per channel
void text_blend_frame(uint8_t* frame, int img_h, int img_w, const uint8_t* atlas, const uint8_t* advance, int first_char, int n_glyphs,
                      int gw, int gh, const char* lines, int n_lines, int line_len, int x0, int y0, int step) {
    for (int l = 0; l < n_lines; ++l) {
        const char* src = lines + (size_t)l * line_len;
        int x = x0;
        bool ended = false;
        for (int k = 0; k < line_len; ++k) {
            const unsigned char ch = (unsigned char)src[k];
            ended = ended || ch == 0;
            const int xk = std::min(x, 32767);                   // the left edge the device path stores as int16
            const int g = (int)ch - first_char;
            if (g < 0 || g >= n_glyphs) continue;
            const int adv = advance[g];
            if (!ended) x += adv;
            const uint8_t* cell = atlas + (size_t)g * gh * gw;
            const int ncol = std::min(adv, gw);
            // the vector form where whole steps of 16 pixels stay inside the row and inside the atlas (LT_TEXT_SCALAR=1: never; A/B, tests)
            static const bool vec_ok = [] { const char* e = LT_EXP_ENV("LT_TEXT_SCALAR"); return !(e && e[0] == '1') && __builtin_cpu_supports("avx2"); }();
            const int span = (ncol + 15) & ~15;
            const bool vec = vec_ok && ncol > 0 && xk >= 0 && xk + span <= img_w &&
                             (size_t)g * gh * gw + (size_t)(gh - 1) * gw + span <= (size_t)n_glyphs * gh * gw;
            for (int gy = 0; gy < gh; ++gy) {
                const int y = y0 + l * step + gy;
                if (y < 0 || y >= img_h) continue;
                uint8_t* row = frame + (size_t)y * img_w * 3;
                if (vec) {
                    blend_row_avx2(row + (size_t)xk * 3, cell + (size_t)gy * gw, ncol);
                    continue;
                }
                for (int gx = 0; gx < adv && gx < gw; ++gx) {
                    const int alpha = cell[(size_t)gy * gw + gx];
                    const int px = xk + gx;
                    if (alpha == 0 || px < 0 || px >= img_w) continue;
                    uint8_t* p = row + (size_t)px * 3;
                    for (int c = 0; c < 3; ++c) {
                        const int v = p[c];
                        p[c] = (uint8_t)(v + ((255 - v) * alpha + 127) / 255);
                    }
                }
            }
        }
    }
}
}  // namespace lt
extern "C" {

int lt_text_blend_host(uint8_t* frames, size_t frame_stride, int n, int img_h, int img_w, const uint8_t* atlas, const uint8_t* advance,
                       int first_char, int n_glyphs, int glyph_w, int glyph_h, const char* lines, int n_lines, int line_len, int x0,
                       int y0, int step) {
    if (n < 0 || img_h < 1 || img_w < 1 || n_glyphs < 1 || glyph_w < 1 || glyph_h < 1 || n_lines < 0 || line_len < 0 || first_char < 0)
        return fail(LT_ERR_INVALID, "lt_text_blend_host: bad geometry");
    if (n == 0 || n_lines == 0 || line_len == 0) return LT_OK;
    if (!frames || !atlas || !advance || !lines) return fail(LT_ERR_INVALID, "lt_text_blend_host: null pointer");
    if (frame_stride < (size_t)img_h * img_w * 3 && n > 1) return fail(LT_ERR_INVALID, "lt_text_blend_host: frame stride below the frame size");
    for (int i = 0; i < n; ++i)
        text_blend_frame(frames + (size_t)i * frame_stride, img_h, img_w, atlas, advance, first_char, n_glyphs, glyph_w, glyph_h,
                         lines + (size_t)i * n_lines * line_len, n_lines, line_len, x0, y0, step);
    return LT_OK;
}

int lt_host_text_async_group(int group, uint8_t* dst, size_t dst_stride, const uint8_t* src, size_t src_stride, int n, const int32_t* rows4,
                             int img_h, int img_w, const uint8_t* atlas, const uint8_t* advance, int first_char, int n_glyphs, int glyph_w,
                             int glyph_h, const char* lines, int n_lines, int line_len, int x0, int y0, int step) {
    int r[4] = {0, 0, 0, 0};
    if (rows4) for (int k = 0; k < 4; ++k) r[k] = rows4[k];
    if (n < 0 || img_h < 1 || img_w < 1 || n_lines < 0 || line_len < 0 || !(0 <= r[0] && r[0] <= r[1] && r[1] <= r[2] && r[2] <= r[3] && r[3] <= img_h))
        return fail(LT_ERR_INVALID, "lt_host_text_async_group: bad geometry");
    if (n == 0) return LT_OK;
    const bool copies = r[1] > r[0] || r[3] > r[2];
    if (!dst || (copies && !src)) return fail(LT_ERR_INVALID, "lt_host_text_async_group: null frames");
    const bool text = n_lines > 0 && line_len > 0;
    if (text && (!atlas || !advance || !lines || n_glyphs < 1 || glyph_w < 1 || glyph_h < 1 || first_char < 0))
        return fail(LT_ERR_INVALID, "lt_host_text_async_group: null or empty font / text");
    // the caller's text need not outlive the call
    auto keep = std::make_shared<std::vector<char>>(text ? lines : nullptr, text ? lines + (size_t)n * n_lines * line_len : nullptr);
    const int parts = std::min(n, host_copy_threads() * 2);
    const size_t row_bytes = (size_t)img_w * 3;
    const int r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
    for (int k = 0; k < parts; ++k) {
        const int f0 = (int)((long long)n * k / parts), f1 = (int)((long long)n * (k + 1) / parts);
        if (f1 <= f0) continue;
        auto fn = [=]() {
            for (int f = f0; f < f1; ++f) {           // a frame at a time: its rows, then its text over them while they are in the cache
                uint8_t* d = dst + (size_t)f * dst_stride;
                const uint8_t* q = src + (size_t)f * src_stride;
                if (r1 > r0) std::memcpy(d + (size_t)r0 * row_bytes, q + (size_t)r0 * row_bytes, (size_t)(r1 - r0) * row_bytes);
                if (r3 > r2) std::memcpy(d + (size_t)r2 * row_bytes, q + (size_t)r2 * row_bytes, (size_t)(r3 - r2) * row_bytes);
                if (text)
                    text_blend_frame(d, img_h, img_w, atlas, advance, first_char, n_glyphs, glyph_w, glyph_h,
                                     keep->data() + (size_t)f * n_lines * line_len, n_lines, line_len, x0, y0, step);
            }
        };
        if (host_submit_fn(group, fn, true)) return fail(LT_ERR_INVALID, "lt_host_text_async_group: unknown group %d", group);
    }
    return LT_OK;
}

// One frame's text lines NOW: wait for the group (the rows under the text are among its copies), then draw -- line 0 on the calling
// thread, the others offered to copy threads that are polling at this moment and drawn here if nobody has claimed them by then.  What
// LaneTracker.process() calls once radius and eccentricity are known: one call and ~5 us of blending on its critical path where a
// job for a copy thread (lt_host_text_async_group) cost the submission, the thread's start and a second wait (22 -> 12 us of the
// frame, tools/process_points.py).
int lt_host_text_now_group(int group, uint8_t* frame, int img_h, int img_w, const uint8_t* atlas, const uint8_t* advance, int first_char,
                           int n_glyphs, int glyph_w, int glyph_h, const char* lines, int n_lines, int line_len, int x0, int y0, int step) {
    if (img_h < 1 || img_w < 1 || n_lines < 0 || n_lines > 8 || line_len < 0) return fail(LT_ERR_INVALID, "lt_host_text_now_group: bad geometry");
    if (host_copier().wait_group(group)) return fail(LT_ERR_INVALID, "lt_host_text_now_group: unknown group %d", group);
    if (n_lines == 0 || line_len == 0) return LT_OK;
    if (!frame || !atlas || !advance || !lines || n_glyphs < 1 || glyph_w < 1 || glyph_h < 1 || first_char < 0)
        return fail(LT_ERR_INVALID, "lt_host_text_now_group: null or empty font / text / frame");
    struct Lines {
        std::atomic<int> claimed[8];
        std::atomic<int> done{0};
        std::vector<char> text;
        std::function<void(int)> draw;
        void run(int l) { draw(l); done.fetch_add(1, std::memory_order_release); }
    };
    auto one = [=](const char* text, int l) {
        text_blend_frame(frame, img_h, img_w, atlas, advance, first_char, n_glyphs, glyph_w, glyph_h, text + (size_t)l * line_len, 1, line_len, x0,
                         y0 + l * step, step);
    };
    const int helpers = std::min(host_copy_pollers(), n_lines - 1);
    if (helpers <= 0) {
        for (int l = 0; l < n_lines; ++l) one(lines, l);
        return LT_OK;
    }
    auto sp = std::make_shared<Lines>();
    sp->text.assign(lines, lines + (size_t)n_lines * line_len);
    for (int l = 0; l < 8; ++l) sp->claimed[l].store(0, std::memory_order_relaxed);
    Lines* raw = sp.get();
    sp->draw = [one, raw](int l) { one(raw->text.data(), l); };
    sp->claimed[0].store(1, std::memory_order_relaxed);
    for (int l = 1; l <= helpers; ++l)
        if (host_submit_fn(0, [sp, l] { if (!sp->claimed[l].exchange(1, std::memory_order_acq_rel)) sp->run(l); }, false) != 0) break;
    sp->run(0);
    for (int l = 1; l < n_lines; ++l)
        if (!sp->claimed[l].exchange(1, std::memory_order_acq_rel)) sp->run(l);
    for (unsigned spins = 0; sp->done.load(std::memory_order_acquire) < n_lines; ++spins) {     // (a claimed line is microseconds of work)
        if ((spins & 0xffffu) == 0xffffu) std::this_thread::yield(); else __builtin_ia32_pause();
    }
    return LT_OK;
}

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

int lt_host_touch_async_group(int group, void* p, size_t bytes) {
    if (bytes == 0) return LT_OK;
    if (!p) return fail(LT_ERR_INVALID, "lt_host_touch_async_group: null pointer");
    const int parts = host_copy_threads() * 2;
    uint8_t* base = static_cast<uint8_t*>(p);
    for (int k = 0; k < parts; ++k) {
        const size_t a = (bytes * (size_t)k / parts) & ~(size_t)4095, b = k + 1 == parts ? bytes : (bytes * (size_t)(k + 1) / parts) & ~(size_t)4095;
        if (b <= a) continue;
        if (host_submit_fn(group, [=]() { for (size_t o = a; o < b; o += 4096) reinterpret_cast<volatile uint8_t*>(base)[o] = 0; }, true))
            return fail(LT_ERR_INVALID, "lt_host_touch_async_group: unknown group %d", group);
    }
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

int lt_host_copy_stats(double* busy_seconds, double* copied_bytes, long long* pieces, int* threads) {
    HostCopier& h = host_copier();
    if (busy_seconds) *busy_seconds = 1e-9 * (double)h.busy_ns.load();
    if (copied_bytes) *copied_bytes = (double)h.bytes_done.load();
    if (pieces) *pieces = (long long)h.jobs_done.load();
    if (threads) *threads = h.threads();
    return LT_OK;
}

int lt_host_memory_stats(size_t* staging_bytes, size_t* queued_pieces, size_t* pending_pieces) {
    if (staging_bytes) {
        PinnedBlocks& pb = pinned_blocks();
        std::lock_guard<std::mutex> lk(pb.m);
        *staging_bytes = pb.allocated;
    }
    if (queued_pieces || pending_pieces) {
        HostCopier& h = host_copier();
        std::lock_guard<std::mutex> lk(h.m);
        if (queued_pieces) *queued_pieces = h.q.size();
        if (pending_pieces) *pending_pieces = h.pending_all;
    }
    return LT_OK;
}

int lt_shutdown(void) {
    if (g_copier) g_copier->shutdown();
    pinned_blocks().trim();
    return LT_OK;
}

}  // extern "C"
