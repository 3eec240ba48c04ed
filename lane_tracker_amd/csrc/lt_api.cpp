// C-ABI layer of liblane_tracker_amd.so (see include/lane_tracker_amd.h).
// Owns the context: HIP stream, calibration tables, frame slots; sequences the kernel chain of
// LaneTracker.find_lane_points() (lane_tracker.py:795-874) for a batch of independent frames.
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

#include "lt_internal.h"

using namespace lt;

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return fail(LT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

enum Stage {
    ST_UNDISTORT = 0, ST_WARP_SPLIT, ST_ERODE_R, ST_TOPHAT_R, ST_ERODE_B, ST_TOPHAT_B, ST_THRESHOLD, ST_MERGE,
    ST_OPEN, ST_SWS_FIT, ST_BAND_FIT, ST_SPLIT_BEV
};
const char* kStageNames[LT_NUM_STAGES] = {"undistort_rows", "warp_split", "erode_r29", "tophat_r29", "erode_b55",
                                          "tophat_b55", "threshold", "merge", "open5", "sws_fit", "band_fit",
                                          "split_bev"};

enum Plane { P_R = 0, P_B, P_THR, P_THB, P_MERGED, P_MASK, P_T0, P_T1, P_T2, P_T3, P_COUNT };

}  // namespace

struct lt_ctx {
    lt_calib calib{};
    int device = 0;
    hipStream_t stream = nullptr;             // = streams[0]
    std::vector<hipStream_t> streams;         // slot s runs on streams[s * nstreams / capacity]
    hipStream_t copy = nullptr;               // lt_upload_frame_rest: the rows the path does not read, off the critical path
    hipStream_t side = nullptr;               // second branch of a one- or two-frame chain (R and b top-hats side by side)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int nstreams = 1;
    hipDeviceProp_t prop{};
    FrontEndGeom fe{};
    int cam_r0 = 0, cam_r1 = 0;               // camera rows the undistortion reads (its taps for rows [fe.r0, fe.r0 + nrows))
    EllipseSE se5{}, se29{}, se55{};
    // device tables
    int16_t *d_uxy = nullptr, *d_wxy = nullptr;
    uint16_t *d_ufrac = nullptr, *d_wfrac = nullptr, *d_gamma = nullptr, *d_cbrt = nullptr;
    int32_t* d_coef = nullptr;
    // slots
    int capacity = 0;
    size_t frame_bytes = 0, und_bytes = 0, plane_bytes = 0, bev_bytes = 0;
    uint8_t *d_frames = nullptr, *d_bev = nullptr;
    uint32_t* d_und = nullptr;        // undistorted camera rows [r0, r0+nrows), one RGBX dword per pixel, slots 2p / 2p+1 interleaved (und_slot_base)
    size_t und_px = 0;                // pixels per slot of d_und
    uint8_t* d_plane[P_COUNT] = {};
    unsigned long long *d_bits_merged = nullptr, *d_bits_eroded = nullptr;   // 1 bit / pixel, wpr words per row
    unsigned long long* d_bits_open = nullptr;    // the opened mask as the mask chain leaves it (what the searches read)
    unsigned long long* d_bits_tmp = nullptr;     // third and fourth partial plane of the walking threshold kernels
    unsigned long long* d_bits_tmp2 = nullptr;
    // Top-hat planes with a 64-byte-multiple row pitch: what the walking threshold kernels read (every 64-byte piece of
    // a row is one aligned sector; with the image width as pitch the horizontal pass fetched every sector twice).
    // th_padded[slot] says which copy of the slot's top-hat planes is current (lt_download_plane).
    uint8_t* d_th_pad[2] = {nullptr, nullptr};
    size_t th_pad_bytes = 0;
    int th_pitch = 0;
    std::vector<uint8_t> th_padded;
    // mask_noise through the walking kernels (allocated by the first such call, ensure_noise_buffers): the raw Lab-b plane
    // in the padded layout (the 55x55 top-hat launch stores its minuend there) and the two greenery-mask bit planes
    uint8_t* d_b_pad = nullptr;
    unsigned long long *d_bits_n1 = nullptr, *d_bits_n2 = nullptr;
    int last_threshold_path = -1;                 // lt_last_threshold_path
    int last_adaptive_path = -1;                  // 'neighborhood' calls: 1 = running box sums (k_adaptive_walk.hip), 0 = per-pixel windows
    // The walking threshold kernels are long serial walks (a wave covers half an image row or column): they win once a
    // call brings enough frames to fill the chip -- measured crossover 70-80 frames of 1100 x 1080 per call
    // (tools/threshold_crossover.py: 64 frames 232 vs 210 us, 96 frames 255 vs 304 us) -- and lose badly on a single
    // frame (164 vs 25 us).  Calls below this many pixels take the tile kernel.  LT_WALK_MIN_FRAMES=<n> (read at
    // lt_create, in frames of this context's bird's-eye size) overrides it; 0 = always walk.
    long long walk_min_pixels = 80LL * 1100 * 1080;
    // per slot: which forms of the mask are current.  The chain writes the bit plane only; the u8 mask
    // (d_plane[P_MASK]) is expanded from it when somebody asks for it; lt_upload_masks provides u8 only.
    std::vector<uint8_t> mask_bits_ok, mask_u8_ok;
    size_t bits_stride = 0;                                                  // u64 words per slot
    lt_lane_record* d_rec = nullptr;
    double* d_prev = nullptr;
    uint32_t* d_pix = nullptr;
    int32_t* d_cent = nullptr;
    uint32_t* d_band_sums = nullptr;  // [slot][band][warp_w] column sums of the search bands
    int maxpix = 0, maxlev = 0, maxbands = 0;
    bool have_mask = false;
    bool brute_tophat = false;
    // presentation stage (lt_overlay_*): inverse-warp table, per-slot row intervals, annotated frames
    int16_t* d_oxy = nullptr;
    uint16_t* d_ofrac = nullptr;
    bool have_overlay = false;
    int16_t* d_spans = nullptr;       // [slot][warp_h] (lo, hi)
    uint8_t* d_annot = nullptr;
    // Page-locked staging with one region PER SLOT (row intervals, text lines, glyph positions), so that an overlay call only
    // enqueues copies and kernels: calls over disjoint slots never wait for each other (the stream pipeline renders a window
    // in pieces while later frames are still searched).  A call over slots whose previous overlay may still be in flight
    // waits for that one first (overlay_lo / overlay_hi / overlay_done).
    int16_t* h_spans = nullptr;       // [capacity][warp_h * 2]
    int h_spans_cap = 0;
    struct StagingBusy { int lo = 0, hi = 0; hipEvent_t done = nullptr; };   // slots whose staging region a copy may still read
    StagingBusy spans_busy, text_busy;
    hipStream_t dl = nullptr;         // lt_download_overlay_async: device-to-host copies beside the compute and upload streams
    // The presentation kernels (spans copy, lane overlay, text) run on a stream of their own: on a slot's compute stream they would
    // queue behind the mask launches of LATER frames, which wait for uploads the bus has not delivered yet (measured: the first
    // overlay of a stream of windows ran 30 ms after its frames were ready).  It waits, per slot range, for the kernels that
    // wrote the slots' masks (hence for their camera rows) and for the copies of the remaining rows.
    hipStream_t present = nullptr;
    // lt_set_urgent: while on, the stage calls run on this stream instead of the slots' streams -- behind what was enqueued
    // for THEIR slots only (slot-range events), not behind the masks of later frames queued on the slots' streams, which
    // wait for uploads still on the bus.  The stateful stream handles a frame whose first try failed this way.
    hipStream_t urgent = nullptr;
    bool urgent_on = false;
    StagingBusy annot_busy;           // annotated frames a copy on `dl` may still read
    // How the annotated frames go back (lt_download_overlay_async): by the copy engine or by a kernel that stores into the
    // page-locked destination.  Both are timed, copy by copy, with an event pair on the download stream; see choose_download().
    struct DlTimed { hipEvent_t a, b; double bytes; int method; };
    std::vector<DlTimed> dl_inflight;
    std::vector<hipEvent_t> dl_event_pool;
    double dl_rate[2] = {0.0, 0.0};   // GB/s, running mean of the last copies: [0] engine, [1] kernel
    int dl_samples[2] = {0, 0};
    int dl_method = 0;                // what the next copy uses
    int dl_since_probe = 0;           // copies since the other method was last tried
    int dl_forced = -1;               // LT_DL_KERNEL=0 / 1, lt_set_download_method: -1 = choose by measurement
    hipEvent_t rest_done = nullptr;   // end of the most recent lt_upload_frame_rest on the copy stream
    bool rest_pending = false;
    // text: glyph atlas (set once) and the per-slot lines of the current call
    uint8_t *d_atlas = nullptr, *d_advance = nullptr, *d_lines = nullptr;
    int16_t* d_xpos = nullptr;
    std::vector<uint8_t> h_advance;
    uint8_t* h_lines = nullptr;       // page-locked, [text_slots][text_per_slot]
    int16_t* h_xpos = nullptr;
    int font_first = 0, font_glyphs = 0, font_gw = 0, font_gh = 0;
    size_t text_per_slot = 0;         // characters per slot the text buffers hold (n_lines * line_len of the largest call)
    int text_slots = 0;
    // ordering events of lt_upload_frame_rows_async (a ring: an event is reused long after its waits were enqueued)
    std::vector<hipEvent_t> order_events;
    size_t order_next = 0;
    // Slot-range bookkeeping of work in flight on the slots' streams, so that other streams wait for exactly what they
    // depend on instead of for the tails of those streams:
    //   readers -- kernels that READ the camera frames (undistortion, overlay): a stream-ordered upload into slots waits for
    //              the readers of those slots only, so the rows of later frames cross the bus while earlier ones are processed;
    //   writers -- kernels that wrote masks / records: a chained search waits for the writers of its own slots only.
    // A ring each; finished entries are dropped as new ones arrive (a long stream never synchronises the whole context).
    // The ring grows with the launches in flight (an outage group adds two or three entries per piece while the head
    // still waits for the bus); beyond 4096 entries it gives up (`overflow`) and waiters fall back to the tails of every
    // stream that can touch the slots, until the next full synchronisation.
    struct RangeEvents {
        struct Entry { int lo, hi; hipEvent_t ev; };
        std::vector<Entry> e = std::vector<Entry>(32, Entry{0, 0, nullptr});
        unsigned head = 0, count = 0;
        bool overflow = false;
        void reset() { head = count = 0; overflow = false; }
    };
    RangeEvents readers, writers;
    RangeEvents rests;                        // lt_upload_frame_rest copies (copy stream): the overlay of a slot waits for ITS rows only
    // The chained band search of a stream (lt_band_fit_chain_run) is one workgroup walking many frames: it runs on a stream
    // of its own, beside the mask chains of later frames on the slots' streams.  A chain leaves its records in page-locked
    // host memory behind an event (lt_band_fit_chain_collect waits for that event only, not for the device).  Work on the
    // slots' streams that touches slots of a chain still in flight waits for it (for_each_slice).
    hipStream_t search = nullptr;
    int search_cus = 0;                       // lt_set_search_cus: CUs the search stream has to itself (0: none reserved)
    struct ChainTicket { int first, n; hipEvent_t done; int own; };   // own: first slot the chain searched itself (first + 1 when slot `first` is only its seed record)
    std::vector<ChainTicket> chains;          // not yet collected, oldest first
    std::vector<hipEvent_t> chain_event_pool;
    int* h_cancel = nullptr;                  // page-locked, device-visible: chains launched with an older epoch stop at their next frame
    int* d_cancel = nullptr;                  // its device address
    uint8_t* h_small = nullptr;               // page-locked scratch of the small downloads (download())
    int ov_r0 = 0, ov_r1 = 0;                 // camera rows the lane overlay can change (lt_overlay_configure)
    lt_lane_record* h_rec = nullptr;          // page-locked mirror of the record of the last ONE-frame search (mirror_record)
    int rec_mirror_slot = -1;                 // the slot whose record the mirror holds once rec_mirror_stream is idle; -1: none
    hipStream_t rec_mirror_stream = nullptr;
    lt_lane_record* h_rec_stage = nullptr;    // capacity records
    int h_rec_stage_cap = 0;
    // timing
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool stage_timing = false;
    std::vector<hipEvent_t> ev_pool;
    struct Pending { int stage; hipEvent_t a, b; };
    std::vector<Pending> pending;
    size_t ev_used = 0;
    float stage_ms[LT_NUM_STAGES] = {};
    int32_t stage_launches[LT_NUM_STAGES] = {};
};

namespace {

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
static DevCache& dev_cache() { static DevCache* c = new DevCache; return *c; }   // (never destroyed: no order problems at exit)

static void* cached_alloc(size_t bytes) {
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
static void cached_free(void* p) {
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

template <class T>
int dev_alloc(T** p, size_t count) {
    if (count == 0) count = 1;
    *p = static_cast<T*>(cached_alloc(count * sizeof(T)));
    if (!*p) return fail(LT_ERR_NOMEM, "hipMalloc(%zu bytes) failed", count * sizeof(T));
    return LT_OK;
}
template <class T>
void dev_free(T*& p) {
    if (p) cached_free(p);
    p = nullptr;
}

int sync_all(lt_ctx* c) {
    for (int i = 0; i < c->nstreams && i < (int)c->streams.size(); ++i) HIP_TRY(hipStreamSynchronize(c->streams[i]));
    if (c->copy) HIP_TRY(hipStreamSynchronize(c->copy));
    if (c->search) HIP_TRY(hipStreamSynchronize(c->search));
    if (c->present) HIP_TRY(hipStreamSynchronize(c->present));
    if (c->urgent) HIP_TRY(hipStreamSynchronize(c->urgent));
    if (c->dl) HIP_TRY(hipStreamSynchronize(c->dl));
    c->spans_busy.lo = c->spans_busy.hi = 0;            // every overlay, every copy of the rest rows
    c->text_busy.lo = c->text_busy.hi = 0;
    c->annot_busy.lo = c->annot_busy.hi = 0;
    c->rest_pending = false;
    c->readers.reset();                                 // every reader / writer enqueued so far is done
    c->writers.reset();
    c->rests.reset();
    return LT_OK;
}

// work touching slots [lo, hi) has just been enqueued on `st`
int note_range(lt_ctx::RangeEvents& r, hipStream_t st, int lo, int hi) {
    unsigned cap = (unsigned)r.e.size();
    while (r.count > 0 && hipEventQuery(r.e[r.head].ev) == hipSuccess) {   // finished: nobody has to wait for it any more
        r.head = (r.head + 1) % cap;
        --r.count;
    }
    if (r.count == cap && cap < 4096) {        // everything in flight: a longer ring (the live entries first, in order)
        std::vector<lt_ctx::RangeEvents::Entry> bigger(2 * (size_t)cap, lt_ctx::RangeEvents::Entry{0, 0, nullptr});
        for (unsigned i = 0; i < r.count; ++i) bigger[i] = r.e[(r.head + i) % cap];
        r.e.swap(bigger);
        r.head = 0;
        cap *= 2;
    }
    if (r.count == cap) {
        r.overflow = true;
        r.head = (r.head + 1) % cap;
        --r.count;
    }
    lt_ctx::RangeEvents::Entry& w = r.e[(r.head + r.count) % cap];
    if (!w.ev && hipEventCreateWithFlags(&w.ev, hipEventDisableTiming) != hipSuccess) return fail(LT_ERR_HIP, "hipEventCreate failed");
    HIP_TRY(hipEventRecord(w.ev, st));
    w.lo = lo;
    w.hi = hi;
    ++r.count;
    return LT_OK;
}
// `waiter` waits for the entries that touch slots [lo, hi); *precise = false if the ring has overflowed (the caller then waits
// for stream tails)
int wait_range(const lt_ctx::RangeEvents& r, hipStream_t waiter, int lo, int hi, bool* precise) {
    *precise = !r.overflow;
    if (r.overflow) return LT_OK;
    for (unsigned i = 0; i < r.count; ++i) {
        const lt_ctx::RangeEvents::Entry& w = r.e[(r.head + i) % (unsigned)r.e.size()];
        if (w.lo < hi && w.hi > lo) HIP_TRY(hipStreamWaitEvent(waiter, w.ev, 0));
    }
    return LT_OK;
}
int note_written(lt_ctx* c, hipStream_t st, int lo, int hi) { return note_range(c->writers, st, lo, hi); }



// Slot -> stream mapping is fixed (contiguous slices of the capacity), so consecutive stages of one
// slot stay ordered on one stream while different slices overlap: the latency-bound search of one
// slice runs under the mask chain of another.  Calls fn(stream, first, n) for every non-empty piece.
hipEvent_t next_order_event(lt_ctx* c);
// `st` waits for the chains still outstanding (not collected) that read or write slots [lo, hi): ticket by ticket, so that work on
// a frame in front of a running chain -- the second try of a failed frame while the frames behind it are already chained -- does
// not wait for that chain
static int wait_chains(lt_ctx* c, hipStream_t st, int lo, int hi) {
    for (const auto& t : c->chains)
        if (t.first < hi && t.first + t.n > lo) HIP_TRY(hipStreamWaitEvent(st, t.done, 0));
    return LT_OK;
}
template <class F>
int for_each_slice(lt_ctx* c, int first, int n, F fn) {
    const int k = std::max(1, std::min(c->nstreams, c->capacity));
    if (c->urgent_on && c->urgent) {
        // one piece on the urgent stream: behind the kernels that wrote these slots (or, with the ring overflowed, the tails of
        // their streams) and a chain still touching them; the slots' own streams then wait for it, so that whatever is
        // enqueued for these slots later stays ordered behind it
        hipStream_t us = c->urgent;
        bool precise = true;
        int rc = wait_range(c->writers, us, first, first + n, &precise);
        if (rc) return rc;
        auto slices = [&](auto g) {
            for (int si = 0; si < k; ++si) {
                const int lo = (int)((long long)c->capacity * si / k) & ~1, hi = si + 1 == k ? c->capacity : (int)((long long)c->capacity * (si + 1) / k) & ~1;
                if (std::min(first + n, hi) > std::max(first, lo)) { int r = g(c->streams[si]); if (r) return r; }
            }
            return (int)LT_OK;
        };
        if (!precise) {
            rc = slices([&](hipStream_t st) {
                hipEvent_t e = next_order_event(c);
                if (!e) return fail(LT_ERR_HIP, "hipEventCreate failed");
                HIP_TRY(hipEventRecord(e, st));
                HIP_TRY(hipStreamWaitEvent(us, e, 0));
                return (int)LT_OK;
            });
            if (rc) return rc;
        }
        if ((rc = wait_chains(c, us, first, first + n))) return rc;
        if ((rc = fn(us, first, n))) return rc;
        hipEvent_t done = next_order_event(c);
        if (!done) return fail(LT_ERR_HIP, "hipEventCreate failed");
        HIP_TRY(hipEventRecord(done, us));
        return slices([&](hipStream_t st) {
            HIP_TRY(hipStreamWaitEvent(st, done, 0));
            return (int)LT_OK;
        });
    }
    for (int si = 0; si < k; ++si) {
        // even boundaries: the undistorted rows of slots 2p and 2p+1 are interleaved, and the warp serves a pair with one load
        const int lo = (int)((long long)c->capacity * si / k) & ~1, hi = si + 1 == k ? c->capacity : (int)((long long)c->capacity * (si + 1) / k) & ~1;
        const int a = std::max(first, lo), b = std::min(first + n, hi);
        if (b <= a) continue;
        int rc = wait_chains(c, c->streams[si], a, b);                     // a chain in flight reads / writes these slots
        if (rc) return rc;
        rc = fn(c->streams[si], a, b - a);
        if (rc) return rc;
    }
    return LT_OK;
}

int flush_stage_events(lt_ctx* c) {
    if (c->pending.empty()) return LT_OK;
    { int rc = sync_all(c); if (rc) return rc; }
    for (auto& p : c->pending) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p.a, p.b));
        c->stage_ms[p.stage] += ms;
        c->stage_launches[p.stage] += 1;
    }
    c->pending.clear();
    c->ev_used = 0;
    return LT_OK;
}

// RAII-free helper pair: bracket one kernel launch with events when stage timing is on
struct StageScope {
    lt_ctx* c;
    int stage;
    hipStream_t st;
    hipEvent_t a = nullptr, b = nullptr;
    StageScope(lt_ctx* c_, int stage_, hipStream_t st_ = nullptr) : c(c_), stage(stage_), st(st_ ? st_ : c_->stream) {
        if (!c->stage_timing) return;
        if (c->ev_used + 2 > c->ev_pool.size()) {
            if (flush_stage_events(c) != LT_OK) return;
            while (c->ev_pool.size() < 256) {
                hipEvent_t e;
                if (hipEventCreate(&e) != hipSuccess) break;
                c->ev_pool.push_back(e);
            }
        }
        if (c->ev_used + 2 > c->ev_pool.size()) return;
        a = c->ev_pool[c->ev_used++];
        b = c->ev_pool[c->ev_used++];
        (void)hipEventRecord(a, st);
    }
    ~StageScope() {
        if (!a) return;
        (void)hipEventRecord(b, st);
        c->pending.push_back({stage, a, b});
    }
};

hipEvent_t next_order_event(lt_ctx* c) {
    constexpr size_t RING = 64;
    if (c->order_events.size() < RING) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr;
        c->order_events.push_back(e);
        return e;
    }
    hipEvent_t e = c->order_events[c->order_next];
    c->order_next = (c->order_next + 1) % RING;
    return e;
}

int check_slots(lt_ctx* c, int first, int n) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    if (first < 0 || n < 0 || first + n > c->capacity)
        return fail(LT_ERR_CAPACITY, "slots [%d, %d) outside reserved capacity %d", first, first + n, c->capacity);
    return LT_OK;
}

int set_device(lt_ctx* c) {
    HIP_TRY(hipSetDevice(c->device));
    return LT_OK;
}

void free_slots(lt_ctx* c) {
    dev_free(c->d_frames);
    dev_free(c->d_und);
    dev_free(c->d_bev);
    for (auto& p : c->d_plane) dev_free(p);
    dev_free(c->d_bits_merged);
    dev_free(c->d_bits_eroded);
    dev_free(c->d_bits_open);
    dev_free(c->d_bits_tmp);
    dev_free(c->d_bits_tmp2);
    dev_free(c->d_th_pad[0]);
    dev_free(c->d_th_pad[1]);
    dev_free(c->d_b_pad);
    dev_free(c->d_bits_n1);
    dev_free(c->d_bits_n2);
    c->th_padded.clear();
    c->mask_bits_ok.clear();
    c->mask_u8_ok.clear();
    dev_free(c->d_rec);
    dev_free(c->d_prev);
    dev_free(c->d_pix);
    dev_free(c->d_cent);
    dev_free(c->d_band_sums);
    dev_free(c->d_spans);
    dev_free(c->d_annot);
    c->maxbands = 0;
    c->capacity = 0;
    c->maxpix = 0;
    c->maxlev = 0;
    c->have_mask = false;
}

// Grow the per-slot result buffers.  Results of earlier searches stay readable (a tracker may fetch its lane
// pixels lazily, after a later search with other parameters enlarged the buffers): the old rows -- one per
// (slot, side) -- are copied to their new positions.
template <class T>
int grow_rows(lt_ctx* c, T** buf, size_t old_row, size_t new_row) {
    T* fresh = nullptr;
    int rc = dev_alloc(&fresh, (size_t)c->capacity * 2 * new_row);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(fresh, 0, (size_t)c->capacity * 2 * new_row * sizeof(T), c->stream));
    if (*buf && old_row)
        HIP_TRY(hipMemcpy2DAsync(fresh, new_row * sizeof(T), *buf, old_row * sizeof(T), old_row * sizeof(T),
                                 (size_t)c->capacity * 2, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    dev_free(*buf);
    *buf = fresh;
    return LT_OK;
}

int ensure_search_buffers(lt_ctx* c, int maxpix, int maxlev) {
    if (maxpix > c->maxpix) {
        { int rc = sync_all(c); if (rc) return rc; }
        int rc = grow_rows(c, &c->d_pix, (size_t)c->maxpix, (size_t)maxpix);
        if (rc) return rc;
        c->maxpix = maxpix;
    }
    if (maxlev > c->maxlev) {
        { int rc = sync_all(c); if (rc) return rc; }
        int rc = grow_rows(c, &c->d_cent, c->maxlev ? (size_t)c->maxlev + 2 : 0, (size_t)maxlev + 2);
        if (rc) return rc;
        c->maxlev = maxlev;
    }
    return LT_OK;
}

int ensure_bev(lt_ctx* c) {
    if (c->d_bev) return LT_OK;
    return dev_alloc(&c->d_bev, (size_t)c->capacity * c->bev_bytes);
}

void mark_masks(lt_ctx* c, int first, int n, int bits_ok, int u8_ok) {
    for (int i = first; i < first + n && i < (int)c->mask_bits_ok.size(); ++i) {
        c->mask_bits_ok[(size_t)i] = (uint8_t)bits_ok;
        c->mask_u8_ok[(size_t)i] = (uint8_t)u8_ok;
    }
}
bool masks_have_bits(const lt_ctx* c, int first, int n) {
    for (int i = first; i < first + n; ++i)
        if (!c->mask_bits_ok[(size_t)i]) return false;
    return true;
}
// make d_plane[P_MASK] current for the slots (expands the bit plane where only that exists)
int ensure_u8_masks(lt_ctx* c, int first, int n) {
    bool any = false;
    for (int i = first; i < first + n; ++i) any = any || !c->mask_u8_ok[(size_t)i];
    if (!any) return LT_OK;
    int rc = sync_all(c);
    if (rc) return rc;
    for (int i = first; i < first + n;) {
        if (c->mask_u8_ok[(size_t)i]) { ++i; continue; }
        int j = i;
        while (j < first + n && !c->mask_u8_ok[(size_t)j]) ++j;
        launch_bits_to_u8(c->stream, c->d_bits_open + (size_t)i * c->bits_stride, c->d_plane[P_MASK] + (size_t)i * c->plane_bytes,
                          c->calib.warp_h, c->calib.warp_w, c->plane_bytes, c->bits_stride, j - i);
        for (int k = i; k < j; ++k) c->mask_u8_ok[(size_t)k] = 1;
        i = j;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LT_OK;
}

int ensure_noise_buffers(lt_ctx* c) {
    if (c->d_b_pad && c->d_bits_n1 && c->d_bits_n2) return LT_OK;
    const size_t n = (size_t)c->capacity;
    int rc;
    if (!c->d_b_pad && (rc = dev_alloc(&c->d_b_pad, n * c->th_pad_bytes))) return rc;
    if (!c->d_bits_n1 && (rc = dev_alloc(&c->d_bits_n1, n * c->bits_stride))) return rc;
    if (!c->d_bits_n2 && (rc = dev_alloc(&c->d_bits_n2, n * c->bits_stride))) return rc;
    return LT_OK;
}

int validate_filter(const lt_filter_params* p) {
    if (!p) return fail(LT_ERR_INVALID, "null filter params");
    if (p->filter_type != 0 && p->filter_type != 1)
        return fail(LT_ERR_INVALID, "Unexpected filter mode. Expected modes are 'bilateral' or 'neighborhood'.");
    if (p->ksize_r < 1 || p->ksize_b < 1 || (p->mask_noise && p->ksize_noise < 1))
        return fail(LT_ERR_INVALID, "filter sizes must be >= 1");
    if (p->filter_type == 1 && ((p->ksize_r & 1) == 0 || (p->ksize_b & 1) == 0))
        return fail(LT_ERR_INVALID, "'neighborhood' block sizes must be odd (cv2.adaptiveThreshold requirement)");
    if (p->ksize_r > 128 || p->ksize_b > 128 || p->ksize_noise > 128)
        return fail(LT_ERR_INVALID, "filter size too large (max 128)");
    return LT_OK;
}

// filter_lane_points() on planes P_R / P_B of the given slots (lane_tracker.py:210-238)
int run_filter_chain(lt_ctx* c, hipStream_t s, int first, int n, const lt_filter_params* p, int h, int w, int call_frames,
                     bool u8_mask = false) {
    const size_t ps = c->plane_bytes, off = (size_t)first * ps;
    uint8_t* R = c->d_plane[P_R] + off;
    uint8_t* B = c->d_plane[P_B] + off;
    uint8_t* thR = c->d_plane[P_THR] + off;
    uint8_t* thB = c->d_plane[P_THB] + off;
    uint8_t* t0 = c->d_plane[P_T0] + off;
    uint8_t* t1 = c->d_plane[P_T1] + off;
    uint8_t* t2 = c->d_plane[P_T2] + off;
    uint8_t* t3 = c->d_plane[P_T3] + off;
    uint8_t* merged = c->d_plane[P_MERGED] + off;
    uint8_t* mask = c->d_plane[P_MASK] + off;
    // the walking threshold kernels read the top-hat planes with a padded row pitch: the dilate launches write them so
    // (the greenery mask, mask_noise, rides along: a third walk with window 65 over the raw Lab-b plane)
    const bool walk = p->filter_type == 0 && !c->brute_tophat && c->d_th_pad[0] && c->d_bits_tmp && c->d_bits_tmp2 &&
                      first + n <= (int)c->th_padded.size() && (long long)call_frames * h * w >= c->walk_min_pixels &&
                      bilateral_walk_supported(p->ksize_r, p->C_r, p->ksize_b, p->C_b, h, w, c->th_pitch) &&
                      (!p->mask_noise || noise_walk_supported(p->ksize_noise, p->C_noise, h, w, c->th_pitch));
    const bool walk_noise = walk && p->mask_noise;
    if (walk_noise) {
        const int rc = ensure_noise_buffers(c);
        if (rc) return rc;
    }
    uint8_t* bpad = walk_noise ? c->d_b_pad + (size_t)first * c->th_pad_bytes : nullptr;
    if (p->filter_type == 0) c->last_threshold_path = walk ? 1 : 0;
    const int dpitch = walk ? c->th_pitch : 0;
    uint8_t* thRd = walk ? c->d_th_pad[0] + (size_t)first * c->th_pad_bytes : thR;
    uint8_t* thBd = walk ? c->d_th_pad[1] + (size_t)first * c->th_pad_bytes : thB;
    // the 55x55 top-hat of the Lab-b plane; with the greenery mask it also leaves the raw plane in the padded layout
    auto tophat_b = [&](hipStream_t st) -> int {
        if (bpad && launch_morph_runs(st, t0, thBd, B, h, w, 55, true, ps, n, dpitch, c->th_pad_bytes, bpad)) return LT_OK;
        launch_morph_runs(st, t0, thBd, B, h, w, 55, true, ps, n, dpitch, c->th_pad_bytes);
        if (bpad)   // that kernel form does not exist for this geometry / A-B switch: plain strided copies
            for (int i = 0; i < n; ++i)
                HIP_TRY(hipMemcpy2DAsync(bpad + (size_t)i * c->th_pad_bytes, (size_t)c->th_pitch, B + (size_t)i * ps, (size_t)w, (size_t)w,
                                         (size_t)h, hipMemcpyDeviceToDevice, st));
        return LT_OK;
    };
    if (p->filter_type == 0 && first + n <= (int)c->th_padded.size())
        for (int i = first; i < first + n; ++i) c->th_padded[(size_t)i] = walk ? 1 : 0;
    unsigned long long* mbits = c->d_bits_merged + (size_t)first * c->bits_stride;
    unsigned long long* ebits = c->d_bits_eroded + (size_t)first * c->bits_stride;
    bool r_verdicts_done = false;                 // the R plane's threshold ran beside the Lab-b top-hats, into ebits
    if (p->filter_type == 0) {
        if (c->brute_tophat) {   // debugging aid (LT_TOPHAT_BRUTE=1): direct footprint evaluation, still on the GPU
            { StageScope t(c, ST_ERODE_R, s);  launch_morph_ellipse(s, R, t0, nullptr, h, w, c->se29, false, ps, n); }
            { StageScope t(c, ST_TOPHAT_R, s); launch_morph_ellipse(s, t0, thR, R, h, w, c->se29, true, ps, n); }
            { StageScope t(c, ST_ERODE_B, s);  launch_morph_ellipse(s, B, t0, nullptr, h, w, c->se55, false, ps, n); }
            { StageScope t(c, ST_TOPHAT_B, s); launch_morph_ellipse(s, t0, thB, B, h, w, c->se55, true, ps, n); }
        } else if (n <= 2 && !c->stage_timing && c->side && s == c->stream) {
            // One or two frames cannot fill the chip (a few hundred waves per top-hat kernel), so the two planes'
            // top-hats, which do not depend on each other, run side by side: the R plane on the side stream, the
            // Lab-b plane here; saves the shorter pair's ~40 us of a 160 us chain.  t3 is free until the merge.
            HIP_TRY(hipEventRecord(c->ev_fork, s));
            HIP_TRY(hipStreamWaitEvent(c->side, c->ev_fork, 0));
            launch_morph_runs(c->side, R, t3, nullptr, h, w, 29, false, ps, n);
            launch_morph_runs(c->side, t3, thRd, R, h, w, 29, true, ps, n, dpitch, c->th_pad_bytes);
            // ... and so does the R plane's threshold: its 29x29 top-hat is done while the 55x55 pair still has half its way
            // to go, and the threshold kernel takes its planes one after the other anyway -- here one per launch, the R
            // verdicts as a partial bit plane the open ORs in (12 us less on the one-frame chain)
            static const bool split_ok = [] { const char* e = std::getenv("LT_THRESHOLD_SPLIT"); return !(e && e[0] == '0'); }();
            if (split_ok && !walk && !p->mask_noise)
                r_verdicts_done = launch_bilateral_bits(c->side, thR, p->ksize_r, p->C_r, nullptr, 1, 0, B, p->ksize_noise, p->C_noise,
                                                        p->noise_thresh, 0, ebits, h, w, ps, c->bits_stride, n) == 0;
            HIP_TRY(hipEventRecord(c->ev_join, c->side));
            launch_morph_runs(s, B, t0, nullptr, h, w, 55, false, ps, n);
            { const int rc = tophat_b(s); if (rc) return rc; }
            HIP_TRY(hipStreamWaitEvent(s, c->ev_join, 0));
        } else {
            { StageScope t(c, ST_ERODE_R, s);  launch_morph_runs(s, R, t0, nullptr, h, w, 29, false, ps, n); }
            { StageScope t(c, ST_TOPHAT_R, s); launch_morph_runs(s, t0, thRd, R, h, w, 29, true, ps, n, dpitch, c->th_pad_bytes); }
            { StageScope t(c, ST_ERODE_B, s);  launch_morph_runs(s, B, t0, nullptr, h, w, 55, false, ps, n); }
            { StageScope t(c, ST_TOPHAT_B, s); const int rc = tophat_b(s); if (rc) return rc; }
        }
    }
    bool merged_done = false, partials = false;   // partials: mbits, ebits, tmp, tmp2 still wait for their OR
    bool two_partials = false;                    // ... only mbits and ebits (the 'neighborhood' walk)
    unsigned long long* tbits = c->d_bits_tmp + (size_t)first * c->bits_stride;
    unsigned long long* ubits = c->d_bits_tmp2 + (size_t)first * c->bits_stride;
    unsigned long long *nbits1 = nullptr, *nbits2 = nullptr;   // the greenery mask of the walking kernels: n1 | n2
    if (p->filter_type == 0) {
        StageScope t(c, ST_THRESHOLD, s);   // both bilateral thresholds, the greenery mask and the OR-merge
        // long-walk kernels for the supported window sizes; their four partial planes are merged on the way into the open
        if (walk) {
            merged_done = launch_bilateral_walk(s, thRd, p->ksize_r, p->C_r, thBd, p->ksize_b, p->C_b, mbits, ebits, tbits, ubits,
                                                h, w, c->th_pitch, c->th_pad_bytes, c->bits_stride, n, false) == 0;
            partials = merged_done;
            if (merged_done && walk_noise) {
                nbits1 = c->d_bits_n1 + (size_t)first * c->bits_stride;
                nbits2 = c->d_bits_n2 + (size_t)first * c->bits_stride;
                if (launch_noise_walk(s, bpad, p->ksize_noise, p->C_noise, p->noise_thresh, nbits1, nbits2, h, w, c->th_pitch,
                                      c->th_pad_bytes, c->bits_stride, n))
                    return fail(LT_ERR_STATE, "the greenery-mask walk refused parameters its own predicate accepted");
            }
        }
        if (!merged_done && r_verdicts_done) {   // the Lab-b plane alone; should that launch be refused, both planes below
            merged_done = launch_bilateral_bits(s, nullptr, 1, 0, thB, p->ksize_b, p->C_b, B, p->ksize_noise, p->C_noise,
                                                p->noise_thresh, 0, mbits, h, w, ps, c->bits_stride, n) == 0;
            partials = two_partials = merged_done;
        }
        if (!merged_done)
          merged_done = launch_bilateral_bits(s, thR, p->ksize_r, p->C_r, thB, p->ksize_b, p->C_b, B, p->ksize_noise,
                                            p->C_noise, p->noise_thresh, p->mask_noise ? 1 : 0, mbits, h, w, ps,
                                            c->bits_stride, n) == 0;
        if (!merged_done) {              // tile + halo exceeds the LDS: one plane at a time
            launch_bilateral(s, thR, t1, h, w, p->ksize_r, p->C_r, 0, 255, 0, ps, n);
            launch_bilateral(s, thB, t2, h, w, p->ksize_b, p->C_b, 0, 255, 0, ps, n);
        }
    } else {
        StageScope t(c, ST_THRESHOLD, s);
        // running box sums, both planes in one launch, bit planes out (merged on the way into the open); the per-pixel
        // window kernel for what that does not take (window > 63, a width that is not a multiple of 4, the greenery mask)
        if (!p->mask_noise && launch_adaptive_walk(s, R, p->ksize_r, p->C_r, mbits, B, p->ksize_b, p->C_b, ebits, h, w, ps, c->bits_stride, n)) {
            merged_done = true;
            partials = true;
            two_partials = true;
        } else {
            launch_adaptive_mean(s, R, t1, h, w, p->ksize_r, p->C_r, ps, n);
            launch_adaptive_mean(s, B, t2, h, w, p->ksize_b, p->C_b, ps, n);
        }
        c->last_adaptive_path = two_partials ? 1 : 0;
    }
    if (!merged_done) {
        if (p->mask_noise) {
            StageScope t(c, ST_THRESHOLD, s);
            launch_bilateral(s, B, t3, h, w, p->ksize_noise, p->C_noise, 0, 255, 0, ps, n);
        }
        StageScope t(c, ST_MERGE, s);
        launch_pack_merge(s, t1, t2, B, t3, p->noise_thresh, p->mask_noise ? 1 : 0, mbits, h, w, ps, c->bits_stride, n);
    }
    { StageScope t(c, ST_OPEN, s);
      unsigned long long* obits = c->d_bits_open + (size_t)first * c->bits_stride;
      bool opened = false;
      // one pass over the words; a handful of frames is latency-bound and better off with the wide, shallow kernels
      // (also with partial planes to OR first: one frame's chain, wall time from the first launch to the record, 134.4 us through
      // k_merge_open5 against 129.9 through k_or4_bits + the two shallow kernels; LT_OPEN_SHALLOW=0 restores the former)
      static const bool deep_small = [] { const char* e = std::getenv("LT_OPEN_SHALLOW"); return e && e[0] == '0'; }();
      if (!u8_mask && (n >= 16 || (partials && deep_small)))
          opened = launch_merge_open5(s, mbits, partials ? ebits : nullptr, two_partials ? nullptr : tbits, two_partials ? nullptr : ubits, obits,
                                      h, w, c->bits_stride, n, nbits1, nbits2);
      if (!opened) {
          if (partials) launch_or4_bits(s, mbits, ebits, two_partials ? ebits : tbits, two_partials ? ebits : ubits, h, w, c->bits_stride, n, nbits1, nbits2);
          if (u8_mask) launch_open5_bits(s, mbits, ebits, mask, h, w, ps, c->bits_stride, n);
          else launch_open5_to_bits(s, mbits, ebits, obits, h, w, c->bits_stride, n);
      } }
    (void)merged; (void)t0;
    HIP_TRY(hipGetLastError());
    return LT_OK;
}

int make_search_geom(lt_ctx* c, const lt_search_params* p, bool band, SearchGeom& g) {
    if (!p) return fail(LT_ERR_INVALID, "null search params");
    const int h = c->calib.warp_h, w = c->calib.warp_w;
    std::memset(&g, 0, sizeof g);
    g.h = h;
    g.w = w;
    if (p->ignore_bottom < 0 || p->ignore_bottom > h) return fail(LT_ERR_INVALID, "ignore_bottom out of range");
    if (!(p->partial >= 0.0 && p->partial <= 1.0)) return fail(LT_ERR_INVALID, "partial must be in [0,1]");
    g.img_height = h - p->ignore_bottom;                                      // lane_tracker.py:277
    if (band) {
        if (p->bandwidth < 0) return fail(LT_ERR_INVALID, "bandwidth must be >= 0");
        g.bandwidth = (double)p->bandwidth;
        g.band_bottom = h - p->ignore_bottom;                                 // :465
        g.band_top = (int)((double)h * (1.0 - p->partial));                   // :466 (2017 NumPy: int())
        if (g.band_top < 0) g.band_top = 0;
        const long long per_row = std::min<long long>(w, 2LL * p->bandwidth + 2);
        long long need = (long long)std::max(g.band_bottom - g.band_top, 0) * per_row;
        g.maxpix = (int)std::min<long long>(std::max<long long>(need, 64), (long long)h * w);
        g.maxlev = 1;
        return LT_OK;
    }
    if (p->window_width < 1 || p->window_height < 1) return fail(LT_ERR_INVALID, "window size must be >= 1");
    if (p->window_height > h) return fail(LT_ERR_INVALID, "window_height exceeds the image height");
    if (p->ignore_sides < 0 || p->search_range < 0) return fail(LT_ERR_INVALID, "negative margin/range");
    g.ww = p->window_width;
    g.wh = p->window_height;
    g.hw = (int)(p->window_width / 2.0);                                      // int(window_width/2)
    g.img_center = (int)(w / 2.0);                                            // :278
    g.y_start = (int)((1.0 - p->start_slice) * g.img_height);                 // :279
    g.nlevels = (int)((p->partial * g.img_height) / p->window_height);        // :282
    if (g.nlevels < 0) g.nlevels = 0;
    g.limit = p->no_success_limit;
    g.ignore_sides = p->ignore_sides;
    g.search_range = p->search_range;
    g.def_left = (int)(w * 0.4);                                              // :308
    g.def_right = (int)(w * 0.6);                                             // :328
    g.mu = p->mu;
    const long long need = (long long)std::max(g.nlevels, 1) * g.wh * std::min(2 * g.hw, w);
    g.maxpix = (int)std::max<long long>(need, 64);
    g.maxlev = std::max(g.nlevels, 1) + 1;
    g.nbands = std::max(g.nlevels, 1);
    return LT_OK;
}

}  // namespace

// ================================================================================================
extern "C" {

const char* lt_last_error(void) { return g_err.c_str(); }
int lt_abi_version(void) { return LT_ABI_VERSION; }

int lt_device_count(int* count) {
    if (!count) return fail(LT_ERR_INVALID, "null count");
    HIP_TRY(hipGetDeviceCount(count));
    return LT_OK;
}

const char* lt_stage_name(int stage) { return stage >= 0 && stage < LT_NUM_STAGES ? kStageNames[stage] : ""; }

// The streams that carry the slot slices' kernels.  The HIP runtime multiplexes the streams of a process onto a
// small pool of hardware queues PER PRIORITY LEVEL (4 by default), in creation order -- so in a process that already
// holds other streams (torch's, RCCL's) two slices can land on one queue and stop overlapping (measured: 9 % of the
// batch rate under torch.distributed).  The slices therefore take the highest priority level, whose pool nothing
// else in the process uses; LT_STREAM_PRIORITY=normal restores plain streams.
// reserved > 0 (lt_set_search_cus): the stream is kept off CUs 0 .. reserved-1 (bits of the CU mask), which the search stream
// has to itself -- see lt_set_search_cus.  hipExtStreamCreateWithCUMask takes no priority, so a CU-masked stream has the
// runtime's default priority and LT_STREAM_PRIORITY has no effect on it: the priority only serves to put the slices of an
// independent-batch context on separate hardware queues (contexts that never call lt_set_search_cus), while a stream
// context runs its slices back to back behind the bus anyway.
static hipError_t create_compute_stream(hipStream_t* st, int reserved = 0) {
    if (reserved > 0) {
        uint32_t mask[8];
        for (auto& w : mask) w = 0xffffffffu;
        for (int i = 0; i < reserved && i < 256; ++i) mask[i >> 5] &= ~(1u << (i & 31));
        return hipExtStreamCreateWithCUMask(st, 8, mask);
    }
    const char* e = getenv("LT_STREAM_PRIORITY");
    int least = 0, greatest = 0;
    if ((e && strcmp(e, "normal") == 0) || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest)
        return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    return hipStreamCreateWithPriority(st, hipStreamNonBlocking, greatest);
}

int lt_create(const lt_calib* calib, int device, lt_ctx** out) {
    if (!calib || !out) return fail(LT_ERR_INVALID, "null argument");
    if (calib->img_w < 2 || calib->img_h < 2 || calib->warp_w < 2 || calib->warp_h < 2 || calib->img_w > 16384 ||
        calib->img_h > 16384 || calib->warp_w > 4096 || calib->warp_h > 16384)
        return fail(LT_ERR_INVALID, "camera size must be in [2, 16384], bird's-eye width in [2, 4096], height in [2, 16384]");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(LT_ERR_HIP, "no HIP device visible: the lane-tracker kernels need a GPU (gfx950)");
    if (device < 0 || device >= ndev) return fail(LT_ERR_INVALID, "device %d out of range (%d visible)", device, ndev);
    lt_ctx* c = new lt_ctx();
    c->calib = *calib;
    c->device = device;
    auto bail = [&](int rc) {
        lt_destroy(c);
        return rc;
    };
    if (const char* e = getenv("LT_WALK_MIN_FRAMES")) c->walk_min_pixels = atoll(e) * calib->warp_w * calib->warp_h;
    if (hipSetDevice(device) != hipSuccess) return bail(fail(LT_ERR_HIP, "hipSetDevice(%d) failed", device));
    if (hipGetDeviceProperties(&c->prop, device) != hipSuccess) return bail(fail(LT_ERR_HIP, "hipGetDeviceProperties failed"));
    if (create_compute_stream(&c->stream) != hipSuccess) return bail(fail(LT_ERR_HIP, "hipStreamCreate failed"));
    c->streams.assign(1, c->stream);
    c->nstreams = 1;
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) return bail(fail(LT_ERR_HIP, "hipEventCreate failed"));
    if (hipStreamCreateWithFlags(&c->copy, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess)
        return bail(fail(LT_ERR_HIP, "side stream / event creation failed"));

    // host tables
    RemapTable warp, und;
    build_warp_table(*calib, warp);
    int r0 = 0, r1 = 0;
    warp_source_rows(*calib, warp, r0, r1);
    build_undistort_table(*calib, r0, r1, und);
    c->fe = FrontEndGeom{calib->img_h, calib->img_w, calib->warp_h, calib->warp_w, r0, r1 - r0};
    {
        int lo = calib->img_h, hi = 0;
        for (size_t o = 0; o < (size_t)und.rows * und.cols; ++o) {   // (the vectors carry one padding entry)
            const int sx = und.xy[o * 2], sy = und.xy[o * 2 + 1];
            if (sy < -1 || sy >= calib->img_h || sx < -1 || sx >= calib->img_w) continue;   // every tap outside: reads as 0
            lo = std::min(lo, std::max(sy, 0));
            hi = std::max(hi, std::min(sy + 2, calib->img_h));
        }
        c->cam_r0 = hi > lo ? lo : 0;
        c->cam_r1 = hi > lo ? hi : 0;
    }
    uint16_t gamma_tab[256], cbrt_tab[3072];
    int32_t coef[9];
    build_lab_tables(gamma_tab, cbrt_tab, coef);
    auto make_se = [](int k, EllipseSE& se) {
        int dx[64];
        ellipse_halfwidths(k, dx);
        se.k = k;
        for (int i = 0; i < 64; ++i) se.dx[i] = (int8_t)(i < k ? dx[i] : 0);
    };
    make_se(5, c->se5);
    make_se(29, c->se29);
    make_se(55, c->se55);
    if (!tophat_tables_match(c->se29, c->se55))
        return bail(fail(LT_ERR_STATE, "compiled-in ellipse run tables disagree with getStructuringElement's formula"));
    {
        const char* e = std::getenv("LT_TOPHAT_BRUTE");
        c->brute_tophat = e && e[0] == '1';
    }

    int rc;
    if ((rc = dev_alloc(&c->d_wxy, warp.xy.size()))) return bail(rc);
    if ((rc = dev_alloc(&c->d_wfrac, warp.frac.size()))) return bail(rc);
    if ((rc = dev_alloc(&c->d_uxy, und.xy.size()))) return bail(rc);
    if ((rc = dev_alloc(&c->d_ufrac, und.frac.size()))) return bail(rc);
    if ((rc = dev_alloc(&c->d_gamma, 256))) return bail(rc);
    if ((rc = dev_alloc(&c->d_cbrt, 3072))) return bail(rc);
    if ((rc = dev_alloc(&c->d_coef, 9))) return bail(rc);
    hipError_t e = hipSuccess;
    auto up = [&](void* dst, const void* src, size_t bytes) {
        if (e == hipSuccess && bytes) e = hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
    };
    up(c->d_wxy, warp.xy.data(), warp.xy.size() * 2);
    up(c->d_wfrac, warp.frac.data(), warp.frac.size() * 2);
    up(c->d_uxy, und.xy.data(), und.xy.size() * 2);
    up(c->d_ufrac, und.frac.data(), und.frac.size() * 2);
    up(c->d_gamma, gamma_tab, sizeof gamma_tab);
    up(c->d_cbrt, cbrt_tab, sizeof cbrt_tab);
    up(c->d_coef, coef, sizeof coef);
    if (e != hipSuccess) return bail(fail(LT_ERR_HIP, "table upload failed: %s", hipGetErrorString(e)));

    c->frame_bytes = (size_t)calib->img_h * calib->img_w * 3;
    c->und_bytes = (size_t)c->fe.nrows * calib->img_w * 3;   // as returned by lt_download_undistorted (RGB)
    c->und_px = (size_t)c->fe.nrows * calib->img_w;
    c->plane_bytes = (size_t)calib->warp_h * calib->warp_w;
    c->bev_bytes = c->plane_bytes * 3;
    *out = c;
    return LT_OK;
}

void lt_destroy(lt_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    // Everything this context has in flight ends here, on EVERY stream it owns: its device memory goes back to the cache below
    // (dev_free), not through hipFree -- which used to wait for the whole device -- and the next context may be handed the very
    // same blocks at once (a cancelled chain still runs one more frame; copies may be queued on the download stream).
    static const bool trace = std::getenv("LT_TRACE_DESTROY") != nullptr;     // where a close() that does not return is waiting
    auto note = [&](const char* what) { if (trace) { std::fprintf(stderr, "lt_destroy: %s\n", what); std::fflush(stderr); } };
    note("streams of the slots");
    for (auto st : c->streams) if (st) (void)hipStreamSynchronize(st);
    {
        const char* names[6] = {"copy", "side", "search", "present", "urgent", "dl"};
        hipStream_t sts[6] = {c->copy, c->side, c->search, c->present, c->urgent, c->dl};
        for (int i = 0; i < 6; ++i) if (sts[i]) { note(names[i]); (void)hipStreamSynchronize(sts[i]); }
    }
    note("slots");
    free_slots(c);
    note("tables and buffers");
    dev_free(c->d_uxy);
    dev_free(c->d_wxy);
    dev_free(c->d_ufrac);
    dev_free(c->d_wfrac);
    dev_free(c->d_gamma);
    dev_free(c->d_cbrt);
    dev_free(c->d_coef);
    dev_free(c->d_oxy);
    dev_free(c->d_ofrac);
    dev_free(c->d_atlas);
    dev_free(c->d_advance);
    dev_free(c->d_lines);
    dev_free(c->d_xpos);
    note("events, streams, page-locked buffers");
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    for (auto e : c->order_events) (void)hipEventDestroy(e);
    if (c->spans_busy.done) (void)hipEventDestroy(c->spans_busy.done);
    if (c->text_busy.done) (void)hipEventDestroy(c->text_busy.done);
    if (c->annot_busy.done) (void)hipEventDestroy(c->annot_busy.done);
    if (c->dl) (void)hipStreamDestroy(c->dl);
    for (auto& d : c->dl_inflight) { (void)hipEventDestroy(d.a); (void)hipEventDestroy(d.b); }
    for (auto e : c->dl_event_pool) (void)hipEventDestroy(e);
    if (c->present) (void)hipStreamDestroy(c->present);
    if (c->urgent) (void)hipStreamDestroy(c->urgent);
    if (c->rest_done) (void)hipEventDestroy(c->rest_done);
    if (c->h_spans) (void)hipHostFree(c->h_spans);
    if (c->h_lines) (void)hipHostFree(c->h_lines);
    if (c->h_xpos) (void)hipHostFree(c->h_xpos);
    for (auto& w : c->readers.e) if (w.ev) (void)hipEventDestroy(w.ev);
    for (auto& w : c->writers.e) if (w.ev) (void)hipEventDestroy(w.ev);
    for (auto& w : c->rests.e) if (w.ev) (void)hipEventDestroy(w.ev);
    for (auto& t : c->chains) (void)hipEventDestroy(t.done);
    for (auto e : c->chain_event_pool) (void)hipEventDestroy(e);
    if (c->h_small) (void)hipHostFree(c->h_small);
    if (c->h_rec) (void)hipHostFree(c->h_rec);
    if (c->h_rec_stage) (void)hipHostFree(c->h_rec_stage);
    if (c->h_cancel) (void)hipHostFree(c->h_cancel);
    if (c->search) (void)hipStreamDestroy(c->search);
    if (c->copy) { (void)hipStreamSynchronize(c->copy); (void)hipStreamDestroy(c->copy); }
    if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    for (auto st : c->streams) if (st) (void)hipStreamDestroy(st);
    if (c->streams.empty() && c->stream) (void)hipStreamDestroy(c->stream);
    note("done");
    delete c;
}

int lt_reserve(lt_ctx* c, int capacity) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    if (capacity < 1) return fail(LT_ERR_INVALID, "capacity must be >= 1");
    int rc = set_device(c);
    if (rc) return rc;
    if (capacity <= c->capacity) return LT_OK;
    if ((rc = sync_all(c))) return rc;
    free_slots(c);
    c->rec_mirror_slot = -1;
    c->capacity = capacity;
    const size_t n = (size_t)capacity;
    if ((rc = dev_alloc(&c->d_frames, n * c->frame_bytes + 16))) { free_slots(c); return rc; }   // +16: k_undistort_rows reads 8-byte windows
    if ((rc = dev_alloc(&c->d_und, (size_t)((n + 1) / 2) * 2 * c->und_px))) { free_slots(c); return rc; }
    for (int i = 0; i < P_COUNT; ++i)
        if ((rc = dev_alloc(&c->d_plane[i], n * c->plane_bytes))) { free_slots(c); return rc; }
    c->bits_stride = (size_t)c->calib.warp_h * ((c->calib.warp_w + 63) / 64);
    if ((rc = dev_alloc(&c->d_bits_merged, n * c->bits_stride))) { free_slots(c); return rc; }
    if ((rc = dev_alloc(&c->d_bits_eroded, n * c->bits_stride))) { free_slots(c); return rc; }
    if ((rc = dev_alloc(&c->d_bits_open, n * c->bits_stride))) { free_slots(c); return rc; }
    if ((rc = dev_alloc(&c->d_bits_tmp, n * c->bits_stride))) { free_slots(c); return rc; }
    if ((rc = dev_alloc(&c->d_bits_tmp2, n * c->bits_stride))) { free_slots(c); return rc; }
    c->th_pitch = (c->calib.warp_w + 63) & ~63;
    c->th_pad_bytes = (size_t)c->calib.warp_h * c->th_pitch;
    for (auto& q : c->d_th_pad)
        if ((rc = dev_alloc(&q, n * c->th_pad_bytes))) { free_slots(c); return rc; }
    c->th_padded.assign(n, 0);
    c->mask_bits_ok.assign(n, 0);
    c->mask_u8_ok.assign(n, 1);          // zero-filled below
    if ((rc = dev_alloc(&c->d_rec, n))) { free_slots(c); return rc; }
    if ((rc = dev_alloc(&c->d_prev, n * 6))) { free_slots(c); return rc; }
    c->capacity = capacity;
    HIP_TRY(hipMemsetAsync(c->d_rec, 0, n * sizeof(lt_lane_record), c->stream));
    HIP_TRY(hipMemsetAsync(c->d_plane[P_MASK], 0, n * c->plane_bytes, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LT_OK;
}

int lt_get_info(lt_ctx* c, lt_info* out) {
    if (!c || !out) return fail(LT_ERR_INVALID, "null argument");
    std::memset(out, 0, sizeof *out);
    out->abi_version = LT_ABI_VERSION;
    out->device = c->device;
    out->capacity = c->capacity;
    out->cu_count = c->prop.multiProcessorCount;
    out->src_row0 = c->fe.r0;
    out->src_row1 = c->fe.r0 + c->fe.nrows;
    out->max_pixels_per_side = c->maxpix;
    out->max_levels = c->maxlev;
    // compulsory input rows (full width, 3 B/px) + the mask written once
    out->alg_bytes_mask = (int64_t)c->fe.nrows * c->calib.img_w * 3 + (int64_t)c->plane_bytes;
    out->alg_bytes_search = (int64_t)c->plane_bytes + (int64_t)sizeof(lt_lane_record);
    std::snprintf(out->device_name, sizeof out->device_name, "%s", c->prop.name[0] ? c->prop.name : c->prop.gcnArchName);
    return LT_OK;
}

int lt_sync(lt_ctx* c) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    return sync_all(c);
}

int lt_set_streams(lt_ctx* c, int nstreams) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    if (nstreams < 1 || nstreams > 8) return fail(LT_ERR_INVALID, "nstreams must be in [1, 8]");
    int rc = set_device(c);
    if (rc) return rc;
    if ((rc = sync_all(c))) return rc;
    if ((rc = flush_stage_events(c))) return rc;
    while ((int)c->streams.size() < nstreams) {
        hipStream_t st = nullptr;
        HIP_TRY(create_compute_stream(&st, c->search_cus));
        c->streams.push_back(st);
    }
    c->nstreams = nstreams;
    return LT_OK;
}

// ---- data movement -------------------------------------------------------------------------------
int lt_upload_frames(lt_ctx* c, const uint8_t* frames, int first, int n) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!frames) return fail(LT_ERR_INVALID, "null frames");
    if ((rc = set_device(c))) return rc;
    if ((rc = sync_all(c))) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_frames + (size_t)first * c->frame_bytes, frames, (size_t)n * c->frame_bytes,
                           hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LT_OK;
}

int lt_get_source_rows(lt_ctx* c, int* row0, int* row1) {
    if (!c || !row0 || !row1) return fail(LT_ERR_INVALID, "null argument");
    *row0 = c->cam_r0;
    *row1 = c->cam_r1;
    return LT_OK;
}

int lt_upload_frame_rows(lt_ctx* c, const uint8_t* frames, int first, int n) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!frames) return fail(LT_ERR_INVALID, "null frames");
    if (n == 0 || c->cam_r1 <= c->cam_r0) return LT_OK;
    if ((rc = set_device(c))) return rc;
    if ((rc = sync_all(c))) return rc;
    const size_t row_bytes = (size_t)c->calib.img_w * 3, off = (size_t)c->cam_r0 * row_bytes;
    const size_t bytes = (size_t)(c->cam_r1 - c->cam_r0) * row_bytes;
    HIP_TRY(hipMemcpy2DAsync(c->d_frames + (size_t)first * c->frame_bytes + off, c->frame_bytes, frames + off, c->frame_bytes,
                             bytes, (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LT_OK;
}

// Fallback of the stream-ordered uploads when the ring of readers has overflowed: `waiter` waits for the tail of every stream
// a kernel that reads camera frames can be on -- the slots' compute streams (undistortion), the presentation stream (overlays)
// and the urgent stream.
static int wait_reader_tails(lt_ctx* c, hipStream_t waiter) {
    auto tail = [&](hipStream_t st) {
        if (!st || st == waiter) return (int)LT_OK;
        hipEvent_t e = next_order_event(c);
        if (!e) return fail(LT_ERR_HIP, "hipEventCreate failed");
        HIP_TRY(hipEventRecord(e, st));
        HIP_TRY(hipStreamWaitEvent(waiter, e, 0));
        return (int)LT_OK;
    };
    for (int i = 0; i < c->nstreams && i < (int)c->streams.size(); ++i) { const int rc = tail(c->streams[i]); if (rc) return rc; }
    int rc = tail(c->present);
    if (!rc) rc = tail(c->urgent);
    return rc;
}

// The same rows, stream-ordered instead of synchronous: the copy runs on the copy stream after the work already
// enqueued on the streams that own these slots (their previous occupants), and those streams wait for it before
// anything enqueued later -- so the upload of one slot range overlaps the chain of every other slot range.
int lt_upload_frame_rows_async(lt_ctx* c, const uint8_t* frames, int first, int n) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!frames) return fail(LT_ERR_INVALID, "null frames");
    if (n == 0 || c->cam_r1 <= c->cam_r0) return LT_OK;
    if ((rc = set_device(c))) return rc;
    // the copy waits for the kernels that still read these slots' camera rows (the undistortion launches over these slots, the
    // overlay) -- not for the rest of their mask chains, and not for launches over other slots
    bool precise = true;
    if ((rc = wait_range(c->readers, c->copy, first, first + n, &precise))) return rc;
    if (!precise && (rc = wait_reader_tails(c, c->copy))) return rc;
    const size_t row_bytes = (size_t)c->calib.img_w * 3, off = (size_t)c->cam_r0 * row_bytes;
    HIP_TRY(hipMemcpy2DAsync(c->d_frames + (size_t)first * c->frame_bytes + off, c->frame_bytes, frames + off, c->frame_bytes,
                             (size_t)(c->cam_r1 - c->cam_r0) * row_bytes, (size_t)n, hipMemcpyHostToDevice, c->copy));
    hipEvent_t up = next_order_event(c);
    if (!up) return fail(LT_ERR_HIP, "hipEventCreate failed");
    HIP_TRY(hipEventRecord(up, c->copy));
    return for_each_slice(c, first, n, [&](hipStream_t st, int, int) {
        HIP_TRY(hipStreamWaitEvent(st, up, 0));
        return (int)LT_OK;
    });
}

// the overlay (on the presentation stream) waits for the copies into its own slots before it reads the frames (or, when the
// ring of slot ranges has overflowed, for the most recent copy)
static int rest_mark(lt_ctx* c, int first, int n) {
    if (!c->rest_done && hipEventCreateWithFlags(&c->rest_done, hipEventDisableTiming) != hipSuccess) return fail(LT_ERR_HIP, "hipEventCreate failed");
    HIP_TRY(hipEventRecord(c->rest_done, c->copy));
    c->rest_pending = true;
    return note_range(c->rests, c->copy, first, first + n);
}

int lt_upload_frame_rest_rows(lt_ctx* c, const uint8_t* frames, int first, int n, const int32_t* rows4) {
    if (!rows4) return lt_upload_frame_rest(c, frames, first, n);
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!frames) return fail(LT_ERR_INVALID, "null frames");
    const int H = c->calib.img_h;
    if (!(0 <= rows4[0] && rows4[0] <= rows4[1] && rows4[1] <= rows4[2] && rows4[2] <= rows4[3] && rows4[3] <= H))
        return fail(LT_ERR_INVALID, "row runs must be ordered and inside the frame");
    if (n == 0) return LT_OK;
    if ((rc = set_device(c))) return rc;
    {
        bool precise = true;
        if ((rc = wait_range(c->readers, c->copy, first, first + n, &precise))) return rc;
        if (!precise && (rc = wait_reader_tails(c, c->copy))) return rc;
    }
    // of the two runs, the rows lt_upload_frame_rows has not brought: below the window of rows the path reads, and above it
    const size_t row_bytes = (size_t)c->calib.img_w * 3;
    const int lo = c->cam_r1 > c->cam_r0 ? c->cam_r0 : 0, hi = c->cam_r1 > c->cam_r0 ? c->cam_r1 : 0;
    uint8_t* dst = c->d_frames + (size_t)first * c->frame_bytes;
    for (int k = 0; k < 4; k += 2) {
        const int piece[2][2] = {{rows4[k], std::min(rows4[k + 1], lo)}, {std::max(rows4[k], hi), rows4[k + 1]}};
        for (const auto& pc : piece) {
            if (pc[1] <= pc[0]) continue;
            const size_t off = (size_t)pc[0] * row_bytes, bytes = (size_t)(pc[1] - pc[0]) * row_bytes;
            HIP_TRY(hipMemcpy2DAsync(dst + off, c->frame_bytes, frames + off, c->frame_bytes, bytes, (size_t)n, hipMemcpyHostToDevice, c->copy));
        }
    }
    return rest_mark(c, first, n);
}

int lt_upload_frame_rest(lt_ctx* c, const uint8_t* frames, int first, int n) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!frames) return fail(LT_ERR_INVALID, "null frames");
    if (n == 0) return LT_OK;
    if ((rc = set_device(c))) return rc;
    // these rows are read by nobody but the overlay: the copy waits for the overlays still reading the frames it replaces
    // (slot-range events; a stream of windows re-uses its slots), and the overlay of these slots waits for it
    {
        bool precise = true;
        if ((rc = wait_range(c->readers, c->copy, first, first + n, &precise))) return rc;
        if (!precise && (rc = wait_reader_tails(c, c->copy))) return rc;
    }
    const size_t row_bytes = (size_t)c->calib.img_w * 3;
    const size_t head = (size_t)c->cam_r0 * row_bytes, tail0 = (size_t)c->cam_r1 * row_bytes;
    uint8_t* dst = c->d_frames + (size_t)first * c->frame_bytes;
    if (c->cam_r1 <= c->cam_r0) {
        HIP_TRY(hipMemcpyAsync(dst, frames, (size_t)n * c->frame_bytes, hipMemcpyHostToDevice, c->copy));
        return rest_mark(c, first, n);
    }
    if (head)
        HIP_TRY(hipMemcpy2DAsync(dst, c->frame_bytes, frames, c->frame_bytes, head, (size_t)n, hipMemcpyHostToDevice, c->copy));
    if (tail0 < c->frame_bytes)
        HIP_TRY(hipMemcpy2DAsync(dst + tail0, c->frame_bytes, frames + tail0, c->frame_bytes, c->frame_bytes - tail0, (size_t)n,
                                 hipMemcpyHostToDevice, c->copy));
    return rest_mark(c, first, n);
}

int lt_upload_masks(lt_ctx* c, const uint8_t* masks, int first, int n) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!masks) return fail(LT_ERR_INVALID, "null masks");
    if ((rc = set_device(c))) return rc;
    if ((rc = sync_all(c))) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_plane[P_MASK] + (size_t)first * c->plane_bytes, masks, (size_t)n * c->plane_bytes,
                           hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->have_mask = true;
    mark_masks(c, first, n, 0, 1);
    return LT_OK;
}

int lt_upload_bev(lt_ctx* c, const uint8_t* bev, int first, int n) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!bev) return fail(LT_ERR_INVALID, "null image");
    if ((rc = set_device(c))) return rc;
    if ((rc = ensure_bev(c))) return rc;
    if ((rc = sync_all(c))) return rc;
    HIP_TRY(hipMemcpyAsync(c->d_bev + (size_t)first * c->bev_bytes, bev, (size_t)n * c->bev_bytes,
                           hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LT_OK;
}

static int download(lt_ctx* c, const void* src, void* dst, size_t bytes) {
    if (!dst) return fail(LT_ERR_INVALID, "null output buffer");
    int rc = set_device(c);
    if (rc) return rc;
    hipStream_t st = c->stream;
    if (c->urgent_on && c->urgent) st = c->urgent;      // lt_set_urgent: what is asked for was produced on the urgent stream (or is complete)
    else if ((rc = sync_all(c))) return rc;             // results may come from any of the context's streams
    // Records, headers, lane-pixel blocks: through a page-locked scratch buffer by a kernel launch, not the copy engine (23 us
    // for 64 bytes; behind the annotated frames of a stream, milliseconds).
    constexpr size_t SMALL = 256 << 10;
    if (bytes <= SMALL) {
        if (!c->h_small && hipHostMalloc(reinterpret_cast<void**>(&c->h_small), SMALL, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            c->h_small = nullptr;
        }
        if (c->h_small && launch_copy_words_to_pinned(st, c->h_small, src, bytes)) {
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipStreamSynchronize(st));
            std::memcpy(dst, c->h_small, bytes);
            return LT_OK;
        }
    }
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return LT_OK;
}

int lt_download_masks(lt_ctx* c, int first, int n, uint8_t* masks) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if ((rc = set_device(c))) return rc;
    if ((rc = ensure_u8_masks(c, first, n))) return rc;
    return download(c, c->d_plane[P_MASK] + (size_t)first * c->plane_bytes, masks, (size_t)n * c->plane_bytes);
}

int lt_download_plane(lt_ctx* c, int plane, int first, int n, uint8_t* out) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    static const int map[6] = {P_R, P_B, P_THR, P_THB, P_MERGED, P_MASK};
    if (plane < 0 || plane > 5) return fail(LT_ERR_INVALID, "unknown plane %d", plane);
    if (plane == LT_PLANE_MASK) {
        if ((rc = set_device(c))) return rc;
        if ((rc = ensure_u8_masks(c, first, n))) return rc;
    }
    if (plane == LT_PLANE_MERGED) {   // kept bit-packed on the device; expand on demand
        if ((rc = set_device(c))) return rc;
        if ((rc = sync_all(c))) return rc;
        launch_bits_to_u8(c->stream, c->d_bits_merged + (size_t)first * c->bits_stride,
                          c->d_plane[P_MERGED] + (size_t)first * c->plane_bytes, c->calib.warp_h, c->calib.warp_w,
                          c->plane_bytes, c->bits_stride, n);
    }
    if (plane == LT_PLANE_TOPHAT_R || plane == LT_PLANE_TOPHAT_B) {   // slot by slot: the current copy may be the padded one
        if (!out) return fail(LT_ERR_INVALID, "null output buffer");
        if ((rc = set_device(c))) return rc;
        if ((rc = sync_all(c))) return rc;
        const int w = c->calib.warp_w, h = c->calib.warp_h, q = plane == LT_PLANE_TOPHAT_R ? 0 : 1;
        for (int i = first; i < first + n; ++i) {
            uint8_t* dst = out + (size_t)(i - first) * c->plane_bytes;
            if (i < (int)c->th_padded.size() && c->th_padded[(size_t)i])
                HIP_TRY(hipMemcpy2DAsync(dst, (size_t)w, c->d_th_pad[q] + (size_t)i * c->th_pad_bytes, (size_t)c->th_pitch, (size_t)w,
                                         (size_t)h, hipMemcpyDeviceToHost, c->stream));
            else
                HIP_TRY(hipMemcpyAsync(dst, c->d_plane[map[plane]] + (size_t)i * c->plane_bytes, c->plane_bytes,
                                       hipMemcpyDeviceToHost, c->stream));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        return LT_OK;
    }
    return download(c, c->d_plane[map[plane]] + (size_t)first * c->plane_bytes, out, (size_t)n * c->plane_bytes);
}

int lt_download_undistorted(lt_ctx* c, int first, int n, uint8_t* out) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!out) return fail(LT_ERR_INVALID, "null output buffer");
    if (n == 0 || c->und_bytes == 0) return LT_OK;
    if ((rc = set_device(c))) return rc;
    uint8_t* tmp = nullptr;
    if ((rc = sync_all(c))) return rc;
    if ((rc = dev_alloc(&tmp, (size_t)n * c->und_bytes))) return rc;
    launch_undistorted_to_rgb(c->stream, c->d_und, c->und_px, first, c->fe.nrows, c->fe.img_w, tmp, n);
    rc = download(c, tmp, out, (size_t)n * c->und_bytes);
    dev_free(tmp);
    return rc;
}

// ---- presentation stage: draw_lane() overlay and the bird's-eye image ----------------
namespace {

// Row intervals of cv2.fillPoly's result for a polygon whose two chains are functions of y: the
// union of the 8-connected edge lines and the even-odd interior is, per row, the hull of the edge
// pixels on that row.  The walk is OpenCV's LineIterator (left end point first, error term
// dx - 2 dy, one major-axis step per pixel).
static inline void span_point(int16_t* spans, int bh, int x, int y) {
    if (y >= 0 && y < bh) {
        const int16_t xc = (int16_t)std::min(std::max(x, -32768), 32767);
        if (xc < spans[2 * y]) spans[2 * y] = xc;
        if (xc > spans[2 * y + 1]) spans[2 * y + 1] = xc;
    }
}

static void span_line(int16_t* spans, int bh, int xa, int ya, int xb, int yb) {
    if (std::abs(xb - xa) <= 1 && std::abs(yb - ya) <= 1) {
        // neighbouring pixels (nearly every edge of a lane polygon: one plot point per row): the line is its two end points
        span_point(spans, bh, xa, ya);
        span_point(spans, bh, xb, yb);
        return;
    }
    if (xb < xa) { std::swap(xa, xb); std::swap(ya, yb); }
    const int adx = xb - xa, ady = std::abs(yb - ya), ystep = yb < ya ? -1 : 1;
    const bool tall = ady > adx;
    const int len = tall ? ady : adx, across = tall ? adx : ady;
    int err = len - 2 * across;
    for (int i = 0, x = xa, y = ya; i <= len; ++i) {
        if (y >= 0 && y < bh) {
            const int16_t xc = (int16_t)std::min(std::max(x, -32768), 32767);
            if (xc < spans[2 * y]) spans[2 * y] = xc;
            if (xc > spans[2 * y + 1]) spans[2 * y + 1] = xc;
        }
        const bool turn = err < 0;
        err -= 2 * across;
        if (turn) err += 2 * len;
        if (tall) { y += ystep; x += turn ? 1 : 0; }
        else { x += 1; y += turn ? ystep : 0; }
    }
}

static void lane_polygon_spans(int16_t* spans, int bh, const int32_t* lyx, int nl, const int32_t* ryx, int nr) {
    for (int y = 0; y < bh; ++y) { spans[2 * y] = 32767; spans[2 * y + 1] = -32768; }
    const int np = nl + nr;
    if (np <= 0) return;
    // vertex k of the closed polygon: the left points in order, then the right points reversed (np.flipud)
    const int32_t* last = nr ? ryx : lyx + 2 * (nl - 1);          // vertex np - 1: the first right point, or the last left one
    int px = last[1], py = last[0];
    for (int k = 0; k < nl; ++k) {
        span_line(spans, bh, px, py, lyx[2 * k + 1], lyx[2 * k]);
        px = lyx[2 * k + 1];
        py = lyx[2 * k];
    }
    for (int k = nr - 1; k >= 0; --k) {
        span_line(spans, bh, px, py, ryx[2 * k + 1], ryx[2 * k]);
        px = ryx[2 * k + 1];
        py = ryx[2 * k];
    }
}

}  // namespace

// ---- host-only views of the calibration tables lt_create builds (no GPU needed) --------------------------------
int lt_calib_source_rows(const lt_calib* calib, int* row0, int* row1) {
    if (!calib || !row0 || !row1) return fail(LT_ERR_INVALID, "null argument");
    RemapTable warp;
    build_warp_table(*calib, warp);
    warp_source_rows(*calib, warp, *row0, *row1);
    return LT_OK;
}

int lt_calib_warp_table(const lt_calib* calib, int16_t* xy, uint16_t* frac) {
    if (!calib || !xy || !frac) return fail(LT_ERR_INVALID, "null argument");
    RemapTable t;
    build_warp_table(*calib, t);
    const size_t n = (size_t)t.rows * t.cols;
    std::memcpy(xy, t.xy.data(), n * 2 * sizeof(int16_t));
    std::memcpy(frac, t.frac.data(), n * sizeof(uint16_t));
    return LT_OK;
}

int lt_calib_undistort_table(const lt_calib* calib, int row0, int row1, int16_t* xy, uint16_t* frac) {
    if (!calib || !xy || !frac) return fail(LT_ERR_INVALID, "null argument");
    if (row0 < 0 || row1 < row0 || row1 > calib->img_h) return fail(LT_ERR_INVALID, "rows [%d, %d) outside the image", row0, row1);
    RemapTable t;
    build_undistort_table(*calib, row0, row1, t);
    const size_t n = (size_t)t.rows * t.cols;
    std::memcpy(xy, t.xy.data(), n * 2 * sizeof(int16_t));
    std::memcpy(frac, t.frac.data(), n * sizeof(uint16_t));
    return LT_OK;
}

int lt_calib_lab_tables(uint16_t* gamma256, uint16_t* cbrt3072, int32_t* coeffs9) {
    if (!gamma256 || !cbrt3072 || !coeffs9) return fail(LT_ERR_INVALID, "null argument");
    build_lab_tables(gamma256, cbrt3072, coeffs9);
    return LT_OK;
}

int lt_calib_ellipse(int k, int32_t* halfwidths, int* taps) {
    if (k < 1 || k > 63 || !(k & 1) || !halfwidths || !taps) return fail(LT_ERR_INVALID, "k must be odd, 1..63");
    int dx[64];
    *taps = ellipse_halfwidths(k, dx);
    for (int i = 0; i < k; ++i) halfwidths[i] = dx[i];
    return LT_OK;
}

int lt_lane_polygon_spans(int warp_h, const int32_t* left_yx, int n_left, const int32_t* right_yx, int n_right,
                          int16_t* spans) {
    if (warp_h < 1 || n_left < 0 || n_right < 0 || !spans || (n_left && !left_yx) || (n_right && !right_yx))
        return fail(LT_ERR_INVALID, "bad polygon arguments");
    lane_polygon_spans(spans, warp_h, left_yx, n_left, right_yx, n_right);
    return LT_OK;
}

int lt_overlay_configure(lt_ctx* c, const double* Minv) {
    if (!c || !Minv) return fail(LT_ERR_INVALID, "null argument");
    int rc = set_device(c);
    if (rc) return rc;
    if ((rc = sync_all(c))) return rc;
    // cv2.warpPerspective(lane, Minv, (img_w, img_h)): the same table builder with M := Minv and the
    // camera frame as the destination
    lt_calib u = c->calib;
    std::memcpy(u.M, Minv, sizeof u.M);
    u.warp_w = c->calib.img_w;
    u.warp_h = c->calib.img_h;
    RemapTable t;
    build_warp_table(u, t);
    dev_free(c->d_oxy);
    dev_free(c->d_ofrac);
    c->have_overlay = false;
    if ((rc = dev_alloc(&c->d_oxy, t.xy.size()))) return rc;
    if ((rc = dev_alloc(&c->d_ofrac, t.frac.size()))) return rc;
    HIP_TRY(hipMemcpy(c->d_oxy, t.xy.data(), t.xy.size() * 2, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_ofrac, t.frac.data(), t.frac.size() * 2, hipMemcpyHostToDevice));
    // Camera rows the lane can reach at all: a pixel's four taps are (sx, sy) .. (sx + 1, sy + 1), so only pixels with
    // -1 <= sx <= bw - 1 and -1 <= sy <= bh - 1 can see the bird's-eye image; every other pixel of the annotated frame is the
    // camera pixel whatever the polygon (lt_overlay_rows, lt_present_frame).
    c->ov_r0 = c->ov_r1 = 0;
    for (int y = 0; y < t.rows; ++y) {
        bool any = false;
        for (int x = 0; x < t.cols && !any; ++x) {
            const int sx = t.xy[2 * ((size_t)y * t.cols + x)], sy = t.xy[2 * ((size_t)y * t.cols + x) + 1];
            any = sx >= -1 && sx <= c->calib.warp_w - 1 && sy >= -1 && sy <= c->calib.warp_h - 1;
        }
        if (any) {
            if (c->ov_r1 == 0) c->ov_r0 = y;
            c->ov_r1 = y + 1;
        }
    }
    c->have_overlay = true;
    return LT_OK;
}

// A call is about to overwrite the page-locked staging regions of slots [first, first + n): if a copy out of those regions
// may still be in flight (an earlier call of the same kind over the same slots), wait for it; then widen the busy range.
static int staging_claim(lt_ctx::StagingBusy& b, int first, int n) {
    if (b.hi > b.lo && first < b.hi && first + n > b.lo && b.done) {
        HIP_TRY(hipEventSynchronize(b.done));
        b.lo = b.hi = 0;
    }
    if (b.hi <= b.lo) { b.lo = first; b.hi = first + n; }
    else { b.lo = std::min(b.lo, first); b.hi = std::max(b.hi, first + n); }
    return LT_OK;
}
static int staging_mark(lt_ctx::StagingBusy& b, hipStream_t st) {
    if (!b.done && hipEventCreateWithFlags(&b.done, hipEventDisableTiming) != hipSuccess) return fail(LT_ERR_HIP, "hipEventCreate failed");
    HIP_TRY(hipEventRecord(b.done, st));
    return LT_OK;
}

static int present_stream(lt_ctx* c) {
    if (!c->present && create_compute_stream(&c->present, c->search_cus) != hipSuccess) return fail(LT_ERR_HIP, "hipStreamCreate failed");
    return LT_OK;
}

int lt_poly_points(int warp_w, int warp_h, const double* coeffs, int n, const double* ploty, const double* ploty2, int n_rows,
                   int32_t* left_n, int32_t* right_n, int32_t* left_yx, int32_t* right_yx) {
    if (warp_w < 1 || warp_h < 1 || n < 0 || n_rows < 0 || !coeffs || !left_n || !right_n || !left_yx || !right_yx ||
        (n_rows && (!ploty || !ploty2)))
        return fail(LT_ERR_INVALID, "bad arguments");
    // get_poly_points (lane_tracker.py:511-528) for n pairs of parabolas: fitx = a * ploty**2 + b * ploty + c evaluated as NumPy
    // does (two products, two sums, no contraction: this file is built with -ffp-contract=off), the points with
    // 0 <= fitx <= W - 1 kept, x truncated (astype(int)), and -- as upstream -- y = H - count .. H - 1 whatever rows they were
    const double xmax = (double)(warp_w - 1);
    size_t ol = 0, orr = 0;
    for (int i = 0; i < n; ++i) {
        for (int side = 0; side < 2; ++side) {
            const double a = coeffs[6 * i + 3 * side], b = coeffs[6 * i + 3 * side + 1], cc = coeffs[6 * i + 3 * side + 2];
            int32_t* out = side ? right_yx + 2 * orr : left_yx + 2 * ol;
            int cnt = 0;
            for (int r = 0; r < n_rows; ++r) {
                const double t1 = a * ploty2[r], t2 = b * ploty[r];
                const double x = (t1 + t2) + cc;
                if (x <= xmax && x >= 0.0) out[2 * cnt++ + 1] = (int32_t)(long long)x;
            }
            for (int k = 0; k < cnt; ++k) out[2 * k] = warp_h - cnt + k;
            if (side) { right_n[i] = cnt; orr += (size_t)cnt; }
            else { left_n[i] = cnt; ol += (size_t)cnt; }
        }
    }
    return LT_OK;
}

// lt_overlay_run; rows4: two runs of camera rows {a0, a1, b0, b1} outside which the annotated frames are not needed
// (lt_present_frame, lt_overlay_run_rows), nullptr = all of them
static int overlay_run_impl(lt_ctx* c, int first, int n, const int32_t* left_n, const int32_t* right_n, const int32_t* left_yx,
                            const int32_t* right_yx, double alpha, const int* rows4) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!c->have_overlay) return fail(LT_ERR_STATE, "lt_overlay_run before lt_overlay_configure");
    if (n == 0) return LT_OK;
    if (!left_n || !right_n) return fail(LT_ERR_INVALID, "null point counts");
    long long tl = 0, tr = 0;
    for (int i = 0; i < n; ++i) {
        if (left_n[i] < 0 || right_n[i] < 0) return fail(LT_ERR_INVALID, "negative point count");
        tl += left_n[i];
        tr += right_n[i];
    }
    if ((tl && !left_yx) || (tr && !right_yx)) return fail(LT_ERR_INVALID, "null point list");
    if ((rc = set_device(c))) return rc;
    const int bh = c->calib.warp_h;
    if (!c->d_spans && (rc = dev_alloc(&c->d_spans, (size_t)c->capacity * bh * 2))) return rc;
    if (!c->d_annot && (rc = dev_alloc(&c->d_annot, (size_t)c->capacity * c->frame_bytes))) return rc;
    // One frame (process()): the intervals travel as a kernel argument -- no staging buffer, no copy launch, no events
    static const bool arg_ok = [] { const char* e = std::getenv("LT_SPANS_ARG"); return !(e && e[0] == '0'); }();
    bool one = arg_ok && n == 1 && bh <= LT_SPAN_ARG_ROWS && (c->calib.img_w & 3) == 0;
    int16_t one_spans[2 * LT_SPAN_ARG_ROWS];
    auto claim_staging = [&]() -> int {
        int r = staging_claim(c->spans_busy, first, n);
        if (r) return r;
        if (c->h_spans_cap < c->capacity) {
            if ((r = sync_all(c))) return r;
            if (c->h_spans) (void)hipHostFree(c->h_spans);
            c->h_spans = nullptr;
            c->h_spans_cap = 0;
            HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_spans), (size_t)c->capacity * bh * 2 * sizeof(int16_t), hipHostMallocDefault));
            c->h_spans_cap = c->capacity;
        }
        return (int)LT_OK;
    };
    if (!one && (rc = claim_staging())) return rc;
    int16_t* hs = one ? one_spans : c->h_spans + (size_t)first * bh * 2;
    static const bool timing = std::getenv("LT_OVERLAY_TIMING") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    // ~18 us of edge walking per polygon: a window's piece of 32 .. 128 polygons is shared among a few threads (the caller is
    // the one thread that feeds the device)
    const int workers = std::max(1, std::min({n / 8, 8, (int)std::thread::hardware_concurrency()}));
    auto some = [&](int w) {
        size_t ol = 0, orr = 0;
        for (int i = 0; i < n; ++i) {
            if (i * (long long)workers / n == w)
                lane_polygon_spans(hs + (size_t)i * bh * 2, bh, left_yx ? left_yx + 2 * ol : nullptr, left_n[i],
                                   right_yx ? right_yx + 2 * orr : nullptr, right_n[i]);
            ol += (size_t)left_n[i];
            orr += (size_t)right_n[i];
        }
    };
    if (workers == 1) some(0);
    else {
        std::vector<std::thread> pool;
        for (int w = 1; w < workers; ++w) pool.emplace_back(some, w);
        some(0);
        for (auto& t : pool) t.join();
    }
    // the rows of the frame the path does not read came on the copy stream (lt_upload_frame_rest): the overlay is their reader
    const auto t1 = std::chrono::steady_clock::now();
    if ((rc = present_stream(c))) return rc;
    hipStream_t ps = c->present;
    if (c->rest_pending) {
        bool precise = true;
        if ((rc = wait_range(c->rests, ps, first, first + n, &precise))) return rc;
        if (!precise) HIP_TRY(hipStreamWaitEvent(ps, c->rest_done, 0));
    }
    {   // the camera rows of these slots: behind the launches that wrote their masks (which waited for the rows' upload)
        bool precise = true;
        if ((rc = wait_range(c->writers, ps, first, first + n, &precise))) return rc;
        if (!precise) {
            rc = for_each_slice(c, first, n, [&](hipStream_t st, int, int) {
                hipEvent_t e = next_order_event(c);
                if (!e) return fail(LT_ERR_HIP, "hipEventCreate failed");
                HIP_TRY(hipEventRecord(e, st));
                HIP_TRY(hipStreamWaitEvent(ps, e, 0));
                return (int)LT_OK;
            });
            if (rc) return rc;
        }
    }
    // an asynchronous download may still be reading the annotated frames this call overwrites
    if (c->annot_busy.hi > c->annot_busy.lo && first < c->annot_busy.hi && first + n > c->annot_busy.lo && c->annot_busy.done)
        HIP_TRY(hipStreamWaitEvent(ps, c->annot_busy.done, 0));
    if (one) {
        if (launch_overlay_lane_one(ps, c->d_frames + (size_t)first * c->frame_bytes, c->d_annot + (size_t)first * c->frame_bytes,
                                    c->d_oxy, c->d_ofrac, hs, c->calib.img_h, c->calib.img_w, bh, c->calib.warp_w, (float)alpha, rows4)) {
            HIP_TRY(hipGetLastError());
            return note_range(c->readers, ps, first, first + n);
        }
        // not launched (the runtime refused the argument block): the staged way after all, with the intervals already built
        one = false;
        if ((rc = claim_staging())) return rc;
        std::memcpy(c->h_spans + (size_t)first * bh * 2, one_spans, (size_t)bh * 2 * sizeof(int16_t));
        hs = c->h_spans + (size_t)first * bh * 2;
    }
    launch_copy_from_pinned(ps, c->d_spans + (size_t)first * bh * 2, hs, (size_t)n * bh * 2 * sizeof(int16_t));
    const auto t2 = std::chrono::steady_clock::now();
    launch_overlay_lane(ps, c->d_frames + (size_t)first * c->frame_bytes, c->d_annot + (size_t)first * c->frame_bytes,
                        c->frame_bytes, c->d_oxy, c->d_ofrac, c->d_spans + (size_t)first * bh * 2, (size_t)bh,
                        c->calib.img_h, c->calib.img_w, bh, c->calib.warp_w, (float)alpha, n, rows4);
    HIP_TRY(hipGetLastError());
    if ((rc = staging_mark(c->spans_busy, ps))) return rc;
    rc = note_range(c->readers, ps, first, first + n);
    if (timing) {
        const auto t3 = std::chrono::steady_clock::now();
        auto us = [](auto a, auto b) { return (long)std::chrono::duration_cast<std::chrono::microseconds>(b - a).count(); };
        std::fprintf(stderr, "overlay_run n=%d: spans %ld us, wait+memcpy %ld us, launch+events %ld us\n", n, us(t0, t1), us(t1, t2), us(t2, t3));
    }
    return rc;
}

int lt_overlay_run(lt_ctx* c, int first, int n, const int32_t* left_n, const int32_t* right_n, const int32_t* left_yx,
                   const int32_t* right_yx, double alpha) {
    return overlay_run_impl(c, first, n, left_n, right_n, left_yx, right_yx, alpha, nullptr);
}

static int ordered_rows(lt_ctx* c, const int32_t* rows4, int r[4]) {
    for (int k = 0; k < 4; ++k) r[k] = rows4[k];
    if (!(0 <= r[0] && r[0] <= r[1] && r[1] <= r[2] && r[2] <= r[3] && r[3] <= c->calib.img_h))
        return fail(LT_ERR_INVALID, "row runs must be ordered and inside the frame");
    return LT_OK;
}

int lt_overlay_run_rows(lt_ctx* c, int first, int n, const int32_t* left_n, const int32_t* right_n, const int32_t* left_yx,
                        const int32_t* right_yx, double alpha, const int32_t* rows4) {
    if (!rows4) return overlay_run_impl(c, first, n, left_n, right_n, left_yx, right_yx, alpha, nullptr);
    if (!c) return fail(LT_ERR_INVALID, "null context");
    int r[4];
    const int rc = ordered_rows(c, rows4, r);
    if (rc) return rc;
    return overlay_run_impl(c, first, n, left_n, right_n, left_yx, right_yx, alpha, r);
}

int lt_overlay_rows(lt_ctx* c, int* row0, int* row1) {
    if (!c || !row0 || !row1) return fail(LT_ERR_INVALID, "null argument");
    if (!c->have_overlay) return fail(LT_ERR_STATE, "lt_overlay_rows before lt_overlay_configure");
    *row0 = c->ov_r0;
    *row1 = c->ov_r1;
    return LT_OK;
}

int lt_overlay_set_font(lt_ctx* c, const uint8_t* atlas, const uint8_t* advance, int first_char, int n_glyphs, int glyph_w,
                        int glyph_h) {
    if (!c || !atlas || !advance) return fail(LT_ERR_INVALID, "null argument");
    if (n_glyphs < 1 || n_glyphs > 256 || glyph_w < 1 || glyph_w > 255 || glyph_h < 1 || glyph_h > 255 || first_char < 0)
        return fail(LT_ERR_INVALID, "bad font geometry");
    int rc = set_device(c);
    if (rc) return rc;
    if ((rc = sync_all(c))) return rc;
    dev_free(c->d_atlas);
    dev_free(c->d_advance);
    c->font_glyphs = 0;
    const size_t bytes = (size_t)n_glyphs * glyph_w * glyph_h;
    if ((rc = dev_alloc(&c->d_atlas, bytes))) return rc;
    if ((rc = dev_alloc(&c->d_advance, (size_t)n_glyphs))) return rc;
    HIP_TRY(hipMemcpy(c->d_atlas, atlas, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(c->d_advance, advance, (size_t)n_glyphs, hipMemcpyHostToDevice));
    c->h_advance.assign(advance, advance + n_glyphs);
    c->font_first = first_char;
    c->font_glyphs = n_glyphs;
    c->font_gw = glyph_w;
    c->font_gh = glyph_h;
    return LT_OK;
}

int lt_overlay_text(lt_ctx* c, int first, int n, const char* lines, int n_lines, int line_len, int x0, int y0, int step) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!c->font_glyphs) return fail(LT_ERR_STATE, "lt_overlay_text before lt_overlay_set_font");
    if (!c->d_annot) return fail(LT_ERR_STATE, "lt_overlay_text before lt_overlay_run");
    if (n == 0 || n_lines <= 0 || line_len <= 0) return LT_OK;
    if (!lines) return fail(LT_ERR_INVALID, "null text");
    if ((rc = set_device(c))) return rc;
    const size_t per = (size_t)n_lines * line_len;
    if (per > c->text_per_slot || c->text_slots < c->capacity) {      // (re)size the per-slot text buffers: rare, synchronises
        if ((rc = sync_all(c))) return rc;
        const size_t per_new = (std::max(per, c->text_per_slot) + 3) & ~(size_t)3, total = per_new * (size_t)c->capacity;
        dev_free(c->d_lines);
        dev_free(c->d_xpos);
        if (c->h_lines) (void)hipHostFree(c->h_lines);
        if (c->h_xpos) (void)hipHostFree(c->h_xpos);
        c->h_lines = nullptr;
        c->h_xpos = nullptr;
        c->text_per_slot = 0;
        c->text_slots = 0;
        if ((rc = dev_alloc(&c->d_lines, total))) return rc;
        if ((rc = dev_alloc(&c->d_xpos, total))) return rc;
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_lines), total, hipHostMallocDefault));
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_xpos), total * sizeof(int16_t), hipHostMallocDefault));
        c->text_per_slot = per_new;
        c->text_slots = c->capacity;
    }
    if ((rc = staging_claim(c->text_busy, first, n))) return rc;
    // A slot's lines sit at the slot's own position in the per-slot buffers, at their FIXED stride text_per_slot -- not at
    // this call's n_lines * line_len: staging_claim orders calls by slot range, and two calls in flight over disjoint
    // slots with different line counts (a 'fail' piece has one line, a lane piece two) must not meet in bytes.
    const size_t stride = c->text_per_slot;
    uint8_t* hl = c->h_lines + (size_t)first * stride;
    int16_t* hx = c->h_xpos + (size_t)first * stride;
    for (int i = 0; i < n; ++i) {
        std::memcpy(hl + (size_t)i * stride, lines + (size_t)i * per, per);
        for (int l = 0; l < n_lines; ++l) {                // left edge of every character: running sum of advances
            const char* src = lines + (size_t)i * per + (size_t)l * line_len;
            int16_t* dst = hx + (size_t)i * stride + (size_t)l * line_len;
            int x = x0;
            bool ended = false;
            for (int k = 0; k < line_len; ++k) {
                const unsigned char ch = (unsigned char)src[k];
                ended = ended || ch == 0;
                dst[k] = (int16_t)std::min(x, 32767);
                const int g = (int)ch - c->font_first;
                if (!ended && g >= 0 && g < c->font_glyphs) x += c->h_advance[(size_t)g];
            }
        }
    }
    uint8_t* dl = c->d_lines + (size_t)first * stride;
    int16_t* dx = c->d_xpos + (size_t)first * stride;
    if ((rc = present_stream(c))) return rc;
    // A frame or two (process(), one frame per call): the kernel reads the few hundred bytes from the page-locked buffers
    // themselves -- two launches fewer between the record and the annotated frame.  A window's worth goes to the device first.
    const uint8_t* kl = dl;
    const int16_t* kx = dx;
    void *pl = nullptr, *px = nullptr;
    static const bool direct_ok = [] { const char* e = std::getenv("LT_TEXT_DIRECT"); return !(e && e[0] == '0'); }();
    if (direct_ok && n <= 2 && hipHostGetDevicePointer(&pl, hl, 0) == hipSuccess && hipHostGetDevicePointer(&px, hx, 0) == hipSuccess &&
        pl && px) {
        kl = static_cast<const uint8_t*>(pl);
        kx = static_cast<const int16_t*>(px);
    } else {
        (void)hipGetLastError();
        launch_copy_from_pinned(c->present, dl, hl, (size_t)n * stride);
        launch_copy_from_pinned(c->present, dx, hx, (size_t)n * stride * sizeof(int16_t));
    }
    launch_overlay_text(c->present, c->d_annot + (size_t)first * c->frame_bytes, c->frame_bytes, c->calib.img_h, c->calib.img_w,
                        c->d_atlas, c->d_advance, c->font_first, c->font_glyphs, c->font_gw, c->font_gh, kl, kx,
                        n_lines, line_len, (int)stride, y0, step, n);
    HIP_TRY(hipGetLastError());
    return staging_mark(c->text_busy, c->present);
}

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

int lt_download_overlay(lt_ctx* c, int first, int n, uint8_t* out) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!c->d_annot) return fail(LT_ERR_STATE, "lt_download_overlay before lt_overlay_run");
    if (!out) return fail(LT_ERR_INVALID, "null output buffer");
    const uint8_t* src = c->d_annot + (size_t)first * c->frame_bytes;
    const size_t bytes = (size_t)n * c->frame_bytes;
    if (!c->present || n == 0) return download(c, src, out, bytes);
    // The annotated frames are written on the presentation stream and nowhere else (lt_overlay_run, lt_overlay_text), behind
    // everything they depend on: the copy is enqueued there, behind them, and the host waits once -- not once for the overlay
    // and once more for a copy it issues only then (10 us of process()'s 0.4 ms per frame).
    if ((rc = set_device(c))) return rc;
    // a frame or two: by a copy kernel (no engine start-up: 11 us less per frame of process()); LT_DL1_KERNEL=0: the engine
    static const bool by_kernel = [] { const char* e = std::getenv("LT_DL1_KERNEL"); return !(e && e[0] == '0'); }();
    if (!(by_kernel && n <= 2 && launch_copy_to_pinned(c->present, out, src, bytes)))
        HIP_TRY(hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, c->present));
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(c->present));
    return LT_OK;
}

// process()'s tail for ONE frame in one call: lt_overlay_run + lt_overlay_text + the way back, the host waiting once.  With
// rows4 = {a0, a1, b0, b1} only those two runs of camera rows are drawn and written to `out` (the rows at their places in the
// frame): a pixel outside the rows the lane can reach (lt_overlay_rows) and outside the text lines is the camera pixel, which
// the caller has -- process() copies those rows from its input while the device is busy, and only half the frame crosses the
// bus behind the overlay.  The runs must cover the text lines and, for a non-empty polygon, lt_overlay_rows.
static int present_copy_rows(lt_ctx* c, int slot, uint8_t* out, int row0, int row1) {
    if (row1 <= row0) return LT_OK;
    const size_t row_bytes = (size_t)c->calib.img_w * 3, off = (size_t)row0 * row_bytes, bytes = (size_t)(row1 - row0) * row_bytes;
    const uint8_t* src = c->d_annot + (size_t)slot * c->frame_bytes;
    if (!launch_copy_to_pinned(c->present, out + off, src + off, bytes))
        HIP_TRY(hipMemcpyAsync(out + off, src + off, bytes, hipMemcpyDeviceToHost, c->present));
    HIP_TRY(hipGetLastError());
    return LT_OK;
}
// rows4 -> r[4] (nullptr: the whole frame as the first run); `split`: the text lines must lie in the first run and the rows the
// lane can reach in the second, the two apart -- the condition for drawing and sending the second run before the text exists
static int present_rows(lt_ctx* c, const int32_t* rows4, bool text, int n_lines, int y0, int step, bool lane, bool split, int r[4]) {
    const int H = c->calib.img_h;
    r[0] = 0; r[1] = H; r[2] = H; r[3] = H;
    if (!rows4) return split ? fail(LT_ERR_INVALID, "two row runs are needed") : (int)LT_OK;
    for (int k = 0; k < 4; ++k) r[k] = rows4[k];
    if (!(0 <= r[0] && r[0] <= r[1] && r[1] <= r[2] && r[2] <= r[3] && r[3] <= H))
        return fail(LT_ERR_INVALID, "row runs must be ordered and inside the frame");
    auto within = [&](int lo, int hi, int a, int b) { lo = std::max(lo, 0); hi = std::min(hi, H); return lo >= hi || (a <= lo && hi <= b); };
    const int t0 = y0, t1 = y0 + (n_lines - 1) * step + c->font_gh;
    if (split) {
        if (text && !within(t0, t1, r[0], r[1])) return fail(LT_ERR_INVALID, "the first row run does not cover the text lines");
        if (!within(c->ov_r0, c->ov_r1, r[2], r[3])) return fail(LT_ERR_INVALID, "the second row run does not cover the rows the lane can reach (lt_overlay_rows)");
        return LT_OK;
    }
    auto covered = [&](int lo, int hi) { return within(lo, hi, r[0], r[1]) || within(lo, hi, r[2], r[3]) || (r[1] == r[2] && within(lo, hi, r[0], r[3])); };
    if (text && !covered(t0, t1)) return fail(LT_ERR_INVALID, "the row runs do not cover the text lines");
    if (lane && !covered(c->ov_r0, c->ov_r1)) return fail(LT_ERR_INVALID, "the row runs do not cover the rows the lane can reach (lt_overlay_rows)");
    return LT_OK;
}

int lt_present_frame(lt_ctx* c, int slot, const int32_t* left_n, const int32_t* right_n, const int32_t* left_yx, const int32_t* right_yx,
                     double alpha, const char* lines, int n_lines, int line_len, int x0, int y0, int step, uint8_t* out,
                     const int32_t* rows4) {
    int rc = check_slots(c, slot, 1);
    if (rc) return rc;
    if (!out) return fail(LT_ERR_INVALID, "null output buffer");
    if (!left_n || !right_n) return fail(LT_ERR_INVALID, "null point counts");
    if (!c->have_overlay) return fail(LT_ERR_STATE, "lt_present_frame before lt_overlay_configure");
    const bool text = lines && n_lines > 0 && line_len > 0 && c->font_glyphs > 0;
    int r[4];
    if ((rc = present_rows(c, rows4, text, n_lines, y0, step, left_n[0] > 0 || right_n[0] > 0, false, r))) return rc;
    if ((rc = overlay_run_impl(c, slot, 1, left_n, right_n, left_yx, right_yx, alpha, rows4 ? r : nullptr))) return rc;
    if (text && (rc = lt_overlay_text(c, slot, 1, lines, n_lines, line_len, x0, y0, step))) return rc;
    if ((rc = present_copy_rows(c, slot, out, r[0], r[1]))) return rc;
    if ((rc = present_copy_rows(c, slot, out, r[2], r[3]))) return rc;
    HIP_TRY(hipStreamSynchronize(c->present));
    return LT_OK;
}

// lt_present_frame in two halves, for a caller that knows the polygon before it knows the text (LaneTracker.process(): the
// averaged curves follow from the record at once, radius, eccentricity and the verdict on the frame take the host another
// 25 us): the first half draws both row runs and sends the second one -- the rows the lane can reach -- on its way without
// waiting; the second half blends the text into the first run, sends that and waits for both.  A first half whose frame turns
// out invalid is simply followed by a whole lt_present_frame (same slot, same `out`): it draws and sends everything again.
int lt_present_lane_async(lt_ctx* c, int slot, const int32_t* left_n, const int32_t* right_n, const int32_t* left_yx,
                          const int32_t* right_yx, double alpha, uint8_t* out, const int32_t* rows4) {
    int rc = check_slots(c, slot, 1);
    if (rc) return rc;
    if (!out) return fail(LT_ERR_INVALID, "null output buffer");
    if (!left_n || !right_n) return fail(LT_ERR_INVALID, "null point counts");
    if (!c->have_overlay) return fail(LT_ERR_STATE, "lt_present_lane_async before lt_overlay_configure");
    int r[4];
    if ((rc = present_rows(c, rows4, false, 0, 0, 0, true, true, r))) return rc;
    if ((rc = overlay_run_impl(c, slot, 1, left_n, right_n, left_yx, right_yx, alpha, r))) return rc;
    return present_copy_rows(c, slot, out, r[2], r[3]);
}

int lt_present_finish(lt_ctx* c, int slot, const char* lines, int n_lines, int line_len, int x0, int y0, int step, uint8_t* out,
                      const int32_t* rows4) {
    int rc = check_slots(c, slot, 1);
    if (rc) return rc;
    if (!out) return fail(LT_ERR_INVALID, "null output buffer");
    if (!c->d_annot || !c->present) return fail(LT_ERR_STATE, "lt_present_finish before lt_present_lane_async");
    const bool text = lines && n_lines > 0 && line_len > 0 && c->font_glyphs > 0;
    int r[4];
    if ((rc = present_rows(c, rows4, text, n_lines, y0, step, false, true, r))) return rc;
    if ((rc = set_device(c))) return rc;
    if (text && (rc = lt_overlay_text(c, slot, 1, lines, n_lines, line_len, x0, y0, step))) return rc;
    if ((rc = present_copy_rows(c, slot, out, r[0], r[1]))) return rc;
    HIP_TRY(hipStreamSynchronize(c->present));
    return LT_OK;
}

static void harvest_downloads(lt_ctx* c);
static int choose_download(lt_ctx* c);
static int download_overlay_async_impl(lt_ctx* c, int first, int n, uint8_t* out, const int* rows4);
int lt_download_overlay_async(lt_ctx* c, int first, int n, uint8_t* out) { return download_overlay_async_impl(c, first, n, out, nullptr); }
int lt_download_overlay_rows_async(lt_ctx* c, int first, int n, uint8_t* out, const int32_t* rows4) {
    if (!rows4) return download_overlay_async_impl(c, first, n, out, nullptr);
    if (!c) return fail(LT_ERR_INVALID, "null context");
    int r[4];
    const int rc = ordered_rows(c, rows4, r);
    if (rc) return rc;
    return download_overlay_async_impl(c, first, n, out, r);
}
// rows4: only these two runs of rows of every frame (at their places in `out`), nullptr: whole frames
static int download_overlay_async_impl(lt_ctx* c, int first, int n, uint8_t* out, const int* rows4) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!out) return fail(LT_ERR_INVALID, "null output buffer");
    if (!c->d_annot) return fail(LT_ERR_STATE, "lt_download_overlay_async before lt_overlay_run");
    if (n == 0) return LT_OK;
    if ((rc = set_device(c))) return rc;
    // on a stream of its own, behind the overlay work enqueued so far: the copy neither holds up the kernels queued behind
    // it on the context's stream nor shares a queue with the uploads
    if (!c->dl) {
        if (c->search_cus >= 2) {                  // the reserved CUs but the first are the copy kernel's (lt_set_search_cus)
            uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int b = 1; b < c->search_cus && b < 256; ++b) mask[b >> 5] |= 1u << (b & 31);
            HIP_TRY(hipExtStreamCreateWithCUMask(&c->dl, 8, mask));
        } else {                                   // highest priority: the copy kernel's few workgroups go ahead of the mask kernels'
            int lo = 0, hi = 0;
            (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
            HIP_TRY(hipStreamCreateWithPriority(&c->dl, hipStreamNonBlocking, hi));
        }
    }
    hipEvent_t e = next_order_event(c);
    if (!e) return fail(LT_ERR_HIP, "hipEventCreate failed");
    HIP_TRY(hipEventRecord(e, c->present ? c->present : c->stream));
    HIP_TRY(hipStreamWaitEvent(c->dl, e, 0));
    // engine or kernel: by measurement (choose_download); LT_DL_KERNEL=1 / 0 and lt_set_download_method pin one of them
    static const int env_method = [] { const char* e = std::getenv("LT_DL_KERNEL"); return !e ? -1 : (e[0] == '0' ? 0 : 1); }();
    if (env_method >= 0 && c->dl_forced < 0) c->dl_forced = env_method;
    harvest_downloads(c);
    int method = choose_download(c);
    auto timing_event = [&]() -> hipEvent_t {
        hipEvent_t ev = nullptr;
        if (!c->dl_event_pool.empty()) { ev = c->dl_event_pool.back(); c->dl_event_pool.pop_back(); }
        else if (hipEventCreate(&ev) != hipSuccess) { (void)hipGetLastError(); ev = nullptr; }
        return ev;
    };
    hipEvent_t ta = timing_event(), tb = timing_event();
    if (ta && tb) HIP_TRY(hipEventRecord(ta, c->dl));
    size_t bytes = (size_t)n * c->frame_bytes;
    const uint8_t* src = c->d_annot + (size_t)first * c->frame_bytes;
    if (!rows4) {
        if (method == 1 && !launch_copy_to_pinned(c->dl, out, src, bytes)) method = 0;   // not page-locked / aligned
        if (method == 0) HIP_TRY(hipMemcpyAsync(out, src, bytes, hipMemcpyDeviceToHost, c->dl));
    } else {
        const size_t row_bytes = (size_t)c->calib.img_w * 3;
        bytes = 0;
        // The kernel takes a run of rows of all the frames in one launch; the engine takes the run of ONE frame as an ordinary
        // copy (a pitched copy over the frames it takes row by row: 120-160 ms for a window of 256 frames), two copies per
        // frame -- 18.4 k frames/s of an annotated 1280x720 stream against the kernel's 22.6 k.  So the kernel, unless
        // lt_set_download_method(0) / LT_DL_KERNEL=0 ask for the engine.
        bool by_kernel = c->dl_forced != 0;
        for (int k = 0; k < 4 && by_kernel; k += 2)       // both runs the same way, so that the timing below means one thing
            by_kernel = rows4[k + 1] <= rows4[k] ||
                        ((((size_t)rows4[k] * row_bytes) | ((size_t)(rows4[k + 1] - rows4[k]) * row_bytes) | c->frame_bytes | (size_t)(uintptr_t)out) & 15) == 0;
        for (int k = 0; k < 4; k += 2) {
            if (rows4[k + 1] <= rows4[k]) continue;
            const size_t off = (size_t)rows4[k] * row_bytes, run = (size_t)(rows4[k + 1] - rows4[k]) * row_bytes;
            if (by_kernel && !launch_copy_rows_to_pinned(c->dl, out, src, c->frame_bytes, off, run, n)) by_kernel = false;
            bytes += run * (size_t)n;
        }
        if (!by_kernel)
            for (int f = 0; f < n; ++f)
                for (int k = 0; k < 4; k += 2) {
                    if (rows4[k + 1] <= rows4[k]) continue;
                    const size_t off = (size_t)f * c->frame_bytes + (size_t)rows4[k] * row_bytes;
                    HIP_TRY(hipMemcpyAsync(out + off, src + off, (size_t)(rows4[k + 1] - rows4[k]) * row_bytes, hipMemcpyDeviceToHost, c->dl));
                }
        method = by_kernel ? 1 : 0;
    }
    HIP_TRY(hipGetLastError());
    if (ta && tb) {
        HIP_TRY(hipEventRecord(tb, c->dl));
        c->dl_inflight.push_back({ta, tb, (double)bytes, method});
    } else {
        if (ta) c->dl_event_pool.push_back(ta);
        if (tb) c->dl_event_pool.push_back(tb);
    }
    if (c->annot_busy.hi <= c->annot_busy.lo) { c->annot_busy.lo = first; c->annot_busy.hi = first + n; }
    else { c->annot_busy.lo = std::min(c->annot_busy.lo, first); c->annot_busy.hi = std::max(c->annot_busy.hi, first + n); }
    return staging_mark(c->annot_busy, c->dl);
}

// Engine or kernel?  The copy engine moves the frames at 48-56 GB/s when the page-locked destination and the device buffer are
// laid out kindly, and at 28-30 GB/s when they are not -- a property of the memory the process happened to get (allocation
// history, the box), not of anything this library orders: tools/copy_engine_probe.py RAW=1 shows one lone download at 29 GB/s
// on the same engine, same code path, beside nothing.  (Rounds 2-3 read the resulting 9.3 k instead of 15 k frames/s of
// the annotated 1280x720 stream as uploads and downloads "taking turns"; they do overlap.)  A kernel storing 16 bytes per lane
// into the same destination is not affected (an annotated 1280x720 stream does 13.3 k frames/s that way in either regime: less
// than the engine at its best, 15 k, because the copy kernel shares the chip with the mask chain, far more than the engine at
// its worst).  So: every copy is timed with an event pair; the engine is the default; when its running rate drops below
// DL_SLOW GB/s the kernel takes over, and one copy in DL_REPROBE goes by the engine again so that a recovery is noticed.
static constexpr double DL_SLOW = 36.0;
static constexpr int DL_REPROBE = 48;
static void harvest_downloads(lt_ctx* c) {
    size_t keep = 0;
    for (size_t i = 0; i < c->dl_inflight.size(); ++i) {
        lt_ctx::DlTimed& d = c->dl_inflight[i];
        float ms = 0.f;
        if (hipEventQuery(d.b) == hipSuccess && hipEventElapsedTime(&ms, d.a, d.b) == hipSuccess) {
            if (ms > 0.f && d.bytes >= 8e6) {              // small copies time the launch, not the bus
                const double r = d.bytes / (ms * 1e-3) / 1e9;
                c->dl_rate[d.method] = c->dl_samples[d.method] ? 0.5 * c->dl_rate[d.method] + 0.5 * r : r;
                ++c->dl_samples[d.method];
            }
            c->dl_event_pool.push_back(d.a);
            c->dl_event_pool.push_back(d.b);
        } else {
            (void)hipGetLastError();
            c->dl_inflight[keep++] = d;
        }
    }
    c->dl_inflight.resize(keep);
}
static int choose_download(lt_ctx* c) {
    if (c->dl_forced >= 0) return c->dl_forced;
    const int cur = c->dl_method, other = 1 - cur;
    ++c->dl_since_probe;
    if (c->dl_samples[cur] >= 4) {       // (the first copies of a stream are short and wait for their overlays: not a verdict)
        const bool never = c->dl_samples[other] == 0;
        if (cur == 0 && c->dl_rate[0] < DL_SLOW && (never || c->dl_rate[1] > c->dl_rate[0])) { c->dl_method = 1; c->dl_since_probe = 0; }
        else if (cur == 1 && c->dl_rate[0] >= DL_SLOW) { c->dl_method = 0; c->dl_since_probe = 0; }   // the engine has recovered
        else if (c->dl_since_probe >= DL_REPROBE && (cur == 1 || c->dl_rate[0] < DL_SLOW)) {   // one copy the other way
            c->dl_since_probe = 0;
            return other;
        }
    }
    return c->dl_method;
}

int lt_set_download_method(lt_ctx* c, int method) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    if (method < -1 || method > 1) return fail(LT_ERR_INVALID, "download method: -1 = measured choice, 0 = copy engine, 1 = kernel");
    c->dl_forced = method;
    return LT_OK;
}

int lt_download_stats(lt_ctx* c, double* engine_gbs, int* engine_copies, double* kernel_gbs, int* kernel_copies, int* method) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    harvest_downloads(c);
    if (engine_gbs) *engine_gbs = c->dl_rate[0];
    if (engine_copies) *engine_copies = c->dl_samples[0];
    if (kernel_gbs) *kernel_gbs = c->dl_rate[1];
    if (kernel_copies) *kernel_copies = c->dl_samples[1];
    if (method) *method = c->dl_forced >= 0 ? c->dl_forced : c->dl_method;
    return LT_OK;
}

int lt_download_overlay_wait(lt_ctx* c) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    int rc = set_device(c);
    if (rc) return rc;
    if (!c->dl) return LT_OK;
    HIP_TRY(hipStreamSynchronize(c->dl));   // every copy of lt_download_overlay_async is behind its slots' overlay kernels
    c->annot_busy.lo = c->annot_busy.hi = 0;
    return LT_OK;
}

int lt_download_bev(lt_ctx* c, int first, int n, uint8_t* out) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!out) return fail(LT_ERR_INVALID, "null output buffer");
    if (!c->have_mask) return fail(LT_ERR_STATE, "lt_download_bev before lt_mask_run");
    if (n == 0) return LT_OK;
    if ((rc = set_device(c))) return rc;
    if ((rc = ensure_bev(c))) return rc;
    if ((rc = sync_all(c))) return rc;
    uint8_t* dst = c->d_bev + (size_t)first * c->bev_bytes;
    if (c->fe.nrows <= 0) HIP_TRY(hipMemsetAsync(dst, 0, (size_t)n * c->bev_bytes, c->stream));
    else
        launch_warp_rgb(c->stream, c->d_und, c->und_px, first, c->d_wxy, c->d_wfrac, c->fe, dst, c->bev_bytes, n);
    HIP_TRY(hipGetLastError());
    return download(c, dst, out, (size_t)n * c->bev_bytes);
}

// A search over ONE slot (process(): one frame per call, the host waiting for its record) sends the record to page-locked
// memory by a launch queued right behind the search kernel, while the device is still busy with the frame: lt_download_records
// then waits for that stream and reads 64 bytes, instead of launching the copy once the search is over (8 us of 0.4 ms).
static lt_lane_record* rec_mirror_device(lt_ctx* c) {      // the mirror as kernels address it (nullptr: there is none)
    if (!c->h_rec && hipHostMalloc(reinterpret_cast<void**>(&c->h_rec), 256, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        c->h_rec = nullptr;
    }
    void* dev = nullptr;
    if (!c->h_rec || hipHostGetDevicePointer(&dev, c->h_rec, 0) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return static_cast<lt_lane_record*>(dev);
}
static void mirror_record(lt_ctx* c, hipStream_t st, int slot) {
    c->rec_mirror_slot = -1;
    if (rec_mirror_device(c) && launch_copy_words_to_pinned(st, c->h_rec, c->d_rec + slot, sizeof(lt_lane_record))) {
        c->rec_mirror_slot = slot;
        c->rec_mirror_stream = st;
    }
}

int lt_download_records(lt_ctx* c, int first, int n, lt_lane_record* out) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (n == 1 && out && c->rec_mirror_slot == first && c->h_rec) {
        if ((rc = set_device(c))) return rc;
        HIP_TRY(hipStreamSynchronize(c->rec_mirror_stream));
        std::memcpy(out, c->h_rec, sizeof(lt_lane_record));
        return LT_OK;
    }
    return download(c, c->d_rec + first, out, (size_t)n * sizeof(lt_lane_record));
}

int lt_download_pixels(lt_ctx* c, int slot, int side, int32_t* ys, int32_t* xs, int cap, int* count) {
    int rc = check_slots(c, slot, 1);
    if (rc) return rc;
    if (side < 0 || side > 1 || !count || cap < 0) return fail(LT_ERR_INVALID, "bad side/count/cap");
    if (!c->d_pix) return fail(LT_ERR_STATE, "no search has run yet");
    lt_lane_record r;
    if ((rc = download(c, c->d_rec + slot, &r, sizeof r))) return rc;
    if (r._pad == 1) {
        // k_sws_fit2 leaves one column mask per window row (lt_internal.h: sws2_mask_offset); the lists
        // self.left_y / left_x (level-major, row-major inside a window, ascending x) are expanded here
        const uint32_t* block = c->d_pix + (size_t)slot * 2 * c->maxpix;
        uint32_t hdr[4];
        if ((rc = download(c, block, hdr, sizeof hdr))) return rc;
        const int nlev = (int)hdr[0], wh = (int)hdr[1], H1 = (int)hdr[2];
        const long long words = sws2_block_words(nlev, wh);
        if (nlev < 1 || wh < 1 || words > 2LL * c->maxpix) return fail(LT_ERR_STATE, "corrupt lane-pixel block");
        std::vector<uint32_t> blk((size_t)words);
        if ((rc = download(c, block, blk.data(), (size_t)words * 4))) return rc;
        const int32_t* roi = reinterpret_cast<const int32_t*>(blk.data() + 4);
        const uint32_t* masks = blk.data() + sws2_mask_offset(nlev);
        int n = 0;
        for (int level = 0; level < nlev; ++level) {
            const int sl = side * nlev + level, a = roi[sl * 2], b = roi[sl * 2 + 1];
            if (b <= a) continue;
            for (int ry = 0; ry < wh; ++ry) {
                const size_t mi = ((size_t)sl * wh + ry) * 2;
                unsigned long long m = (unsigned long long)masks[mi] | ((unsigned long long)masks[mi + 1] << 32);
                const int y = H1 - (1 + level) * wh + ry;
                while (m) {
                    const int j = __builtin_ctzll(m);
                    m &= m - 1;
                    if (n < cap && ys && xs) { ys[n] = y; xs[n] = a + j; }
                    ++n;
                }
            }
        }
        *count = n;
        if (n > 0 && cap > 0 && (!ys || !xs)) return fail(LT_ERR_INVALID, "null pixel buffers");
        return LT_OK;
    }
    if (r._pad == 2) {
        // k_band_fit2: one column mask and one first column per (side, row); row-major, ascending x
        const uint32_t* block = c->d_pix + (size_t)slot * 2 * c->maxpix;
        uint32_t hdr[4];
        if ((rc = download(c, block, hdr, sizeof hdr))) return rc;
        const int nrows = (int)hdr[0], top = (int)hdr[1];
        const long long words = band2_block_words(nrows);
        if (nrows < 0 || words > 2LL * c->maxpix) return fail(LT_ERR_STATE, "corrupt lane-pixel block");
        std::vector<uint32_t> blk((size_t)words);
        if ((rc = download(c, block, blk.data(), (size_t)words * 4))) return rc;
        const int32_t* row_a = reinterpret_cast<const int32_t*>(blk.data() + 4) + (size_t)side * nrows;
        const uint32_t* masks = blk.data() + band2_mask_offset(nrows) + (size_t)side * nrows * 2;
        int n = 0;
        for (int ry = 0; ry < nrows; ++ry) {
            unsigned long long m = (unsigned long long)masks[2 * ry] | ((unsigned long long)masks[2 * ry + 1] << 32);
            while (m) {
                const int j = __builtin_ctzll(m);
                m &= m - 1;
                if (n < cap && ys && xs) { ys[n] = top + ry; xs[n] = row_a[ry] + j; }
                ++n;
            }
        }
        *count = n;
        if (n > 0 && cap > 0 && (!ys || !xs)) return fail(LT_ERR_INVALID, "null pixel buffers");
        return LT_OK;
    }
    int n = side == 0 ? r.n_left : r.n_right;
    if (n > c->maxpix) n = c->maxpix;
    *count = n;
    if (n > cap) n = cap;
    if (n <= 0) return LT_OK;
    if (!ys || !xs) return fail(LT_ERR_INVALID, "null pixel buffers");
    std::vector<uint32_t> tmp((size_t)n);
    if ((rc = download(c, c->d_pix + ((size_t)slot * 2 + side) * c->maxpix, tmp.data(), (size_t)n * 4))) return rc;
    for (int i = 0; i < n; ++i) {
        ys[i] = (int32_t)(tmp[i] >> 16);
        xs[i] = (int32_t)(tmp[i] & 0xffffu);
    }
    return LT_OK;
}

int lt_download_centroids(lt_ctx* c, int slot, int side, int32_t* out, int cap, int* count) {
    int rc = check_slots(c, slot, 1);
    if (rc) return rc;
    if (side < 0 || side > 1 || !count || cap < 0 || (cap > 0 && !out)) return fail(LT_ERR_INVALID, "bad side/count/cap/out");
    if (!c->d_cent) return fail(LT_ERR_STATE, "no sliding-window search has run yet");
    std::vector<int32_t> tmp((size_t)c->maxlev + 2);
    if ((rc = download(c, c->d_cent + ((size_t)slot * 2 + side) * (c->maxlev + 2), tmp.data(), tmp.size() * 4))) return rc;
    int n = tmp[0];
    if (n < 0) n = 0;
    if (n > c->maxlev + 1) n = c->maxlev + 1;
    *count = n;
    for (int i = 0; i < n && i < cap; ++i) out[i] = tmp[1 + i];
    return LT_OK;
}

int lt_copy_records_to_device(lt_ctx* c, int first, int n, void* dst) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!dst) return fail(LT_ERR_INVALID, "null destination");
    if ((rc = set_device(c))) return rc;
    if ((rc = sync_all(c))) return rc;
    HIP_TRY(hipMemcpyAsync(dst, c->d_rec + first, (size_t)n * sizeof(lt_lane_record), hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LT_OK;
}

int lt_enqueue_records_to_device(lt_ctx* c, int first, int n, void* dst) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!dst) return fail(LT_ERR_INVALID, "null destination");
    if ((rc = set_device(c))) return rc;
    // stream-ordered behind the searches of each slot slice; no host synchronisation
    rc = for_each_slice(c, first, n, [&](hipStream_t st, int f0, int m) {
        HIP_TRY(hipMemcpyAsync(static_cast<lt_lane_record*>(dst) + (f0 - first), c->d_rec + f0, (size_t)m * sizeof(lt_lane_record),
                               hipMemcpyDeviceToDevice, st));
        return (int)LT_OK;
    });
    return rc;
}

int lt_set_frame_base(lt_ctx* c, int first, int n, int first_frame) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    c->rec_mirror_slot = -1;                  // the records change: the page-locked mirror of a one-frame search is stale
    if ((rc = set_device(c))) return rc;
    std::vector<lt_lane_record> tmp((size_t)n);
    if (n == 0) return LT_OK;
    if ((rc = download(c, c->d_rec + first, tmp.data(), tmp.size() * sizeof(lt_lane_record)))) return rc;
    for (int i = 0; i < n; ++i) tmp[i].frame = first_frame + i;
    HIP_TRY(hipMemcpyAsync(c->d_rec + first, tmp.data(), tmp.size() * sizeof(lt_lane_record), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LT_OK;
}

// ---- compute ------------------------------------------------------------------------------------------
int lt_mask_run(lt_ctx* c, int first, int n, const lt_filter_params* p) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if ((rc = validate_filter(p))) return rc;
    if ((rc = set_device(c))) return rc;
    if (n == 0) return LT_OK;
    const size_t ps = c->plane_bytes;
    rc = for_each_slice(c, first, n, [&](hipStream_t st, int f0, int m) {
        { StageScope t(c, ST_UNDISTORT, st);
          launch_undistort_rows(st, c->d_frames + (size_t)f0 * c->frame_bytes, c->frame_bytes, c->d_uxy, c->d_ufrac,
                                c->fe, c->d_und, c->und_px, f0, m); }
        { int mrc = note_range(c->readers, st, f0, f0 + m); if (mrc) return mrc; }
        { StageScope t(c, ST_WARP_SPLIT, st);
          launch_warp_split(st, c->d_und, c->und_px, f0, c->d_wxy, c->d_wfrac, c->fe, c->d_gamma,
                            c->d_cbrt, c->d_coef, c->d_plane[P_R] + (size_t)f0 * ps, c->d_plane[P_B] + (size_t)f0 * ps,
                            ps, m); }
        int frc = run_filter_chain(c, st, f0, m, p, c->calib.warp_h, c->calib.warp_w, n);
        return frc ? frc : note_written(c, st, f0, f0 + m);
    });
    if (rc) return rc;
    c->have_mask = true;
    mark_masks(c, first, n, 1, 0);
    return LT_OK;
}

int lt_filter_run(lt_ctx* c, int first, int n, const lt_filter_params* p) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if ((rc = validate_filter(p))) return rc;
    if ((rc = set_device(c))) return rc;
    if (!c->d_bev) return fail(LT_ERR_STATE, "lt_upload_bev has not been called");
    if (n == 0) return LT_OK;
    const size_t ps = c->plane_bytes;
    rc = for_each_slice(c, first, n, [&](hipStream_t st, int f0, int m) {
        { StageScope t(c, ST_SPLIT_BEV, st);
          launch_split_bev(st, c->d_bev + (size_t)f0 * c->bev_bytes, c->bev_bytes, (int)ps, c->d_gamma, c->d_cbrt,
                           c->d_coef, c->d_plane[P_R] + (size_t)f0 * ps, c->d_plane[P_B] + (size_t)f0 * ps, ps, m); }
        int frc = run_filter_chain(c, st, f0, m, p, c->calib.warp_h, c->calib.warp_w, n);
        return frc ? frc : note_written(c, st, f0, f0 + m);
    });
    if (rc) return rc;
    c->have_mask = true;
    mark_masks(c, first, n, 1, 0);
    return LT_OK;
}

static int ensure_search_stream(lt_ctx* c) {
    if (c->search) return LT_OK;
    if (c->search_cus > 0) {                  // the CUs the slots' streams were kept off (lt_set_search_cus)
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int mine = c->search_cus >= 2 ? 1 : c->search_cus;   // with two or more, the others are the download stream's
        for (int i = 0; i < mine && i < 256; ++i) mask[i >> 5] |= 1u << (i & 31);
        HIP_TRY(hipExtStreamCreateWithCUMask(&c->search, 8, mask));
    } else HIP_TRY(create_compute_stream(&c->search));
    return LT_OK;
}

int lt_sws_fit_run(lt_ctx* c, int first, int n, const lt_search_params* p) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    c->rec_mirror_slot = -1;                  // the records change: the page-locked mirror of a one-frame search is stale
    if ((rc = set_device(c))) return rc;
    if (!c->have_mask) return fail(LT_ERR_STATE, "no mask in the slots: run lt_mask_run or lt_upload_masks first");
    SearchGeom g;
    if ((rc = make_search_geom(c, p, false, g))) return rc;
    if ((rc = ensure_search_buffers(c, g.maxpix, g.maxlev))) return rc;
    if (g.nbands > c->maxbands) {
        if ((rc = sync_all(c))) return rc;
        dev_free(c->d_band_sums);
        if ((rc = dev_alloc(&c->d_band_sums, (size_t)c->capacity * g.nbands * c->calib.warp_w))) return rc;
        c->maxbands = g.nbands;
    }
    g.maxpix = c->maxpix;
    g.maxlev = c->maxlev;
    if (n == 0) return LT_OK;
    // the searches read the opened bit plane when the slots have one and the kernel that will run takes it
    const bool use_bits = masks_have_bits(c, first, n) && sws_fit_takes_bits(g, c->plane_bytes);
    if (!use_bits && (rc = ensure_u8_masks(c, first, n))) return rc;
    const int wpr = (c->calib.warp_w + 63) / 64;
    rc = for_each_slice(c, first, n, [&](hipStream_t st, int f0, int m) {
        StageScope t(c, ST_SWS_FIT, st);
        const MaskBits mb{use_bits ? c->d_bits_open + (size_t)f0 * c->bits_stride : nullptr, c->bits_stride, wpr};
        launch_sws_fit(st, c->d_plane[P_MASK] + (size_t)f0 * c->plane_bytes, c->plane_bytes, mb, g,
                       c->d_band_sums + (size_t)f0 * g.nbands * c->calib.warp_w, c->d_pix + (size_t)f0 * 2 * c->maxpix,
                       c->d_cent + (size_t)f0 * 2 * (c->maxlev + 2), c->d_rec + f0, m);
        if (n == 1) mirror_record(c, st, f0);
        return note_written(c, st, f0, f0 + m);
    });
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return LT_OK;
}

int lt_band_fit_run(lt_ctx* c, int first, int n, const lt_search_params* p, const double* prev) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    c->rec_mirror_slot = -1;                  // the records change: the page-locked mirror of a one-frame search is stale
    if (!prev) return fail(LT_ERR_INVALID, "band search needs the previous coefficients (last_left_coeffs/last_right_coeffs)");
    if ((rc = set_device(c))) return rc;
    if (!c->have_mask) return fail(LT_ERR_STATE, "no mask in the slots: run lt_mask_run or lt_upload_masks first");
    SearchGeom g;
    if ((rc = make_search_geom(c, p, true, g))) return rc;
    if ((rc = ensure_search_buffers(c, g.maxpix, 1))) return rc;
    g.maxpix = c->maxpix;
    if (n == 0) return LT_OK;
    BandPrev bp;
    std::memset(&bp, 0, sizeof bp);
    bool one_seed = true;     // every frame around the same curves (a group of frames behind a failure: the last valid fits)
    for (int i = 1; i < n && one_seed; ++i) one_seed = std::memcmp(prev, prev + (size_t)i * 6, 6 * sizeof(double)) == 0;
    if (one_seed) {   // the stateful stream: coefficients by value, no copy to wait for
        std::memcpy(bp.c, prev, sizeof bp.c);
        bp.by_value = 1;
    } else {
        if ((rc = sync_all(c))) return rc;
        HIP_TRY(hipMemcpyAsync(c->d_prev + (size_t)first * 6, prev, (size_t)n * 6 * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));  // prev is caller memory: do not keep reading it after return
    }
    const bool use_bits = masks_have_bits(c, first, n) && band_fit_takes_bits(g, c->plane_bytes);
    if (!use_bits && (rc = ensure_u8_masks(c, first, n))) return rc;
    const int wpr = (c->calib.warp_w + 63) / 64;
    rc = for_each_slice(c, first, n, [&](hipStream_t st, int f0, int m) {
        StageScope t(c, ST_BAND_FIT, st);
        const MaskBits mb{use_bits ? c->d_bits_open + (size_t)f0 * c->bits_stride : nullptr, c->bits_stride, wpr};
        // one frame (process()): the chain kernel with a chain of one, a third of the latency (LT_BAND_ONE=0: k_band_fit2)
        const char* one_env = n == 1 ? std::getenv("LT_BAND_ONE") : nullptr;
        lt_lane_record* mirror = n == 1 ? rec_mirror_device(c) : nullptr;
        if (n == 1 && !(one_env && one_env[0] == '0') &&
            launch_band_fit_one(st, mb, g, bp, c->d_pix + (size_t)f0 * 2 * c->maxpix, c->d_rec + f0, c->plane_bytes,
                                reinterpret_cast<const int*>(c->d_prev), mirror)) {
            if (mirror) {                    // the kernel itself leaves a copy of the record in page-locked memory
                c->rec_mirror_slot = f0;
                c->rec_mirror_stream = st;
            }
        } else {
            launch_band_fit(st, c->d_plane[P_MASK] + (size_t)f0 * c->plane_bytes, c->plane_bytes, mb, g, c->d_prev + (size_t)f0 * 6, bp,
                            c->d_pix + (size_t)f0 * 2 * c->maxpix, c->d_rec + f0, m);
            if (n == 1) mirror_record(c, st, f0);
        }
        return note_written(c, st, f0, f0 + m);
    });
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return LT_OK;
}

int lt_band_fit_chain_run(lt_ctx* c, int first, int n, const lt_search_params* p, const double* seed) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    c->rec_mirror_slot = -1;                  // the records change: the page-locked mirror of a one-frame search is stale
    if (!seed && first < 1) return fail(LT_ERR_INVALID, "a chain without seed coefficients continues from the record of slot first - 1");
    if ((rc = set_device(c))) return rc;
    if (!c->have_mask) return fail(LT_ERR_STATE, "no mask in the slots: run lt_mask_run or lt_upload_masks first");
    SearchGeom g;
    if ((rc = make_search_geom(c, p, true, g))) return rc;
    if ((rc = ensure_search_buffers(c, g.maxpix, 1))) return rc;
    g.maxpix = c->maxpix;
    if (!band_chain_supported(g, c->plane_bytes))
        return fail(LT_ERR_STATE, "chained band search needs a band of at most 64 columns (2 * bandwidth + 2) and a mask width that is a multiple of 4");
    if (n == 0) return LT_OK;
    BandPrev bp;
    std::memset(&bp, 0, sizeof bp);
    if (seed) {
        std::memcpy(bp.c, seed, sizeof bp.c);
        bp.by_value = 1;
    }
    const bool use_bits = masks_have_bits(c, first, n) && band_fit_takes_bits(g, c->plane_bytes);
    if (!use_bits && (rc = ensure_u8_masks(c, first, n))) return rc;
    // The chain runs on the context's search stream, behind whatever the slots' streams hold so far (the masks of these
    // slots, the search that wrote the seed record); those streams do not wait for it -- the mask chains of later frames run
    // beside it -- unless they touch its slots (for_each_slice).
    if ((rc = ensure_search_stream(c))) return rc;
    if (!c->h_cancel) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_cancel), 64, hipHostMallocMapped));
        *c->h_cancel = 0;
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_cancel), c->h_cancel, 0));
    }
    const int lo = seed ? first : first - 1, cnt = seed ? n : n + 1;     // with a device seed the seed record is collected too
    bool precise = true;
    // everything the slots' streams wrote into these slots and may not have finished (their masks; the search that left the
    // seed record); a seed record left by an earlier chain is ordered by the search stream itself
    if ((rc = wait_range(c->writers, c->search, lo, lo + cnt, &precise))) return rc;
    if (!precise) {
        rc = for_each_slice(c, lo, cnt, [&](hipStream_t st, int, int) {      // the ring has overflowed: wait for the streams' tails
            hipEvent_t e = next_order_event(c);
            if (!e) return fail(LT_ERR_HIP, "hipEventCreate failed");
            HIP_TRY(hipEventRecord(e, st));
            HIP_TRY(hipStreamWaitEvent(c->search, e, 0));
            return (int)LT_OK;
        });
        if (rc) return rc;
    }
    if (c->h_rec_stage_cap < c->capacity) {
        HIP_TRY(hipStreamSynchronize(c->search));
        if (c->h_rec_stage) (void)hipHostFree(c->h_rec_stage);
        c->h_rec_stage = nullptr;
        c->h_rec_stage_cap = 0;
        for (auto& t : c->chains) c->chain_event_pool.push_back(t.done);   // tickets of the old staging block: nothing to collect any more
        c->chains.clear();
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_rec_stage), (size_t)c->capacity * sizeof(lt_lane_record), hipHostMallocDefault));
        c->h_rec_stage_cap = c->capacity;
    }
    const int wpr = (c->calib.warp_w + 63) / 64;
    {
        StageScope t(c, ST_BAND_FIT, c->search);
        const MaskBits mb{use_bits ? c->d_bits_open + (size_t)first * c->bits_stride : nullptr, c->bits_stride, wpr};
        launch_band_chain(c->search, c->d_plane[P_MASK] + (size_t)first * c->plane_bytes, c->plane_bytes, mb, g, seed ? nullptr : c->d_rec + first - 1,
                          bp, c->d_pix + (size_t)first * 2 * c->maxpix, c->d_rec + first, n, c->d_cancel, *c->h_cancel);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->h_rec_stage + lo, c->d_rec + lo, (size_t)cnt * sizeof(lt_lane_record), hipMemcpyDeviceToHost, c->search));
    hipEvent_t done = nullptr;
    if (!c->chain_event_pool.empty()) { done = c->chain_event_pool.back(); c->chain_event_pool.pop_back(); }
    else if (hipEventCreateWithFlags(&done, hipEventDisableTiming) != hipSuccess) return fail(LT_ERR_HIP, "hipEventCreate failed");
    HIP_TRY(hipEventRecord(done, c->search));
    while (c->chains.size() >= 32) {            // tickets nobody collected: the oldest goes -- once its chain has really ended
        // (a cancelled chain runs one more frame, and later work on its slots is ordered behind tickets only: dropping the
        // ticket of a chain still running would let mask / search launches race with it)
        HIP_TRY(hipEventSynchronize(c->chains.front().done));
        c->chain_event_pool.push_back(c->chains.front().done);
        c->chains.erase(c->chains.begin());
    }
    c->chains.push_back({lo, cnt, done, first});
    return LT_OK;
}

int lt_set_urgent(lt_ctx* c, int on) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    int rc = set_device(c);
    if (rc) return rc;
    if (on && !c->urgent && create_compute_stream(&c->urgent, c->search_cus) != hipSuccess) return fail(LT_ERR_HIP, "hipStreamCreate failed");
    if (!on && c->urgent_on && c->urgent) HIP_TRY(hipStreamSynchronize(c->urgent));   // leaving: nothing of it is left in flight unseen
    c->urgent_on = on != 0;
    return LT_OK;
}

int lt_set_search_cus(lt_ctx* c, int n) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    if (n < 0 || n > 64) return fail(LT_ERR_INVALID, "the search stream can have 0 .. 64 CUs to itself");
    int rc = set_device(c);
    if (rc) return rc;
    if (n == c->search_cus) return LT_OK;
    if ((rc = sync_all(c))) return rc;
    if ((rc = flush_stage_events(c))) return rc;
    // every stream that carries slot kernels is recreated with (or without) the reservation; the search stream follows on its
    // next use
    std::vector<hipStream_t> fresh;
    for (size_t i = 0; i < c->streams.size(); ++i) {
        hipStream_t st = nullptr;
        if (create_compute_stream(&st, n) != hipSuccess) {
            for (auto f : fresh) (void)hipStreamDestroy(f);
            return fail(LT_ERR_HIP, "stream with a CU mask could not be created");
        }
        fresh.push_back(st);
    }
    for (auto st : c->streams) if (st) (void)hipStreamDestroy(st);
    c->streams = fresh;
    c->stream = c->streams.empty() ? c->stream : c->streams[0];
    if (c->search) { (void)hipStreamDestroy(c->search); c->search = nullptr; }
    if (c->present) { (void)hipStreamDestroy(c->present); c->present = nullptr; }
    if (c->urgent) { (void)hipStreamDestroy(c->urgent); c->urgent = nullptr; c->urgent_on = false; }
    if (c->dl) { (void)hipStreamDestroy(c->dl); c->dl = nullptr; }
    c->search_cus = n;
    return LT_OK;
}

int lt_band_fit_chain_cancel(lt_ctx* c) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    if (c->h_cancel) __atomic_fetch_add(c->h_cancel, 1, __ATOMIC_RELEASE);   // chains enqueued so far carry an older epoch
    return LT_OK;
}

int lt_band_fit_chain_collect(lt_ctx* c, int first, int n, lt_lane_record* out) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!out) return fail(LT_ERR_INVALID, "null output buffer");
    if (n == 0) return LT_OK;
    if ((rc = set_device(c))) return rc;
    // the most recent chain that covers the range decides (an older, superseded chain over the same slots is dropped) -- a chain
    // that searched the range's first slot itself before one that only holds it as its seed record (a one-frame chain with
    // another chain behind it: collecting the first must not use up the ticket of the second)
    int hit = -1;
    for (int pass = 0; pass < 2 && hit < 0; ++pass)
        for (int i = (int)c->chains.size() - 1; i >= 0; --i) {
            const lt_ctx::ChainTicket& t = c->chains[(size_t)i];
            if ((pass ? t.first : t.own) <= first && first + n <= t.first + t.n) { hit = i; break; }
        }
    if (hit < 0) return fail(LT_ERR_STATE, "no chained search covers slots [%d, %d)", first, first + n);
    HIP_TRY(hipEventSynchronize(c->chains[(size_t)hit].done));
    std::memcpy(out, c->h_rec_stage + first, (size_t)n * sizeof(lt_lane_record));
    for (int i = 0; i <= hit; ++i) c->chain_event_pool.push_back(c->chains[(size_t)i].done);   // this ticket and everything older
    c->chains.erase(c->chains.begin(), c->chains.begin() + hit + 1);
    return LT_OK;
}

// ---- host-buffer wrappers --------------------------------------------------------------------------------
int lt_mask_batch(lt_ctx* c, const uint8_t* frames, int n, const lt_filter_params* p, uint8_t* masks) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    int rc;
    if (n > c->capacity && (rc = lt_reserve(c, n))) return rc;
    if ((rc = lt_upload_frames(c, frames, 0, n))) return rc;
    if ((rc = lt_mask_run(c, 0, n, p))) return rc;
    return lt_download_masks(c, 0, n, masks);
}

int lt_sws_fit_batch(lt_ctx* c, const uint8_t* masks, int n, const lt_search_params* p, lt_lane_record* out) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    int rc;
    if (n > c->capacity && (rc = lt_reserve(c, n))) return rc;
    if (masks && (rc = lt_upload_masks(c, masks, 0, n))) return rc;
    if ((rc = lt_sws_fit_run(c, 0, n, p))) return rc;
    return lt_download_records(c, 0, n, out);
}

int lt_band_fit_batch(lt_ctx* c, const uint8_t* masks, int n, const lt_search_params* p, const double* prev,
                      lt_lane_record* out) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    int rc;
    if (n > c->capacity && (rc = lt_reserve(c, n))) return rc;
    if (masks && (rc = lt_upload_masks(c, masks, 0, n))) return rc;
    if ((rc = lt_band_fit_run(c, 0, n, p, prev))) return rc;
    return lt_download_records(c, 0, n, out);
}

// ---- single-image operators ------------------------------------------------------------------------------
int lt_bilateral_adaptive_threshold(lt_ctx* c, const uint8_t* img, int h, int w, int ksize, int C, int mode, int tv,
                                    int fv, uint8_t* out) {
    if (!c || !img || !out) return fail(LT_ERR_INVALID, "null argument");
    if (mode != 0 && mode != 1) return fail(LT_ERR_INVALID, "Unexpected mode value. Expected value is 'floor' or 'ceil'.");
    if (h < 1 || w < 1 || ksize < 1 || ksize > 128) return fail(LT_ERR_INVALID, "bad image size or ksize (1..128)");
    if (tv < 0 || tv > 255 || fv < 0 || fv > 255) return fail(LT_ERR_INVALID, "true/false values must be in [0,255]");
    int rc = set_device(c);
    if (rc) return rc;
    const size_t n = (size_t)h * w;
    uint8_t *d_in = nullptr, *d_out = nullptr;
    if ((rc = dev_alloc(&d_in, n))) return rc;
    if ((rc = dev_alloc(&d_out, n))) { dev_free(d_in); return rc; }
    hipError_t e = hipMemcpyAsync(d_in, img, n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        launch_bilateral(c->stream, d_in, d_out, h, w, ksize, C, mode, tv, fv, n, 1);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(d_in);
    dev_free(d_out);
    if (e != hipSuccess) return fail(LT_ERR_HIP, "bilateral threshold failed: %s", hipGetErrorString(e));
    return LT_OK;
}

int lt_filter_lane_points(lt_ctx* c, const uint8_t* bev, int h, int w, const lt_filter_params* p, uint8_t* mask) {
    if (!c || !bev || !mask) return fail(LT_ERR_INVALID, "null argument");
    int rc = validate_filter(p);
    if (rc) return rc;
    if (h < 1 || w < 1 || h > 16384 || w > 4096) return fail(LT_ERR_INVALID, "bad image size (width <= 4096, height <= 16384)");
    if ((rc = set_device(c))) return rc;
    // a private one-slot arena of the requested size (the image may differ from the calibration's BEV size)
    lt_ctx tmp;
    tmp.device = c->device;
    tmp.stream = c->stream;
    tmp.se5 = c->se5; tmp.se29 = c->se29; tmp.se55 = c->se55;
    tmp.brute_tophat = c->brute_tophat;
    tmp.plane_bytes = (size_t)h * w;
    tmp.bits_stride = (size_t)h * ((w + 63) / 64);
    tmp.capacity = 1;
    uint8_t* d_bev = nullptr;
    auto cleanup = [&]() {
        for (auto& q : tmp.d_plane) dev_free(q);
        dev_free(tmp.d_bits_merged);
        dev_free(tmp.d_bits_eroded);
        dev_free(d_bev);
        tmp.stream = nullptr;
    };
    for (int i = 0; i < P_COUNT; ++i)
        if ((rc = dev_alloc(&tmp.d_plane[i], tmp.plane_bytes))) { cleanup(); return rc; }
    if ((rc = dev_alloc(&d_bev, tmp.plane_bytes * 3))) { cleanup(); return rc; }
    if ((rc = dev_alloc(&tmp.d_bits_merged, tmp.bits_stride))) { cleanup(); return rc; }
    if ((rc = dev_alloc(&tmp.d_bits_eroded, tmp.bits_stride))) { cleanup(); return rc; }
    hipError_t e = hipMemcpyAsync(d_bev, bev, tmp.plane_bytes * 3, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        launch_split_bev(c->stream, d_bev, tmp.plane_bytes * 3, (int)tmp.plane_bytes, c->d_gamma, c->d_cbrt, c->d_coef,
                         tmp.d_plane[P_R], tmp.d_plane[P_B], tmp.plane_bytes, 1);
        rc = run_filter_chain(&tmp, c->stream, 0, 1, p, h, w, 1, true);
        if (rc == LT_OK) e = hipMemcpyAsync(mask, tmp.d_plane[P_MASK], tmp.plane_bytes, hipMemcpyDeviceToHost, c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    cleanup();
    if (rc) return rc;
    if (e != hipSuccess) return fail(LT_ERR_HIP, "filter_lane_points failed: %s", hipGetErrorString(e));
    return LT_OK;
}

int lt_morph_ellipse(lt_ctx* c, const uint8_t* img, int h, int w, int k, int op, int direct, uint8_t* out) {
    if (!c || !img || !out) return fail(LT_ERR_INVALID, "null argument");
    if (h < 1 || w < 1 || h > 16384 || w > 16384) return fail(LT_ERR_INVALID, "bad image size");
    if (k != 5 && k != 29 && k != 55) return fail(LT_ERR_INVALID, "structuring element size must be 5, 29 or 55");
    if (op < 0 || op > 3) return fail(LT_ERR_INVALID, "op must be 0 erode, 1 dilate, 2 tophat, 3 open");
    int rc = set_device(c);
    if (rc) return rc;
    const size_t n = (size_t)h * w;
    uint8_t *d_in = nullptr, *d_t = nullptr, *d_out = nullptr;
    if ((rc = dev_alloc(&d_in, n)) || (rc = dev_alloc(&d_t, n)) || (rc = dev_alloc(&d_out, n))) {
        dev_free(d_in); dev_free(d_t); dev_free(d_out);
        return rc;
    }
    const EllipseSE& se = k == 5 ? c->se5 : (k == 29 ? c->se29 : c->se55);
    auto pass = [&](const uint8_t* src, uint8_t* dst, const uint8_t* minuend, bool dilate) {
        if (k == 5 || direct) launch_morph_ellipse(c->stream, src, dst, minuend, h, w, se, dilate, n, 1);
        else launch_morph_runs(c->stream, src, dst, minuend, h, w, k, dilate, n, 1);
    };
    hipError_t e = hipMemcpyAsync(d_in, img, n, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        if (op == 0) pass(d_in, d_out, nullptr, false);
        else if (op == 1) pass(d_in, d_out, nullptr, true);
        else { pass(d_in, d_t, nullptr, false); pass(d_t, d_out, op == 2 ? d_in : nullptr, true); }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(d_in); dev_free(d_t); dev_free(d_out);
    if (e != hipSuccess) return fail(LT_ERR_HIP, "morph_ellipse failed: %s", hipGetErrorString(e));
    return LT_OK;
}

int lt_fit_poly2(lt_ctx* c, const int32_t* ys, const int32_t* xs, int n, int h, int w, double coef[3], int* rank_deficient) {
    if (!c || !coef || !rank_deficient || n < 0 || (n > 0 && (!ys || !xs))) return fail(LT_ERR_INVALID, "bad argument");
    if (h < 1 || w < 1) return fail(LT_ERR_INVALID, "bad image size");
    int rc = set_device(c);
    if (rc) return rc;
    std::vector<uint32_t> packed((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (ys[i] < 0 || ys[i] > 65535 || xs[i] < 0 || xs[i] > 65535) return fail(LT_ERR_INVALID, "pixel coordinate outside [0, 65535]");
        packed[i] = ((uint32_t)ys[i] << 16) | (uint32_t)xs[i];
    }
    uint32_t* d_pix = nullptr;
    double* d_out = nullptr;
    if ((rc = dev_alloc(&d_pix, (size_t)n))) return rc;
    if ((rc = dev_alloc(&d_out, 4))) { dev_free(d_pix); return rc; }
    double out[4] = {0, 0, 0, 1};
    hipError_t e = n ? hipMemcpyAsync(d_pix, packed.data(), (size_t)n * 4, hipMemcpyHostToDevice, c->stream) : hipSuccess;
    if (e == hipSuccess) {
        launch_fit_list(c->stream, d_pix, n, h, w, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, sizeof out, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    dev_free(d_pix);
    dev_free(d_out);
    if (e != hipSuccess) return fail(LT_ERR_HIP, "fit_poly2 failed: %s", hipGetErrorString(e));
    coef[0] = out[0]; coef[1] = out[1]; coef[2] = out[2];
    *rank_deficient = out[3] != 0.0;
    return LT_OK;
}

// ---- measurement ---------------------------------------------------------------------------------------
int lt_timer_start(lt_ctx* c) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    { int rc = sync_all(c); if (rc) return rc; }
    HIP_TRY(hipEventRecord(c->ev0, c->stream));
    return LT_OK;
}

int lt_timer_stop(lt_ctx* c, float* ms) {
    if (!c || !ms) return fail(LT_ERR_INVALID, "null argument");
    { int rc = sync_all(c); if (rc) return rc; }
    HIP_TRY(hipEventRecord(c->ev1, c->stream));
    HIP_TRY(hipEventSynchronize(c->ev1));
    HIP_TRY(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return LT_OK;
}

int lt_last_threshold_path(lt_ctx* c) {
    if (!c) { (void)fail(LT_ERR_INVALID, "null context"); return LT_NO_CONTEXT; }
    return c->last_threshold_path;
}

int lt_last_adaptive_path(lt_ctx* c) {
    if (!c) { (void)fail(LT_ERR_INVALID, "null context"); return LT_NO_CONTEXT; }
    return c->last_adaptive_path;
}

int lt_set_stage_timing(lt_ctx* c, int enabled) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    int rc = flush_stage_events(c);
    c->stage_timing = enabled != 0;
    return rc;
}

int lt_stage_reset(lt_ctx* c) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    int rc = flush_stage_events(c);
    std::memset(c->stage_ms, 0, sizeof c->stage_ms);
    std::memset(c->stage_launches, 0, sizeof c->stage_launches);
    return rc;
}

int lt_stage_ms(lt_ctx* c, float* ms, int32_t* launches, int n) {
    if (!c || !ms) return fail(LT_ERR_INVALID, "null argument");
    int rc = flush_stage_events(c);
    for (int i = 0; i < n && i < LT_NUM_STAGES; ++i) {
        ms[i] = c->stage_ms[i];
        if (launches) launches[i] = c->stage_launches[i];
    }
    return rc;
}

}  // extern "C"

namespace lt {
int set_error(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
int ctx_device(lt_ctx* c) { return c->device; }
int ctx_streams(lt_ctx* c, hipStream_t* out, int cap) {
    int n = 0;
    for (int i = 0; i < c->nstreams && i < (int)c->streams.size() && n < cap; ++i) out[n++] = c->streams[(size_t)i];
    return n;
}
int ctx_sync(lt_ctx* c) { return lt_sync(c); }
int ctx_enqueue_records(lt_ctx* c, int first, int n, lt_lane_record* dst) { return lt_enqueue_records_to_device(c, first, n, dst); }
}  // namespace lt
