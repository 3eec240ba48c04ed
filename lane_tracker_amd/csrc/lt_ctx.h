// The context behind the C ABI (struct lt_ctx) and the helpers its translation units share:
//   lt_api.cpp     -- create / destroy / reserve, uploads, downloads, the mask chain, the searches, measurement
//   lt_memory.cpp  -- device-memory cache, page-locked host memory, the host copy threads
//   lt_present.cpp -- presentation stage: lane overlay, text, annotated frames on their way back
//   lt_chain.cpp   -- the chained band search of a stream (tickets, cancel, collect)
// Not installed; the public ABI is include/lane_tracker_amd.h.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <functional>
#include <memory>
#include <vector>

#include "lt_internal.h"

namespace lt {

int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));   // fills lt_last_error(), returns code

#define HIP_TRY(expr)                                                                               \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return lt::fail(LT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

enum Stage {
    ST_UNDISTORT = 0, ST_WARP_SPLIT, ST_ERODE_R, ST_TOPHAT_R, ST_ERODE_B, ST_TOPHAT_B, ST_THRESHOLD, ST_MERGE,
    ST_OPEN, ST_SWS_FIT, ST_BAND_FIT, ST_SPLIT_BEV
};

enum Plane { P_R = 0, P_B, P_THR, P_THB, P_MERGED, P_MASK, P_T0, P_T1, P_T2, P_T3, P_COUNT };

}  // namespace lt

using lt::P_COUNT;

struct lt_ctx {
    lt_calib calib{};
    int device = 0;
    hipStream_t stream = nullptr;             // = streams[0]
    std::vector<hipStream_t> streams;         // slot s runs on streams[s * nstreams / capacity]
    hipStream_t copy = nullptr;               // lt_upload_frame_rest: the rows the path does not read, off the critical path
    // One frame's rows written by the calling thread straight into the slot through the PCIe aperture (lt_upload_frame_rows_enqueue of
    // a small call; lt_set_direct_upload): -1 = not yet decided for the current frame buffer, 0 = no (no large BAR, the buffer is
    // not mapped into this process, or switched off), 1 = yes.
    int direct_upload = -1;
    bool direct_upload_wanted = true;
    unsigned long long lane_spec_reader_seq = 0;   // readers.lazy_seq when the completion word of lt_present_lane_from_fit_async was launched
    unsigned long long direct_uploads = 0;    // calls that took the aperture (lt_set_direct_upload(ctx, -1) reports the state, tests read this through it)
    int nstreams = 1;
    hipDeviceProp_t prop{};
    lt::FrontEndGeom fe{};
    int cam_r0 = 0, cam_r1 = 0;               // camera rows the undistortion reads (its taps for rows [fe.r0, fe.r0 + nrows))
    lt::EllipseSE se5{}, se29{}, se55{};
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
    uint8_t* d_plane[P_COUNT] = {};   // P_R, P_B, P_THR, P_THB, P_T0 with the slots; the others on first use (ensure_plane)
    uint8_t* d_side_scratch = nullptr;     // the eroded R plane of a one- or two-frame chain: two planes per stream that can run one (run_filter_chain)
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
    // per slot: the WHOLE camera frame is on the device (lt_upload_frames, or lt_upload_frame_rows + lt_upload_frame_rest; the row-run
    // uploads bring parts only) / the WHOLE annotated frame has been drawn (lt_overlay_run; the row-run and strip overlays draw
    // parts).  Whole-frame overlays and downloads refuse slots that are not (LT_ERR_STATE).
    std::vector<uint8_t> frame_full, annot_full;
    // per slot: the R / Lab-b planes (and the undistorted rows) are those of the camera rows the slot holds now -- set by lt_mask_run's
    // front end, cleared by every upload of camera rows and by lt_filter_run.  A second lt_mask_run over such slots with other
    // filter parameters (the second try of a frame, lane_tracker.py:1081-1101) skips the undistortion and the warp: same planes.
    std::vector<uint8_t> front_ok;
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
    uint8_t* d_strip = nullptr;       // strip mode (lt_overlay_run_strip): rows [ov_r0, ov_r1) of every slot's annotated frame, packed
    size_t strip_bytes = 0;
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
    // lt_present_lane_from_fit_async: the plot rows (ploty, ploty ** 2) the device evaluates the averaged curves at, as last sent
    double* d_ploty = nullptr;        // [2][ploty_rows]
    std::vector<double> h_ploty;      // what d_ploty holds (compared on every call: 2 x 9 KB)
    hipStream_t lane_spec_stream = nullptr;   // the stream a speculative overlay of process() runs on, until lt_present_finish has waited for it
    unsigned lane_spec_ticket = 0;            // ... and the ticket a one-thread launch behind it stores at h_rec + 128 bytes (0: none, wait for the stream)
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
        // work on the slots' streams that was NOT given an event of its own (note_range_frame: the one-frame calls of a context of
        // one or two slots -- LaneTracker.process() -- where every kernel of a frame runs on one stream and an event record between
        // two of them costs the frame ~6 us of device time each: profiles/r06_process_timeline.txt).  A waiter on another stream
        // then waits for the tails of the slots' streams, as after an overflow; cleared by the next full synchronisation.
        bool lazy = false;
        // ... and how much of it the HOST has seen finished: every lazy note counts (`lazy_seq`); a completion word the host polls
        // (lt_present_finish: the word stored behind the frame's overlay) carries the count at its launch, and seeing it moves
        // `lazy_seen` there -- everything noted up to then ran in front of it on the same stream.  lazy_seq == lazy_seen: nothing
        // unrecorded is in flight (the direct upload's question, lt_upload_frame_rows_enqueue).
        unsigned long long lazy_seq = 0, lazy_seen = 0;
        void reset() { head = count = 0; overflow = false; lazy = false; lazy_seen = lazy_seq; }
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
    uint8_t* h_lists = nullptr;               // page-locked landing area of lt_download_lane_lists (record + list region + centroids)
    size_t h_lists_bytes = 0;
    int ov_r0 = 0, ov_r1 = 0;                 // camera rows the lane overlay can change (lt_overlay_configure)
    lt_lane_record* h_rec = nullptr;          // page-locked mirror of the record of the last ONE-frame search (mirror_record)
    int rec_mirror_slot = -1;                 // the slot whose record the mirror holds once rec_mirror_stream is idle; -1: none
    hipStream_t rec_mirror_stream = nullptr;
    // the kernel that fills the mirror stores this ticket behind the record (k_band_chain3): lt_download_records polls for it
    // instead of waiting for the stream (0: the mirror is filled by a copy launch, wait for the stream)
    unsigned rec_ticket = 0, rec_ticket_counter = 0;
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

namespace lt {

// ---- LT_TRACE_START=1 (lt_memory.cpp): one line per set-up phase on stderr -------------------------------
//   lt_start <seconds on CLOCK_MONOTONIC, = Python's time.monotonic()> <what> <ms> [<bytes>]
// What a first window of a stream or the first calls of process() spend outside the kernels: context growth (lt_reserve,
// hipMalloc, cache hits and misses), page-locked allocations, table builds and uploads.  tools/cold_start.py reads them.
bool trace_on();
double trace_now();                                          // seconds, CLOCK_MONOTONIC
void trace_line(const char* what, double t0, size_t bytes = 0);   // prints the phase that started at t0 (trace_now())
struct TraceScope {
    const char* what;
    size_t bytes;
    double t0;
    TraceScope(const char* w, size_t b = 0) : what(w), bytes(b), t0(trace_on() ? trace_now() : 0.0) {}
    ~TraceScope() { if (trace_on()) trace_line(what, t0, bytes); }
};

// ---- host copy threads, staging blocks (lt_memory.cpp), for lt_present.cpp ----------------------------------
int host_submit_copy2d(int group, uint8_t* dst, size_t dpitch, const uint8_t* src, size_t spitch, size_t width, size_t height,
                       std::shared_ptr<void> hold);          // `hold` is released when the last piece of the copy has run
int host_submit_fn(int group, std::function<void()> fn, bool many);
int host_reserve(int group);                                 // a piece that will be submitted later: the group waits for it
void host_unreserve(int group);
void host_after_event(hipEvent_t ev, int device, std::function<void()> then);   // `then` runs on the library's waiter thread once `ev` has fired
int host_copy_threads();
int host_copy_pollers();                                     // copy threads polling for work right now (they take a piece within a microsecond)
void* pinned_block_acquire(size_t bytes);                    // page-locked staging, pooled per size (nullptr: allocation failed)
void pinned_block_release(void* p, size_t bytes);
hipEvent_t pooled_event();                                   // process-wide events (they outlive the context that recorded them)
void pooled_event_release(hipEvent_t e);

// ---- device memory (lt_memory.cpp): a cache in front of hipMalloc / hipFree ------------------------------
void* cached_alloc(size_t bytes);
void cached_free(void* p);
// Frees of one thread bracketed by this wait for `device` ONCE and enter the cache together when the scope closes -- behind
// the allocations made inside it (lt_reserve: the larger blocks first, then the old ones into the cache; lt_destroy).
struct FreeScope {
    explicit FreeScope(int device);
    ~FreeScope();
    FreeScope(const FreeScope&) = delete;
    FreeScope& operator=(const FreeScope&) = delete;
};

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

// ---- streams, slot ranges, ordering (lt_api.cpp) ---------------------------------------------------------
int sync_all(lt_ctx* c);
int note_range(lt_ctx::RangeEvents& r, hipStream_t st, int lo, int hi);
int note_range_frame(lt_ctx* c, lt_ctx::RangeEvents& r, hipStream_t st, int lo, int hi);   // ... without an event where the context is one frame's (RangeEvents::lazy)
int wait_range(const lt_ctx::RangeEvents& r, hipStream_t waiter, int lo, int hi, bool* precise);
int note_written(lt_ctx* c, hipStream_t st, int lo, int hi);
hipEvent_t next_order_event(lt_ctx* c);
int wait_chains(lt_ctx* c, hipStream_t st, int lo, int hi);
int flush_stage_events(lt_ctx* c);
int check_slots(lt_ctx* c, int first, int n);
int set_device(lt_ctx* c);
hipError_t create_compute_stream(hipStream_t* st, int reserved = 0);
// (stream_get / stream_put -- the per-process stream pool -- are declared in lt_internal.h)
int download(lt_ctx* c, const void* src, void* dst, size_t bytes);
int ensure_bev(lt_ctx* c);
int ensure_plane(lt_ctx* c, int idx);
int ensure_search_buffers(lt_ctx* c, int maxpix, int maxlev);
bool masks_have_bits(const lt_ctx* c, int first, int n);
int ensure_u8_masks(lt_ctx* c, int first, int n);
int make_search_geom(lt_ctx* c, const lt_search_params* p, bool band, SearchGeom& g);
int ensure_band_sums(lt_ctx* c, int nbands);
void mark_frames(lt_ctx* c, int first, int n, int full);
void mark_annot(lt_ctx* c, int first, int n, int full);
int first_partial(const std::vector<uint8_t>& v, int first, int n);
int ensure_search_stream(lt_ctx* c);                          // lt_chain.cpp
int ensure_chain_buffers(lt_ctx* c);                          // lt_chain.cpp
int warm_presentation(lt_ctx* c, bool strips);                // lt_present.cpp




// Slot -> stream mapping is fixed (contiguous slices of the capacity), so consecutive stages of one
// slot stay ordered on one stream while different slices overlap: the latency-bound search of one
// slice runs under the mask chain of another.  Calls fn(stream, first, n) for every non-empty piece.
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

}  // namespace lt
