// C-ABI layer of liblane_tracker_amd.so (see include/lane_tracker_amd.h).
// Owns the context: HIP stream, calibration tables, frame slots; sequences the kernel chain of
// LaneTracker.find_lane_points() (lane_tracker.py:795-874) for a batch of independent frames.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <tuple>
#include <vector>
#include <immintrin.h>
#include <sys/mman.h>
#include <unistd.h>

#include <hip/hip_ext.h>

#include "lt_ctx.h"

using namespace lt;

namespace {

thread_local std::string g_err;

}  // namespace

namespace lt {

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

const char* kStageNames[LT_NUM_STAGES] = {"undistort_rows", "warp_split", "erode_r29", "tophat_r29", "erode_b55",
                                          "tophat_b55", "threshold", "merge", "open5", "sws_fit", "band_fit",
                                          "split_bev"};

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
// One-frame calls of a context of one or two slots (LaneTracker.process(): capacity 2, one slot stream, the frame's kernels one
// behind the other on it): no event -- stream order is all the ordering the frame needs, and a hipEventRecord between two
// dependent kernels kept the second one waiting ~6 us (three of them on the way to the record: undistortion -> warp, open ->
// search, search -> lane spans).  Whoever waits from ANOTHER stream finds `lazy` set and waits for the slot stream's tail.
int note_range_frame(lt_ctx* c, lt_ctx::RangeEvents& r, hipStream_t st, int lo, int hi) {
    if (c->capacity <= 2 && !c->urgent_on && !c->stage_timing && c->nstreams == 1 && st == c->stream) {
        r.lazy = true;
        ++r.lazy_seq;
        return LT_OK;
    }
    return note_range(r, st, lo, hi);
}
// `waiter` waits for the entries that touch slots [lo, hi); *precise = false if the ring has overflowed or work was enqueued
// without an event (the caller then waits for stream tails)
int wait_range(const lt_ctx::RangeEvents& r, hipStream_t waiter, int lo, int hi, bool* precise) {
    *precise = !r.overflow && !r.lazy;
    if (r.overflow) return LT_OK;
    for (unsigned i = 0; i < r.count; ++i) {
        const lt_ctx::RangeEvents::Entry& w = r.e[(r.head + i) % (unsigned)r.e.size()];
        if (w.lo < hi && w.hi > lo) HIP_TRY(hipStreamWaitEvent(waiter, w.ev, 0));
    }
    return LT_OK;
}
int note_written(lt_ctx* c, hipStream_t st, int lo, int hi) { return note_range(c->writers, st, lo, hi); }
static int note_written_frame(lt_ctx* c, hipStream_t st, int lo, int hi, int n_call) {
    return n_call == 1 ? note_range_frame(c, c->writers, st, lo, hi) : note_range(c->writers, st, lo, hi);
}

// `st` waits for the chains still outstanding (not collected) that read or write slots [lo, hi): ticket by ticket, so that work on
// a frame in front of a running chain -- the second try of a failed frame while the frames behind it are already chained -- does
// not wait for that chain
int wait_chains(lt_ctx* c, hipStream_t st, int lo, int hi) {
    for (const auto& t : c->chains)
        if (t.first < hi && t.first + t.n > lo) HIP_TRY(hipStreamWaitEvent(st, t.done, 0));
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
    c->frame_full.clear();
    c->annot_full.clear();
    c->front_ok.clear();
    dev_free(c->d_rec);
    dev_free(c->d_prev);
    dev_free(c->d_pix);
    dev_free(c->d_cent);
    dev_free(c->d_band_sums);
    dev_free(c->d_spans);
    dev_free(c->d_ploty);
    c->h_ploty.clear();
    dev_free(c->d_annot);
    dev_free(c->d_strip);
    dev_free(c->d_side_scratch);
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

// Which slots hold their whole camera frame / whole annotated frame (see lt_ctx.h): the row-run entry points leave the other rows
// of a slot as whatever the block held before -- device memory comes back from the cache dirty -- and a whole-frame call on such
// a slot would hand out another stream's pixels.
void mark_frames(lt_ctx* c, int first, int n, int full) {
    for (int i = first; i < first + n && i < (int)c->frame_full.size(); ++i) c->frame_full[(size_t)i] = (uint8_t)full;
}
static void front_stale(lt_ctx* c, int first, int n) {      // new camera rows in these slots: their planes are the old frames'
    for (int i = first; i < first + n && i < (int)c->front_ok.size(); ++i) c->front_ok[(size_t)i] = 0;
}
void mark_annot(lt_ctx* c, int first, int n, int full) {
    for (int i = first; i < first + n && i < (int)c->annot_full.size(); ++i) c->annot_full[(size_t)i] = (uint8_t)full;
}
int first_partial(const std::vector<uint8_t>& v, int first, int n) {
    for (int i = first; i < first + n && i < (int)v.size(); ++i)
        if (!v[(size_t)i]) return i;
    return -1;
}

int ensure_band_sums(lt_ctx* c, int nbands) {
    if (nbands <= c->maxbands) return LT_OK;
    int rc = sync_all(c);
    if (rc) return rc;
    dev_free(c->d_band_sums);
    if ((rc = dev_alloc(&c->d_band_sums, (size_t)c->capacity * nbands * c->calib.warp_w))) return rc;
    c->maxbands = nbands;
    return LT_OK;
}

// one no-op launch per kernel translation unit, once per process and device: their code objects load now (a few ms each), not under
// the first window of a stream or the first frame of a video
static void preload_kernels(int device, hipStream_t s) {
    static std::mutex m;
    static std::vector<int> done;
    std::lock_guard<std::mutex> g(m);
    if (std::find(done.begin(), done.end(), device) != done.end()) return;
    done.push_back(device);
    TraceScope ts_("lt_create:preload_kernels");
    preload_k_frontend(s);
    preload_k_filter(s);
    preload_k_tophat(s);
    preload_k_threshold(s);
    preload_k_threshold_walk(s);
    preload_k_adaptive_walk(s);
    preload_k_search(s);
    preload_k_overlay(s);
    (void)hipStreamSynchronize(s);
    (void)hipGetLastError();
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
    int rc = ensure_plane(c, P_MASK);
    if (rc) return rc;
    if (!any) return LT_OK;
    if ((rc = sync_all(c))) return rc;
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

// Planes only some paths use are allocated when one of those paths runs first (the whole capacity at once): the u8 mask
// (d_plane[P_MASK]: lt_upload_masks, lt_download_masks, searches outside the bit-plane kernels' limits), the expanded merged
// plane (lt_download_plane), the scratch planes of the older threshold kernels (P_T1 .. P_T3).  A 768-slot context is 10.4 GB
// instead of 14.6 -- and device memory that has been used before costs ~16 ms per GB to allocate (the driver clears it: NOTES D.2).
int ensure_plane(lt_ctx* c, int idx) {
    if (c->d_plane[idx]) return LT_OK;
    int rc = dev_alloc(&c->d_plane[idx], (size_t)c->capacity * c->plane_bytes);
    if (rc) return rc;
    if (idx == P_MASK) {             // slots nobody has written a mask to read as zeros
        HIP_TRY(hipMemsetAsync(c->d_plane[idx], 0, (size_t)c->capacity * c->plane_bytes, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return LT_OK;
}

constexpr int SIDE_LANES = 10;   // eroded-R scratch of the one- and two-frame chain: one per slice stream (up to 8), one for every other stream

// filter_lane_points() on planes P_R / P_B of the given slots (lane_tracker.py:210-238)
int run_filter_chain(lt_ctx* c, hipStream_t s, int first, int n, const lt_filter_params* p, int h, int w, int call_frames,
                     bool u8_mask = false) {
    const size_t ps = c->plane_bytes, off = (size_t)first * ps;
    uint8_t* R = c->d_plane[P_R] + off;
    uint8_t* B = c->d_plane[P_B] + off;
    uint8_t* thR = c->d_plane[P_THR] + off;
    uint8_t* thB = c->d_plane[P_THB] + off;
    uint8_t* t0 = c->d_plane[P_T0] + off;
    uint8_t *t1 = nullptr, *t2 = nullptr, *t3 = nullptr;      // allocated by the paths that use them (ensure_plane)
    auto scratch = [&](int idx, uint8_t*& q) -> int {
        const int rc = ensure_plane(c, idx);
        if (!rc) q = c->d_plane[idx] + off;
        return rc;
    };
    uint8_t* mask = nullptr;
    if (u8_mask) { const int rc = scratch(P_MASK, mask); if (rc) return rc; }
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
    unsigned long long* tbits = c->d_bits_tmp ? c->d_bits_tmp + (size_t)first * c->bits_stride : nullptr;
    unsigned long long* ubits = c->d_bits_tmp2 ? c->d_bits_tmp2 + (size_t)first * c->bits_stride : nullptr;
    // One or two frames (process()): both planes' thresholds in ONE launch with the H and the V phases of a tile in workgroups of
    // their own (one frame: 81 tiles on 256 CUs) -- H verdicts of both planes into mbits, V verdicts into ebits; the open ORs them
    bool both_split = false;
    if (p->filter_type == 0) {
        if (c->brute_tophat) {   // debugging aid of the experiments build (LT_TOPHAT_BRUTE=1): direct footprint evaluation, still on the GPU
            { StageScope t(c, ST_ERODE_R, s);  launch_morph_ellipse(s, R, t0, nullptr, h, w, c->se29, false, ps, n); }
            { StageScope t(c, ST_TOPHAT_R, s); launch_morph_ellipse(s, t0, thR, R, h, w, c->se29, true, ps, n); }
            { StageScope t(c, ST_ERODE_B, s);  launch_morph_ellipse(s, B, t0, nullptr, h, w, c->se55, false, ps, n); }
            { StageScope t(c, ST_TOPHAT_B, s); launch_morph_ellipse(s, t0, thB, B, h, w, c->se55, true, ps, n); }
        } else if (n <= 2 && !c->stage_timing && !walk) {
            // One or two frames cannot fill the chip (a few hundred waves per top-hat kernel), and the two planes' top-hats do not
            // depend on each other: the 55x55 erode of the Lab-b plane and the 29x29 erode of the R plane are ONE launch, the two
            // top-hats the next (k_morph_one_pair).  (Round 5 ran the R plane's chain on a side stream: a fork, a join that cost the
            // frame 11-12 us of signalling, and three more launches.)  The eroded R plane has a scratch of its own, two planes per context.
            // (per STREAM, not per slot: a context of a thousand slots runs these one- and two-frame calls on a handful of streams --
            // the slices' streams and the urgent one -- and calls on one stream are ordered)
            if (!c->d_side_scratch) { const int rc = dev_alloc(&c->d_side_scratch, (size_t)SIDE_LANES * 2 * ps); if (rc) return rc; }
            int lane_of_stream = SIDE_LANES - 1;
            for (int i = 0; i < (int)c->streams.size() && i < SIDE_LANES - 1; ++i)
                if (c->streams[(size_t)i] == s) { lane_of_stream = i; break; }
            uint8_t* ts = c->d_side_scratch + (size_t)lane_of_stream * 2 * ps;
            if (launch_morph_one_pair(s, B, t0, nullptr, R, ts, nullptr, h, w, false, ps, n, 0, 0)) {
                if (!launch_morph_one_pair(s, t0, thB, B, ts, thR, R, h, w, true, ps, n, 0, 0)) {
                    launch_morph_runs(s, t0, thB, B, h, w, 55, true, ps, n);
                    launch_morph_runs(s, ts, thR, R, h, w, 29, true, ps, n);
                }
            } else {             // (a geometry the one-frame kernel does not take: an image width that is not a multiple of four)
                launch_morph_runs(s, R, t0, nullptr, h, w, 29, false, ps, n);
                launch_morph_runs(s, t0, thR, R, h, w, 29, true, ps, n);
                launch_morph_runs(s, B, t0, nullptr, h, w, 55, false, ps, n);
                launch_morph_runs(s, t0, thB, B, h, w, 55, true, ps, n);
            }
            both_split = !p->mask_noise;
        } else {
            { StageScope t(c, ST_ERODE_R, s);  launch_morph_runs(s, R, t0, nullptr, h, w, 29, false, ps, n); }
            { StageScope t(c, ST_TOPHAT_R, s); launch_morph_runs(s, t0, thRd, R, h, w, 29, true, ps, n, dpitch, c->th_pad_bytes); }
            { StageScope t(c, ST_ERODE_B, s);  launch_morph_runs(s, B, t0, nullptr, h, w, 55, false, ps, n); }
            { StageScope t(c, ST_TOPHAT_B, s); const int rc = tophat_b(s); if (rc) return rc; }
        }
    }
    bool merged_done = false, partials = false;   // partials: mbits, ebits, tmp, tmp2 still wait for their OR
    bool two_partials = false;                    // ... only mbits and ebits (the 'neighborhood' walk)
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
        if (!merged_done && both_split) {        // (refused when the packed arithmetic does not fit the parameters: one workgroup per tile below)
            merged_done = launch_bilateral_bits(s, thR, p->ksize_r, p->C_r, thB, p->ksize_b, p->C_b, B, p->ksize_noise, p->C_noise,
                                                p->noise_thresh, 0, mbits, h, w, ps, c->bits_stride, n, ebits) == 0;
            partials = two_partials = merged_done;
        }
        if (!merged_done)
          merged_done = launch_bilateral_bits(s, thR, p->ksize_r, p->C_r, thB, p->ksize_b, p->C_b, B, p->ksize_noise,
                                            p->C_noise, p->noise_thresh, p->mask_noise ? 1 : 0, mbits, h, w, ps,
                                            c->bits_stride, n) == 0;
        if (!merged_done) {              // tile + halo exceeds the LDS: one plane at a time
            { int rc = scratch(P_T1, t1); if (!rc) rc = scratch(P_T2, t2); if (rc) return rc; }
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
            { int rc = scratch(P_T1, t1); if (!rc) rc = scratch(P_T2, t2); if (rc) return rc; }
            launch_adaptive_mean(s, R, t1, h, w, p->ksize_r, p->C_r, ps, n);
            launch_adaptive_mean(s, B, t2, h, w, p->ksize_b, p->C_b, ps, n);
        }
        c->last_adaptive_path = two_partials ? 1 : 0;
    }
    if (!merged_done) {
        { const int rc = scratch(P_T3, t3); if (rc) return rc; }
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
      if (!u8_mask && n >= 16)
          opened = launch_merge_open5(s, mbits, partials ? ebits : nullptr, two_partials ? nullptr : tbits, two_partials ? nullptr : ubits, obits,
                                      h, w, c->bits_stride, n, nbits1, nbits2);
      // a few frames: the OR and the open in one launch of small workgroups (the one-frame chain is made of launch gaps: three
      // kernels of 5 us here; LT_OPEN_SMALL=0 restores them)
      if (!opened && !u8_mask && n <= 4)
          opened = launch_or_open5_small(s, mbits, partials ? ebits : nullptr, (!partials || two_partials) ? nullptr : tbits,
                                         (!partials || two_partials) ? nullptr : ubits, obits, h, w, c->bits_stride, n, nbits1, nbits2);
      if (!opened) {
          if (partials) launch_or4_bits(s, mbits, ebits, two_partials ? ebits : tbits, two_partials ? ebits : ubits, h, w, c->bits_stride, n, nbits1, nbits2);
          if (u8_mask) launch_open5_bits(s, mbits, ebits, mask, h, w, ps, c->bits_stride, n);
          else launch_open5_to_bits(s, mbits, ebits, obits, h, w, c->bits_stride, n);
      } }
    (void)t0;
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
static hipError_t make_compute_stream(hipStream_t* st, int reserved) {
    if (reserved > 0) {
        uint32_t mask[8];
        for (auto& w : mask) w = 0xffffffffu;
        for (int i = 0; i < reserved && i < 256; ++i) mask[i >> 5] &= ~(1u << (i & 31));
        return hipExtStreamCreateWithCUMask(st, 8, mask);
    }
    const char* e = LT_EXP_ENV("LT_STREAM_PRIORITY");
    int least = 0, greatest = 0;
    if ((e && strcmp(e, "normal") == 0) || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest)
        return hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    return hipStreamCreateWithPriority(st, hipStreamNonBlocking, greatest);
}

// Streams are never destroyed: a context takes them from a per-process pool (by device, kind and CU reservation) and gives them
// back, idle, when it goes.  hipStreamDestroy can hang on this runtime: destroy one stream and, within a few milliseconds,
// another one that was created with hipExtStreamCreateWithCUMask -- the second call waits in AMDKFD_IOC_WAIT_EVENTS for good
// (tools/microbench/close_hang.hip: 9 of 9 runs; with 50 ms between the last work and the destroys, or the masked stream
// destroyed first, 0 of 8).  That was the close() of NOTES C.8 (round 4's library: 8 of 10 runs of tools/close_hang.py).  No
// ordering inside lt_destroy is safe against a second tracker's streams, so none is destroyed; a stream that has been idle in
// the pool is as good as a new one, and lt_create saves 20 ms per priority stream it no longer creates.
namespace {
struct StreamPool {
    struct Key { int device, kind, param; bool operator<(const Key& o) const { return std::tie(device, kind, param) < std::tie(o.device, o.kind, o.param); } };
    std::mutex m;
    std::map<Key, std::vector<hipStream_t>> idle;
    std::map<hipStream_t, Key> out;
};
StreamPool& stream_pool() { static StreamPool* p = new StreamPool; return *p; }
}  // namespace

hipError_t stream_get(hipStream_t* st, StreamKind kind, int param) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    StreamPool& sp = stream_pool();
    const StreamPool::Key key{dev, (int)kind, param};
    {
        std::lock_guard<std::mutex> g(sp.m);
        auto& v = sp.idle[key];
        if (!v.empty()) {
            *st = v.back();
            v.pop_back();
            sp.out[*st] = key;
            return hipSuccess;
        }
    }
    hipError_t e = hipSuccess;
    if (kind == SK_COMPUTE) e = make_compute_stream(st, param);
    else if (kind == SK_PLAIN) e = hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    else if (kind == SK_CU_SET) {        // the first `param` CUs and nothing else (the chained search), or -- param < 0 -- CUs 1 .. -param - 1 (the copy kernel)
        uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (param >= 0) for (int i = 0; i < param && i < 256; ++i) mask[i >> 5] |= 1u << (i & 31);
        else for (int b = 1; b < -param && b < 256; ++b) mask[b >> 5] |= 1u << (b & 31);
        e = hipExtStreamCreateWithCUMask(st, 8, mask);
    } else {                             // SK_PRIORITY: the highest priority level
        int lo = 0, hi = 0;
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        e = hipStreamCreateWithPriority(st, hipStreamNonBlocking, hi);
    }
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> g(sp.m);
        sp.out[*st] = key;
    }
    return e;
}

void stream_put(hipStream_t st) {
    if (!st) return;
    (void)hipStreamSynchronize(st);
    StreamPool& sp = stream_pool();
    std::lock_guard<std::mutex> g(sp.m);
    auto it = sp.out.find(st);
    if (it == sp.out.end()) return;      // not ours: left alone
    sp.idle[it->second].push_back(st);
    sp.out.erase(it);
}

hipError_t create_compute_stream(hipStream_t* st, int reserved) { return stream_get(st, SK_COMPUTE, reserved); }

}  // namespace lt

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

int lt_create(const lt_calib* calib, int device, lt_ctx** out) {
    if (!calib || !out) return fail(LT_ERR_INVALID, "null argument");
    if (calib->img_w < 2 || calib->img_h < 2 || calib->warp_w < 2 || calib->warp_h < 2 || calib->img_w > 16384 ||
        calib->img_h > 16384 || calib->warp_w > 4096 || calib->warp_h > 16384)
        return fail(LT_ERR_INVALID, "camera size must be in [2, 16384], bird's-eye width in [2, 4096], height in [2, 16384]");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(LT_ERR_HIP, "no HIP device visible: the lane-tracker kernels need a GPU (gfx950)");
    if (device < 0 || device >= ndev) return fail(LT_ERR_INVALID, "device %d out of range (%d visible)", device, ndev);
    TraceScope ts_all("lt_create");
    lt_ctx* c = new lt_ctx();
    c->calib = *calib;
    c->device = device;
    auto bail = [&](int rc) {
        lt_destroy(c);
        return rc;
    };
    if (const char* e = LT_EXP_ENV("LT_WALK_MIN_FRAMES")) c->walk_min_pixels = atoll(e) * calib->warp_w * calib->warp_h;   // (experiments build; the API: lt_set_walk_min_frames)
    if (hipSetDevice(device) != hipSuccess) return bail(fail(LT_ERR_HIP, "hipSetDevice(%d) failed", device));
    if (hipGetDeviceProperties(&c->prop, device) != hipSuccess) return bail(fail(LT_ERR_HIP, "hipGetDeviceProperties failed"));
    if (create_compute_stream(&c->stream) != hipSuccess) return bail(fail(LT_ERR_HIP, "hipStreamCreate failed"));
    c->streams.assign(1, c->stream);
    c->nstreams = 1;
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) return bail(fail(LT_ERR_HIP, "hipEventCreate failed"));
    if (stream_get(&c->copy, SK_PLAIN, 0) != hipSuccess) return bail(fail(LT_ERR_HIP, "copy stream creation failed"));

    // host tables
    const double t_tables = trace_on() ? trace_now() : 0.0;
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
        const char* e = LT_EXP_ENV("LT_TOPHAT_BRUTE");
        c->brute_tophat = e && e[0] == '1';
    }

    if (trace_on()) trace_line("lt_create:host_tables", t_tables);
    TraceScope ts_up("lt_create:table_upload");
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

    preload_kernels(device, c->stream);
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
        const char* names[5] = {"copy", "search", "present", "urgent", "dl"};
        hipStream_t sts[5] = {c->copy, c->search, c->present, c->urgent, c->dl};
        for (int i = 0; i < 5; ++i) if (sts[i]) { note(names[i]); (void)hipStreamSynchronize(sts[i]); }
    }
    // Streams, events and page-locked buffers go FIRST, device memory after them.  Round 4 released the memory first, and when the
    // cache then handed blocks back to the driver (hipFree of several GB), the hipStreamDestroy of the presentation stream -- a
    // stream with a CU mask -- that followed did not return: the thread sat in AMDKFD_IOC_WAIT_EVENTS for good (5 of 6 runs of
    // tools/close_hang.py with round 4's library; never once the stream is destroyed before the hipFree; NOTES D.5).
    // (LT_TRACE_DESTROY names every class of call.)
    note("events: timing pool, order ring, staging");
    for (auto e : c->ev_pool) (void)hipEventDestroy(e);
    for (auto e : c->order_events) (void)hipEventDestroy(e);
    if (c->spans_busy.done) (void)hipEventDestroy(c->spans_busy.done);
    if (c->text_busy.done) (void)hipEventDestroy(c->text_busy.done);
    if (c->annot_busy.done) (void)hipEventDestroy(c->annot_busy.done);
    note("streams back to the pool: dl");
    stream_put(c->dl);
    note("events: download timing");
    for (auto& d : c->dl_inflight) { (void)hipEventDestroy(d.a); (void)hipEventDestroy(d.b); }
    for (auto e : c->dl_event_pool) (void)hipEventDestroy(e);
    note("streams back to the pool: present, urgent");
    stream_put(c->present);
    stream_put(c->urgent);
    if (c->rest_done) (void)hipEventDestroy(c->rest_done);
    note("hipHostFree(spans, lines, xpos)");
    if (c->h_spans) (void)hipHostFree(c->h_spans);
    if (c->h_lines) (void)hipHostFree(c->h_lines);
    if (c->h_xpos) (void)hipHostFree(c->h_xpos);
    note("events: slot-range rings, chain tickets");
    for (auto& w : c->readers.e) if (w.ev) (void)hipEventDestroy(w.ev);
    for (auto& w : c->writers.e) if (w.ev) (void)hipEventDestroy(w.ev);
    for (auto& w : c->rests.e) if (w.ev) (void)hipEventDestroy(w.ev);
    for (auto& t : c->chains) (void)hipEventDestroy(t.done);
    for (auto e : c->chain_event_pool) (void)hipEventDestroy(e);
    note("hipHostFree(small, rec, rec_stage, cancel)");
    if (c->h_small) (void)hipHostFree(c->h_small);
    if (c->h_lists) (void)hipHostFree(c->h_lists);
    if (c->h_rec) (void)hipHostFree(c->h_rec);
    if (c->h_rec_stage) (void)hipHostFree(c->h_rec_stage);
    if (c->h_cancel) (void)hipHostFree(c->h_cancel);
    note("streams back to the pool: search, copy");
    stream_put(c->search);
    stream_put(c->copy);
    note("events: timer");
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    note("streams back to the pool: slot streams");
    for (auto st : c->streams) stream_put(st);
    if (c->streams.empty()) stream_put(c->stream);
    note("slots");
    FreeScope frees(c->device);          // one wait for the device instead of one per block; the blocks enter the cache together
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
    note("done");
    delete c;
}

int lt_reserve(lt_ctx* c, int capacity) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    if (capacity < 1) return fail(LT_ERR_INVALID, "capacity must be >= 1");
    int rc = set_device(c);
    if (rc) return rc;
    if (capacity <= c->capacity) return LT_OK;
    TraceScope ts_all("lt_reserve", (size_t)capacity);
    if ((rc = sync_all(c))) return rc;
    // the old blocks are parked until the new ones are allocated and enter the device cache behind them (FreeScope, lt_memory.cpp):
    // the cache must not evict, to make room for the small blocks, the large ones this call is about to ask for
    FreeScope frees(c->device);
    {
        TraceScope ts_("lt_reserve:free_slots", (size_t)c->capacity);
        free_slots(c);
    }
    c->rec_mirror_slot = -1;
    c->capacity = capacity;
    const size_t n = (size_t)capacity;
    if ((rc = dev_alloc(&c->d_frames, n * c->frame_bytes + 16))) { free_slots(c); return rc; }   // +16: k_undistort_rows reads 8-byte windows
    c->direct_upload = -1;               // a new frame buffer: whether the host can write it is found out by the first small upload
    if ((rc = dev_alloc(&c->d_und, (size_t)((n + 1) / 2) * 2 * c->und_px))) { free_slots(c); return rc; }
    for (int i : {(int)P_R, (int)P_B, (int)P_THR, (int)P_THB, (int)P_T0})     // the others when a path that uses them runs (ensure_plane)
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
    c->frame_full.assign(n, 0);
    c->annot_full.assign(n, 0);
    c->front_ok.assign(n, 0);
    if ((rc = dev_alloc(&c->d_rec, n))) { free_slots(c); return rc; }
    if ((rc = dev_alloc(&c->d_prev, n * 6))) { free_slots(c); return rc; }
    c->capacity = capacity;
    TraceScope ts_ms("lt_reserve:memset");
    HIP_TRY(hipMemsetAsync(c->d_rec, 0, n * sizeof(lt_lane_record), c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LT_OK;
}

// Everything a stream's first window (or a video's first frame) would otherwise set up on the way -- streams, page-locked
// staging, search buffers sized for these parameters, the presentation stage's buffers -- now, for the current capacity
// (lt_reserve first).  Nothing changes in what later calls compute; they find their buffers in place.
int lt_warm(lt_ctx* c, const lt_search_params* sws, const lt_search_params* band, int annotate) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    if (c->capacity < 1) return fail(LT_ERR_STATE, "lt_warm before lt_reserve");
    int rc = set_device(c);
    if (rc) return rc;
    TraceScope ts_all("lt_warm", (size_t)c->capacity);
    if ((rc = ensure_chain_buffers(c))) return rc;
    if (!c->urgent && create_compute_stream(&c->urgent, c->search_cus) != hipSuccess) return fail(LT_ERR_HIP, "hipStreamCreate failed");
    constexpr size_t SMALL = 256 << 10;
    if (!c->h_small && hipHostMalloc(reinterpret_cast<void**>(&c->h_small), SMALL, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        c->h_small = nullptr;
    }
    int maxpix = 0, maxlev = 0;
    if (sws) {
        SearchGeom g;
        if ((rc = make_search_geom(c, sws, false, g))) return rc;
        maxpix = std::max(maxpix, g.maxpix);
        maxlev = std::max(maxlev, g.maxlev);
        if ((rc = ensure_band_sums(c, g.nbands))) return rc;
    }
    if (band) {
        SearchGeom g;
        if ((rc = make_search_geom(c, band, true, g))) return rc;
        maxpix = std::max(maxpix, g.maxpix);
        maxlev = std::max(maxlev, 1);
    }
    if (maxpix && (rc = ensure_search_buffers(c, maxpix, maxlev))) return rc;
    if (annotate && (rc = warm_presentation(c, annotate == 2))) return rc;
    return sync_all(c);
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
    mark_frames(c, first, n, 1);
    front_stale(c, first, n);
    return LT_OK;
}

int lt_get_source_rows(lt_ctx* c, int* row0, int* row1) {
    if (!c || !row0 || !row1) return fail(LT_ERR_INVALID, "null argument");
    *row0 = c->cam_r0;
    *row1 = c->cam_r1;
    return LT_OK;
}

static int wait_reader_tails(lt_ctx* c, hipStream_t waiter);

// ---- one frame's rows through the PCIe aperture ----------------------------------------------------------------------------------
// With a large BAR the whole device memory is mapped into the process, write-combining, at the addresses hipMalloc hands out: the
// calling thread can store a frame's rows into the slot itself.  For the ONE frame of a LaneTracker.process() call that beats the
// copy engine: hipMemcpy2DAsync costs the call 8-24 us (by box) before the engine even starts (23 us of copy + 6 us until the
// first kernel behind it), the stores take the bus's 20 us for the 914 KB of a 1280x720 frame's rows and the undistortion can be
// launched the moment they are out (tools/microbench/upload_latency.hip: rows out of cold memory + a dependent kernel 66 -> 43 us;
// NOTES_r06 E.6 for the frame).  The bytes are the same bytes; only their way differs.
// Ordering: the stores end with an sfence and are posted writes of the thread that then rings the launch's doorbell (or of
// threads it has waited for); the kernels behind them start with an acquire that drops what the XCDs' L2s still hold of the
// slot's previous frame (as between any two kernels), and the memory-side cache sees the bus's writes.  In front: nothing is
// waited for -- the aperture is taken only when the library already KNOWS the slot's readers are done (camera_rows_known_idle).
// Up to 1.5 MB per call: the 914 KB of a 1280x720 frame's rows take the calling thread (and two polling copy threads) 25 us against
// the engine's 8 + 29 us; the 2.0 MB of a 1920x1080 frame's take them 52-60 us against the engine's 9 + 51 us, most of which the
// engine spends beside the mask chain's launches -- no gain there, and a busy host (tools/process_points.py, NOTES_r06 E.6).
constexpr size_t APERTURE_MAX_BYTES = (size_t)3 << 19;
static bool host_has_mapped(const void* p, size_t n) {
    const uintptr_t pg = (uintptr_t)sysconf(_SC_PAGESIZE);
    unsigned char v = 0;
    auto mapped = [&](uintptr_t a) { return mincore((void*)(a & ~(pg - 1)), 1, &v) == 0; };
    return n > 0 && mapped((uintptr_t)p) && mapped((uintptr_t)p + n - 1);
}
static bool direct_upload_possible(lt_ctx* c) {
    if (c->direct_upload < 0)
        c->direct_upload = c->prop.isLargeBar && c->d_frames && host_has_mapped(c->d_frames, (size_t)c->capacity * c->frame_bytes) ? 1 : 0;
    return c->direct_upload == 1 && c->direct_upload_wanted;
}
__attribute__((target("avx2"))) static void store_stream_avx2(uint8_t* dst, const uint8_t* src, size_t n) {
    size_t head = (32 - ((uintptr_t)dst & 31)) & 31;
    if (head > n) head = n;
    if (head) { std::memcpy(dst, src, head); dst += head; src += head; n -= head; }
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 32));
        const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 64));
        const __m256i e = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i), a);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 64), d);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 96), e);
    }
    if (i < n) std::memcpy(dst + i, src + i, n - i);
}
static void store_piece(uint8_t* dst, const uint8_t* src, size_t n) {
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) store_stream_avx2(dst, src, n);
    else std::memcpy(dst, src, n);
    _mm_sfence();
}
// One thread moves a frame's rows at what it can READ from memory the caller's frame lies cold in (30 us for 914 KB); two or three
// reach the bus's 20 us (tools/microbench/upload_latency.hip, COLD=1).  The copy threads that are polling for work at this moment
// (LaneTracker.process() keeps two or three of them warm with the rows of its output frames) are offered a piece each; pieces
// nobody has claimed when the calling thread is through with its own are the calling thread's again -- no piece waits for a
// sleeper, and none for a queue another tracker has filled.
static void store_through_aperture(uint8_t* dst, const uint8_t* src, size_t n) {
    const int helpers = n >= ((size_t)256 << 10) ? std::min(host_copy_pollers(), 2) : 0;
    if (helpers <= 0) { store_piece(dst, src, n); return; }
    struct Split {
        std::atomic<int> claimed[3];
        std::atomic<int> done{0};
        uint8_t* dst; const uint8_t* src; size_t lo[4];
        void run(int k) { store_piece(dst + lo[k], src + lo[k], lo[k + 1] - lo[k]); done.fetch_add(1, std::memory_order_release); }
    };
    auto sp = std::make_shared<Split>();
    const int pieces = helpers + 1;
    sp->dst = dst; sp->src = src;
    for (int k = 0; k <= pieces; ++k) sp->lo[k] = k == pieces ? n : (n * (size_t)k / (size_t)pieces) & ~(size_t)127;
    for (int k = 0; k < 3; ++k) sp->claimed[k].store(0, std::memory_order_relaxed);
    sp->claimed[0].store(1, std::memory_order_relaxed);
    for (int k = 1; k < pieces; ++k)
        if (host_submit_fn(0, [sp, k] { if (!sp->claimed[k].exchange(1, std::memory_order_acq_rel)) sp->run(k); }, false) != 0) break;   // (not submitted: claimed below)
    sp->run(0);
    for (int k = 1; k < pieces; ++k)
        if (!sp->claimed[k].exchange(1, std::memory_order_acq_rel)) sp->run(k);
    // (a helper that has claimed a piece finishes it in microseconds -- unless the scheduler has taken its core: then give ours up too)
    for (unsigned spins = 0; sp->done.load(std::memory_order_acquire) < pieces; ++spins) {
        if ((spins & 0xffffu) == 0xffffu) std::this_thread::yield(); else __builtin_ia32_pause();
    }
}
// Does the host KNOW that no kernel reads the camera rows of slots [first, first + n) any more?  Asked without a call that could
// cost the frame anything: hipStreamQuery on a stream whose last kernel carries no signal makes the runtime enqueue a marker and
// wait for it (measured: process() 172 -> 198 us per frame with two such queries in front of the stores) -- so the answer comes
// from what the library has seen itself: the recorded readers' events (a query of an existing signal), and for the unrecorded
// ones of a one-frame context the completion word the host polled at the end of the previous frame (RangeEvents::lazy_seen).
// "Not known" is not "busy": the caller then takes the engine's stream-ordered copy, which needs no answer.
static bool camera_rows_known_idle(lt_ctx* c, int first, int n) {
    const lt_ctx::RangeEvents& r = c->readers;
    if (r.overflow || r.lazy_seen != r.lazy_seq) return false;
    for (unsigned i = 0; i < r.count; ++i) {
        const lt_ctx::RangeEvents::Entry& w = r.e[(r.head + i) % (unsigned)r.e.size()];
        if (w.lo < first + n && w.hi > first && hipEventQuery(w.ev) != hipSuccess) { (void)hipGetLastError(); return false; }
    }
    return true;
}

int lt_set_direct_upload(lt_ctx* c, int on) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    if (on >= 0) c->direct_upload_wanted = on != 0;
    return direct_upload_possible(c) ? 1 : 0;
}
unsigned long long lt_direct_upload_count(lt_ctx* c) { return c ? c->direct_uploads : 0; }

static int upload_frame_rows_impl(lt_ctx* c, const uint8_t* frames, int first, int n, bool enqueue) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!frames) return fail(LT_ERR_INVALID, "null frames");
    if (n == 0 || c->cam_r1 <= c->cam_r0) return LT_OK;
    if ((rc = set_device(c))) return rc;
    // The enqueued form waits for nothing on the host: the copy goes onto the slots' own streams (behind everything launched over
    // these slots there) and, like lt_upload_frame_rows_async's, behind the kernels of OTHER streams that still read the slots'
    // camera rows -- overlays on the presentation stream (slot-range events).  LT_UPLOAD_SYNC=1: wait for the whole context first, as
    // lt_upload_frame_rows does (A/B; 5-10 us of a process() frame, on its critical path).
    static const bool enqueue_syncs = [] { const char* e = LT_EXP_ENV("LT_UPLOAD_SYNC"); return e && e[0] == '1'; }();
    if ((!enqueue || enqueue_syncs) && (rc = sync_all(c))) return rc;
    mark_frames(c, first, n, 0);         // a new frame's rows: the others are the previous occupant's until lt_upload_frame_rest
    front_stale(c, first, n);
    const size_t row_bytes = (size_t)c->calib.img_w * 3, off = (size_t)c->cam_r0 * row_bytes;
    const size_t bytes = (size_t)(c->cam_r1 - c->cam_r0) * row_bytes;
    // (one frame, measured in round 5 against this pitched copy, 279-286 us per frame of process(): a plain copy of the contiguous
    // run 315-320; the rows onto page-locked staging by the copy threads and an asynchronous engine copy from there 295-329, a copy
    // kernel from there 320 -- the runtime's pageable path is the fastest of the four)
    // enqueue (lt_upload_frame_rows_enqueue): the copy on the slots' own streams, ahead of the kernels lt_mask_run puts there, and no
    // wait.  From the caller's pageable frame the call returns once the runtime has the bytes on their way (22-27 us for one
    // 1280x720 frame's rows against 49-54 with the wait: the engine's 18 us run under the mask chain's launches).
    if (enqueue && !enqueue_syncs && (size_t)n * bytes <= APERTURE_MAX_BYTES && direct_upload_possible(c) && camera_rows_known_idle(c, first, n)) {
        // a frame or two, and nothing left on the device that reads these slots' rows: by this thread's own stores (above); the
        // caller's array is free again when the call returns
        for (int k = 0; k < n; ++k)
            store_through_aperture(c->d_frames + (size_t)(first + k) * c->frame_bytes + off, frames + (size_t)k * c->frame_bytes + off, bytes);
        ++c->direct_uploads;
        return LT_OK;
    }
    if (enqueue)
        return for_each_slice(c, first, n, [&](hipStream_t st, int f0, int m) {
            if (!enqueue_syncs) {
                bool precise = true;
                int wrc = wait_range(c->readers, st, f0, f0 + m, &precise);
                if (wrc) return wrc;
                if (!precise && (wrc = wait_reader_tails(c, st))) return wrc;
            }
            // (the rows in two to four pieces, so that the engine's copy of one runs under the runtime's staging of the next: measured
            // SLOWER, 195-218 against 183-186 us per 1280x720 frame -- every piece pays the call again; NOTES_r06 E.2)
            HIP_TRY(hipMemcpy2DAsync(c->d_frames + (size_t)f0 * c->frame_bytes + off, c->frame_bytes, frames + off, c->frame_bytes,
                                     bytes, (size_t)m, hipMemcpyHostToDevice, st));
            // (where a later small call could take the aperture, this copy counts as work on the slots' rows that the host has not
            // seen finished: stores from the host must not be overtaken by it)
            if (c->direct_upload == 1 && (size_t)m * bytes <= APERTURE_MAX_BYTES) return note_range_frame(c, c->readers, st, f0, f0 + m);
            return (int)LT_OK;
        });
    HIP_TRY(hipMemcpy2DAsync(c->d_frames + (size_t)first * c->frame_bytes + off, c->frame_bytes, frames + off, c->frame_bytes,
                             bytes, (size_t)n, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return LT_OK;
}

int lt_upload_frame_rows(lt_ctx* c, const uint8_t* frames, int first, int n) { return upload_frame_rows_impl(c, frames, first, n, false); }
int lt_upload_frame_rows_enqueue(lt_ctx* c, const uint8_t* frames, int first, int n) { return upload_frame_rows_impl(c, frames, first, n, true); }

// Fallback of the stream-ordered uploads when the ring of readers has overflowed: `waiter` waits for the tail of every stream
// a kernel that reads camera frames can be on -- the slots' compute streams (undistortion), the presentation stream (overlays)
// and the urgent stream.
static int wait_reader_tails(lt_ctx* c, hipStream_t waiter) {
    const bool slices_only = !c->readers.overflow;     // (only `lazy`: the unrecorded readers are on the slots' streams)
    auto tail = [&](hipStream_t st) {
        if (!st || st == waiter) return (int)LT_OK;
        hipEvent_t e = next_order_event(c);
        if (!e) return fail(LT_ERR_HIP, "hipEventCreate failed");
        HIP_TRY(hipEventRecord(e, st));
        HIP_TRY(hipStreamWaitEvent(waiter, e, 0));
        return (int)LT_OK;
    };
    for (int i = 0; i < c->nstreams && i < (int)c->streams.size(); ++i) { const int rc = tail(c->streams[i]); if (rc) return rc; }
    if (slices_only) return LT_OK;
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
    mark_frames(c, first, n, 0);
    front_stale(c, first, n);
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
        mark_frames(c, first, n, 1);
        return rest_mark(c, first, n);
    }
    if (head)
        HIP_TRY(hipMemcpy2DAsync(dst, c->frame_bytes, frames, c->frame_bytes, head, (size_t)n, hipMemcpyHostToDevice, c->copy));
    if (tail0 < c->frame_bytes)
        HIP_TRY(hipMemcpy2DAsync(dst + tail0, c->frame_bytes, frames + tail0, c->frame_bytes, c->frame_bytes - tail0, (size_t)n,
                                 hipMemcpyHostToDevice, c->copy));
    mark_frames(c, first, n, 1);
    return rest_mark(c, first, n);
}

int lt_upload_masks(lt_ctx* c, const uint8_t* masks, int first, int n) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (!masks) return fail(LT_ERR_INVALID, "null masks");
    if ((rc = set_device(c))) return rc;
    if ((rc = sync_all(c))) return rc;
    if ((rc = ensure_plane(c, P_MASK))) return rc;
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

}  // extern "C"
namespace lt {
int download(lt_ctx* c, const void* src, void* dst, size_t bytes) {
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
}  // namespace lt
extern "C" {

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
        if ((rc = ensure_plane(c, P_MERGED))) return rc;
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

// A search over ONE slot (process(): one frame per call, the host waiting for its record) sends the record to page-locked
// memory by a launch queued right behind the search kernel, while the device is still busy with the frame: lt_download_records
// then waits for that stream and reads 64 bytes, instead of launching the copy once the search is over (8 us of 0.4 ms).
static lt_lane_record* rec_mirror_device(lt_ctx* c) {      // the mirror as kernels address it (nullptr: there is none)
    if (!c->h_rec && hipHostMalloc(reinterpret_cast<void**>(&c->h_rec), 256, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        c->h_rec = nullptr;
    } else if (c->h_rec && !c->rec_ticket_counter) {
        std::memset(c->h_rec, 0, 256);       // (the ticket word behind the record: no stale match)
        c->rec_ticket_counter = 1;
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
    c->rec_ticket = 0;
    static_assert(sizeof(lt_lane_record) == 64, "k_mirror_record copies 16 words and stores the ticket behind them");
    unsigned ticket = ++c->rec_ticket_counter;
    if (!ticket) ticket = ++c->rec_ticket_counter;          // 0 means "no ticket"
    if (rec_mirror_device(c) && launch_mirror_record(st, c->h_rec, c->d_rec + slot, ticket)) {
        c->rec_mirror_slot = slot;
        c->rec_mirror_stream = st;
        c->rec_ticket = ticket;
    }
}

int lt_download_records(lt_ctx* c, int first, int n, lt_lane_record* out) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if (n == 1 && out && c->rec_mirror_slot == first && c->h_rec) {
        if ((rc = set_device(c))) return rc;
        // The search kernel stores its ticket behind the record: poll for it (hipStreamSynchronize returns 15-20 us after the
        // kernel's last store, the word is there within 2: tools/microbench/sync_latency.hip).  LT_RECORD_POLL=0: wait for the
        // stream (A/B); after 2 ms without the ticket likewise (an error would show there).
        static const bool poll = [] { const char* e = LT_EXP_ENV("LT_RECORD_POLL"); return !(e && e[0] == '0'); }();
        bool seen = false;
        if (poll && c->rec_ticket) {
            const volatile unsigned* word = reinterpret_cast<const volatile unsigned*>(c->h_rec + 1);
            const auto t0 = std::chrono::steady_clock::now();
            for (unsigned spins = 0; !(seen = *word == c->rec_ticket); ++spins) {
                __builtin_ia32_pause();
                if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
            }
            std::atomic_thread_fence(std::memory_order_acquire);
        }
        if (!seen) HIP_TRY(hipStreamSynchronize(c->rec_mirror_stream));
        std::memcpy(out, c->h_rec, sizeof(lt_lane_record));
        return LT_OK;
    }
    return download(c, c->d_rec + first, out, (size_t)n * sizeof(lt_lane_record));
}

// The lane pixels of one side of a slot, expanded from the form the search kernels leave on the device (`region`: a host copy of the
// slot's list region, at least the words the record's format uses) into the reference's (y, x) lists, in the reference's order.
static int expand_pixels(const lt_ctx* c, const lt_lane_record& r, const uint32_t* region, size_t region_words, int side, int32_t* ys,
                         int32_t* xs, int cap, int* count) {
    if (r._pad == 1) {
        // k_sws_fit2 leaves one column mask per window row (lt_internal.h: sws2_mask_offset); the lists
        // self.left_y / left_x (level-major, row-major inside a window, ascending x) are expanded here
        const int nlev = (int)region[0], wh = (int)region[1], H1 = (int)region[2];
        const long long words = sws2_block_words(nlev, wh);
        if (nlev < 1 || wh < 1 || words > (long long)region_words) return fail(LT_ERR_STATE, "corrupt lane-pixel block");
        const int32_t* roi = reinterpret_cast<const int32_t*>(region + 4);
        const uint32_t* masks = region + sws2_mask_offset(nlev);
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
        const int nrows = (int)region[0], top = (int)region[1];
        const long long words = band2_block_words(nrows);
        if (nrows < 0 || words > (long long)region_words) return fail(LT_ERR_STATE, "corrupt lane-pixel block");
        const int32_t* row_a = reinterpret_cast<const int32_t*>(region + 4) + (size_t)side * nrows;
        const uint32_t* masks = region + band2_mask_offset(nrows) + (size_t)side * nrows * 2;
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
    // first-version kernels: packed (y << 16 | x) lists, side by side
    int n = side == 0 ? r.n_left : r.n_right;
    if (n > c->maxpix) n = c->maxpix;
    *count = n;
    if (n > cap) n = cap;
    if (n <= 0) return LT_OK;
    if (!ys || !xs) return fail(LT_ERR_INVALID, "null pixel buffers");
    const uint32_t* tmp = region + (size_t)side * c->maxpix;
    for (int i = 0; i < n; ++i) {
        ys[i] = (int32_t)(tmp[i] >> 16);
        xs[i] = (int32_t)(tmp[i] & 0xffffu);
    }
    return LT_OK;
}

int lt_download_pixels(lt_ctx* c, int slot, int side, int32_t* ys, int32_t* xs, int cap, int* count) {
    int rc = check_slots(c, slot, 1);
    if (rc) return rc;
    if (side < 0 || side > 1 || !count || cap < 0) return fail(LT_ERR_INVALID, "bad side/count/cap");
    if (!c->d_pix) return fail(LT_ERR_STATE, "no search has run yet");
    lt_lane_record r;
    if ((rc = download(c, c->d_rec + slot, &r, sizeof r))) return rc;
    const uint32_t* block = c->d_pix + (size_t)slot * 2 * c->maxpix;
    size_t words = 0;
    if (r._pad == 1 || r._pad == 2) {
        uint32_t hdr[4];
        if ((rc = download(c, block, hdr, sizeof hdr))) return rc;
        const long long w = r._pad == 1 ? sws2_block_words((int)hdr[0], (int)hdr[1]) : band2_block_words((int)hdr[0]);
        if (w < 4 || w > 2LL * c->maxpix) return fail(LT_ERR_STATE, "corrupt lane-pixel block");
        words = (size_t)w;
    } else {
        words = (size_t)c->maxpix + (size_t)std::min(std::max((int)r.n_right, 0), c->maxpix);     // both sides' lists, the right one as far as it goes
        if (side == 0) words = (size_t)std::min(std::max((int)r.n_left, 0), c->maxpix);
    }
    std::vector<uint32_t> blk(std::max<size_t>(words, 4));
    if (words && (rc = download(c, block, blk.data(), words * 4))) return rc;
    return expand_pixels(c, r, blk.data(), blk.size(), side, ys, xs, cap, count);
}

// Both sides' lane pixels and (want_centroids) both window-centroid lists of a slot in ONE round trip to the device: the record, the
// slot's list region and its centroid lists are copied into page-locked memory by three launches behind each other, the host waits
// once and expands.  LaneTracker fetches the lists of a search lazily -- when somebody reads lt.left_x, or when the NEXT search over
// the same slot is about to overwrite them (the second try of a frame, lane_tracker.py:1101: the first try's lists stay the
// tracker's if the second finds nothing) -- and did so through lt_download_pixels / lt_download_centroids: fourteen round trips,
// 300 us of a two-try frame (profiles/r06_config1_timeline.txt).  counts[2] / cent_counts[2] return the full lengths; lists longer
// than cap / cent_cap are cut (call again with larger buffers).  LT_ERR_CAPACITY: the slot's region does not fit the staging
// buffer (huge search windows) -- use lt_download_pixels / lt_download_centroids.
int lt_download_lane_lists(lt_ctx* c, int slot, int32_t* ly, int32_t* lx, int32_t* ry, int32_t* rx, int cap, int* counts, int want_centroids,
                           int32_t* cent_l, int32_t* cent_r, int cent_cap, int* cent_counts) {
    int rc = check_slots(c, slot, 1);
    if (rc) return rc;
    if (!counts || cap < 0 || (want_centroids && (!cent_counts || cent_cap < 0))) return fail(LT_ERR_INVALID, "lt_download_lane_lists: bad arguments");
    if (!c->d_pix) return fail(LT_ERR_STATE, "no search has run yet");
    if (want_centroids && !c->d_cent) return fail(LT_ERR_STATE, "no sliding-window search has run yet");
    if ((rc = set_device(c))) return rc;
    const size_t region_words = 2 * (size_t)c->maxpix, cent_words = want_centroids ? 2 * ((size_t)c->maxlev + 2) : 0;
    const size_t bytes = sizeof(lt_lane_record) + (region_words + cent_words) * 4;
    if (bytes > ((size_t)2 << 20)) return fail(LT_ERR_CAPACITY, "lt_download_lane_lists: the slot's list region (%zu bytes) exceeds the staging buffer", bytes);
    if (c->h_lists_bytes < bytes) {
        if (c->h_lists) (void)hipHostFree(c->h_lists);
        c->h_lists = nullptr;
        c->h_lists_bytes = 0;
        if (hipHostMalloc(reinterpret_cast<void**>(&c->h_lists), bytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            c->h_lists = nullptr;
            return fail(LT_ERR_NOMEM, "hipHostMalloc(%zu) failed", bytes);
        }
        c->h_lists_bytes = bytes;
    }
    hipStream_t st = c->stream;
    if (c->urgent_on && c->urgent) st = c->urgent;      // as in download(): what is asked for was produced on the urgent stream (or is complete)
    else if ((rc = sync_all(c))) return rc;
    uint8_t* h = c->h_lists;
    bool ok = launch_copy_words_to_pinned(st, h, c->d_rec + slot, sizeof(lt_lane_record)) &&
              launch_copy_words_to_pinned(st, h + sizeof(lt_lane_record), c->d_pix + (size_t)slot * region_words, region_words * 4);
    if (ok && cent_words) ok = launch_copy_words_to_pinned(st, h + sizeof(lt_lane_record) + region_words * 4, c->d_cent + (size_t)slot * cent_words, cent_words * 4);
    if (!ok) return fail(LT_ERR_HIP, "lt_download_lane_lists: the copy launches were refused");
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(st));
    lt_lane_record r;
    std::memcpy(&r, h, sizeof r);
    const uint32_t* region = reinterpret_cast<const uint32_t*>(h + sizeof(lt_lane_record));
    if ((rc = expand_pixels(c, r, region, region_words, 0, ly, lx, cap, &counts[0]))) return rc;
    if ((rc = expand_pixels(c, r, region, region_words, 1, ry, rx, cap, &counts[1]))) return rc;
    if (want_centroids) {
        const int32_t* cw = reinterpret_cast<const int32_t*>(region + region_words);
        for (int side = 0; side < 2; ++side) {
            const int32_t* t = cw + (size_t)side * (c->maxlev + 2);
            int n = t[0];
            if (n < 0) n = 0;
            if (n > c->maxlev + 1) n = c->maxlev + 1;
            cent_counts[side] = n;
            int32_t* out = side == 0 ? cent_l : cent_r;
            for (int i = 0; i < n && i < cent_cap && out; ++i) out[i] = t[1 + i];
        }
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
static int mask_run_impl(lt_ctx* c, int first, int n, const lt_filter_params* p, bool reuse_front) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if ((rc = validate_filter(p))) return rc;
    if ((rc = set_device(c))) return rc;
    if (n == 0) return LT_OK;
    const size_t ps = c->plane_bytes;
    rc = for_each_slice(c, first, n, [&](hipStream_t st, int f0, int m) {
        // the front end -- unless the caller asked for a RE-run (lt_mask_rerun: the second try of a frame, other filter parameters
        // over the same bird's-eye planes) and these slots' planes are those of the frames they hold
        bool have_front = reuse_front && !c->stage_timing && f0 + m <= (int)c->front_ok.size();
        for (int i = f0; have_front && i < f0 + m; ++i) have_front = c->front_ok[(size_t)i] != 0;
        if (!have_front) {
            { StageScope t(c, ST_UNDISTORT, st);
              launch_undistort_rows(st, c->d_frames + (size_t)f0 * c->frame_bytes, c->frame_bytes, c->d_uxy, c->d_ufrac,
                                    c->fe, c->d_und, c->und_px, f0, m); }
            { int mrc = n == 1 ? note_range_frame(c, c->readers, st, f0, f0 + m) : note_range(c->readers, st, f0, f0 + m); if (mrc) return mrc; }
            { StageScope t(c, ST_WARP_SPLIT, st);
              launch_warp_split(st, c->d_und, c->und_px, f0, c->d_wxy, c->d_wfrac, c->fe, c->d_gamma,
                                c->d_cbrt, c->d_coef, c->d_plane[P_R] + (size_t)f0 * ps, c->d_plane[P_B] + (size_t)f0 * ps,
                                ps, m); }
            for (int i = f0; i < f0 + m && i < (int)c->front_ok.size(); ++i) c->front_ok[(size_t)i] = 1;
        }
        int frc = run_filter_chain(c, st, f0, m, p, c->calib.warp_h, c->calib.warp_w, n);
        return frc ? frc : note_written_frame(c, st, f0, f0 + m, n);
    });
    if (rc) return rc;
    c->have_mask = true;
    mark_masks(c, first, n, 1, 0);
    return LT_OK;
}

int lt_mask_run(lt_ctx* c, int first, int n, const lt_filter_params* p) { return mask_run_impl(c, first, n, p, false); }
int lt_mask_rerun(lt_ctx* c, int first, int n, const lt_filter_params* p) { return mask_run_impl(c, first, n, p, true); }

int lt_filter_run(lt_ctx* c, int first, int n, const lt_filter_params* p) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    if ((rc = validate_filter(p))) return rc;
    if ((rc = set_device(c))) return rc;
    if (!c->d_bev) return fail(LT_ERR_STATE, "lt_upload_bev has not been called");
    if (n == 0) return LT_OK;
    const size_t ps = c->plane_bytes;
    front_stale(c, first, n);            // the planes become the uploaded bird's-eye image's
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

int lt_sws_fit_run(lt_ctx* c, int first, int n, const lt_search_params* p) {
    int rc = check_slots(c, first, n);
    if (rc) return rc;
    c->rec_mirror_slot = -1;                  // the records change: the page-locked mirror of a one-frame search is stale
    if ((rc = set_device(c))) return rc;
    if (!c->have_mask) return fail(LT_ERR_STATE, "no mask in the slots: run lt_mask_run or lt_upload_masks first");
    SearchGeom g;
    if ((rc = make_search_geom(c, p, false, g))) return rc;
    if ((rc = ensure_search_buffers(c, g.maxpix, g.maxlev))) return rc;
    if ((rc = ensure_band_sums(c, g.nbands))) return rc;
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
        return note_written_frame(c, st, f0, f0 + m, n);
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
        const char* one_env = n == 1 ? LT_EXP_ENV("LT_BAND_ONE") : nullptr;
        lt_lane_record* mirror = n == 1 ? rec_mirror_device(c) : nullptr;
        unsigned ticket = ++c->rec_ticket_counter;
        if (!ticket) ticket = ++c->rec_ticket_counter;      // 0 means "no ticket"
        if (n == 1 && !(one_env && one_env[0] == '0') &&
            launch_band_fit_one(st, mb, g, bp, c->d_pix + (size_t)f0 * 2 * c->maxpix, c->d_rec + f0, c->plane_bytes,
                                reinterpret_cast<const int*>(c->d_prev), mirror, ticket)) {
            if (mirror) {                    // the kernel itself leaves a copy of the record in page-locked memory, and its ticket
                c->rec_mirror_slot = f0;
                c->rec_mirror_stream = st;
                c->rec_ticket = ticket;
            }
        } else {
            launch_band_fit(st, c->d_plane[P_MASK] + (size_t)f0 * c->plane_bytes, c->plane_bytes, mb, g, c->d_prev + (size_t)f0 * 6, bp,
                            c->d_pix + (size_t)f0 * 2 * c->maxpix, c->d_rec + f0, m);
            if (n == 1) mirror_record(c, st, f0);
        }
        return note_written_frame(c, st, f0, f0 + m, n);
    });
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
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

int lt_set_walk_min_frames(lt_ctx* c, int frames) {
    if (!c) return fail(LT_ERR_INVALID, "null context");
    c->walk_min_pixels = frames < 0 ? 80LL * 1100 * 1080 : (long long)frames * c->calib.warp_w * c->calib.warp_h;
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
            for (auto f : fresh) stream_put(f);
            return fail(LT_ERR_HIP, "stream with a CU mask could not be created");
        }
        fresh.push_back(st);
    }
    for (auto st : c->streams) stream_put(st);
    c->streams = fresh;
    c->stream = c->streams.empty() ? c->stream : c->streams[0];
    stream_put(c->search); c->search = nullptr;
    stream_put(c->present); c->present = nullptr;
    stream_put(c->urgent); c->urgent = nullptr; c->urgent_on = false;
    stream_put(c->dl); c->dl = nullptr;
    c->search_cus = n;
    c->rec_mirror_slot = -1;             // the stream the mirror of a one-frame search was queued on is gone
    c->rec_mirror_stream = nullptr;
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