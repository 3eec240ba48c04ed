// The chained band search of one stream (lt_band_fit_chain_run / _cancel / _collect; lane_tracker.py:449-509, 851-872):
// one workgroup walks the resident masks of consecutive frames and hands the fit on.  See lt_ctx.h.
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

extern "C" {

}  // extern "C"
namespace lt {
int ensure_search_stream(lt_ctx* c) {
    if (c->search) return LT_OK;
    if (c->search_cus > 0) {                  // the CUs the slots' streams were kept off (lt_set_search_cus)
        const int mine = c->search_cus >= 2 ? 1 : c->search_cus;   // with two or more, the others are the download stream's
        HIP_TRY(stream_get(&c->search, SK_CU_SET, mine));
    } else HIP_TRY(create_compute_stream(&c->search));
    return LT_OK;
}

// what a chain needs besides its slots: the search stream, the cancel word (page-locked, device-visible) and the page-locked
// staging of the records, one entry per slot.  The first chain of a stream used to pay for these (9 ms: lt_warm does it ahead).
int ensure_chain_buffers(lt_ctx* c) {
    int rc = ensure_search_stream(c);
    if (rc) return rc;
    if (!c->h_cancel) {
        HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_cancel), 64, hipHostMallocMapped));
        *c->h_cancel = 0;
        HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_cancel), c->h_cancel, 0));
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
    return LT_OK;
}
}  // namespace lt
extern "C" {

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
    if ((rc = ensure_chain_buffers(c))) return rc;
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

}  // extern "C"
