/*
 * lane_tracker_amd.h -- C ABI of the MI355X-native lane-tracker hot path.
 *
 * The upstream reference (pierluigiferrari/lane_tracker) is pure Python and has no FFI of its own;
 * its boundary for this path is the Python API used by process_video.py.  This header is what a
 * maintainer binds with ctypes to replace the cv2/NumPy call sites on that path (INTEGRATION.md
 * shows the binding).  Each entry point cites the reference lines it replaces.
 *
 * Conventions: every function returns 0 on success and a negative lt_status on failure;
 * lt_last_error() returns a thread-local description.  No exceptions cross the ABI, no torch or
 * HIP types appear in signatures.  A context owns one HIP stream and all of its device buffers;
 * contexts are not thread-safe (one per tracker / per rank).  Host buffers are caller-owned.
 * The library is HIP-only: there is no CPU fallback behind any entry point.
 *
 * Frame "slots": a context holds `capacity` (lt_reserve) independent frame slots in HBM.  Slot i
 * keeps the camera frame, every intermediate plane, the mask, the lane-pixel lists and the 64-byte
 * lane record of frame i.  The *_run functions enqueue kernels on the context's stream and return
 * immediately; downloads synchronise.
 */
#ifndef LANE_TRACKER_AMD_H
#define LANE_TRACKER_AMD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden; the declarations below are its whole export list
 * (tests/test_native_abi.py compares `nm -D` with this header). */
#pragma GCC visibility push(default)

/* 2: + lt_gather_*, lt_band_fit_chain_run, lt_calib_*, lt_upload_frame_rows_async & co.; lt_debug_cycles removed;
 *    lt_last_threshold_path(NULL) returns LT_NO_CONTEXT instead of -1. */
/* 3: + lt_present_frame, lt_present_lane_async, lt_present_finish, lt_overlay_rows, lt_upload_frame_rest_rows (one frame per
 *    call, the host waiting: LaneTracker.process()), lt_set_download_method, lt_download_stats, lt_device_cache_trim,
 *    lt_last_adaptive_path, lt_host_copy_async, lt_host_copy2d_async, lt_host_copy_wait, lt_overlay_run_rows,
 *    lt_download_overlay_rows_async.  Nothing removed or changed. */
/* 4: + lt_host_copy_group_create / _destroy, lt_host_copy_async_group, lt_host_copy2d_async_group, lt_host_copy_wait_group
 *    (completion per group instead of per process), lt_shutdown, lt_device_cache_stats, lt_warm (a stream's set-up ahead of its
 *    first window), lt_overlay_run_strip + lt_strip_download_async (annotated frames as one packed strip of rows per frame through
 *    page-locked staging blocks into ordinary memory), lt_text_blend_host + lt_host_text_async_group (the text lines on the host),
 *    lt_host_copy_stats, lt_host_touch_async_group; lt_upload_frame_rows_enqueue (an upload nobody waits for),
 *    lt_present_lane_from_fit_async + lt_lane_spans_from_fit (the lane of a frame drawn by the device behind its search),
 *    lt_overlay_run_strip_coeffs (strips from averaged coefficients).  Nothing removed or changed. */
/* 5: + lt_device_cache_counters (hits / misses / evictions of the device-memory cache since the process started),
 *    lt_set_walk_min_frames (the batch size from which the threshold stage takes its walking kernels; was an environment
 *    switch), lt_host_memory_stats, lt_mask_rerun (a second parameter set over frames whose front end has run), lt_download_lane_lists (a search's lists in one
 *    round trip), lt_set_direct_upload + lt_direct_upload_count (one frame's rows stored through the PCIe aperture),
 *    lt_host_text_now_group (a frame's text lines drawn at once, behind the group's copies), lt_frame_tail (validity, average,
 *    plot points, radius and eccentricity of a valid first try in one host call).  Nothing removed or changed. */
#define LT_ABI_VERSION 5

typedef enum lt_status {
    LT_OK = 0,
    LT_ERR_INVALID = -1,     /* bad argument (also the reference's ValueError cases) */
    LT_ERR_HIP = -2,         /* a HIP runtime call failed */
    LT_ERR_NOMEM = -3,
    LT_ERR_CAPACITY = -4,    /* slot range outside lt_reserve()'d capacity */
    LT_ERR_STATE = -5        /* e.g. search requested before a mask exists */
} lt_status;

typedef struct lt_ctx lt_ctx;

/* LaneTracker.__init__ arguments that shape the kernels (lane_tracker.py:101-137). */
typedef struct lt_calib {
    int32_t img_w, img_h;        /* img_size     (width, height) */
    int32_t warp_w, warp_h;      /* warped_size  (width, height) */
    double  cam_matrix[9];       /* row major */
    double  dist_coeffs[5];      /* k1 k2 p1 p2 k3 */
    double  M[9];                /* warp_matrices[0]: camera -> bird's eye */
} lt_calib;

/* filter_lane_points() keyword arguments (lane_tracker.py:183-193). */
typedef struct lt_filter_params {
    int32_t filter_type;         /* 0 'bilateral', 1 'neighborhood'; anything else -> LT_ERR_INVALID (:220) */
    int32_t ksize_r, C_r, ksize_b, C_b;
    int32_t mask_noise, noise_thresh, ksize_noise, C_noise;
} lt_filter_params;

/* sliding_window_search() / band_search() arguments (lane_tracker.py:242-253, 449). */
typedef struct lt_search_params {
    int32_t window_width, window_height, search_range, no_success_limit;
    int32_t ignore_sides, ignore_bottom, bandwidth, _pad;
    double  mu, start_slice, partial;
} lt_search_params;

/* One fitted frame, 64 bytes; this is the unit the multi-GPU gather moves. */
typedef struct lt_lane_record {
    double  left_coeffs[3];      /* np.polyfit order: x = a*y^2 + b*y + c (lane_tracker.py:506) */
    double  right_coeffs[3];     /* (:507) */
    int32_t n_left, n_right;     /* lane-pixel counts */
    uint8_t detected;            /* self.detected_pixels (:438, :496) */
    uint8_t fit_flags;           /* bit0: left fit rank-deficient (<3 distinct y), bit1: right */
    uint8_t mode;                /* 0 sliding-window, 1 band, 255 not searched (lt_band_fit_chain_run stopped before it) */
    uint8_t _pad;                /* reserved (the library records how the slot's lane pixels are stored) */
    int32_t frame;               /* caller-defined global frame index */
} lt_lane_record;

typedef struct lt_info {
    int32_t abi_version, device, capacity, cu_count;
    int32_t src_row0, src_row1;  /* camera rows [row0,row1) that influence the bird's-eye view */
    int32_t max_pixels_per_side; /* capacity of each lane-pixel list */
    int32_t max_levels;          /* capacity of each centroid list */
    int64_t alg_bytes_mask;      /* algorithmic bytes / frame of the warp+threshold stage */
    int64_t alg_bytes_search;    /* algorithmic bytes / frame of search+fit, excluding emitted pixels */
    char    device_name[64];
} lt_info;

enum lt_plane {                  /* lt_download_plane selectors (bird's-eye planes, h*w bytes each) */
    LT_PLANE_R = 0,              /* rgb_r_channel        (:207) */
    LT_PLANE_LAB_B = 1,          /* lab_b_channel        (:208) */
    LT_PLANE_TOPHAT_R = 2,       /* rgb_r_tophat         (:210) */
    LT_PLANE_TOPHAT_B = 3,       /* lab_b_tophat         (:211) */
    LT_PLANE_MERGED = 4,         /* merged               (:229-235) */
    LT_PLANE_MASK = 5            /* opened = return value (:238) */
};

#define LT_NUM_STAGES 12         /* lt_stage_ms slots, see lt_stage_name() */

/* ---- lifecycle ------------------------------------------------------------------------------- */
const char* lt_last_error(void);
int  lt_abi_version(void);
int  lt_device_count(int* count);
/* Builds the calibration-constant tables (remap tables, Lab LUTs, structuring elements) on the
 * host in f64 and uploads them.  Replaces the per-call table work inside cv2.undistort /
 * cv2.warpPerspective / cv2.cvtColor / cv2.getStructuringElement (:203-205, :208, :832, :834). */
int  lt_create(const lt_calib* calib, int device, lt_ctx** out);
void lt_destroy(lt_ctx* ctx);
int  lt_reserve(lt_ctx* ctx, int capacity);
/* Set-up ahead of use, for the current capacity (call lt_reserve first): the streams, page-locked staging and device buffers that
 * the first searches / the first chained search / the first overlay would otherwise create on the way (20-35 ms of the first
 * window of a stream).  sws / band: the search parameters to size the result buffers for (either may be NULL); annotate: 0 = no
 * presentation stage, 1 = whole annotated frames (lt_overlay_run), 2 = strips (lt_overlay_run_strip); needs lt_overlay_configure
 * for 1 and 2.  Optional: every entry point still sets up what it finds missing. */
int  lt_warm(lt_ctx* ctx, const lt_search_params* sws, const lt_search_params* band, int annotate);
int  lt_get_info(lt_ctx* ctx, lt_info* out);
int  lt_sync(lt_ctx* ctx);
/* Number of HIP streams (1..8, default 1) the context spreads its slots over.  Slot s always runs on
 * stream s*k/capacity, so the stages of one frame stay ordered while slices overlap: the latency-bound
 * search of one slice hides under the mask chain of another.  Every transfer / sync waits for all streams. */
int  lt_set_streams(lt_ctx* ctx, int nstreams);

/* ---- frames in, results out ------------------------------------------------------------------ */
/* frames: n * img_h * img_w * 3 bytes, RGB interleaved, as LaneTracker.process() receives them (:876) */
int  lt_upload_frames(lt_ctx* ctx, const uint8_t* frames_rgb, int first_slot, int n);
/* Camera rows [row0, row1) that undistort + warp actually read (a third of a 720-row frame at the reference
 * calibration).  lt_upload_frame_rows takes the same full-size host frames as lt_upload_frames but moves only
 * those rows: enough for lt_mask_run and the searches, not for lt_overlay_run, which shows the whole frame. */
int  lt_get_source_rows(lt_ctx* ctx, int* row0, int* row1);
int  lt_upload_frame_rows(lt_ctx* ctx, const uint8_t* frames_rgb, int first_slot, int n);
/* The same copy, waited for by nobody: enqueued on the slots' own streams -- behind everything launched over these slots so far
 * (and behind overlays of other streams that still read them), ahead of whatever is launched over them next -- and the call
 * returns (from pageable memory: once the runtime has the bytes on their way).  frames_rgb must stay valid and unchanged until a call that
 * waits for work launched over these slots afterwards has returned (lt_download_records of a search, lt_sync).  What
 * LaneTracker.process() uses (:876: one frame per call, the caller's array): the engine's copy runs under the mask chain's launches. */
int  lt_upload_frame_rows_enqueue(lt_ctx* ctx, const uint8_t* frames_rgb, int first_slot, int n);
/* A small lt_upload_frame_rows_enqueue (at most 1.5 MB of rows: the one 1280x720 frame of a process() call) does not go to the copy engine
 * where the device's memory is mapped into the process (large BAR): the calling thread stores the rows into the slot itself,
 * through the PCIe aperture -- 20 us of bus time for a 1280x720 frame's 914 KB instead of 22-24 us of call + 23 us of engine
 * copy + 6 us until the first kernel behind it.  The call then returns with the copy DONE (frames_rgb is the caller's again) after
 * waiting, on the host, for kernels that still read those slots' camera rows.  Same bytes, same results.
 * lt_set_direct_upload(ctx, on): 1 / 0 allows (the default) / forbids it, negative leaves the setting; returns 1 when such calls take
 * the aperture on this context, 0 when not (no large BAR, memory not mapped, forbidden).  lt_direct_upload_count: calls that did. */
int  lt_set_direct_upload(lt_ctx* ctx, int on);
unsigned long long lt_direct_upload_count(lt_ctx* ctx);
/* The same rows without the host wait: the copy is enqueued on the context's copy stream behind the work already
 * enqueued for these slots, and everything enqueued for them afterwards waits for it -- the upload of one slot range
 * runs under the chain of the others (double-buffered host-fed pipeline).  frames_rgb must stay valid (and should be
 * page-locked, lt_host_alloc) until the next lt_sync. */
int  lt_upload_frame_rows_async(lt_ctx* ctx, const uint8_t* frames_rgb, int first_slot, int n);
/* The complement: every other row of the same frames, enqueued on a copy stream of its own so that it runs beside
 * the mask chain (call it after lt_mask_run).  The host buffer must stay valid until the next lt_sync or download;
 * lt_overlay_run waits for it. */
int  lt_upload_frame_rest(lt_ctx* ctx, const uint8_t* frames_rgb, int first_slot, int n);
/* Of that complement, only the rows inside the two runs rows4 = {a0, a1, b0, b1} (NULL: all of it, as above): what
 * lt_present_frame with the same runs reads.  The other rows of the slots keep whatever they held -- an overlay over the whole
 * frame needs lt_upload_frame_rest first. */
int  lt_upload_frame_rest_rows(lt_ctx* ctx, const uint8_t* frames_rgb, int first_slot, int n, const int32_t* rows4);
/* masks: n * warp_h * warp_w bytes; lets the search stages run on caller-supplied binary images */
int  lt_upload_masks(lt_ctx* ctx, const uint8_t* masks, int first_slot, int n);
int  lt_download_masks(lt_ctx* ctx, int first_slot, int n, uint8_t* masks);
int  lt_download_plane(lt_ctx* ctx, int plane, int first_slot, int n, uint8_t* out);
/* the undistorted camera rows [src_row0, src_row1) as RGB interleaved: n * rows * img_w * 3 */
int  lt_download_undistorted(lt_ctx* ctx, int first_slot, int n, uint8_t* out);
int  lt_download_records(lt_ctx* ctx, int first_slot, int n, lt_lane_record* out);
/* side 0 = left, 1 = right.  Writes at most cap (y,x) pairs in reference order; returns the count
 * through *count (self.left_y/left_x/right_y/right_x, :434-437, :492-495). */
int  lt_download_pixels(lt_ctx* ctx, int slot, int side, int32_t* ys, int32_t* xs, int cap, int* count);
/* self.left_window_centroids / right_window_centroids (:439-440) */
int  lt_download_centroids(lt_ctx* ctx, int slot, int side, int32_t* out, int cap, int* count);
/* Both sides' lane pixels and, with want_centroids, both window-centroid lists of a slot in ONE round trip to the device (the lists
 * of lane_tracker.py:434-440 / :492-495 as lt_download_pixels and lt_download_centroids return them).  counts[2] / cent_counts[2]:
 * the full lengths (left, right); lists longer than cap / cent_cap are cut.  LT_ERR_CAPACITY when the slot's list region does not
 * fit the staging buffer (very large search windows): use the two calls above. */
int  lt_download_lane_lists(lt_ctx* ctx, int slot, int32_t* left_y, int32_t* left_x, int32_t* right_y, int32_t* right_x, int cap, int* counts,
                            int want_centroids, int32_t* cent_left, int32_t* cent_right, int cent_cap, int* cent_counts);
/* device-to-device copy of n records into caller-owned device memory (e.g. a collective's send buffer) */
int  lt_copy_records_to_device(lt_ctx* ctx, int first_slot, int n, void* dst_device);
/* the same copy enqueued behind the slots' searches on the context's streams, without waiting: the records are in
 * dst_device after the next lt_sync (several steps can fill one send buffer and be gathered once) */
int  lt_enqueue_records_to_device(lt_ctx* ctx, int first_slot, int n, void* dst_device);

/* ---- the hot path, device resident ----------------------------------------------------------- */
/* find_lane_points() part 1 (:832-846): undistort -> warpPerspective -> filter_lane_points. */
int  lt_mask_run(lt_ctx* ctx, int first_slot, int n, const lt_filter_params* p);
/* The same over slots whose frames have been through lt_mask_run already, with OTHER filter parameters -- the second try of a frame
 * (lane_tracker.py:1081-1101: the 'neighborhood' set over the same bird's-eye image): the undistortion and the warp are skipped
 * where a slot's R / Lab-b planes are still those of the frame it holds (any upload of camera rows into the slot, or lt_filter_run,
 * ends that), and run as in lt_mask_run where they are not.  Results are those of lt_mask_run. */
int  lt_mask_rerun(lt_ctx* ctx, int first_slot, int n, const lt_filter_params* p);
/* filter_lane_points() only, on bird's-eye RGB images already uploaded with lt_upload_bev (:183-240) */
int  lt_upload_bev(lt_ctx* ctx, const uint8_t* bev_rgb, int first_slot, int n);
int  lt_filter_run(lt_ctx* ctx, int first_slot, int n, const lt_filter_params* p);
/* sliding_window_search() + fit_poly() (:242-447, :502-509) on the slots' masks */
int  lt_sws_fit_run(lt_ctx* ctx, int first_slot, int n, const lt_search_params* p);
/* band_search() + fit_poly() (:449-509); prev_coeffs: n * 6 doubles (last_left_coeffs, last_right_coeffs) */
int  lt_band_fit_run(lt_ctx* ctx, int first_slot, int n, const lt_search_params* p, const double* prev_coeffs);
/* The warm path of ONE stateful stream, chained on the device (band_search :449-500 + fit_poly :502-509 per frame, with
 * the cross-frame dependence of :474-489 / :1182-1183 kept in HBM): the slots first_slot .. first_slot + n - 1 hold
 * consecutive frames of one video; the band of the first is drawn around seed_coeffs (6 doubles: last_left_coeffs,
 * last_right_coeffs) or, if seed_coeffs is NULL, around the fit in the record of slot first_slot - 1; the band of every
 * later frame around the fit of the frame before it -- what the reference does as long as every frame is found valid.
 * Validity (check_validity, :561-627) stays on the host: the caller collects the n records once, keeps those up to the
 * first frame it rejects and discards the rest (speculation; results never differ from the frame-by-frame calls).  The
 * walk stops by itself behind a frame without both lanes or with a rank-deficient fit; the slots it did not search get
 * detected = 0 and mode = 255.  LT_ERR_STATE if 2 * bandwidth + 2 > 64 or the mask width is not a multiple of 4
 * (use lt_band_fit_run frame by frame then).
 * The chain is one workgroup: it runs on a stream of its own behind the work already enqueued for its slots, beside the
 * mask chains of later slots, and leaves its records (with a NULL seed also the seed record of slot first_slot - 1) in
 * page-locked host memory.  lt_band_fit_chain_collect copies records [first_slot, first_slot + n) of the most recent
 * chain covering that range to `out`, waiting for that chain only -- not for the device; every lt_download_* / lt_sync
 * also waits for all chains. */
int  lt_band_fit_chain_run(lt_ctx* ctx, int first_slot, int n, const lt_search_params* p, const double* seed_coeffs);
int  lt_band_fit_chain_collect(lt_ctx* ctx, int first_slot, int n, lt_lane_record* out);
/* Give the chained search n CUs of its own (0 = none, the default): the context's compute streams are recreated with a CU
 * mask that keeps them off CUs 0 .. n-1, and the search stream is restricted to those.  Why: the kernels of the mask chain
 * spread their workgroups over the chip once, statically; a workgroup of the chain kernel sharing ONE CU with them slows that
 * CU's share of every mask kernel, and each kernel then ends with that CU -- measured: the mask chain of 128-frame launches
 * takes 18.6 us per frame beside a running chain, 15.1 with one CU set aside (12.8 alone).  With n >= 2 the search has CU 0
 * and the others belong to the download stream (used when LT_DL_KERNEL=1 copies annotated frames back with a kernel).  Call it
 * on an idle context (it synchronises); a context that only processes independent batches has no use for it. */
int  lt_set_search_cus(lt_ctx* ctx, int n);
/* The bilateral threshold (lane_tracker.py:14-83, :214-215) has two kernels: walks down half rows / columns with running sums
 * (k_bilateral_walk_hv), which win once a call brings enough frames to fill the chip, and 128 x 128 tiles in LDS
 * (k_bilateral_tile2) for the few frames of a process() call or a short chunk.  A call of at least `frames` frames (of this
 * context's bird's-eye size) takes the walks; default: 80 frames' worth of 1100 x 1080 pixels, the crossover measured on MI355X
 * (64 frames: 232 vs 210 us, 96 frames: 255 vs 304 us).  0: always walk; negative: the default again.  Results never depend on it. */
int  lt_set_walk_min_frames(lt_ctx* ctx, int frames);
/* Urgent mode, for the frame of a stateful stream whose first try failed (lane_tracker.py:1071-1128: second parameter set,
 * second search) while masks of later frames are already queued: between lt_set_urgent(ctx, 1) and lt_set_urgent(ctx, 0),
 * lt_mask_run / lt_filter_run / lt_sws_fit_run / lt_band_fit_run run on a stream of their own, behind the work enqueued for
 * THEIR slots only, and the lt_download_* calls wait for that stream only (they must ask for results produced in urgent mode,
 * or already complete).  Work enqueued for those slots afterwards is ordered behind it.  Leaving the mode waits for it. */
int  lt_set_urgent(lt_ctx* ctx, int on);
/* The caller has rejected a frame: every chain enqueued so far ON THIS CONTEXT -- whatever its slot range -- stops at its
 * next frame (the slots it has not searched get mode 255) instead of finishing its speculation.  Chains enqueued
 * afterwards are not affected.  One context serves one stream; independent streams take separate contexts. */
int  lt_band_fit_chain_cancel(lt_ctx* ctx);
/* tag records with global frame indices first_frame, first_frame+1, ... */
int  lt_set_frame_base(lt_ctx* ctx, int first_slot, int n, int first_frame);

/* ---- host-buffer convenience wrappers (upload + run + download) -------------------------------- */
int  lt_mask_batch(lt_ctx* ctx, const uint8_t* frames_rgb, int n, const lt_filter_params* p, uint8_t* masks);
/* masks == NULL: use the device-resident masks of slots [0,n) */
int  lt_sws_fit_batch(lt_ctx* ctx, const uint8_t* masks, int n, const lt_search_params* p, lt_lane_record* out);
int  lt_band_fit_batch(lt_ctx* ctx, const uint8_t* masks, int n, const lt_search_params* p,
                       const double* prev_coeffs, lt_lane_record* out);

/* ---- single-image operators with the reference's module-level signatures ----------------------- */
/* bilateral_adaptive_threshold(img, ksize, C, mode, true_value, false_value) (:14-83).
 * mode 0 'floor', 1 'ceil', anything else -> LT_ERR_INVALID (the reference's ValueError, :71). */
int  lt_bilateral_adaptive_threshold(lt_ctx* ctx, const uint8_t* img, int h, int w, int ksize, int C,
                                     int mode, int true_value, int false_value, uint8_t* out);
/* LaneTracker.filter_lane_points(img, ...) on one bird's-eye RGB image of any size (:183-240) */
int  lt_filter_lane_points(lt_ctx* ctx, const uint8_t* bev_rgb, int h, int w, const lt_filter_params* p,
                           uint8_t* mask);

/* One elliptical morphology operator on a single-channel image of any size: the building block of
 * morphologyEx (:210-211, :238).  k in {5, 29, 55}; op 0 erode, 1 dilate, 2 top-hat, 3 open.
 * direct != 0 evaluates the footprint tap by tap instead of using the run decomposition (k = 5 always does). */
int  lt_morph_ellipse(lt_ctx* ctx, const uint8_t* img, int h, int w, int k, int op, int direct, uint8_t* out);

/* LaneTracker.fit_poly() on explicit pixel lists (np.polyfit(ys, xs, 2), :506-507): n (y,x) pairs with
 * coordinates in [0, 65535].  *rank_deficient is set when fewer than 3 distinct y exist (coef = 0). */
int  lt_fit_poly2(lt_ctx* ctx, const int32_t* ys, const int32_t* xs, int n, int h, int w, double coef[3],
                  int* rank_deficient);

/* ---- presentation stage (next row after the hot path) ------------------------ */
/* Minv: the pickled inverse perspective matrix the reference hands to warpPerspective in draw_lane
 * (lane_tracker.py:648); builds the camera-sized remap table once.  Call it BEFORE uploading frames whose annotated form will be
 * asked for: when the rows the lane can reach (lt_overlay_rows) stick out of the rows lt_upload_frame_rows brings by a few rows
 * (458-696 against 457-695 of 720 with the reference calibration), that run is widened to cover them (lt_get_source_rows reports
 * the new run), so that an annotated frame's lane rows need no upload of their own; a frame uploaded before lacks those rows. */
int  lt_overlay_configure(lt_ctx* ctx, const double* Minv /* 9 */);
/* draw_lane() without the text (lane_tracker.py:637-662) for the frames in slots [first, first+n):
 * fillPoly of the polygon left points + reversed right points in (0,255,0), warpPerspective with Minv,
 * addWeighted(frame, 1, lane, alpha, 0).  left_n/right_n: points per slot; left_yx/right_yx: (y, x)
 * int32 pairs of all slots, concatenated.  A slot with no points yields a copy of its frame. */
int  lt_overlay_run(lt_ctx* ctx, int first_slot, int n, const int32_t* left_n, const int32_t* right_n,
                    const int32_t* left_yx, const int32_t* right_yx, double alpha);
/* The same, drawing only two runs of camera rows rows4 = {a0, a1, b0, b1} of every frame (NULL: whole frames): for annotated
 * frames that travel back as row runs (lt_download_overlay_rows_async; lt_overlay_rows says which rows the lane can reach). */
int  lt_overlay_run_rows(lt_ctx* ctx, int first_slot, int n, const int32_t* left_n, const int32_t* right_n,
                         const int32_t* left_yx, const int32_t* right_yx, double alpha, const int32_t* rows4);
/* Text on the annotated frames (putText in draw_lane / print_failure, :653-672).  OpenCV's Hershey glyphs are
 * not reproduced: the caller supplies its own glyph atlas once -- n_glyphs alpha cells of glyph_w x glyph_h bytes
 * for the characters first_char, first_char + 1, ..., and each character's advance width (<= glyph_w). */
int  lt_overlay_set_font(lt_ctx* ctx, const uint8_t* atlas, const uint8_t* advance, int first_char, int n_glyphs,
                         int glyph_w, int glyph_h);
/* Blend n_lines white text lines into the annotated frame of every slot in [first, first+n) (after lt_overlay_run).
 * lines: n * n_lines * line_len bytes, zero-padded; line i of a slot starts at (x0, y0 + i * step). */
int  lt_overlay_text(lt_ctx* ctx, int first_slot, int n, const char* lines, int n_lines, int line_len, int x0, int y0,
                     int step);
/* The camera rows [*row0, *row1) in which draw_lane()'s inverse warp (lane_tracker.py:648) can place a lane pixel at all, from
 * the table lt_overlay_configure built: every pixel of an annotated frame outside these rows and outside the text lines equals
 * the camera pixel, whatever the polygon. */
int  lt_overlay_rows(lt_ctx* ctx, int* row0, int* row1);
/* draw_lane() / print_failure() (lane_tracker.py:629-673) for ONE resident frame in one call: lt_overlay_run with one polygon
 * (left_n / right_n: one count each), lt_overlay_text when lines != NULL, and the annotated frame into `out` (img_h * img_w * 3
 * bytes, page-locked memory preferred); returns when it is there.  rows4 = NULL: the whole frame.  rows4 = {a0, a1, b0, b1},
 * two ordered runs of camera rows: only these rows are drawn and written, at their places in `out` -- the caller fills the
 * others from the camera frame it holds (LaneTracker.process() does so while the device is busy, and half the frame crosses the
 * bus).  The runs must cover the text lines and, for a non-empty polygon, lt_overlay_rows; LT_ERR_INVALID otherwise. */
int  lt_present_frame(lt_ctx* ctx, int slot, const int32_t* left_n, const int32_t* right_n, const int32_t* left_yx,
                      const int32_t* right_yx, double alpha, const char* lines, int n_lines, int line_len, int x0, int y0,
                      int step, uint8_t* out, const int32_t* rows4);
/* lt_present_frame in two halves, for a caller that knows the polygon before it knows the text (radius, eccentricity and the
 * verdict on the frame take LaneTracker.process() another 25 us of host work behind the record).  lt_present_lane_async draws
 * both row runs (rows4 is required: the text lines in the first run, lt_overlay_rows in the second, the two apart) and sends the
 * second run on its way without waiting; lt_present_finish blends the text into the first run, sends it and waits for both.  A
 * first half that turns out to be for nothing (the frame was invalid) is followed by a whole lt_present_frame on the same slot
 * and `out`, which draws and sends everything again. */
int  lt_present_lane_async(lt_ctx* ctx, int slot, const int32_t* left_n, const int32_t* right_n, const int32_t* left_yx,
                           const int32_t* right_yx, double alpha, uint8_t* out, const int32_t* rows4);
int  lt_present_finish(lt_ctx* ctx, int slot, const char* lines, int n_lines, int line_len, int x0, int y0, int step,
                       uint8_t* out, const int32_t* rows4);
/* The first half without the host: the averaged lane of the frame in `slot` (draw_lane's polygon, :629-662, of the running average
 * :1182-1189 that this frame's fit would give) drawn by the device itself, enqueued on the slot's stream right behind its search --
 * call it after lt_band_fit_run / lt_sws_fit_run over that one slot, before waiting for the record.  prev_sum: the sum, in order,
 * of the older fits (left a, b, c, right a, b, c) that stay in the average, count: fits in the average with this frame's (1: this
 * frame's alone, prev_sum ignored); ploty / ploty2: get_poly_points' rows and their squares (:514).  One workgroup forms average,
 * plot points and the polygon's row intervals from the fit in the slot's record with the host's f64 operations in the host's order;
 * the overlay stores the rows of rows4's second run (its first run must be empty: text on the host) into the page-locked `out`.
 * A record without a usable fit (nothing detected, fit_flags != 0) draws nothing; lt_present_finish waits for the rows.  Returns
 * LT_ERR_STATE where this form does not exist (out not page-locked / 16-byte aligned, too many plot rows): draw with
 * lt_present_lane_async then.  lt_lane_spans_from_fit: the same workgroup for a fit given by value, its row intervals
 * (warp_h x (lo, hi) int16) brought back -- for tests. */
int  lt_present_lane_from_fit_async(lt_ctx* ctx, int slot, const double* prev_sum, int count, const double* ploty,
                                    const double* ploty2, int n_rows, double alpha, uint8_t* out, const int32_t* rows4);
int  lt_lane_spans_from_fit(lt_ctx* ctx, const double* fit6, int detected, int fit_flags, const double* prev_sum, int count,
                            const double* ploty, const double* ploty2, int n_rows, int16_t* spans_out);
/* Host-only helper (no GPU needed): the (lo, hi) column interval per bird's-eye row that cv2.fillPoly
 * paints for that polygon; empty rows are (32767, -32768).  spans: warp_h * 2 int16. */
int  lt_lane_polygon_spans(int warp_h, const int32_t* left_yx, int n_left, const int32_t* right_yx, int n_right,
                           int16_t* spans);
/* Host-only helper (no GPU needed): get_poly_points (lane_tracker.py:511-528) for n pairs of parabolas, in the packed form
 * lt_overlay_run takes.  coeffs: n * 6 doubles (left a, b, c, right a, b, c); ploty / ploty2: the n_rows plot rows and their
 * squares, as the caller's NumPy computed them.  fitx = a * ploty2 + b * ploty + c in exactly these IEEE operations; points
 * with 0 <= fitx <= warp_w - 1 are kept, x truncated, y = warp_h - count .. warp_h - 1 (upstream's rule).  left_n / right_n:
 * n counts; left_yx / right_yx: room for n * n_rows (y, x) pairs each, written back to back. */
int  lt_poly_points(int warp_w, int warp_h, const double* coeffs, int n, const double* ploty, const double* ploty2, int n_rows,
                    int32_t* left_n, int32_t* right_n, int32_t* left_yx, int32_t* right_yx);
/* Host-only (no GPU needed): what LaneTracker.process() computes between a frame's record and its text lines when the frame's first
 * try is valid -- check_validity (:561-627), the running average of the fits (:1186-1187), get_poly_points of the averaged curves
 * (:511-528), get_curve_radius through the pixel fit (:530-549) and get_eccentricity (:551-559) -- in the host's operations and
 * order, in one call.  in: 22 doubles {left fit a b c, right fit a b c, sum of the window's other valid fits (6), divisor, the seven
 * validity limits, metres per pixel vertical, horizontal}; ploty_v / ploty2_v: plot rows of partial = 1, ploty / ploty2: of the
 * frame's partial; avg6, left_n .. right_yx: the averaged coefficients and their points as lt_poly_points leaves them; out: 5
 * doubles {valid, not-reproduced flag, left radius, right radius, eccentricity}.  With the flag set the caller computes the frame the
 * long way (a radius within 1e-8 of an integer, where upstream's refit decides int(); no plot point inside the image; ...). */
int  lt_frame_tail(int warp_w, int warp_h, const double* in, const double* ploty_v, const double* ploty2_v, int n_rows_v, const double* ploty,
                   const double* ploty2, int n_rows, double* avg6, int32_t* left_n, int32_t* right_n, int32_t* left_yx, int32_t* right_yx,
                   double* out);
/* annotated frames, RGB interleaved, n * img_h * img_w * 3 bytes */
int  lt_download_overlay(lt_ctx* ctx, int first_slot, int n, uint8_t* out);
/* The same copy enqueued behind the slots' overlay work without waiting: `out` (page-locked memory from lt_host_alloc, or
 * the copy is not asynchronous) holds the frames after the next lt_sync / lt_download_*.  lt_overlay_run and lt_overlay_text
 * themselves only enqueue (their staging is per slot), so a window can be rendered and downloaded in pieces while later
 * frames are still searched; a call over slots whose previous overlay is still in flight waits for that one. */
int  lt_download_overlay_async(lt_ctx* ctx, int first_slot, int n, uint8_t* out);
/* The same for two runs of rows of every frame (rows4 = {a0, a1, b0, b1}; NULL: whole frames), at their places in `out`: the
 * other rows of an annotated frame equal the camera frame (lt_overlay_rows) and need not cross the bus -- the caller copies them
 * from the frames it holds (lt_host_copy2d_async). */
int  lt_download_overlay_rows_async(lt_ctx* ctx, int first_slot, int n, uint8_t* out, const int32_t* rows4);
/* Wait until every copy enqueued by lt_download_overlay_async has landed -- and for nothing else: uploads and masks of
 * later frames keep running (lt_sync would drain them too). */
int  lt_download_overlay_wait(lt_ctx* ctx);
/* How lt_download_overlay_async moves the frames: 0 = the copy engine, 1 = a kernel storing into the (page-locked, 16-byte
 * aligned) destination, -1 (default) = chosen by measurement: every copy is timed, the engine is used while its copies
 * reach ~42 GB/s, otherwise whichever of the two measures faster (the engine's rate depends on how the process's memory
 * happens to be laid out: 28-56 GB/s; the kernel reaches 38-40 GB/s regardless).  lt_download_stats reports the running
 * rates (GB/s), the number of timed copies per method and the method the next copy would use; any pointer may be NULL. */
int  lt_set_download_method(lt_ctx* ctx, int method);
int  lt_download_stats(lt_ctx* ctx, double* engine_gbs, int* engine_copies, double* kernel_gbs, int* kernel_copies, int* method);
/* ---- annotated frames of a window as strips (round 5) ----------------------------------------------------------------------
 * Outside the rows the lane can reach (lt_overlay_rows) an annotated frame is the camera frame plus the text lines -- both of
 * which the host has.  So of an annotated frame only that run of rows is drawn on the device, packed ("strip", rows * img_w * 3
 * bytes per slot), and it comes back as ONE contiguous copy per block of 32 slots into page-locked staging blocks the library
 * pools (copy engine, full rate), from where the library's copy threads put the rows into the caller's frames -- ordinary
 * memory: no window-sized page-locked output array (0.13 s of page-locking per 0.7 GB, the first window's largest cost).  The
 * other rows and the text the caller adds with lt_host_copy2d_async_group / lt_host_text_async_group in the same group;
 * lt_host_copy_wait_group(group) returns when the frames are complete.  Needs img_w % 4 == 0.
 *   lt_overlay_run_strip     lt_overlay_run for rows [lt_overlay_rows) only, into the context's strip buffer
 *   lt_strip_download_async  the strips of slots [first, first + n) -> rows [row0, row1) of out + i * out_frame_stride (i < n)
 *   lt_overlay_run_strip_coeffs  lt_overlay_run_strip from the lanes' AVERAGED coefficients (n x 6 doubles: left a, b, c, right
 *                            a, b, c; :1182-1189) instead of their plot points: get_poly_points (:511-528; ploty / ploty2 as for
 *                            lt_present_lane_from_fit_async) and the polygons' row intervals are formed on the device, one
 *                            workgroup per frame.  draw: n bytes, 0 = no lane in that frame (nullptr: all drawn).  LT_ERR_STATE
 *                            where that form does not exist (an odd bird's-eye height): use the points form */
int  lt_overlay_run_strip(lt_ctx* ctx, int first_slot, int n, const int32_t* left_n, const int32_t* right_n, const int32_t* left_yx,
                          const int32_t* right_yx, double alpha);
int  lt_overlay_run_strip_coeffs(lt_ctx* ctx, int first_slot, int n, const double* coeffs, const uint8_t* draw, const double* ploty,
                                 const double* ploty2, int n_rows, double alpha);
int  lt_strip_download_async(lt_ctx* ctx, int first_slot, int n, uint8_t* out, size_t out_frame_stride, int group);
/* The text lines of draw_lane() / print_failure() (lane_tracker.py:652-661, 664-673; glyph atlas as for lt_overlay_set_font) drawn on
 * the HOST, with lt_overlay_text's arithmetic bit for bit (white over the frame: v + ((255 - v) * alpha + 127) / 255): `lines` holds
 * n_lines * line_len bytes per frame (NUL-padded), frame after frame.  No GPU, no context.
 *   lt_text_blend_host        n frames in place, on the calling thread (LaneTracker.process(): ~10 us per frame)
 *   lt_host_text_async_group  on the library's copy threads, in `group`: per frame, first copy two runs of rows {a0, a1, b0, b1}
 *                             (rows4; NULL or empty runs: nothing) of the source frame into the destination frame -- every row
 *                             the device does not deliver: above and below the lane's run -- then draw the lines over them.
 *                             `lines` is copied; the frames and the atlas must stay valid until the group's wait.
 *   lt_host_text_now_group    ONE frame, now: waits for `group` (whose copies bring the rows under the text), then draws the lines --
 *                             the first on the calling thread, the others on copy threads that are polling at that moment (or here);
 *                             returns with the text drawn.  LaneTracker.process(): no job, no second wait. */
int  lt_text_blend_host(uint8_t* frames, size_t frame_stride, int n, int img_h, int img_w, const uint8_t* atlas, const uint8_t* advance,
                        int first_char, int n_glyphs, int glyph_w, int glyph_h, const char* lines, int n_lines, int line_len, int x0,
                        int y0, int step);
int  lt_host_text_async_group(int group, uint8_t* dst, size_t dst_stride, const uint8_t* src, size_t src_stride, int n, const int32_t* rows4,
                              int img_h, int img_w, const uint8_t* atlas, const uint8_t* advance, int first_char, int n_glyphs,
                              int glyph_w, int glyph_h, const char* lines, int n_lines, int line_len, int x0, int y0, int step);
int  lt_host_text_now_group(int group, uint8_t* frame, int img_h, int img_w, const uint8_t* atlas, const uint8_t* advance, int first_char,
                            int n_glyphs, int glyph_w, int glyph_h, const char* lines, int n_lines, int line_len, int x0, int y0, int step);
/* Page-locked host memory for buffers passed to the upload / download entry points (copies from or to pageable
 * memory run at a fraction of the PCIe rate).  Needs a GPU; lt_host_free(NULL) is a no-op.  The reference has no
 * counterpart: its frames are NumPy arrays on the host (lane_tracker.py:876, :662). */
int  lt_host_alloc(size_t bytes, void** out);
/* Plain host-to-host copies on a second host thread of the library (one per process), for a caller that has launches to issue
 * meanwhile: LaneTracker.process() fills the rows of its output that no overlay can touch from the camera frame this way.
 * lt_host_copy_async returns at once; both buffers must stay valid until lt_host_copy_wait() has returned, which is when every
 * copy requested so far (by any thread) is complete. */
int  lt_host_copy_async(void* dst, const void* src, size_t bytes);
/* height pieces of width bytes, dst_pitch / src_pitch bytes apart (a run of rows of every frame of a window), shared among
 * the library's copy threads (LT_COPY_THREADS, default 4) */
int  lt_host_copy2d_async(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width, size_t height);
int  lt_host_copy_wait(void);
/* The same copies with completion per GROUP: a copy belongs to the group it is submitted to, and lt_host_copy_wait_group(g)
 * returns when the copies of g are complete -- whatever other trackers, threads or windows have queued meanwhile (with one
 * counter per process, two trackers on two threads waited for each other's copies and a short wait could be starved by
 * another tracker's 360 MB window).  Group 0 is the default group lt_host_copy_async / lt_host_copy2d_async submit to;
 * lt_host_copy_wait() waits for every group.  lt_host_copy_group_create hands out a fresh id (> 0, never reused);
 * lt_host_copy_group_destroy waits for the group and forgets it.  An unknown group is LT_ERR_INVALID.  No GPU needed.
 * LaneTracker keeps one group per tracker for process() and one per window of a stream. */
int  lt_host_copy_group_create(int* group);
int  lt_host_copy_group_destroy(int group);
int  lt_host_copy_async_group(int group, void* dst, const void* src, size_t bytes);
int  lt_host_copy2d_async_group(int group, void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t width, size_t height);
int  lt_host_copy_wait_group(int group);
/* Write one byte into every 4 KB page of [p, p + bytes) on the copy threads: first touch of fresh memory ahead of its use (the
 * output frames of a stream's first windows; ~10 GB/s whatever the thread count).  The contents are undefined afterwards. */
int  lt_host_touch_async_group(int group, void* p, size_t bytes);
/* Finish the queued host copies and join the copy threads now (they are also joined when the library is unloaded, and start
 * again with the next request).  For hosts that must not have library threads alive at a point of their choosing -- before a
 * fork(), at interpreter shutdown.  (The child of a fork() gets fresh workers by itself: pthread_atfork.) */
int  lt_shutdown(void);
/* Since the process started: seconds the copy threads spent on pieces (summed over the threads), bytes of plain copies, pieces
 * run, and the number of threads requests are shared among (LT_COPY_THREADS; default half of the CPUs the process may use,
 * 2 .. 8).  Any pointer may be NULL.  bench.py reports the copy threads' share of an annotated stream from these. */
int  lt_host_copy_stats(double* busy_seconds, double* copied_bytes, long long* pieces, int* threads);
/* What the library holds on the host right now: page-locked staging blocks it has allocated (bytes; strips of annotated frames
 * travel through them), pieces waiting in the copy threads' queue, and pieces submitted or reserved but not finished.  A
 * long-lived process sees the first flat and the other two back at 0 between windows (tests/test_gpu_soak.py).  Any pointer
 * may be NULL. */
int  lt_host_memory_stats(size_t* staging_bytes, size_t* queued_pieces, size_t* pending_pieces);
int  lt_host_free(void* p);
/* Device memory a context gives up (lt_destroy, lt_reserve growing) is kept in a per-process cache, by device and exact
 * size, and reused by later allocations.  Why: memory handed back to the driver is wiped in the background on an SDMA engine,
 * and for that time the process's device-to-host copies run at half speed (csrc/lt_memory.cpp, DevCache).  The cache keeps at
 * most the high-water mark of what the process's live contexts have held at once, and at most 16 GB (LT_DEVICE_CACHE_GB=<n>:
 * another limit; 0: no cache); beyond that the blocks that have waited longest go back to the driver.  lt_device_cache_trim
 * returns everything beyond keep_bytes NOW -- for a process that shares the GPU (several ranks on one device, another
 * allocator in the same process): call it with 0 after closing trackers whose memory somebody else should have.  A failed
 * hipMalloc inside the library trims by itself and tries once more.  lt_device_cache_stats: bytes the cache holds, bytes handed
 * out to live contexts, the current limit, the number of cached blocks (any pointer may be NULL).  No context needed. */
int  lt_device_cache_trim(size_t keep_bytes);
int  lt_device_cache_stats(size_t* kept_bytes, size_t* live_bytes, size_t* limit_bytes, int* kept_blocks);
/* Since the process started: allocations served from the cache, allocations that went to hipMalloc, and what the cache gave
 * back to the driver (evictions over its limit, lt_device_cache_trim) -- a long-lived process that keeps evicting pays the
 * driver's wipe with half-rate device-to-host copies each time (tests/test_gpu_soak.py holds these flat).  Any pointer may be NULL. */
int  lt_device_cache_counters(unsigned long long* hits, unsigned long long* misses, unsigned long long* evicted_blocks,
                              unsigned long long* evicted_bytes);
/* the bird's-eye RGB image of the slots' frames (lane_tracker.py:834, :1035): n * warp_h * warp_w * 3;
 * needs lt_mask_run on those slots first (it reuses their undistorted rows) */
int  lt_download_bev(lt_ctx* ctx, int first_slot, int n, uint8_t* out);

/* ---- host-only views of the calibration tables (no GPU needed) --------------------------------------- */
/* What lt_create derives from the calibration, exposed so that it can be checked against independent
 * generators (tests/test_tables_independent.py): the cv::remap fixed-point maps -- xy: (sx, sy) int16 pairs,
 * frac: fy*32 + fx -- of cv2.warpPerspective (:834; warp_h*warp_w entries) and of cv2.undistort (:832; rows
 * [row0,row1) x img_w entries), the camera rows the warp reads, the RGB2LAB tables (:208) and the half-widths
 * of getStructuringElement(MORPH_ELLIPSE, (k,k)) (:203-205). */
int  lt_calib_source_rows(const lt_calib* calib, int* row0, int* row1);
int  lt_calib_warp_table(const lt_calib* calib, int16_t* xy, uint16_t* frac);
int  lt_calib_undistort_table(const lt_calib* calib, int row0, int row1, int16_t* xy, uint16_t* frac);
int  lt_calib_lab_tables(uint16_t* gamma256, uint16_t* cbrt3072, int32_t* coeffs9);
int  lt_calib_ellipse(int k, int32_t* halfwidths /* k */, int* taps);

/* ---- multi-GPU: the gather of the lane records ------------------------------------------------------ */
/* Independent frames shard over the GPUs of one node by contiguous index blocks with no data-path exchange
 * (SURVEY 8(e)); the only collective is this all-gather of the 64-byte lane records, run on RCCL (librccl.so is
 * opened by lt_gather_init, so single-GPU users never load it).  The reference has no counterpart: it is one
 * Python process (process_video.py:41-44).  One lt_gather per rank = per process = per GPU, bound to the
 * context whose records it moves.
 *
 * lt_gather_init: rank 0 creates the RCCL id and publishes it at id_path (temporary name + rename); the other
 * ranks wait up to timeout_s (<= 0: 120 s) for that file.  Every rank of the job must call it (collective). */
typedef struct lt_gather lt_gather;
int  lt_gather_init(lt_ctx* ctx, int rank, int world, const char* id_path, int timeout_s, lt_gather** out);
int  lt_gather_world(lt_gather* g, int* rank, int* world);
/* staging capacity: records per rank (e.g. steps * frames per step); identical on every rank */
int  lt_gather_reserve(lt_gather* g, int records_per_rank);
/* Copy the records of context slots [first_slot, first_slot + n) to position `at` of the send buffer,
 * enqueued behind the slots' searches on the context's streams -- no host wait. */
int  lt_gather_stage(lt_gather* g, int first_slot, int n, int at);
/* ONE ncclAllGather of the first n_records staged records of every rank (the same n_records everywhere; pad
 * uneven shards), after the context's streams have drained the staged copies.  out_host receives
 * world * n_records records, rank-major.  Collective; synchronises. */
int  lt_gather_records(lt_gather* g, int n_records, lt_lane_record* out_host);
/* All-gather of `bytes` host bytes per rank (timings, checksums): out holds world * bytes.  Collective. */
int  lt_gather_host(lt_gather* g, const void* in, size_t bytes, void* out);
/* lt_sync of the bound context, then a collective round trip: no rank returns before every rank has arrived */
int  lt_gather_barrier(lt_gather* g);
void lt_gather_destroy(lt_gather* g);

/* ---- measurement ------------------------------------------------------------------------------ */
/* hipEvent pair on the context's stream */
int  lt_timer_start(lt_ctx* ctx);
int  lt_timer_stop(lt_ctx* ctx, float* ms);
/* when enabled, every *_run records a hipEvent after each kernel; lt_stage_ms returns the
 * accumulated milliseconds and launch counts per stage since the last lt_stage_reset */
int  lt_set_stage_timing(lt_ctx* ctx, int enabled);
int  lt_stage_reset(lt_ctx* ctx);
int  lt_stage_ms(lt_ctx* ctx, float* ms, int32_t* launches, int n);
const char* lt_stage_name(int stage);
/* Which kernels evaluated the bilateral thresholds in the context's last 'bilateral' lt_mask_run / lt_filter_run:
 * 1 = the long-walk kernels (window sizes 15 / 20 / 35, width a multiple of 4; with the greenery mask -- mask_noise,
 * lane_tracker.py:221-231 -- when its window is 65), 0 = the tile kernel, -1 = none yet; LT_NO_CONTEXT for a null
 * context (distinct from every status and from "none yet").  Both give identical masks; tests use this to know which
 * one they have exercised. */
#define LT_NO_CONTEXT (-2147483647 - 1)
int  lt_last_threshold_path(lt_ctx* ctx);
/* The same for the last 'neighborhood' call (cv2.adaptiveThreshold, lane_tracker.py:217-218): 1 = running box sums
 * (odd windows up to 63, width a multiple of 4, no greenery mask), 0 = the per-pixel window kernel, -1 = none yet. */
int  lt_last_adaptive_path(lt_ctx* ctx);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
