#!/usr/bin/env python3
"""Headline benchmark: frames/s of the per-frame hot path on synthetic 1280x720 frames, plus the
achieved HBM GB/s of the warp+threshold stage against the MI355X memory roofline.

  python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher: this process stays off the GPU and starts N rank processes itself (one per GPU).
Under `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` the ranks are already there
(RANK / LOCAL_RANK / WORLD_SIZE in the environment).  Either way the ranks talk through `lt_gather_*` of the C ABI
(RCCL over xGMI); PyTorch is not imported.  A rank count that does not match --gpus, or more ranks than visible
GPUs, is an error -- never a silent single-GPU run.

One step = one pass of the hot path over the rank's frames, resident in HBM before the timed region
(undistort -> warp -> filter_lane_points -> sliding_window_search (26 levels) -> fit_poly):
  default        BASELINE config 3: 256 independent frames per GPU per step           ("scaling": "weak")
  --frames 4096  BASELINE config 4: a 4096-frame stream sharded over the GPUs          ("scaling": "strong")
No data-path collective; every step's 64-byte lane records are staged device to device and the timed region ends
with ONE RCCL all-gather of all of them.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_COPY_CEILING_GBS = 6290.0  # measured float4-copy ceiling, same guide
MASK_STAGES = ["undistort_rows", "warp_split", "erode_r29", "tophat_r29", "erode_b55", "tophat_b55", "threshold",
               "merge", "open5"]
PROFILE_TAGS = ("r06", "r05")   # profiles/<tag>_traffic.json, <tag>_kernel_stats_summary.json: the committed counter run of this code (the newest that exists)
PROFILE_TAG = next((t for t in PROFILE_TAGS if os.path.exists(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", t + "_traffic.json"))), PROFILE_TAGS[-1])
CALIB_TAG = "r02"              # profiles/<tag>_valu_issue.json, <tag>_fetch_calib.json: the issue-rate / counter calibrations (hardware facts)


from lane_tracker_amd.hostcpu import cpu_quota, usable_cpus      # (what the cgroup grants; the GPU boxes show 256 CPUs and grant 16)


def _render_one(i):
    from lane_tracker_amd import synth
    global _RENDERER
    try:
        r = _RENDERER
    except NameError:
        r = _RENDERER = synth.SceneRenderer()
    return r.render(i)[0]


def render_frames(indices):
    """Fresh synthetic scene per frame index (lane_tracker_amd/synth.py), rendered on the host cores BEFORE this
    process touches the GPU (worker processes are forked)."""
    indices = list(indices)
    cpus = usable_cpus()                 # not the 256 the box shows: worker processes beyond the quota only add set-up cost
    world = int(os.environ.get("WORLD_SIZE", "1"))
    workers = max(1, min(32, cpus // max(world, 1), len(indices)))
    if workers == 1:
        return np.stack([_render_one(i) for i in indices], 0)
    from concurrent.futures import ProcessPoolExecutor
    with ProcessPoolExecutor(workers) as ex:
        return np.stack(list(ex.map(_render_one, indices, chunksize=max(1, len(indices) // (workers * 4)))), 0)


def _render_stream_one(args):
    from lane_tracker_amd import calib, synth
    scale, sd, lc, rc, ph = args
    global _STREAM_RENDERERS
    try:
        cache = _STREAM_RENDERERS
    except NameError:
        cache = _STREAM_RENDERERS = {}
    if scale not in cache:
        cache[scale] = synth.SceneRenderer(calib.reference_calibration() if scale == 1.0 else calib.scaled_calibration(scale))
    return cache[scale].render(sd, lc, rc, dashed_phase=ph)[0]


def render_streams(n_base=96):
    """One drifting-lane stream per camera size (1280x720, and 1920x1080 = BASELINE config 5), rendered before the GPU is
    touched: n_base DISTINCT frames each; the stream leg plays them forwards and backwards (smooth at the turning points)
    into separately allocated windows."""
    from concurrent.futures import ProcessPoolExecutor
    from lane_tracker_amd import synth
    jobs = [(scale,) + prm for scale in (1.0, 1.5) for prm in synth.stream_lane_params(n_base, seed=5)]
    with ProcessPoolExecutor(max(1, min(32, usable_cpus(), len(jobs)))) as ex:
        got = list(ex.map(_render_stream_one, jobs, chunksize=2))
    return {"1280x720": np.stack(got[:n_base], 0), "1920x1080": np.stack(got[n_base:], 0)}


def stream_windows(base, window, count):
    """`count` consecutive windows of the endless forwards / backwards playback of `base`, each one a separately allocated
    pageable array whose pages this call touches for the first time (what a decoder hands over: memory the runtime has
    never pinned or staged)."""
    from concurrent.futures import ThreadPoolExecutor
    period = np.concatenate([np.arange(len(base)), np.arange(len(base))[::-1]])

    def one(w):
        idx = period[(np.arange(window) + w * window) % len(period)]
        return np.take(base, idx, axis=0)                 # a fresh array; NumPy releases the GIL while it copies
    with ThreadPoolExecutor(min(count, max(1, usable_cpus() // 2))) as ex:
        return list(ex.map(one, range(count)))


def _call_trace(tracker):
    """LT_BENCH_TRACE=1: wall time of the driving thread per library call of `tracker` (and of the page-locked pool) -> a
    function returning {call: [count, ms]} sorted by time (tools/cold_start.py has the finer tool)."""
    import collections
    from lane_tracker_amd import _native
    acc = collections.defaultdict(lambda: [0, 0.0])

    def timed(obj, name):
        fn = getattr(obj, name)

        def w(*a, **k):
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                acc[name][0] += 1
                acc[name][1] += (time.perf_counter() - t0) * 1e3
        setattr(obj, name, w)
    ctx = tracker._ctx
    for m in ("reserve", "sync", "upload_frame_rows_async", "upload_frame_rest", "mask_run", "sws_fit_run", "band_fit_run", "band_fit_chain_run",
              "band_fit_chain_collect", "download_records", "overlay_configure", "overlay_run_packed", "overlay_text", "download_overlay_async",
              "download_overlay_wait"):
        if hasattr(ctx, m):
            timed(ctx, m)
    return lambda: {k: [v[0], round(v[1], 2)] for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])}


def spread(values):
    """min / median / max of a few repetitions (every stream figure of the line is a median of three with its spread)."""
    v = sorted(values)
    return {"min": v[0], "median": v[len(v) // 2], "max": v[-1]}


def stream_leg(streams, window=256, seconds=1.0, nwin=8):
    """The stateful stream (BASELINE config 5) through the drop-in API, host-fed: frames/s of process() frame
    by frame (annotated frame back, as process_video.py uses it), of process_batch() (device-chained searches) and of
    process_stream() over `nwin` DISTINCT windows -- the first pass over them (pages the runtime has never seen) and a second
    pass, separately."""
    from lane_tracker_amd import _native, calib, settings
    from lane_tracker_amd.lane_tracker import LaneTracker
    out = {}
    for name, base in streams.items():
        cal = calib.reference_calibration() if name == "1280x720" else calib.scaled_calibration(1.5)
        wins = stream_windows(base, window, nwin)
        frames = wins[0]
        lt = LaneTracker(**cal)
        try:
            for f in frames[:32]:        # (page-locked output blocks, copy threads, the plot rows on the device: first uses)
                lt.process(f)
            chunks, k, per_frame = [], 0, []
            for _ in range(3):           # three stretches, the median: a stretch of 0.15 s is 600 frames, and one stall of the host shows
                t0, k0 = time.perf_counter(), k
                t_prev = t0
                while t_prev - t0 < seconds * 0.15:
                    lt.process(frames[32 + k % (window - 32)])
                    k += 1
                    t_now = time.perf_counter()
                    per_frame.append(t_now - t_prev)
                    t_prev = t_now
                chunks.append(round((k - k0) / (t_prev - t0), 1))
            per_frame.sort()
            pct = lambda q: round(per_frame[min(int(q * len(per_frame)), len(per_frame) - 1)] * 1e6, 1)
            res = {"process_fps": sorted(chunks)[1], "process_fps_stretches": chunks,
                   # (process_fps is frames per wall time, i.e. the MEAN frame; the frame itself:)
                   "process_frame_us": {"median": pct(0.5), "p10": pct(0.1), "p90": pct(0.9), "p99": pct(0.99), "frames": len(per_frame)},
                   # how the frame's rows reached the device: stored through the PCIe aperture by the calling thread (a large-BAR box,
                   # at most 1.5 MB per frame: lt_set_direct_upload) or copied by the engine
                   "process_rows_through_the_aperture": bool(lt._ctx.direct_upload_count() > 0)}
            # ... and with the caller's frames in page-locked memory (lt_host_alloc): measures whether the upload's cost is a staging
            # copy the caller could spare the runtime (it is not: within a few % of process_fps).  Not the reference's
            # call pattern (moviepy hands over ordinary arrays): a separate key, never `process_fps`.
            try:
                pin = _native.pinned_empty((64,) + frames.shape[1:])
                pin[...] = frames[32:96]
                for f in pin[:8]:
                    lt.process(f)
                t0, k0 = time.perf_counter(), 0
                while time.perf_counter() - t0 < seconds * 0.15:
                    lt.process(pin[k0 % 64])
                    k0 += 1
                res["process_page_locked_input_fps"] = round(k0 / (time.perf_counter() - t0), 1)
                del pin
            except Exception as e:
                res["process_page_locked_input_fps"] = repr(e)
            for key, ann in (("process_batch_fps", False), ("process_batch_annotated_fps", True)):
                lt.process_batch(frames, annotate=ann)
                reps, k = [], 0
                for _ in range(3):
                    t0, k0 = time.perf_counter(), k
                    while time.perf_counter() - t0 < seconds * 0.07 or k == k0:
                        lt.process_batch(wins[1 + k % (nwin - 1)], annotate=ann)
                        k += 1
                    reps.append(round((k - k0) * window / (time.perf_counter() - t0), 1))
                res[key + "_passes"] = spread(reps)
                res[key] = res[key + "_passes"]["median"]

            def stream_rate(ws, ann, tracker=lt, first=None, **kw):
                t0 = time.perf_counter()
                for _ in tracker.process_stream(ws, annotate=ann, **kw):
                    if first is not None and not first:
                        first.append((time.perf_counter() - t0) * 1e3)
                return round(len(ws) * window / (time.perf_counter() - t0), 1)

            def first_passes(ann, trackers=3):
                """The first pass of a stream -- what a single video pays (process_video.py:41-44) -- from `trackers` FRESH trackers,
                each over freshly allocated windows the runtime has never seen: frames/s and the time to the first window, after
                `LaneTracker.warm(window, annotate)` (its own time beside them: the slot regions, the search / chain / presentation
                buffers, the output pool) -- and once more from a tracker that was NOT warmed, which does that work inside the pass."""
                rates, ttfw, warm_ms = [], [], []
                reuse = bool(os.environ.get("LT_BENCH_OLD_FIRST"))     # diagnosis: round 4's sequence (the tracker of the legs above)
                cold_rate = None
                for k in range(1 if reuse else trackers + 1):
                    ws = stream_windows(base, window, nwin)
                    fresh = lt if reuse else LaneTracker(**cal)
                    trace = _call_trace(fresh) if os.environ.get("LT_BENCH_TRACE") else None
                    try:
                        first = []
                        if k < trackers and not reuse:
                            warm_ms.append(round(fresh.warm(window, ann) * 1e3, 1))
                        r = stream_rate(ws, ann, fresh, first)
                        if k < trackers:
                            rates.append(r)
                            ttfw.append(round(first[0], 2))
                        else:
                            cold_rate = {"frames_per_s": r, "time_to_first_window_ms": round(first[0], 2)}
                        if trace is not None:
                            print("LT_BENCH_TRACE %s annotate=%s %.1f frames/s ttfw %.1f ms: %s" % (name, ann, r, first[0], trace()), file=sys.stderr)
                    finally:
                        if not reuse:
                            fresh.close()
                    del ws
                mid = lambda v: sorted(v)[len(v) // 2]
                return {"frames_per_s": {"min": min(rates), "median": mid(rates), "max": max(rates)},
                        "time_to_first_window_ms": {"min": min(ttfw), "median": mid(ttfw), "max": max(ttfw)},
                        "warm_ms": {"min": min(warm_ms), "median": mid(warm_ms), "max": max(warm_ms)} if warm_ms else None,
                        "fresh_trackers": len(rates), "not_warmed": cold_rate}
            # consecutive windows of one video: process_stream keeps the device busy across window boundaries.
            fp_plain = first_passes(False)
            res["process_stream_first_pass"] = fp_plain
            res["process_stream_first_pass_fps"] = fp_plain["frames_per_s"]["median"]           # (after LaneTracker.warm(); as in round 5)
            res["process_stream_first_pass_not_warmed_fps"] = fp_plain["not_warmed"]["frames_per_s"]   # what one short video pays (round 4's meaning of the key above)
            cold = stream_windows(base, window, nwin)
            stream_rate(cold, False)
            res["process_stream_passes"] = spread([stream_rate(cold + cold, False) for _ in range(3)])      # 4096 frames per pass
            res["process_stream_fps"] = res["process_stream_passes"]["median"]
            # ... and with every annotated frame rendered and copied back (what process_video.py consumes)
            fp_ann = first_passes(True)
            res["process_stream_annotated_first_pass"] = fp_ann
            res["process_stream_annotated_first_pass_fps"] = fp_ann["frames_per_s"]["median"]
            res["process_stream_annotated_first_pass_not_warmed_fps"] = fp_ann["not_warmed"]["frames_per_s"]
            # the later passes of the LONG-LIVED tracker (the one that ran process(), process_batch and the plain stream above): what a
            # deployed process does all day.  Three passes, median with the spread; the device cache's traffic with the driver beside
            # them (an eviction = a wipe by the driver = half-rate downloads for the next half second: NOTES_r06 E.1)
            cc0 = _native.device_cache_counters()
            stream_rate(cold, True)
            cs0, t_ann = _native.host_copy_stats(), time.perf_counter()
            res["process_stream_annotated_passes"] = spread([stream_rate(cold, True) for _ in range(3)])
            res["process_stream_annotated_fps"] = res["process_stream_annotated_passes"]["median"]
            cs1, t_ann = _native.host_copy_stats(), time.perf_counter() - t_ann
            cc1 = _native.device_cache_counters()
            res["copy_threads_busy_share_annotated_stream"] = round((cs1["busy_s"] - cs0["busy_s"]) / max(t_ann * cs1["threads"], 1e-9), 3)
            res["device_cache_during_annotated_passes"] = {k: cc1[k] - cc0[k] for k in cc1}
            # ... and drawn into the caller's own windows (annotate="inplace", not in the reference): windows of their own, restored
            # from `cold` before every pass (outside the clock); like the line above a later pass -- memory the runtime has seen
            work = [w.copy() for w in cold]
            rates = []
            for k in range(4):
                r = stream_rate(work, "inplace")
                if k:
                    rates.append(r)
                for w, c0 in zip(work, cold):
                    np.copyto(w, c0)
            res["process_stream_annotated_inplace_passes"] = spread(rates)
            res["process_stream_annotated_inplace_fps"] = res["process_stream_annotated_inplace_passes"]["median"]
            del work
            res["annotated_frames_came_back_by"] = lt._ctx.download_stats()
            rows = lt._present_rows() if lt.host_copies_rows else None
            res["annotated_frames_travel_as"] = ("whole frames" if rows is None else
                                                 ("strips: rows %s of %d (the rows the lane can reach) drawn on the device, back through page-locked staging "
                                                  "blocks into ordinary memory; the other rows from the caller's window and the text lines by %d host threads"
                                                  % (list(rows[4][2]), cal["img_size"][1], _native.host_copy_stats()["threads"])) if rows[4] is not None else
                                                 "row runs %s of %d rows (text lines, rows the lane can reach: copy kernel); the other rows are copied "
                                                 "from the caller's window by host threads" % (rows[2], cal["img_size"][1]))
            res["success_ratio"] = round(lt.get_success_ratio()[0], 4)
            # ... and with outages: every 64th frame starts 16 frames of noise / flat grey / black (tools/outage_profile.py)
            broken = cold[0].copy()
            for j, s0 in enumerate(range(40, window, 64)):
                for i in range(s0, min(window, s0 + 16)):
                    broken[i] = (np.random.default_rng(4000 + i).integers(0, 256, broken[i].shape, dtype=np.uint8) if j % 3 == 0
                                 else (128 if j % 3 == 1 else 0))
            list(lt.process_stream([broken] * 2, annotate=False))
            res["process_stream_outages_passes"] = spread([stream_rate([broken] * 4, False) for _ in range(3)])
            res["process_stream_outages_fps"] = res["process_stream_outages_passes"]["median"]
            if name == "1280x720":       # the author's Demo 1 settings (tracker_settings.md:1-33: the greenery mask) on the same stream
                lt1 = LaneTracker(**cal)
                try:
                    kw = settings.apply(lt1, settings.DEMO_1)
                    stream_rate(cold[:2], False, lt1, **kw)
                    res["process_stream_demo1_passes"] = spread([stream_rate(cold + cold, False, lt1, **kw) for _ in range(3)])   # 4096 frames, as process_stream_fps
                    res["process_stream_demo1_fps"] = res["process_stream_demo1_passes"]["median"]
                    res["demo1_success_ratio"] = round(lt1.get_success_ratio()[0], 4)
                finally:
                    lt1.close()
            out[name] = res
        finally:
            lt.close()
        del wins, cold
    out["window"] = window
    out["distinct_windows"] = nwin
    out["note"] = ("one stateful stream, host-fed (pageable NumPy frames in, PCIe included), %d distinct rendered frames per size played "
                   "forwards and backwards into %d separately allocated windows of %d frames: process() = one frame per call, annotated frame "
                   "returned; process_batch() = one window per call, searches chained on the device (lt_band_fit_chain_run), check_validity / "
                   "history on the host; process_stream() = the same over consecutive windows, the next windows' uploads and masks under the "
                   "current one's searches; *_first_pass = three fresh trackers over pages the runtime has never seen, after LaneTracker.warm() "
                   "(its time in warm_ms; not_warmed / *_not_warmed_fps: a fourth fresh tracker without it), the figure beside it the later passes of the long-lived tracker: median of three, spread in *_passes "
                   "(the *_annotated figures return every annotated frame; only the rows an overlay can touch cross the bus, annotated_frames_travel_as; *_outages: four outages of 16 "
                   "frames per window, handled in speculative groups; *_demo1: settings.DEMO_1, mask_noise = True); success_ratio is that of "
                   "the clean streams; 1920x1080 is BASELINE config 5" % (len(next(iter(streams.values()))), nwin, window))
    return out


FILTER_KEYS = ("filter_type", "ksize_r", "C_r", "ksize_b", "C_b", "mask_noise", "noise_thresh", "ksize_noise", "C_noise")
SEARCH_KEYS = ("window_width", "window_height", "search_range", "mu", "no_success_limit", "start_slice", "ignore_sides",
               "ignore_bottom", "bandwidth", "partial")


def parameter_sets():
    """The parameter sets the reference itself documents: process()'s defaults (lane_tracker.py:876-900), the author's three
    demo configurations (tracker_settings.md:1-111) and the hard-coded second try (lane_tracker.py:1081-1099)."""
    from lane_tracker_amd import settings
    try2 = dict(ksize_r=15, C_r=5, ksize_b=35, C_b=5, filter_type="neighborhood", mask_noise=False, noise_thresh=140,
                ksize_noise=65, C_noise=10, window_width=30, window_height=40, search_range=20, mu=0.1, no_success_limit=50,
                start_slice=0.25, ignore_sides=360, ignore_bottom=30, bandwidth=30, partial=1.0)
    return [("process_defaults", {}), ("demo1", settings.DEMO_1["process"]), ("demo2", settings.DEMO_2["process"]),
            ("demo3", settings.DEMO_3["process"]), ("second_try", try2)]


def settings_leg(ctx, NL, streams, steps=5):
    """Batch frames/s and the mask-stage time per launch for every documented parameter set, on the frames already
    resident in `ctx` (slots [0, NL)): the same step as `value` -- mask chain + sliding-window search + fit -- on one
    copy of the batch."""
    from lane_tracker_amd import _native
    out = {}
    for name, kw in parameter_sets():
        fp = _native.filter_params(**{k: v for k, v in kw.items() if k in FILTER_KEYS})
        sp = _native.search_params(**{k: v for k, v in kw.items() if k in SEARCH_KEYS})
        ctx.set_streams(streams)
        for _ in range(2):
            ctx.mask_run(NL, fp)
            ctx.sws_fit_run(NL, sp)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            ctx.mask_run(NL, fp)
            ctx.sws_fit_run(NL, sp)
        ctx.sync()
        fps = NL * steps / (time.perf_counter() - t0)
        path = ctx.last_threshold_path() if kw.get("filter_type", "bilateral") == "bilateral" else ctx.last_adaptive_path()
        rec = ctx.download_records(NL)
        ctx.set_streams(1)
        ctx.set_stage_timing(True)
        ctx.stage_reset()
        for _ in range(3):
            ctx.mask_run(NL, fp)
        ctx.sync()
        st = ctx.stage_ms()
        ctx.set_stage_timing(False)
        out[name] = {"frames_per_s": round(fps, 1), "mask_stage_ms": round(sum(st[k][0] for k in MASK_STAGES) / 3, 4),
                     "threshold_ms": round((st["threshold"][0] + st["merge"][0]) / 3, 4),
                     "walking_threshold_kernels": bool(path == 1),
                     "detected_fraction": round(float(np.mean(rec["detected"])), 4),
                     "kernels_ms": {k: round(st[k][0] / 3, 4) for k in MASK_STAGES if st[k][1]}}
    out["note"] = ("per parameter set: frames_per_s = mask chain + sliding-window search + fit over the %d resident frames, every step on the "
                   "same slots, %d HIP streams; mask_stage_ms = the stage's kernels on one stream between hipEvents; second_try = the "
                   "'neighborhood' filter of lane_tracker.py:1081-1099 (no top-hats)" % (NL, streams))
    return out


def coeff_close(got, want, h=1100, tol=1e-4):
    """north_star tolerance: 1e-4 relative per coefficient with the absolute floor of SURVEY 8(a)."""
    lim = tol * max(1.0, abs(float(want[2])))
    return (abs(got[0] - want[0]) * h * h <= lim and abs(got[1] - want[1]) * h <= lim and abs(got[2] - want[2]) <= lim)


def cpu_baseline(frames, cal, gpu_records, gpu_mask_of, max_seconds=25.0):
    """The oracle (CPU port of the same path) on a bounded sample of the same workload, all host cores -- and, since
    its results are there anyway, the in-run parity check of the GPU records and masks against them."""
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle as O
    oc = O.make_calib(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0])
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    quota = cpu_quota()
    cores = max(1, min(avail, int(quota + 0.999)) if quota else avail)   # threads actually used: one frame per thread
    try:
        # The port allocates its planes per frame; with glibc's defaults every megabyte-sized block is an mmap / munmap pair
        # and 256 threads serialise on the process's mmap lock.  Keep freed blocks in the per-thread arenas instead
        # (M_MMAP_THRESHOLD = -3, M_TRIM_THRESHOLD = -1): 2.2x faster on 8 threads, far more on 256.
        import ctypes
        libc = ctypes.CDLL("libc.so.6")
        libc.mallopt(-3, 1 << 30)
        libc.mallopt(-1, 1 << 30)
    except Exception:
        pass
    O.frame_sws_fit(oc, frames[0])                      # warms the per-calibration tables
    t0 = time.perf_counter()
    O.frame_sws_fit(oc, frames[0])
    one_naive = time.perf_counter() - t0
    t0 = time.perf_counter()
    O.frame_sws_fit(oc, frames[0], fast=True)
    one = time.perf_counter() - t0
    # The timed port: the same restatement with the two thresholds as running sums (O(1) per pixel; lto_*_fast, checked equal
    # to the loops in tests/test_oracle_units.py) -- the loops the parity checker uses cost O(k) per pixel and would make
    # the stated baseline an unfair one.  Sample: every frame of the batch at least once, cycled up to 8 frames per thread
    # (bounded by max_seconds of ideal scaling) so that thread start-up and the slowest straggler do not dominate.
    n = int(max(len(frames), min(8 * cores, (max_seconds * 0.6 * cores) / max(one, 1e-3) * 0.5)))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:               # ctypes releases the GIL
        list(ex.map(lambda f: O.frame_sws_fit(oc, f, fast=True), [frames[i % len(frames)] for i in range(n)]))
    dt = time.perf_counter() - t0
    n_all, n = n, min(n, len(frames))
    # ... and the parity checker proper (the loops) once over the batch: what the GPU results are compared with below
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        want = list(ex.map(lambda f: O.frame_sws_fit(oc, f), [frames[i] for i in range(n)]))
    dt_naive = time.perf_counter() - t0

    # ---- parity: GPU records of the same frames vs the oracle's; masks of a subset bit for bit ----
    bad = []
    for i, w in enumerate(want):
        g = gpu_records[i]
        ok = bool(g["detected"]) == w["detected"] and (int(g["n_left"]), int(g["n_right"])) == (w["n_left"], w["n_right"])
        if ok and w["detected"]:
            ok = coeff_close(g["left_coeffs"], w["coeffs"][0]) and coeff_close(g["right_coeffs"], w["coeffs"][1])
        if not ok:
            bad.append(i)
    n_masks = min(n, 32)
    with ThreadPoolExecutor(cores) as ex:
        crc = list(ex.map(lambda i: zlib.crc32(O.mask_from_frame(oc, frames[i]).tobytes()), range(n_masks)))
    mask_bad = [i for i in range(n_masks) if zlib.crc32(gpu_mask_of(i).tobytes()) != crc[i]]
    parity = {"parity_checked": n, "record_mismatches": len(bad), "masks_checked_bit_exact": n_masks,
              "mask_mismatches": len(mask_bad), "first_bad": (bad + mask_bad)[:4],
              "rule": "detected flag and lane-pixel counts equal, coefficients within 1e-4 (SURVEY 8(a) floor); masks by CRC32"}

    def ms(fn, *args):                                  # one call on one thread
        t = time.perf_counter()
        r = fn(*args)
        return r, round((time.perf_counter() - t) * 1e3, 2)
    f0 = frames[0]
    und, t_und = ms(O.undistort, oc, f0)
    bev, t_warp = ms(O.warp, oc, und)
    R = np.ascontiguousarray(bev[:, :, 0])
    b, t_lab = ms(O.lab_b, bev)
    tr, t_th29 = ms(O.tophat, R, 29)
    tb, t_th55 = ms(O.tophat, b, 55)
    _, t_thr = ms(lambda: (O.bilateral_adaptive_threshold(tr, 15, 8, fast=True), O.bilateral_adaptive_threshold(tb, 35, 5, fast=True)))
    _, t_thr_naive = ms(lambda: (O.bilateral_adaptive_threshold(tr, 15, 8), O.bilateral_adaptive_threshold(tb, 35, 5)))
    mask, _ = ms(O.mask_from_frame, oc, f0)
    _, t_open = ms(O.morph_open, mask, 5)
    (_, _, _), t_sws = ms(lambda: (O.sliding_window_search(mask), None, None))
    stages = {"undistort": t_und, "warp": t_warp, "lab_b": t_lab, "tophat_r29": t_th29, "tophat_b55": t_th55,
              "thresholds": t_thr, "thresholds_as_loops_(parity_checker)": t_thr_naive, "open5": t_open,
              "sliding_window_search+fit": t_sws}
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    out = {"value": round(n_all / dt, 3), "unit": "frames/s", "cores": cores, "kind": "port", "cpu_model": model,
           "nproc": os.cpu_count(), "cpus_in_affinity_mask": avail, "cgroup_cpu_quota": quota, "stage_ms_one_thread": stages,
           "single_thread_frames_per_s": round(1.0 / one, 2),
           "parity_checker_form": {"frames_per_s": round(n / dt_naive, 3), "single_thread_frames_per_s": round(1.0 / one_naive, 2),
                                   "what": "the same port with the two thresholds as the O(k)-per-pixel loops that read like the reference's "
                                           "filter2D kernels: the form the GPU results are checked against, not the baseline"},
           "note": "plain-C restatement written for fidelity (scalar, one frame per thread; thresholds as running sums, top-hats by run "
                   "decomposition), not a tuned CPU path; a stated baseline, never a ratio to quote",
           "sample": "%d passes over frames of the same synthetic batch through oracle/lt_oracle.c (mask + sliding window + fit), one frame "
                     "per thread on %d threads (%d CPUs visible, cgroup quota %s); single-thread %.1f ms/frame = %.1f frames/s, so the "
                     "threaded run is %.1fx one thread" % (n_all, cores, avail, quota, one * 1e3, 1.0 / one, (n_all / dt) * one)}
    try:   # the reference's own NumPy stages (a5-a8), timed in the build container where the reference can be imported
        out["reference_numpy_container"] = json.load(open(os.path.join(ROOT, "profiles", "reference_numpy_timings.json")))
    except Exception:
        pass
    return out, parity


def config1_leg(calls=200):
    """BASELINE config 1: ONE 1280x720 photo (the reference's test_images/test4.jpg, kept losslessly as tests/golden/photo_test4.png)
    through process() with its defaults.  The reference's hard-coded validity limits reject this frame (tests/test_photos.py:
    pinned by the reference's own run), so every call is the whole two-try path -- undistort + warp + 'bilateral' filter +
    sliding windows + fit + check_validity, then the same with the 'neighborhood' filter, then the failure frame -- and leaves the
    tracker in the state it started in (`last_detection > n_reset`: sliding windows again): median of `calls` calls.  Beside it
    the oracle's two tries of the same frame on ONE CPU thread (masks + searches + fits + validity; no drawing), and the first
    frame of a video under the author's Demo 1 limits, which accept the frame (first try valid, lane drawn)."""
    from PIL import Image
    from lane_tracker_amd import calib, settings
    from lane_tracker_amd.lane_tracker import LaneTracker
    from oracle import oracle as O
    path = os.path.join(ROOT, "tests", "golden", "photo_test4.png")
    frame = np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"), np.uint8))
    cal = calib.reference_calibration()
    out = {"frame": "tests/golden/photo_test4.png (= test_images/test4.jpg decoded; SURVEY 8(d))", "calls": calls}
    lt = LaneTracker(**cal)
    try:
        for _ in range(16):
            lt.process(frame)
        ts = []
        for _ in range(calls):
            t0 = time.perf_counter()
            lt.process(frame)
            ts.append(time.perf_counter() - t0)
        ts.sort()
        out["process_defaults"] = {"median_ms": round(ts[len(ts) // 2] * 1e3, 4), "p10_ms": round(ts[len(ts) // 10] * 1e3, 4), "p90_ms": round(ts[len(ts) * 9 // 10] * 1e3, 4),
                                   "frames_per_s": round(1.0 / ts[len(ts) // 2], 1), "valid": bool(lt.valid_lane_lines), "detected": bool(lt.detected_pixels),
                                   "what": "two tries per call (both rejected by check_validity, as in the reference's own run of this frame), failure frame returned"}
    finally:
        lt.close()
    lt = LaneTracker(**cal)
    try:
        kw = settings.apply(lt, settings.DEMO_1)
        fresh = lt.get_state()
        ts = []
        for k in range(16 + min(calls, 100)):
            lt.set_state(fresh)              # the first frame of a video, every time (outside the clock)
            t0 = time.perf_counter()
            lt.process(frame, **kw)
            if k >= 16:
                ts.append(time.perf_counter() - t0)
        ts.sort()
        out["process_demo1_first_frame"] = {"median_ms": round(ts[len(ts) // 2] * 1e3, 4), "frames_per_s": round(1.0 / ts[len(ts) // 2], 1),
                                            "valid": bool(lt.valid_lane_lines),
                                            "what": "settings.DEMO_1 (tracker_settings.md:1-33: greenery mask, wider validity limits): first try valid, lane drawn"}
    finally:
        lt.close()
    oc = O.make_calib(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0])
    tries = [(O.filter_params(), O.search_params()),
             (O.filter_params(filter_type="neighborhood", C_r=5), O.search_params(no_success_limit=50, bandwidth=30))]

    def two_tries(fast):
        ok = []
        for fp, sp in tries:
            r = O.frame_sws_fit(oc, frame, fp, sp, fast=fast)
            ok.append(bool(r["detected"]) and O.check_validity(cal["warped_size"], r["coeffs"][0], r["coeffs"][1]))
        return ok
    two_tries(True)
    reps = []
    for _ in range(5):
        t0 = time.perf_counter()
        verdicts = two_tries(True)
        reps.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    two_tries(False)
    loops = time.perf_counter() - t0
    out["cpu_port_one_thread"] = {"median_ms": round(sorted(reps)[2] * 1e3, 2), "frames_per_s": round(1.0 / sorted(reps)[2], 2), "valid": verdicts,
                                  "parity_checker_form_ms": round(loops * 1e3, 2), "cores": 1, "kind": "port",
                                  "what": "oracle/lt_oracle.c: both tries of the same frame (mask + sliding windows + fit) + check_validity, thresholds as "
                                          "running sums; no drawing.  A stated baseline, never a ratio to quote"}
    return out


def host_fed_overlapped(cal, frames, fp, sp, streams, resident_records, batches=12):
    """Double-buffered host-fed pipeline: two page-locked host buffers, the camera rows of batch k+1 cross PCIe on the
    copy stream while the chain of batch k runs (lt_upload_frame_rows_async).  Returns frames/s, PCIe included."""
    from lane_tracker_amd import _native
    B = len(frames)
    ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"],
                          cal["warp_matrices"][0], device=0, capacity=2 * B)
    try:
        pin = [_native.pinned_empty(frames.shape), _native.pinned_empty(frames.shape)]
        pin[0][...] = frames
        pin[1][...] = frames[::-1]
        ctx.set_streams(streams)

        def run(nb):
            for k in range(nb):
                half = k & 1
                ctx.upload_frame_rows_async(pin[half], first=half * B)
                ctx.mask_run(B, fp, first=half * B)
                ctx.sws_fit_run(B, sp, first=half * B)
            ctx.sync()
        run(2)
        t0 = time.perf_counter()
        run(batches)
        dt = time.perf_counter() - t0
        rec = ctx.download_records(2 * B)
        same = all(rec[i][f].tobytes() == resident_records[i][f].tobytes() and
                   rec[B + i][f].tobytes() == resident_records[B - 1 - i][f].tobytes()
                   for i in range(B) for f in ("left_coeffs", "right_coeffs", "n_left", "n_right", "detected"))
        return {"overlapped_frames_per_s": round(batches * B / dt, 2), "batches": batches, "frames_per_batch": B,
                "records_equal_resident_run": bool(same),
                "how": "two pinned host buffers; rows %d..%d of batch k+1 uploaded on the copy stream under the chain of "
                       "batch k (slots split in two halves of the context)" % tuple(ctx.source_rows())}
    finally:
        ctx.close()


def launcher(a):
    """--gpus N > 1 and no RANK in the environment: start the N ranks.  This process never touches the GPU."""
    from lane_tracker_amd import distributed
    have = distributed.visible_gpu_count()
    if a.gpus > have and not (distributed.shares_devices() and have > 0):
        print("bench.py: --gpus %d requested but %d GPU(s) visible; refusing to report a %d-GPU number from fewer devices"
              % (a.gpus, have, a.gpus), file=sys.stderr)
        return 2
    return distributed.spawn_ranks(a.gpus, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="frames per GPU per step (weak scaling, BASELINE config 3)")
    ap.add_argument("--frames", type=int, default=0,
                    help="total frames of one stream, sharded over the GPUs (strong scaling; 4096 = BASELINE config 4)")
    ap.add_argument("--single-copy", action="store_true", help="do not keep a second resident copy of the batch (no overlapped_batches figure)")
    ap.add_argument("--streams", type=int, default=4, help="HIP streams per context (slot slices overlap each other's stages)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-host-fed", action="store_true")
    ap.add_argument("--no-stream", action="store_true", help="skip the stateful-stream leg (process / process_batch frames/s)")
    ap.add_argument("--no-settings", action="store_true", help="skip the parameter-set leg (defaults, demo 1-3, second try)")
    ap.add_argument("--only-settings", action="store_true", help="development probe: print the parameter-set leg alone")
    a = ap.parse_args()
    if a.gpus < 1 or a.steps < 1 or a.warmup < 0:
        ap.error("--gpus >= 1, --steps >= 1, --warmup >= 0")

    in_rank = "RANK" in os.environ
    if not in_rank and a.gpus > 1:
        sys.exit(launcher(a))

    from lane_tracker_amd import _native, calib, distributed
    rank, local_rank, world = distributed.env_rank()
    if world != a.gpus:
        print("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE); refusing to mislabel the run"
              % (a.gpus, world), file=sys.stderr)
        sys.exit(2)

    cal = calib.reference_calibration()
    strong = a.frames > 0
    if strong:
        lo, hi = distributed.shard_range(a.frames, rank, world)
        if a.frames % world:
            print("bench.py: --frames %d is not a multiple of %d ranks (the one all-gather moves equal blocks)" % (a.frames, world),
                  file=sys.stderr)
            sys.exit(2)
        indices = range(lo, hi)
    else:
        indices = range(rank * a.batch, (rank + 1) * a.batch)
    NL = len(indices)                                   # frames this rank processes per step
    frames = render_frames(indices)                     # before anything initialises the GPU (forked workers)
    streams = render_streams() if (world == 1 and not strong and not a.no_stream and not a.only_settings) else None

    try:
        device = distributed.local_device(local_rank)   # LOCAL_RANK (LT_DEVICE_MODULO: test runs that share a GPU, labelled below)
    except RuntimeError as e:
        print("bench.py: rank %d: %s" % (rank, e), file=sys.stderr)
        sys.exit(2)
    # `value`: every step over the SAME NL resident frames (slots [0, NL)); each step waits in order behind the previous one on
    # its slots.  A second resident copy of the batch (slots [NL, 2 NL)) serves one extra figure outside the timed region,
    # overlapped_batches_frames_per_s: steps alternating between the copies, so that the tail of step k and the head of step
    # k+1 overlap the way consecutive batches of a stream do (rounds 1-3 reported that one as `value`).
    two_copies = NL <= 1024 and not a.single_copy
    ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"],
                          cal["warp_matrices"][0], device=device, capacity=2 * NL if two_copies else NL)
    info = ctx.info()
    t0 = time.perf_counter()
    for c0 in range(0, NL, 256):
        ctx.upload_frames(frames[c0:c0 + 256], first=c0)
    h2d_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    for c0 in range(0, NL, 256):
        ctx.upload_frame_rows(frames[c0:c0 + 256], first=c0)   # what a host-fed pipeline moves: the rows the path reads
    h2d_rows_s = time.perf_counter() - t0
    ctx.set_frame_base(NL, indices[0])
    if two_copies:
        for c0 in range(0, NL, 256):
            ctx.upload_frames(frames[c0:c0 + 256], first=NL + c0)
        ctx.set_frame_base(NL, indices[0], first=NL)
    fp, sp = _native.filter_params(), _native.search_params()
    if a.only_settings:
        print(json.dumps(settings_leg(ctx, NL, a.streams), indent=1))
        ctx.close()
        return

    # ranks: every step's records are staged, stream-ordered and without a host wait, into the gather's send buffer;
    # the timed region ends with ONE RCCL all-gather of all of them (the path has no other exchange step)
    nsteps = max(a.steps, a.warmup, 1)
    gather = None
    if in_rank:
        gather = distributed.init_gather(ctx)
        gather.reserve(nsteps * NL)

    def step(k=0, alternate=False, stage=True):
        first = NL * (k & 1) if two_copies and alternate else 0
        ctx.mask_run(NL, fp, first=first)
        ctx.sws_fit_run(NL, sp, first=first)
        if gather is not None and stage:
            gather.stage(NL, at=k * NL, first=first)

    marks = {}
    gathered = [None]

    def fence(n_gather=0):
        ctx.sync()
        marks["drained"] = time.perf_counter()
        if gather is not None:
            if n_gather:
                gathered[0] = gather.records(n_gather * NL)
            marks["gathered"] = time.perf_counter()
            gather.barrier()
        marks["fenced"] = time.perf_counter()

    ctx.set_streams(a.streams)
    for k in range(a.warmup):
        step(k)
    fence(a.warmup)
    t0 = time.perf_counter()
    for k in range(a.steps):
        step(k)
    fence(a.steps)
    dt = time.perf_counter() - t0
    if os.environ.get("LT_BENCH_VERBOSE") and rank == 0:
        print("timed region: drained %.3f ms, +gather %.3f ms, +barrier %.3f ms" % (
            (marks["drained"] - t0) * 1e3, (marks.get("gathered", marks["drained"]) - marks["drained"]) * 1e3,
            (marks["fenced"] - marks.get("gathered", marks["drained"])) * 1e3), file=sys.stderr)

    overlapped_fps = None
    if two_copies and not in_rank:     # not part of the timed region: the same steps alternating between two resident copies of
        for k in range(2):             # the batch, so that consecutive steps overlap the way consecutive batches of a stream do
            step(k, True, False)
        ctx.sync()
        t1 = time.perf_counter()
        for k in range(a.steps):
            step(k, True, False)
        ctx.sync()
        overlapped_fps = NL * a.steps / (time.perf_counter() - t1)

    # Per-kernel durations for the roofline: the same steps again on ONE stream with a hipEvent pair
    # around every kernel (with several streams the kernels of different slices overlap, so their
    # individual durations are not separable).  These agree with the rocprofv3 --stats averages.
    ctx.set_streams(1)
    ctx.set_stage_timing(True)
    ctx.stage_reset()
    KS = max(1, min(a.steps, 5))
    for _ in range(KS):
        ctx.mask_run(NL, fp)
        ctx.sws_fit_run(NL, sp)
    ctx.sync()
    stages = ctx.stage_ms()
    ctx.set_stage_timing(False)

    my_records = ctx.download_records(NL)
    if gather is not None:
        dt = float(gather.host(np.array([dt], np.float64)).max())           # MAX over ranks
        sizes = gather.host(np.array([NL], np.int64)).reshape(-1)
        rec_all = gathered[0].reshape(world, a.steps, NL)[:, a.steps - 1, :].reshape(-1)   # last timed step of every rank
        assert rec_all[rank * NL:(rank + 1) * NL].tobytes() == my_records.tobytes(), "gathered records differ from the rank's own"
        total_frames = int(sizes.sum())
        first_frame = indices[0] - rank * NL              # rank-major = frame order: record i of the job is frame first_frame + i
        assert list(rec_all["frame"]) == list(range(first_frame, first_frame + total_frames)), "gathered records are not rank-major"
    else:
        rec_all, total_frames = my_records, NL
    if gather is not None:            # the collective part is over: release the communicator before rank 0 reports
        gather.barrier()
        gather.close()

    if rank == 0:
        K = a.steps
        ms_step = dt / K * 1e3
        value = total_frames * K / dt
        mask_ms = sum(stages[s][0] for s in MASK_STAGES) / KS              # per launch of NL frames, serial pass
        search_ms = stages["sws_fit"][0] / KS
        alg = info.alg_bytes_mask * NL                                     # algorithmic bytes of the stage per launch
        achieved = alg / (mask_ms * 1e-3) / 1e9 if mask_ms > 0 else 0.0
        dom = max(MASK_STAGES, key=lambda s: stages[s][0])
        # HBM-side traffic and VALU instruction counts come from a committed rocprofv3 --pmc run of this same command
        # (profiles/<tag>_traffic.json); they are labelled as such and only attached when the launch shape matches.
        from_profile = None
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", PROFILE_TAG + "_traffic.json")))
            if tj["frames_per_launch"] == NL:
                traffic = int(tj["mask_stage_traffic_bytes_per_launch"])
                vi = json.load(open(os.path.join(ROOT, "profiles", CALIB_TAG + "_valu_issue.json")))
                peak_rate = float(vi["peak_wave_insts_per_cycle_per_simd"])
                lane_ops = float(tj["mask_stage_valu_wave_insts_per_launch"]) * 64.0
                peak = info.cu_count * 4 * peak_rate * 64.0 * 2.4e9       # CUs x SIMDs x wave-insts/clk x 64 lanes x 2.4 GHz
                # per kernel: the committed instruction / byte counts over THIS run's kernel times
                stage_kernels = {"undistort_rows": ["k_undistort_rows"], "warp_split": ["k_warp_split4"],
                                 "erode_r29": ["SE29, false"], "tophat_r29": ["SE29, true"], "erode_b55": ["SE55, false"],
                                 "tophat_b55": ["SE55, true"], "threshold": ["k_bilateral_walk", "k_bilateral_tile"],
                                 "open5": ["k_merge_open5", "k_erode5_bits", "k_dilate5_bits", "k_or4_bits"]}
                per_kernel = {}
                for st, keys in stage_kernels.items():
                    ms = stages.get(st, (0.0, 0))[0] / KS
                    ks = [v for n, v in tj.get("per_kernel", {}).items() if any(k in n for k in keys)]
                    if ms <= 0 or not ks:
                        continue
                    insts = sum(v["valu_wave_insts"] for v in ks)
                    byts = sum(v["fetch_bytes"] + v["write_bytes"] for v in ks)
                    per_kernel[st] = {"ms": round(ms, 4),
                                      "valu_issue_frac": round(insts / (ms * 1e-3 * 2.4e9 * info.cu_count * 4 * peak_rate), 3),
                                      "hbm_side_GBs": round(byts / (ms * 1e-3) / 1e9, 1)}
                rocprof = None
                try:
                    ks = json.load(open(os.path.join(ROOT, "profiles", PROFILE_TAG + "_kernel_stats_summary.json")))
                    rocprof = {"stage_ms": ks["mask_stage_ms"], "this_run_over_rocprof": round(mask_ms / ks["mask_stage_ms"], 3),
                               "note": "sum of the rocprofv3 --kernel-trace average durations of the same kernels in the committed "
                                       "profile run (kernels under the tracer run slower than between hipEvents)"}
                except Exception:
                    pass
                from_profile = {
                    "from_profile": PROFILE_TAG, "commit": tj.get("commit"), "rocprof_kernel_trace": rocprof,
                    "source": tj.get("source"), "per_kernel": per_kernel,
                    "traffic_bytes_per_launch": traffic, "traffic_over_algorithmic": round(traffic / float(alg), 2),
                    "fetch_write_calibration": tj.get("calibration"),
                    "traffic_over_compulsory": round(traffic / float(tj["mask_stage_compulsory_bytes_per_launch"]), 2) if tj.get("mask_stage_compulsory_bytes_per_launch") else None,
                    "lds_pipe": ({"busy_frac": round(float(tj["mask_stage_lds_idx_active_cycles_per_launch"]) / (mask_ms * 1e-3 * 2.4e9 * info.cu_count), 4),
                                  "what": "SQ_LDS_IDX_ACTIVE cycles of the stage's kernels (summed over the CUs) / (CUs x this run's stage time x 2.4 GHz)"}
                                 if tj.get("mask_stage_lds_idx_active_cycles_per_launch") else None),
                    "valu_issue": {"achieved": round(lane_ops / (mask_ms * 1e-3) / 1e12, 3), "peak": round(peak / 1e12, 3),
                                   "unit": "T lane-ops/s", "frac": round(lane_ops / (mask_ms * 1e-3) / peak, 4),
                                   "peak_from": "profiles/%s_valu_issue.json: %s wave64 instructions per cycle per SIMD measured for "
                                                "the packed-16 / integer instructions of these kernels" % (CALIB_TAG, peak_rate)}}
        except Exception:
            pass
        workload = ("BASELINE config 4: %d-frame synthetic stream sharded over %d GPU(s) (%d frames on rank 0), HBM-resident"
                    % (a.frames, world, NL)) if strong else \
                   ("BASELINE config 3: batch of %d synthetic lane-like 1280x720 frames per GPU, HBM-resident" % NL)
        out = {
            "metric": "frames/sec at 1280x720 (end-to-end hot path: undistort+warp+filter_lane_points+sliding_window_search+fit_poly)",
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_step, 4), "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload + "; warp+filter_lane_points chain + sliding_window_search (26 levels) + fit_poly",
                       "frames_per_step_all_gpus": total_frames, "frames_per_gpu_per_step": NL, "bev": "1080x1100",
                       "parallelism": "frames sharded x%d, one process per GPU" % world,
                       "collective": ("one RCCL all-gather (lt_gather_records) of %d x 64-byte records per rank at the end of the "
                                      "timed region" % (a.steps * NL)) if gather is not None else "none (single process)",
                       "gathered_records_checked": int(len(rec_all)) if gather is not None else 0,
                       "ranks_share_devices": bool(distributed.shares_devices()) if world > 1 else False,
                       "rank_environment": {distributed.IPC_ENV[0]: os.environ.get(distributed.IPC_ENV[0])} if world > 1 else None,
                       "streams_per_gpu": a.streams, "resident_copies_of_the_batch_in_the_timed_region": 1,
                       "detected_fraction": round(float(np.mean(rec_all["detected"])), 4)},
            "roofline": {"bound": "hbm", "binding_unit": "valu+lds",
                         "binding_unit_fracs": ({"valu_issue_frac": from_profile["valu_issue"]["frac"],
                                                 "lds_pipe_busy_frac": (from_profile.get("lds_pipe") or {}).get("busy_frac"),
                                                 "from": "the counters of profile.from_profile over this run's stage time"} if from_profile else None),
                         "kernel": "warp+threshold stage (%d kernels: %s)" % (len(MASK_STAGES), ",".join(MASK_STAGES)),
                         "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic,
                         "frac_of_copy_ceiling": round(achieved / HBM_COPY_CEILING_GBS, 6),
                         "alg_bytes_per_frame": int(info.alg_bytes_mask), "frames_per_launch": NL,
                         "stage_ms_per_launch": round(mask_ms, 4), "dominant_kernel": dom,
                         "profile": from_profile,
                         "note": "achieved = algorithmic bytes / hipEvent time of the stage's kernels, measured in this run; "
                                 "`traffic` and `profile` are NOT measured in this run: they are the PMC counters of the committed "
                                 "rocprofv3 run named in profile.from_profile (null when the launch shape differs). `bound` names the roofline "
                                 "the METRIC asks for (HBM); what binds the stage is binding_unit: the integer-VALU issue port and the CUs' "
                                 "LDS pipes together (binding_unit_fracs), see DESIGN.md"},
            "kernels_ms_per_step": {k: round(v[0] / KS, 4) for k, v in stages.items() if v[1]},
            "timing_note": "value / ms_per_step: %d steps over the same resident frames on %d HIP streams per GPU (slot slices overlap: the "
                           "latency-bound search of one slice hides under the mask chain of another)%s. kernels_ms_per_step, roofline and "
                           "search_fit: %d further steps on one stream with hipEvents around every kernel; their sum (%.3f ms) "
                           "is the un-overlapped step" % (a.steps, a.streams,
                                                          "; overlapped_batches_frames_per_s: the same steps alternating between two resident copies "
                                                          "of the batch (what rounds 1-3 reported as value)" if overlapped_fps else "", KS, mask_ms + search_ms),
            "overlapped_batches_frames_per_s": round(overlapped_fps, 2) if overlapped_fps else None,
            "search_fit": {"ms_per_step": round(search_ms, 4),
                           "achieved_GBs": round(info.alg_bytes_search * NL / (search_ms * 1e-3) / 1e9, 3) if search_ms > 0 else None},
            "host_fed": {"h2d_seconds_whole_frames": round(h2d_s, 4),
                         "serial_frames_per_s_whole_frames": round(NL / (h2d_s + dt / K), 2),
                         "h2d_seconds_source_rows_only": round(h2d_rows_s, 4), "source_rows": list(ctx.source_rows()),
                         "serial_frames_per_s_source_rows_only": round(NL / (h2d_rows_s + dt / K), 2)},
            "device": info.device_name.decode(errors="replace"),
        }
        single = world == 1 and not strong
        if single and not a.no_cpu_baseline:
            masks_cache = {}

            def gpu_mask_of(i):
                if i not in masks_cache:
                    masks_cache[i] = ctx.download_masks(1, first=i)[0]
                return masks_cache.pop(i)
            out["cpu_baseline"], out["parity"] = cpu_baseline(frames, cal, rec_all, gpu_mask_of)
        if single and not a.no_settings:
            try:
                out["settings"] = settings_leg(ctx, NL, a.streams)
            except Exception as e:
                out["settings"] = {"error": repr(e)}
        ctx.close()
        ctx = None
        if single and not a.no_host_fed:
            try:
                out["host_fed"].update(host_fed_overlapped(cal, frames, fp, sp, a.streams, rec_all))
            except Exception as e:   # the resident number stands on its own
                out["host_fed"]["overlapped_error"] = repr(e)
        if single and not a.no_cpu_baseline and not a.no_stream:
            try:
                out["config1"] = config1_leg()
            except Exception as e:
                out["config1"] = {"error": repr(e)}
        if single and streams is not None:
            try:
                out["stream"] = stream_leg(streams)
            except Exception as e:
                out["stream"] = {"error": repr(e)}
        print(json.dumps(out))
        sys.stdout.flush()
    if ctx is not None:
        ctx.close()


if __name__ == "__main__":
    main()
