#!/usr/bin/env python3
"""Where an annotated stream spends its time (round 5, strips): wall time per pass, the driving thread's time per call, the copy
threads' busy share and their copy rate; plus the raw rate of the copy threads on a window-sized 2-D copy.
  python tools/annot_probe.py [1280x720|1920x1080] [passes]        (LT_COPY_THREADS=n to sweep the thread count)"""
import collections, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
size = sys.argv[1] if len(sys.argv) > 1 else "1280x720"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
base = bench.render_streams(96)[size]
from lane_tracker_amd import _native, calib, lane_tracker as LTM
from lane_tracker_amd.lane_tracker import LaneTracker
cal = calib.reference_calibration() if size == "1280x720" else calib.scaled_calibration(1.5)
acc = collections.defaultdict(lambda: [0, 0.0])
def timed(obj, name, label=None):
    fn = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            e = acc[label or name]; e[0] += 1; e[1] += (time.perf_counter() - t0) * 1e3
    setattr(obj, name, w)
for m in ("upload_frame_rows_async", "upload_frame_rest", "mask_run", "sws_fit_run", "band_fit_chain_run", "band_fit_chain_collect",
          "overlay_run_strip_packed", "strip_download_async", "overlay_run_packed", "overlay_text", "download_overlay_async", "download_overlay_wait"):
    timed(_native.Context, m)
for m in ("host_text_async", "text_bytes", "frames_empty", "pinned_empty", "host_copy_group", "host_copy_group_release", "poly_points"):
    timed(_native, m)
from lane_tracker_amd import stream as STM
timed(STM, "_pack_deferred")
for m in ("_copies_done", "_valid_many", "_record_successes", "_commit_valid_run"):
    timed(LaneTracker, m)
# raw rate of the copy threads
lib = _native.load()
src = np.ones((256,) + base.shape[1:], np.uint8); dst = np.zeros_like(src)
g = _native.host_copy_group()
rates = []
for _ in range(4):
    t0 = time.perf_counter()
    lib.lt_host_copy2d_async_group(g, dst.ctypes.data, src[0].nbytes, src.ctypes.data, src[0].nbytes, src[0].nbytes, 256)
    lib.lt_host_copy_wait_group(g)
    rates.append(src.nbytes / (time.perf_counter() - t0) / 1e9)
print(json.dumps({"copy_threads": _native.host_copy_stats()["threads"], "raw_copy_GBs_window_sized": [round(r, 1) for r in rates]}))
del src, dst
wins = bench.stream_windows(base, 256, 8)
lt = LaneTracker(**cal)
lt.warm(256, True)
for p in range(passes):
    acc.clear()
    s0 = _native.host_copy_stats()
    t0 = time.perf_counter()
    for _ in lt.process_stream(wins, annotate=True):
        pass
    dt = time.perf_counter() - t0
    s1 = _native.host_copy_stats()
    busy = s1["busy_s"] - s0["busy_s"]
    print(json.dumps({"pass": p, "fps": round(2048 / dt, 1), "ms": round(dt * 1e3, 1), "copy_busy_s": round(busy, 4),
                      "copy_threads_busy_share": round(busy / (dt * s1["threads"]), 3), "copied_GB": round((s1["bytes"] - s0["bytes"]) / 1e9, 2),
                      "copy_GBs_while_busy": round((s1["bytes"] - s0["bytes"]) / 1e9 / max(busy / s1["threads"], 1e-9), 1),
                      "main_thread_ms": {k: [v[0], round(v[1], 1)] for k, v in sorted(acc.items(), key=lambda kv: -kv[1][1])}}))
lt.close()
