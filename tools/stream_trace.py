#!/usr/bin/env python3
"""Host wall time per context call inside process_batch(annotate=False): which calls block and for how long."""
import collections, json, sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from lane_tracker_amd import _native, calib, synth
from lane_tracker_amd.lane_tracker import LaneTracker

cal = calib.reference_calibration() if len(sys.argv) < 2 else calib.scaled_calibration(1.5)
n = 256
base = synth.stream_lanes(32, seed=5, cal=cal)
frames = np.concatenate([base, base[::-1]] * (n // 64 + 1), 0)[:n].copy()
lt = LaneTracker(**cal)
lt.process_batch(frames, annotate=False)
acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
ctx = lt._ctx
for name in ("upload_frame_rows_async", "mask_run", "sws_fit_run", "band_fit_chain_run", "band_fit_chain_collect", "download_records", "download_pixels",
             "reserve", "sync", "upload_frame_rest"):
    fn = getattr(ctx, name)
    def wrap(fn=fn, name=name):
        def w(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            acc[name] += time.perf_counter() - t0
            cnt[name] += 1
            return r
        return w
    setattr(ctx, name, wrap())
for name in ("_valid_many", "_record_success", "_materialise_pending", "_step"):
    fn = getattr(lt, name)
    def wrap(fn=fn, name=name):
        def w(*a, **k):
            t0 = time.perf_counter()
            r = fn(*a, **k)
            acc[name] += time.perf_counter() - t0
            cnt[name] += 1
            return r
        return w
    setattr(lt, name, wrap())
t0 = time.perf_counter()
lt.process_batch(frames, annotate=False)
total = time.perf_counter() - t0
print(json.dumps({"total_ms": round(total * 1e3, 3), "fps": round(n / total, 1),
                  "calls": {k: [cnt[k], round(v * 1e3, 3)] for k, v in sorted(acc.items(), key=lambda kv: -kv[1])}}))
