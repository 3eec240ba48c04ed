#!/usr/bin/env python3
"""Host wall time per call inside process_batch(annotate=True), window of 256."""
import collections, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import calib, synth
from lane_tracker_amd.lane_tracker import LaneTracker
cal = calib.reference_calibration() if len(sys.argv) < 2 else calib.scaled_calibration(1.5)
n = 256
base = synth.stream_lanes(32, seed=5, cal=cal)
frames = np.concatenate([base, base[::-1]] * (n // 64 + 1), 0)[:n].copy()
lt = LaneTracker(**cal)
if os.environ.get("PINNED"):
    from lane_tracker_amd._native import pinned_empty
    pf = pinned_empty(frames.shape); pf[...] = frames; frames = pf
lt.process_batch(frames)
acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
def wrap(obj, name):
    fn = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); acc[name] += time.perf_counter() - t0; cnt[name] += 1; return r
    setattr(obj, name, w)
for name in ("upload_frame_rows_async", "upload_frame_rest", "mask_run", "band_fit_chain_run", "band_fit_chain_collect", "overlay_run", "overlay_text", "download_overlay", "download_pixels"):
    wrap(lt._ctx, name)
for name in ("_valid_many", "_record_success", "_lane_text", "_render_window", "get_poly_points", "get_curve_radius"):
    wrap(lt, name)
t0 = time.perf_counter()
out = lt.process_batch(frames)
total = time.perf_counter() - t0
print(json.dumps({"total_ms": round(total * 1e3, 2), "fps": round(n / total, 1), "calls": {k: [cnt[k], round(v * 1e3, 2)] for k, v in sorted(acc.items(), key=lambda kv: -kv[1])}}))
