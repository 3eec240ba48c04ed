#!/usr/bin/env python3
"""The spread of LaneTracker.process() frame times: percentiles, mean over median, and whether the slow frames come with a period
(slot parity, every n-th frame) or with the interpreter's garbage collector.   python tools/process_tail_probe.py [size] [frames] [nogc]"""
import gc, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lane_tracker_amd import calib
from lane_tracker_amd.lane_tracker import LaneTracker
size = sys.argv[1] if len(sys.argv) > 1 else "1280x720"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
cal = calib.reference_calibration() if size == "1280x720" else calib.scaled_calibration(1.5)
frames = bench.stream_windows(bench.render_streams(96)[size], 256, 1)[0]
lt = LaneTracker(**cal)
for f in frames[:32]:
    lt.process(f)
if "nogc" in sys.argv[3:]:
    gc.disable()
if "freeze" in sys.argv[3:]:
    gc.collect(); gc.freeze()
gcs = []
gc.callbacks.append(lambda phase, info: gcs.append((phase, info["generation"], time.perf_counter())))
t = np.empty(n)
now = time.perf_counter
cur = {}
def wrap(obj, name):
    fn = getattr(obj, name)
    def w(*a, **k):
        t0 = now(); r = fn(*a, **k); cur[name] = cur.get(name, 0.0) + (now() - t0); return r
    setattr(obj, name, w)
for name in ("upload_frame_rows", "mask_run", "band_fit_run", "sws_fit_run", "download_record", "present_finish", "present_lane_from_fit_async"):
    wrap(lt._ctx, name)
for name in ("_prepare_out", "_text_early", "_tail_fast", "_present"):
    wrap(lt, name)
from lane_tracker_amd import _native as _nat
_pe = _nat.pinned_empty
def _pe_t(*a, **k):
    t0 = now(); r = _pe(*a, **k); cur["pinned_empty"] = cur.get("pinned_empty", 0.0) + (now() - t0); return r
_nat.pinned_empty = _pe_t
stalls = []
for k in range(n):
    cur.clear()
    t0 = now()
    lt.process(frames[32 + k % 224])
    t[k] = now() - t0
    if t[k] > 1e-3 and len(stalls) < 12:
        stalls.append({"frame": k, "ms": round(t[k] * 1e3, 2), "ms_by_call": {kk: round(v * 1e3, 2) for kk, v in cur.items() if v > 2e-4}})
t *= 1e6
med = float(np.median(t))
slow = np.where(t > 1.25 * med)[0]
starts = [x for x in gcs if x[0] == "start"]
out = {"size": size, "frames": n, "flags": sys.argv[3:], "us_mean": round(float(t.mean()), 1), "us_median": round(med, 1),
       "mean_over_median": round(float(t.mean() / med), 4),
       "percentiles_us": {str(p): round(float(np.percentile(t, p)), 1) for p in (1, 10, 25, 50, 75, 90, 95, 99, 99.9)},
       "frames_over_1.25x_median": int(len(slow)), "their_share_of_time": round(float(t[slow].sum() / t.sum()), 4),
       "excess_us_per_frame_from_them": round(float((t[slow] - med).sum() / n), 2),
       "frames_over_1ms": stalls, "frames_over_1ms_count": int((t > 1000).sum()), "their_indices": [int(i) for i in np.where(t > 1000)[0][:40]],
       "gc_runs_by_generation": {g: sum(1 for x in starts if x[1] == g) for g in (0, 1, 2)},
       "mean_us_by_frame_mod_2": [round(float(t[i::2].mean()), 1) for i in range(2)],
       "mean_us_by_frame_mod_224_first_8": [round(float(t[np.arange(n) % 224 == i].mean()), 1) for i in range(8)],
       "gaps_between_slow_frames_most_common": [int(v) for v in np.bincount(np.diff(slow)).argsort()[::-1][:6]] if len(slow) > 2 else []}
print(json.dumps(out))
lt.close()
