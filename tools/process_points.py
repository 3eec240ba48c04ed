#!/usr/bin/env python3
"""Where inside a LaneTracker.process() call the time goes, as a TIMELINE: entry and exit of a handful of calls in microseconds from
the start of the frame (medians over the frames of bench.py's process() leg).  Seven wrapped calls, two clock reads each: ~2 us per
frame, against the ~25 us of tools/process_trace.py (which wraps every library call and answers "how long", not "when").
  python tools/process_points.py [1280x720|1920x1080] [seconds] [engine] [switch=0|1 ...]
engine: the frame's rows by the copy engine (lt_set_direct_upload(ctx, 0)); switch: a class switch of LaneTracker by name
(fast_tail=0, text_now=0, host_text=0 ...)"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lane_tracker_amd import calib
from lane_tracker_amd.lane_tracker import LaneTracker

size = sys.argv[1] if len(sys.argv) > 1 else "1280x720"
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 1.5
flags = sys.argv[3:]
cal = calib.reference_calibration() if size == "1280x720" else calib.scaled_calibration(1.5)
frames = bench.stream_windows(bench.render_streams(96)[size], 256, 1)[0]
lt = LaneTracker(**cal)
if "engine" in flags:
    lt._ctx.set_direct_upload(False)
for f in flags:
    if "=" in f:                                   # class switches by name: host_text=0, fast_tail=0 ...
        k, v = f.split("=")
        setattr(lt, k, bool(int(v)))
for f in frames[:8]:
    lt.process(f)

now = time.perf_counter
t_frame = [0.0]
marks = {}


def wrap(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def w(*a, **k):
        t0 = now()
        r = fn(*a, **k)
        t1 = now()
        marks.setdefault(label + " in", []).append((t0 - t_frame[0]) * 1e6)
        marks.setdefault(label + " out", []).append((t1 - t_frame[0]) * 1e6)
        return r
    setattr(obj, name, w)


for n in ("upload_frame_rows", "mask_run", "band_fit_run", "download_record", "present_finish"):
    wrap(lt._ctx, n)
for n in ("_prepare_out", "_record_success", "_text_early", "_present"):
    wrap(lt, n)
total = []
t_end = now() + seconds
k = 0
while now() < t_end:
    t_frame[0] = t0 = now()
    lt.process(frames[8 + k % 248])
    total.append((now() - t0) * 1e6)
    k += 1
parity = {}
if "parity" in flags:                          # the same timeline for even and odd frames apart (the two slots / the two output blocks)
    for par in (0, 1):
        parity[str(par)] = {"us_median": round(float(np.median(total[par::2])), 1),
                            "timeline_us": {kk: round(float(np.median(v[par::2])), 1) for kk, v in sorted(marks.items(), key=lambda kv: np.median(kv[1]))
                                            if len(v) == len(total)}}
out = {"size": size, "flags": flags, "frames": k, "by_frame_parity": parity or None, "rows_through_the_aperture": lt._ctx.direct_upload_count() > 0,
       "us_median": round(float(np.median(total)), 1), "us_p10": round(float(np.percentile(total, 10)), 1),
       "timeline_us": {kk: round(float(np.median(v)), 1) for kk, v in sorted(marks.items(), key=lambda kv: np.median(kv[1]))}}
print(json.dumps(out))
lt.close()
