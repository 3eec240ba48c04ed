#!/usr/bin/env python3
"""The stateful stream under outages: windows of 256 frames in which every 64th frame starts an outage of `length` frames
(noise, flat grey or black, in turn).  frames/s of process_stream(annotate=False) and process() calls per outage length."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import calib, synth
from lane_tracker_amd.lane_tracker import LaneTracker
cal = calib.reference_calibration()
n, W = 256, 4
clean = synth.stream_lanes(n, seed=5, cal=cal)
out = {}
for length in (0, 1, 4, 16, 32):
    frames = clean.copy()
    for k, s in enumerate(range(40, n, 64)):
        for i in range(s, min(n, s + length)):
            frames[i] = synth.frame_uniform(4000 + i) if k % 3 == 0 else (128 if k % 3 == 1 else 0)
    lt = LaneTracker(**cal)
    list(lt.process_stream([frames] * 2, annotate=False))
    t0 = time.perf_counter()
    list(lt.process_stream([frames] * W, annotate=False))
    dt = time.perf_counter() - t0
    out["outage_%d" % length] = {"frames_per_s": round(W * n / dt, 1), "success_ratio": round(lt.get_success_ratio()[0], 3),
                                 "ms_per_outage_frame": None if not length else round((dt - W * n / 50000.0) / (W * 4 * length) * 1e3, 3)}
    lt.close()
print(json.dumps(out, indent=1))
