#!/usr/bin/env python3
"""process_batch(annotate=False) frames/s over chain_chunk x chain_depth x window size."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import calib, synth
from lane_tracker_amd.lane_tracker import LaneTracker

def t(fn, reps=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps

for name, cal in (("720p", calib.reference_calibration()), ("1080p", calib.scaled_calibration(1.5))):
    base = synth.stream_lanes(32, seed=5, cal=cal)
    for n in (128, 256, 512):
        frames = np.concatenate([base, base[::-1]] * (n // 64 + 1), 0)[:n].copy()
        lt = LaneTracker(**cal)
        row = {}
        for chunk in (32, 64, 128):
            for depth in (1, 3):
                lt.chain_chunk, lt.chain_depth = chunk, depth
                row["c%d_d%d" % (chunk, depth)] = round(n / t(lambda: lt.process_batch(frames, annotate=False)) / 1e3, 1)
        print(name, n, json.dumps(row))
        lt.close()
