#!/usr/bin/env python3
"""process_stream(annotate=False) frames/s over chain_chunk x chain_depth (windows of 256 frames, 8 windows per call)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import calib, synth
from lane_tracker_amd.lane_tracker import LaneTracker
for name, cal in (("720p", calib.reference_calibration()), ("1080p", calib.scaled_calibration(1.5))):
    base = synth.stream_lanes(32, seed=5, cal=cal)
    for n in (64, 256):
        frames = np.concatenate([base, base[::-1]] * (n // 64 + 1), 0)[:n].copy()
        lt = LaneTracker(**cal)
        row = {}
        for chunk in (32, 64, 128):
            for depth in (1, 3):
                lt.chain_chunk, lt.chain_depth = chunk, depth
                list(lt.process_stream([frames] * 2, annotate=False))
                t0 = time.perf_counter()
                list(lt.process_stream([frames] * 8, annotate=False))
                row["c%d_d%d" % (chunk, depth)] = round(8 * n / (time.perf_counter() - t0) / 1e3, 1)
        print(name, n, json.dumps(row))
        lt.close()
