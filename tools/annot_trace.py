#!/usr/bin/env python3
"""process_stream(annotate=True) over 3 windows of 256 frames (for a rocprofv3 kernel + memory-copy timeline)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import calib, synth
from lane_tracker_amd.lane_tracker import LaneTracker
cal = calib.reference_calibration() if len(sys.argv) < 2 else calib.scaled_calibration(1.5)
n = 256
base = synth.stream_lanes(32, seed=5, cal=cal)
frames = np.concatenate([base, base[::-1]] * (n // 64 + 1), 0)[:n].copy()
lt = LaneTracker(**cal)
for o in lt.process_stream([frames] * 4):
    pass
time.sleep(0.05)
t0 = time.perf_counter()
for o in lt.process_stream([frames] * 5):
    pass
print("fps", 5 * n / (time.perf_counter() - t0))
