#!/usr/bin/env python3
"""process_stream over 4 windows of 256 frames (for a rocprofv3 timeline)."""
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
list(lt.process_stream([frames] * 2, annotate=False))
time.sleep(0.05)
t0 = time.perf_counter()
list(lt.process_stream([frames] * 8, annotate=False))
print("fps", 8 * n / (time.perf_counter() - t0))
