#!/usr/bin/env python3
"""process() over a short stream, nothing else (the program rocprofv3 traces for tools/process_timeline.sh)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import calib, synth
from lane_tracker_amd.lane_tracker import LaneTracker
cal = calib.scaled_calibration(1.5) if "x" in sys.argv[1:] else calib.reference_calibration()
frames = synth.stream_lanes(24, seed=5, cal=cal)
frames = np.concatenate([frames, frames[::-1]] * 4, 0)
lt = LaneTracker(**cal)
if "engine" in sys.argv[1:]:                 # the frame's rows by the copy engine, not through the PCIe aperture
    lt._ctx.set_direct_upload(False)
for f in frames[:8]:
    lt.process(f)
t0 = time.perf_counter()
for f in frames[8:]:
    lt.process(f)
print("us per frame %.1f" % ((time.perf_counter() - t0) / (len(frames) - 8) * 1e6))
lt.close()
