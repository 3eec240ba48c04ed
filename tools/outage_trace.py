#!/usr/bin/env python3
"""cProfile of process_stream over windows with 32-frame outages."""
import cProfile, io, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import calib, synth
from lane_tracker_amd.lane_tracker import LaneTracker
cal = calib.reference_calibration()
n, length = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 32
frames = synth.stream_lanes(n, seed=5, cal=cal).copy()
for k, s in enumerate(range(40, n, 64)):
    for i in range(s, min(n, s + length)):
        frames[i] = synth.frame_uniform(4000 + i) if k % 3 == 0 else (128 if k % 3 == 1 else 0)
lt = LaneTracker(**cal)
list(lt.process_stream([frames] * 2, annotate=False))
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
list(lt.process_stream([frames] * 4, annotate=False))
dt = time.perf_counter() - t0
pr.disable()
print("fps", 4 * n / dt)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(16); print(s.getvalue()[:4000])
