#!/usr/bin/env python3
"""process_stream(annotate=True) over 6 windows of 256 frames: frames/s and where the host thread spends its time (cProfile)."""
import cProfile, io, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import calib, synth
from lane_tracker_amd.lane_tracker import LaneTracker
cal = calib.reference_calibration() if len(sys.argv) < 2 or sys.argv[1] != "1080" else calib.scaled_calibration(1.5)
n, W = 256, 6
base = synth.stream_lanes(32, seed=5, cal=cal)
frames = np.concatenate([base, base[::-1]] * (n // 64 + 1), 0)[:n].copy()
if os.environ.get("CUS"):
    LaneTracker.search_cus = int(os.environ["CUS"])
if os.environ.get("SECOND"):                 # not the first tracker of the process (the copy engines are handed out differently then)
    t0 = LaneTracker(**cal); t0.process(frames[0]); t0.close()
lt = LaneTracker(**cal)
for out in lt.process_stream([frames] * 4):      # (three page-locked output windows are alive at a time: let the pool get them)
    pass
t0 = time.perf_counter()
for out in lt.process_stream([frames] * W):
    pass
dt = time.perf_counter() - t0
print("annotated stream: %.0f frames/s (%.1f us per frame)" % (W * n / dt, dt / (W * n) * 1e6))
t0 = time.perf_counter()
lt.process_batch(frames)
print("annotated stand-alone window: %.0f frames/s" % (n / (time.perf_counter() - t0)))
pr = cProfile.Profile()
pr.enable()
for out in lt.process_stream([frames] * W):
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
