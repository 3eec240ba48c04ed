#!/usr/bin/env python3
"""cProfile of LaneTracker.process() one frame at a time: which Python functions the host spends its share in
(the figures include the profiler's own overhead, a few tenths of a microsecond per call; read them as a ranking)."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import calib, synth
from lane_tracker_amd.lane_tracker import LaneTracker
cal = calib.reference_calibration() if len(sys.argv) < 2 else calib.scaled_calibration(1.5)
frames = synth.stream_lanes(24, seed=5, cal=cal)
frames = np.concatenate([frames, frames[::-1]] * 6, 0)
lt = LaneTracker(**cal)
for f in frames[:8]:
    lt.process(f)
n = len(frames) - 8
t0 = time.perf_counter()
for f in frames[8:]:
    lt.process(f)
plain = (time.perf_counter() - t0) / n * 1e6
pr = cProfile.Profile()
pr.enable()
for f in frames[8:]:
    lt.process(f)
pr.disable()
print("plain: %.1f us per frame (%.0f frames/s), %d frames" % (plain, 1e6 / plain, n))
st = pstats.Stats(pr)
rows = sorted(((tt, ct, nc, fn) for fn, (cc, nc, tt, ct, callers) in st.stats.items()), reverse=True)[:40]
print("%9s %9s %7s  function" % ("self us", "cum us", "calls"))
for tt, ct, nc, fn in rows:
    print("%9.1f %9.1f %7.1f  %s:%d %s" % (tt / n * 1e6, ct / n * 1e6, nc / n, os.path.basename(fn[0]), fn[1], fn[2]))
