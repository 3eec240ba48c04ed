#!/usr/bin/env python3
"""Which earlier use of a tracker slows process_stream(annotate=True) down (bench.py's stream leg measured 9.4 k frames/s where a
fresh tracker does 15.5 k)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import calib, synth, _native
from lane_tracker_amd.lane_tracker import LaneTracker
cal = calib.reference_calibration() if "1080" not in sys.argv else calib.scaled_calibration(1.5)
if os.environ.get("CUS"):
    LaneTracker.search_cus = int(os.environ["CUS"])
n = 256
base = synth.stream_lanes(32, seed=5, cal=cal)
frames = np.concatenate([base, base[::-1]] * (n // 64 + 1), 0)[:n].copy()

def measure(lt, tag):
    for _ in lt.process_stream([frames] * 4, annotate=True):
        pass
    t0 = time.perf_counter()
    for _ in lt.process_stream([frames] * 6, annotate=True):
        pass
    print("%-40s %.0f frames/s   pinned outstanding %.2f GB" % (tag, 6 * n / (time.perf_counter() - t0), _native._pinned.outstanding / 2**30))

steps = {
    "process": lambda lt: [lt.process(f) for f in frames[:40]],
    "batch": lambda lt: lt.process_batch(frames, annotate=False),
    "batch_annotated": lambda lt: lt.process_batch(frames, annotate=True),
    "stream": lambda lt: list(lt.process_stream([frames] * 6, annotate=False)),
}
lt = LaneTracker(**cal); measure(lt, "fresh"); lt.close()
for name, fn in list(steps.items())[:1]:
    lt = LaneTracker(**cal)
    fn(lt)
    measure(lt, "after " + name)
    lt.close()
