#!/usr/bin/env python3
"""Engine downloads (LT_DL_ENGINE=1): does re-creating the context's streams (set_search_cus round trip) after earlier use bring
the fast overlap back?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import calib, synth, _native
from lane_tracker_amd.lane_tracker import LaneTracker
cal = calib.reference_calibration()
n = 256
base = synth.stream_lanes(32, seed=5, cal=cal)
frames = np.concatenate([base, base[::-1]] * (n // 64 + 1), 0)[:n].copy()
def measure(lt, tag):
    for _ in lt.process_stream([frames] * 4, annotate=True):
        pass
    t0 = time.perf_counter()
    for _ in lt.process_stream([frames] * 6, annotate=True):
        pass
    print("%-50s %.0f frames/s" % (tag, 6 * n / (time.perf_counter() - t0)))
lt = LaneTracker(**cal)
lt.process_batch(frames, annotate=False)
measure(lt, "after a plain batch")
lt._ctx.set_search_cus(0); lt._ctx.set_search_cus(LaneTracker.search_cus)
measure(lt, "... streams re-created")
lt.close()
lt = LaneTracker(**cal)
lt._ctx.reserve(1024)
lt.process_batch(frames, annotate=False)
measure(lt, "after a plain batch, context sized first")
lt.close()
lt = LaneTracker(**cal)
lt.process_batch(frames[:8], annotate=True)
measure(lt, "after a small annotated batch")
lt.close()
