#!/usr/bin/env python3
"""process_stream(annotate=False) at 1280x720 for process()'s defaults, settings.DEMO_1, and DEMO_1 with single keywords put
back to the defaults: which of them costs the stream its rate?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import calib, settings, synth
from lane_tracker_amd.lane_tracker import LaneTracker
cal = calib.reference_calibration()
n = 256
base = synth.stream_lanes(32, seed=5, cal=cal)
frames = np.concatenate([base, base[::-1]] * (n // 64 + 1), 0)[:n].copy()
wins = [frames] * 8
def rate(kw, limits=None):
    lt = LaneTracker(**cal)
    if limits:
        lt.validity_limits = dict(limits)
    list(lt.process_stream(wins[:3], annotate=False, **kw))
    best = 0
    for _ in range(2):
        t0 = time.perf_counter()
        list(lt.process_stream(wins, annotate=False, **kw))
        best = max(best, len(wins) * n / (time.perf_counter() - t0))
    r = lt.get_success_ratio()[0]
    lt.close()
    return "%.0f frames/s (success %.3f)" % (best, r)
D = dict(settings.DEMO_1["process"])
print("defaults           ", rate({}))
print("DEMO_1             ", rate(D, settings.DEMO_1["validity"]))
print("DEMO_1, no greenery", rate(dict(D, mask_noise=False), settings.DEMO_1["validity"]))
print("DEMO_1, bandwidth 25", rate(dict(D, bandwidth=25), settings.DEMO_1["validity"]))
print("defaults + greenery ", rate(dict(mask_noise=True)))
