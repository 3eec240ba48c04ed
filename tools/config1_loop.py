#!/usr/bin/env python3
"""BASELINE config 1 in a loop (the program rocprofv3 traces for the timeline of a two-try frame): tests/golden/photo_test4.png
through process() with its defaults -- both tries rejected by check_validity, the failure frame returned."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image
from lane_tracker_amd import calib
from lane_tracker_amd.lane_tracker import LaneTracker
frame = np.ascontiguousarray(np.asarray(Image.open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "photo_test4.png")).convert("RGB"), np.uint8))
lt = LaneTracker(**calib.reference_calibration())
for _ in range(16):
    lt.process(frame)
t0 = time.perf_counter()
for _ in range(150):
    lt.process(frame)
print("us per frame %.1f" % ((time.perf_counter() - t0) / 150 * 1e6))
lt.close()
