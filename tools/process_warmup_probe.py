#!/usr/bin/env python3
"""The first 0.15-0.2 s of process() in a process run at half speed (bench.py's first stretch).  One-time set-up, or a power state
that comes back after every idle spell?  Frame times from the first call on, as medians over blocks of 50 frames: (1) a fresh
tracker, 1500 frames; (2) the same tracker after 2 s of idle, 800 frames; (3) after 2 s in which only the GPU was kept busy (a mask
chain over resident slots in a loop), 800 frames; (4) after 2 s in which only this thread spun, 800 frames.
  python tools/process_warmup_probe.py [1280x720]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lane_tracker_amd import calib, _native
from lane_tracker_amd.lane_tracker import LaneTracker
size = sys.argv[1] if len(sys.argv) > 1 else "1280x720"
base = bench.render_streams(96)[size]
cal = calib.reference_calibration() if size == "1280x720" else calib.scaled_calibration(1.5)
frames = bench.stream_windows(base, 256, 1)[0]
time.sleep(1.0)
lt = LaneTracker(**cal)


def run(n, label):
    t = np.empty(n)
    for k in range(n):
        t0 = time.perf_counter()
        lt.process(frames[k % 256])
        t[k] = time.perf_counter() - t0
    blocks = [round(float(np.median(t[i:i + 50])) * 1e6, 1) for i in range(0, n, 50)]
    print(json.dumps({"phase": label, "median_us_per_block_of_50_frames": blocks, "first_10_frames_us": [round(v * 1e6) for v in t[:10]]}), flush=True)


run(1500, "fresh tracker")
time.sleep(2.0)
run(800, "after 2 s idle")
ctx2 = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=64)
ctx2.upload_frames(frames[:64])
t_end = time.perf_counter() + 2.0
while time.perf_counter() < t_end:
    ctx2.mask_run(64)
    ctx2.sync()
run(800, "after 2 s of GPU work only")
t_end = time.perf_counter() + 2.0
while time.perf_counter() < t_end:
    pass
run(800, "after 2 s of this thread spinning")
ctx2.close()
lt.close()
