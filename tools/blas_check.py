#!/usr/bin/env python3
"""Does hostcpu.blas_limited() take hold in THIS process (GPU box: 256 CPUs shown, HIP runtime loaded)?  Prints the BLAS pools
threadpoolctl sees before / inside / behind the block, with a tracker alive."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import threadpoolctl
from lane_tracker_amd import calib, hostcpu
from lane_tracker_amd.lane_tracker import LaneTracker
info = lambda: [(p.get("internal_api"), p.get("num_threads"), os.path.basename(p.get("filepath", ""))) for p in threadpoolctl.threadpool_info()]
print("at import", info())
lt = LaneTracker(**calib.reference_calibration())
print("pools found", hostcpu.find_blas_pools(), "with a tracker", info())
with hostcpu.blas_limited():
    print("inside", info())
    t = time.perf_counter(); np.polyfit(np.arange(13000.0), np.arange(13000.0) ** 2, 2); print("polyfit ms", round((time.perf_counter() - t) * 1e3, 2))
print("behind", info())
print("threads", len(os.listdir("/proc/self/task")))
lt.close()
