#!/usr/bin/env python3
"""Is process() slowed down by the cgroup's CPU quota?  bench.py's process() leg (a window of 256 rendered frames, one frame
per call) at one size, with /sys/fs/cgroup/cpu.stat (nr_throttled, throttled_usec) read before and after, the per-call wall
times of tools/process_trace.py and the spread of the frame times.
  python tools/process_throttle_probe.py 1920x1080 [seconds]     (environment: LT_COPY_THREADS, LT_COPY_SPIN_US, ...)"""
import collections, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lane_tracker_amd import calib, _native
from lane_tracker_amd.lane_tracker import LaneTracker


def cpu_stat():
    out = {}
    for path in ("/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.stat"):
        try:
            for line in open(path):
                k, v = line.split()
                out[k] = int(v)
            break
        except OSError:
            continue
    return out


size = sys.argv[1] if len(sys.argv) > 1 else "1920x1080"
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
base = bench.render_streams(96)[size]
cal = calib.reference_calibration() if size == "1280x720" else calib.scaled_calibration(1.5)
frames = bench.stream_windows(base, 256, 1)[0]
time.sleep(1.0)
lt = LaneTracker(**cal)
for f in frames[:4]:
    lt.process(f)
s0 = cpu_stat()
times = []
t_end = time.perf_counter() + seconds
k = 0
while time.perf_counter() < t_end:
    t0 = time.perf_counter()
    lt.process(frames[4 + k % 252])
    times.append(time.perf_counter() - t0)
    k += 1
s1 = cpu_stat()
t = np.array(times) * 1e6
print(json.dumps({"size": size, "frames": k, "fps": round(k / t.sum() * 1e6, 1), "us_median": round(float(np.median(t)), 1),
                  "us_p10": round(float(np.percentile(t, 10)), 1), "us_p90": round(float(np.percentile(t, 90)), 1),
                  "us_p99": round(float(np.percentile(t, 99)), 1), "us_max": round(float(t.max()), 1),
                  "share_of_time_in_frames_over_2x_median": round(float(t[t > 2 * np.median(t)].sum() / t.sum()), 3),
                  "cgroup_cpu_stat_delta": {kk: s1[kk] - s0.get(kk, 0) for kk in s1 if s1[kk] != s0.get(kk, 0)},
                  "usable_cpus": bench.usable_cpus(), "copy_threads": _native.host_copy_stats()["threads"],
                  "env": {kk: vv for kk, vv in os.environ.items() if kk.startswith("LT_")}}))
lt.close()
