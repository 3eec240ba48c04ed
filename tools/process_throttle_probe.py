#!/usr/bin/env python3
"""Is process() slowed down by the cgroup's CPU quota?  bench.py's process() leg (a window of 256 rendered frames, one frame
per call) at one size, with /sys/fs/cgroup/cpu.stat (nr_throttled, throttled_usec) read before and after, the per-call wall
times of tools/process_trace.py and the spread of the frame times.
  python tools/process_throttle_probe.py 1920x1080 [seconds] [engine]     (environment: LT_COPY_THREADS, LT_COPY_SPIN_US, ...;
  engine: the frame's rows by the copy engine instead of through the PCIe aperture, lt_set_direct_upload(ctx, 0))"""
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
if len(sys.argv) > 3 and sys.argv[3] == "engine":
    lt._ctx.set_direct_upload(False)
for f in frames[:4]:
    lt.process(f)
def thread_cpu():
    """{tid: (comm, cpu seconds)} of every thread of this process"""
    out = {}
    tick = os.sysconf("SC_CLK_TCK")
    for tid in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % tid).read()
            comm = f[f.index("(") + 1:f.rindex(")")]
            v = f[f.rindex(")") + 2:].split()
            out[tid] = (comm, (int(v[11]) + int(v[12])) / tick)
        except Exception:
            pass
    return out


def burners(a, b, floor=0.02):
    """threads that used more than `floor` CPU seconds between two snapshots, by name"""
    acc = collections.Counter()
    for tid, (comm, cpu) in b.items():
        d = cpu - a.get(tid, (comm, 0.0))[1]
        if d >= floor:
            acc[comm] += 1
    return dict(acc)


s0 = cpu_stat()
snap, snap_k, stalls = thread_cpu(), 0, []
times = []
t_end = time.perf_counter() + seconds
k = 0
while time.perf_counter() < t_end:
    t0 = time.perf_counter()
    lt.process(frames[4 + k % 252])
    times.append(time.perf_counter() - t0)
    k += 1
    if times[-1] > 0.02 and len(stalls) < 6:      # a frozen frame: which threads burned CPU since the last snapshot?
        now = thread_cpu()
        stalls.append({"frame": k, "ms": round(times[-1] * 1e3, 1), "frames_since_snapshot": k - snap_k, "threads_that_burned_20ms": burners(snap, now),
                       "threads": len(now)})
        snap, snap_k = now, k
    elif k - snap_k >= 400:
        snap, snap_k = thread_cpu(), k
s1 = cpu_stat()
t = np.array(times) * 1e6
print(json.dumps({"size": size, "rows_through_the_aperture": lt._ctx.direct_upload_count() > 0, "frames": k, "fps": round(k / t.sum() * 1e6, 1), "us_median": round(float(np.median(t)), 1),
                  "us_p10": round(float(np.percentile(t, 10)), 1), "us_p90": round(float(np.percentile(t, 90)), 1),
                  "us_p99": round(float(np.percentile(t, 99)), 1), "us_max": round(float(t.max()), 1),
                  "share_of_time_in_frames_over_2x_median": round(float(t[t > 2 * np.median(t)].sum() / t.sum()), 3),
                  "cgroup_cpu_stat_delta": {kk: s1[kk] - s0.get(kk, 0) for kk in s1 if s1[kk] != s0.get(kk, 0)},
                  "stalls": stalls, "usable_cpus": bench.usable_cpus(), "copy_threads": _native.host_copy_stats()["threads"],
                  "env": {kk: vv for kk, vv in os.environ.items() if kk.startswith("LT_")}}))
lt.close()
