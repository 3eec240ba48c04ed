#!/usr/bin/env python3
"""LaneTracker.process() one frame per call for a while: the one-call tail, the text drawn at once and the aperture uploads under
a long run -- resident set, page-locked staging, device cache, the copy threads' backlog and the rate, one sample per interval;
failure frames and a closed / reopened tracker in between.   python tools/process_soak.py [seconds] [size]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lane_tracker_amd import _native, calib
from lane_tracker_amd.lane_tracker import LaneTracker

seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
size = sys.argv[2] if len(sys.argv) > 2 else "1280x720"
cal = calib.reference_calibration() if size == "1280x720" else calib.scaled_calibration(1.5)
frames = bench.stream_windows(bench.render_streams(96)[size], 256, 1)[0].copy()
frames[100:103] = 0                       # a short outage in every lap: failure frames, the lane redrawn, band search again
frames[180] = 128


def rss_mb():
    for line in open("/proc/self/status"):
        if line.startswith("VmRSS:"):
            return int(line.split()[1]) / 1024.0
    return 0.0


lt = LaneTracker(**cal)
t_end = time.perf_counter() + seconds
interval = max(seconds / 12.0, 1.0)
k = n_total = 0
samples = []
t_int, n_int = time.perf_counter(), 0
reopened = 0
while time.perf_counter() < t_end:
    out = lt.process(frames[k % 256])
    k += 1
    n_int += 1
    now = time.perf_counter()
    if now - t_int >= interval:
        samples.append({"t_s": round(now - (t_end - seconds), 1), "frames_per_s": round(n_int / (now - t_int), 1), "rss_mb": round(rss_mb(), 1),
                        "host": _native.host_memory_stats(), "device_cache": _native.device_cache_stats(),
                        "evicted_bytes": _native.device_cache_counters()["evicted_bytes"],
                        "aperture_uploads": lt._ctx.direct_upload_count(), "success": [lt.success, lt.counter]})
        if len(samples) in (4, 8):        # a new tracker continues the stream from the old one's state
            st = lt.get_state()
            lt.close()
            lt = LaneTracker(**cal)
            lt.set_state(st)
            reopened += 1
        n_total += n_int
        t_int, n_int = time.perf_counter(), 0
lt.close()
rss = [s["rss_mb"] for s in samples]
print(json.dumps({"size": size, "seconds": seconds, "frames": n_total + n_int, "reopened": reopened, "samples": samples,
                  "rss_mb_first_last_max": [rss[0], rss[-1], max(rss)] if rss else None}))
