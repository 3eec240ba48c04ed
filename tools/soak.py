#!/usr/bin/env python3
"""Soak run of the drop-in: ONE process, `process_stream(annotate=True)` over windows of a drifting-lane video
with outages, for `seconds` (or until `frames`), with two tracker close / reopen cycles (state carried over by get_state /
set_state); every `interval` seconds one JSON line: frames/s of the interval, RSS, page-locked bytes (the Python pool's and the
library's staging), the copy threads' backlog, the device cache's size and its traffic with the driver, live threads.  The last
line is the verdict the test reads (tests/test_gpu_soak.py).

    python tools/soak.py [--seconds 60] [--frames 0] [--interval 5] [--size 1280x720] [--window 256]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument("--seconds", type=float, default=60.0)
ap.add_argument("--frames", type=int, default=0)
ap.add_argument("--interval", type=float, default=5.0)
ap.add_argument("--size", default="1280x720")
ap.add_argument("--window", type=int, default=256)
ap.add_argument("--nwin", type=int, default=6)
a = ap.parse_args()

import bench
from lane_tracker_amd import _native, calib
from lane_tracker_amd.lane_tracker import LaneTracker

base = bench.render_streams(96)[a.size]
cal = calib.reference_calibration() if a.size == "1280x720" else calib.scaled_calibration(1.5)
wins = bench.stream_windows(base, a.window, a.nwin)
# outages in two of the windows: 16 frames of noise / grey / black every 64 (the stream falls back to the second try and to
# sliding windows, handled in speculative groups)
for w in (wins[1], wins[4 % a.nwin]):
    for j, s0 in enumerate(range(40, a.window, 64)):
        for i in range(s0, min(a.window, s0 + 16)):
            w[i] = (np.random.default_rng(4000 + i).integers(0, 256, w[i].shape, dtype=np.uint8) if j % 3 == 0 else (128 if j % 3 == 1 else 0))


def rss():
    return int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE")


def sample():
    hm, dc, cc = _native.host_memory_stats(), _native.device_cache_stats(), _native.device_cache_counters()
    return {"rss": rss(), "pinned_pool_outstanding": _native._pinned.outstanding, "pinned_pool_idle": sum(k * len(v) for k, v in _native._pinned.free.items()),
            "frame_pool_idle": _native._frames.idle_bytes, "staging_bytes": hm["staging_bytes"], "queued_pieces": hm["queued_pieces"],
            "pending_pieces": hm["pending_pieces"], "cache_kept": dc["kept_bytes"], "cache_live": dc["live_bytes"], "evicted_bytes": cc["evicted_bytes"],
            "cache_misses": cc["misses"], "threads": len(os.listdir("/proc/self/task"))}


lt = LaneTracker(**cal)
lt.warm(a.window, True)
t_start = time.perf_counter()
reopen_at = [a.seconds / 3.0, 2.0 * a.seconds / 3.0] if not a.frames else []
reopen_frames = [a.frames // 3, 2 * a.frames // 3] if a.frames else []
lines, frames_done, reopened = [], 0, 0
t_iv, f_iv = time.perf_counter(), 0
checksum = 0
while True:
    for out in lt.process_stream(wins, annotate=True):
        frames_done += len(out)
        f_iv += len(out)
        checksum ^= int(out[-1][500, 640, 1])            # (the frames are really read)
    now = time.perf_counter()
    if now - t_iv >= a.interval:
        line = dict(sample(), t=round(now - t_start, 2), frames=frames_done, fps=round(f_iv / (now - t_iv), 1), success_ratio=round(lt.get_success_ratio()[0], 4))
        lines.append(line)
        print(json.dumps(line), flush=True)
        t_iv, f_iv = time.perf_counter(), 0
    due = (reopened < len(reopen_at) and now - t_start >= reopen_at[reopened]) or (reopened < len(reopen_frames) and frames_done >= reopen_frames[reopened])
    if due:                                  # close / reopen: the stream goes on in a NEW tracker from the old one's state
        st = json.loads(json.dumps(lt.get_state()))
        lt.close()
        lt = LaneTracker(**cal)
        lt.set_state(st)
        reopened += 1
        print(json.dumps({"reopened": reopened, "t": round(time.perf_counter() - t_start, 2)}), flush=True)
        t_iv, f_iv = time.perf_counter(), 0       # (the interval that holds a reopen is not a rate sample)
    if (a.frames and frames_done >= a.frames) or (not a.frames and now - t_start >= a.seconds):
        break
lt.close()
print("VERDICT " + json.dumps({"frames": frames_done, "seconds": round(time.perf_counter() - t_start, 1), "reopened": reopened, "samples": len(lines), "checksum": checksum}), flush=True)
