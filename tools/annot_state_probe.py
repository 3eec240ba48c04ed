#!/usr/bin/env python3
"""Which of bench.py's legs leaves the state that makes the later annotated pass of the LONG-LIVED tracker slower than the
first pass of a fresh one in the same process?  (19.9 k / 10.5 k later pass against 27.7 k / 13.5 k first
pass, four of four bench runs.)

ONE process.  The sequence of bench.py::stream_leg, one leg at a time; after every leg the annotated pass of the SAME
long-lived tracker over the SAME windows is measured again (3 passes), with, per pass: frames/s, the copy threads' own speed
(bytes per busy second of plain copies) and busy share, the driving thread's waits (copy group, chain records), pool sizes
(frame pool, page-locked pool, device cache), live threads, RSS, cgroup throttling.  At the end the cross checks that separate
"the tracker" from "the windows" from "the process": a fresh tracker over the old windows, the old tracker over fresh windows,
the pools trimmed.

    python tools/annot_state_probe.py [1280x720|1920x1080] [--prelude] [--passes 3]

--prelude: first what bench.py's main() runs before the stream leg (resident batch, CPU baseline on all threads incl. its
mallopt, parameter sets, host-fed leg)."""
import argparse
import gc
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from lane_tracker_amd import _native, calib
from lane_tracker_amd.lane_tracker import LaneTracker

ap = argparse.ArgumentParser()
ap.add_argument("size", nargs="?", default="1280x720")
ap.add_argument("--prelude", action="store_true")
ap.add_argument("--passes", type=int, default=3)
ap.add_argument("--window", type=int, default=256)
ap.add_argument("--nwin", type=int, default=8)
a = ap.parse_args()
W, NW = a.window, a.nwin


def cpu_stat():
    out = {}
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, v = line.split()
            out[k] = int(v)
    except Exception:
        pass
    return out


def proc_state():
    rss = 0
    try:
        rss = int(open("/proc/self/statm").read().split()[1]) * os.sysconf("SC_PAGE_SIZE")
    except Exception:
        pass
    cpus = set()
    tasks = os.listdir("/proc/self/task")
    for tid in tasks:
        try:
            cpus.add(int(open("/proc/self/task/%s/stat" % tid).read().rsplit(")", 1)[1].split()[36]))
        except Exception:
            pass
    fp, pp = _native._frames, _native._pinned
    return {"threads": len(tasks), "rss_GB": round(rss / 1e9, 2), "cpus_last_run_on": sorted(cpus)[:48],
            "frame_pool_blocks": {str(k >> 20) + "MB": len(v) for k, v in fp.free.items() if v}, "frame_pool_idle_GB": round(fp.idle_bytes / 1e9, 2),
            "pinned_pool_blocks": {str(k >> 20) + "MB": len(v) for k, v in pp.free.items() if v}, "pinned_outstanding_MB": pp.outstanding >> 20,
            "device_cache": {k: (round(v / 1e9, 2) if k.endswith("bytes") else v) for k, v in _native.device_cache_stats().items()}}


def timed_attr(obj, name, acc):
    fn = getattr(obj, name)

    def w(*x, **k):
        t0 = time.perf_counter()
        try:
            return fn(*x, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    setattr(obj, name, w)
    return fn


def ann_pass(tracker, wins, ann=True):
    """One pass of process_stream over `wins` -> the figures of the pass."""
    acc = {}
    ctx = tracker._ctx
    saved = [(tracker, "_copies_done", timed_attr(tracker, "_copies_done", acc))]
    for nm in ("band_fit_chain_collect", "overlay_run_strip_coeffs", "overlay_run_strip_packed", "strip_download_async", "upload_frame_rows_async", "mask_run"):
        saved.append((ctx, nm, timed_attr(ctx, nm, acc)))
    c0, s0, t0 = _native.host_copy_stats(), cpu_stat(), time.perf_counter()
    n = 0
    for out in tracker.process_stream(wins, annotate=ann):
        n += len(out)
    dt = time.perf_counter() - t0
    c1, s1 = _native.host_copy_stats(), cpu_stat()
    for obj, nm, fn in saved:
        try:
            delattr(obj, nm)             # (the instance attribute that shadows the method)
        except AttributeError:
            setattr(obj, nm, fn)
    busy = c1["busy_s"] - c0["busy_s"]
    return {"fps": round(n / dt), "copy_GBps_per_busy_thread": round((c1["bytes"] - c0["bytes"]) / max(busy, 1e-9) / 1e9, 2),
            "busy_share": round(busy / (dt * c1["threads"]), 3), "pieces": c1["pieces"] - c0["pieces"],
            "us_per_frame": {k: round(v / n * 1e6, 2) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])},
            "throttled": [s1.get("nr_throttled", 0) - s0.get("nr_throttled", 0), (s1.get("throttled_usec", 0) - s0.get("throttled_usec", 0)) // 1000]}


def report(tag, tracker, wins, passes=None, ann=True, extra=None):
    rs = [ann_pass(tracker, wins, ann) for _ in range(passes or a.passes)]
    line = {"after": tag, "fps": [r["fps"] for r in rs], "median_fps": sorted(r["fps"] for r in rs)[len(rs) // 2],
            "copy_GBps_per_busy_thread": [r["copy_GBps_per_busy_thread"] for r in rs], "busy_share": [r["busy_share"] for r in rs],
            "throttled_events_ms": [r["throttled"] for r in rs], "us_per_frame_last": rs[-1]["us_per_frame"], "state": proc_state()}
    if extra:
        line.update(extra)
    print(json.dumps(line), flush=True)
    return line


# ---- frames are rendered before the GPU is touched (forked workers) ----
t_start = time.perf_counter()
frames_batch = bench.render_frames(range(256)) if a.prelude else None
base = bench.render_streams(96)[a.size]
cal = calib.reference_calibration() if a.size == "1280x720" else calib.scaled_calibration(1.5)

if a.prelude:
    # bench.py main() up to the stream leg, shortened only in its repetition counts
    cal0 = calib.reference_calibration()
    ctx = _native.Context(cal0["img_size"], cal0["warped_size"], cal0["cam_matrix"], cal0["dist_coeffs"], cal0["warp_matrices"][0], device=0, capacity=512)
    ctx.upload_frames(frames_batch, first=0)
    ctx.upload_frame_rows(frames_batch, first=0)
    ctx.upload_frames(frames_batch, first=256)
    ctx.set_frame_base(256, 0)
    ctx.set_frame_base(256, 0, first=256)
    fp, sp = _native.filter_params(), _native.search_params()
    ctx.set_streams(4)
    for k in range(12):
        ctx.mask_run(256, fp)
        ctx.sws_fit_run(256, sp)
    ctx.sync()
    rec = ctx.download_records(256)
    t0 = time.perf_counter()
    cb, par = bench.cpu_baseline(frames_batch, cal0, rec, lambda i: ctx.download_masks(1, first=i)[0])
    print(json.dumps({"prelude": "cpu_baseline", "s": round(time.perf_counter() - t0, 1), "value": cb["value"], "cores": cb["cores"], "parity": par["record_mismatches"]}), flush=True)
    bench.settings_leg(ctx, 256, 4)
    ctx.close()
    bench.host_fed_overlapped(cal0, frames_batch, fp, sp, 4, rec)
    print(json.dumps({"prelude": "done", "state": proc_state()}), flush=True)

wins = bench.stream_windows(base, W, NW)
frames = wins[0]
lt = LaneTracker(**cal)
lt.warm(W, True)


def stream_rate(ws, ann, tracker):
    t0 = time.perf_counter()
    for _ in tracker.process_stream(ws, annotate=ann):
        pass
    return round(len(ws) * W / (time.perf_counter() - t0), 1)


cold = bench.stream_windows(base, W, NW)
base_line = report("nothing (fresh tracker, warmed): first pass is pass 0", lt, cold, passes=5)

# --- leg 1: process() one frame at a time (bench.py:147-156)
for f in frames[:32]:
    lt.process(f)
k = 0
for _ in range(3):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.15:
        lt.process(frames[32 + k % (W - 32)])
        k += 1
report("process() x %d" % (k + 32), lt, cold)

# --- leg 2 / 3: process_batch plain, annotated (bench.py:157-163)
for ann in (False, True):
    lt.process_batch(frames, annotate=ann)
    t0, k = time.perf_counter(), 0
    while time.perf_counter() - t0 < 0.2 or k == 0:
        lt.process_batch(wins[1 + k % (NW - 1)], annotate=ann)
        k += 1
    report("process_batch(annotate=%s) x %d" % (ann, k + 1), lt, cold)


def first_passes(ann, trackers=4):
    rates = []
    for k in range(trackers):
        ws = bench.stream_windows(base, W, NW)
        fresh = LaneTracker(**cal)
        try:
            if k < trackers - 1:
                fresh.warm(W, ann)
            rates.append(stream_rate(ws, ann, fresh))
        finally:
            fresh.close()
        del ws
    return rates


# --- leg 4: four fresh trackers, plain first passes (bench.py:206)
r = first_passes(False)
report("4 fresh trackers opened and closed, plain first passes", lt, cold, extra={"their_fps": r})
# --- leg 5: the plain stream on the long-lived tracker (bench.py:209-211)
stream_rate(cold, False, lt)
r = [stream_rate(cold + cold, False, lt) for _ in range(2)]
report("plain stream on the long-lived tracker", lt, cold, extra={"plain_fps": r})
# --- leg 6: four fresh trackers, annotated first passes (bench.py:213)
r = first_passes(True)
report("4 fresh trackers opened and closed, annotated first passes", lt, cold, extra={"their_fps": r})

# ---- cross checks ----
lt2 = LaneTracker(**cal)
lt2.warm(W, True)
report("CROSS: a FRESH tracker over the OLD windows", lt2, cold)
ws = bench.stream_windows(base, W, NW)
report("CROSS: the OLD tracker over FRESH windows", lt, ws)
report("CROSS: the fresh tracker over the fresh windows", lt2, ws)
lt2.close()
del ws
report("CROSS: the old tracker over the old windows again", lt, cold)
_native._frames.trim()
gc.collect()
report("CROSS: frame pool trimmed", lt, cold)
_native.device_cache_trim(0)
report("CROSS: device cache trimmed", lt, cold)
_native.load().lt_shutdown()
report("CROSS: copy threads restarted + staging blocks given back (lt_shutdown)", lt, cold)
lt.close()
lt = LaneTracker(**cal)
lt.warm(W, True)
report("CROSS: the long-lived tracker closed, a new one", lt, cold)
lt.close()
print(json.dumps({"total_s": round(time.perf_counter() - t_start, 1)}), flush=True)
