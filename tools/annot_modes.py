#!/usr/bin/env python3
"""Why does the annotated stream run at 30 k frames/s in one process and at 22 k in the next (same box, same settings)?
One process, many passes; per pass the rate, the copy threads' own speed (bytes per busy second: memory placement, CPU) and
their busy share (below 1: they wait for the device's strips); between passes, in turn: nothing / the copy threads restarted
(lt_shutdown: new threads, new CPUs) / fresh windows / a fresh tracker.   python tools/annot_modes.py [1280x720] [passes]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lane_tracker_amd import calib, _native
from lane_tracker_amd.lane_tracker import LaneTracker
size = sys.argv[1] if len(sys.argv) > 1 else "1280x720"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 16
base = bench.render_streams(96)[size]
cal = calib.reference_calibration() if size == "1280x720" else calib.scaled_calibration(1.5)
wins = bench.stream_windows(base, 256, 8)
lt = LaneTracker(**cal)


def cpus_of_threads():
    out = {}
    for tid in os.listdir("/proc/self/task"):
        try:
            st = open("/proc/self/task/%s/stat" % tid).read().rsplit(")", 1)[1].split()
            comm = open("/proc/self/task/%s/comm" % tid).read().strip()
            out[tid] = (comm, int(st[36]))          # processor the thread last ran on
        except Exception:
            pass
    return out


def one():
    c0, t0 = _native.host_copy_stats(), time.perf_counter()
    n = 0
    for out in lt.process_stream(wins, annotate=True):
        n += len(out)
    dt = time.perf_counter() - t0
    c1 = _native.host_copy_stats()
    busy = c1["busy_s"] - c0["busy_s"]
    return {"fps": round(n / dt), "copy_GBps_per_busy_thread": round((c1["bytes"] - c0["bytes"]) / max(busy, 1e-9) / 1e9, 2),
            "busy_share": round(busy / (dt * c1["threads"]), 3)}


one()
for k in range(passes):
    what = ("nothing", "nothing", "copy threads restarted", "fresh windows", "fresh tracker")[k % 5] if k else "first"
    if what == "copy threads restarted":
        _native.load().lt_shutdown()
    elif what == "fresh windows":
        wins = bench.stream_windows(base, 256, 8)
        one()
    elif what == "fresh tracker":
        lt.close()
        lt = LaneTracker(**cal)
        one()
    r = one()
    r["before_this_pass"] = what
    th = cpus_of_threads()
    r["cpus_last_run_on"] = sorted(set(v[1] for v in th.values()))[:40]
    print(json.dumps(r), flush=True)
lt.close()
