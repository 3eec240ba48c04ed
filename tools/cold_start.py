#!/usr/bin/env python3
"""Where the FIRST pass of a stream (and the first calls of process()) spend their time.

  python tools/cold_start.py [--size 1280x720|1920x1080] [--annotate] [--trackers 3] [--windows 8] [--prefix none|bench]
                             [--warm] [--json out.json]

For each of `--trackers` fresh LaneTrackers in this process: process_stream over `--windows` freshly allocated pageable
windows of 256 frames; wall time of every library call of the driving thread (a wrapper around _native.Context and the
page-locked pool) merged with the library's own LT_TRACE_START lines (hipMalloc, cache hits, hipHostMalloc, lt_reserve, table
builds).  Prints time_to_first_window_ms, the first pass and a later pass in frames/s, and the phases of the first window
sorted by their share.  --prefix bench: the calls bench.py's stream leg makes on the tracker before its first pass
(process() x N, two process_batch legs).  --warm: LaneTracker.warm() before the stream (round 5)."""
import argparse
import collections
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1280x720")
    ap.add_argument("--annotate", action="store_true")
    ap.add_argument("--trackers", type=int, default=3)
    ap.add_argument("--windows", type=int, default=8)
    ap.add_argument("--window", type=int, default=256)
    ap.add_argument("--prefix", default="none")
    ap.add_argument("--warm", action="store_true")
    ap.add_argument("--json", default=None)
    ap.add_argument("--events", type=int, default=0, help="print the first N timeline events of the first tracker")
    a = ap.parse_args()

    trace_path = tempfile.mktemp(prefix="lt_start_", suffix=".log")
    os.environ["LT_TRACE_START"] = "1"
    # the library writes its lines to stderr: point fd 2 at a file for the duration and read it back
    import bench
    streams = bench.render_streams(96)
    base = streams[a.size]
    saved = os.dup(2)
    fd = os.open(trace_path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC)
    os.dup2(fd, 2)

    from lane_tracker_amd import _native, calib
    from lane_tracker_amd.lane_tracker import LaneTracker
    cal = calib.reference_calibration() if a.size == "1280x720" else calib.scaled_calibration(1.5)

    events = []                      # (t0, name, ms)

    def timed(obj, name, label=None):
        fn = getattr(obj, name)

        def wrapper(*args, **kw):
            t0 = time.monotonic()
            try:
                return fn(*args, **kw)
            finally:
                events.append((t0, label or name, (time.monotonic() - t0) * 1e3))
        setattr(obj, name, wrapper)
    for m in ("reserve", "sync", "upload_frame_rows_async", "upload_frame_rows", "upload_frame_rest", "mask_run", "sws_fit_run",
              "band_fit_run", "band_fit_chain_run", "band_fit_chain_collect", "download_records", "download_record", "overlay_configure",
              "overlay_set_font", "overlay_run_packed", "overlay_text", "download_overlay_async", "download_overlay_wait",
              "present_frame", "present_lane_async", "present_finish", "set_search_cus", "close"):
        timed(_native.Context, m)
    timed(_native, "pinned_empty")
    timed(LaneTracker, "_copies_done")
    timed(LaneTracker, "_valid_many")

    results = []
    for k in range(a.trackers):
        del events[:]
        wins = bench.stream_windows(base, a.window, a.windows)
        t_new = time.monotonic()
        lt = LaneTracker(**cal)
        t_made = time.monotonic()
        try:
            if a.prefix == "bench":
                for f in wins[0][:4]:
                    lt.process(f)
                for i in range(200):
                    lt.process(wins[0][4 + i % 252])
                for ann in (False, True):
                    lt.process_batch(wins[0], annotate=ann)
                    lt.process_batch(wins[1], annotate=ann)
                wins = bench.stream_windows(base, a.window, a.windows)
            if a.warm:
                t0 = time.monotonic()
                lt.warm(a.window, annotate=a.annotate)
                events.append((t0, "warm", (time.monotonic() - t0) * 1e3))
            t0 = time.monotonic()
            first = None
            for _ in lt.process_stream(wins, annotate=a.annotate):
                if first is None:
                    first = time.monotonic()
            t1 = time.monotonic()
            first_events = [e for e in events if t0 <= e[0] <= first]
            pass_events = [e for e in events if t0 <= e[0] <= t1]
            t2 = time.monotonic()
            for _ in lt.process_stream(wins, annotate=a.annotate):
                pass
            t3 = time.monotonic()
            n = a.windows * a.window
            res = {"tracker": k, "size": a.size, "annotate": a.annotate, "prefix": a.prefix, "warm": a.warm,
                   "constructor_ms": round((t_made - t_new) * 1e3, 2),
                   "time_to_first_window_ms": round((first - t0) * 1e3, 2),
                   "first_pass_fps": round(n / (t1 - t0), 1), "later_pass_fps": round(n / (t3 - t2), 1),
                   "first_pass_ms": round((t1 - t0) * 1e3, 2), "later_pass_ms": round((t3 - t2) * 1e3, 2),
                   "t0": t0, "t_first": first, "t1": t1}
            by = collections.defaultdict(lambda: [0, 0.0])
            for _, nm, ms in first_events:
                by[nm][0] += 1
                by[nm][1] += ms
            res["host_calls_until_first_window"] = {nm: [c, round(ms, 2)] for nm, (c, ms) in sorted(by.items(), key=lambda kv: -kv[1][1])}
            by = collections.defaultdict(lambda: [0, 0.0])
            for _, nm, ms in pass_events:
                by[nm][0] += 1
                by[nm][1] += ms
            res["host_calls_first_pass"] = {nm: [c, round(ms, 2)] for nm, (c, ms) in sorted(by.items(), key=lambda kv: -kv[1][1])}
            if k == 0 and a.events:
                res["timeline"] = [(round((e[0] - t0) * 1e3, 3), e[1], round(e[2], 3)) for e in pass_events[:a.events]]
            results.append(res)
        finally:
            lt.close()
        del wins

    os.dup2(saved, 2)
    os.close(fd)
    lib = []
    for line in open(trace_path):
        p = line.split()
        if len(p) >= 4 and p[0] == "lt_start":
            lib.append((float(p[1]), p[2], float(p[3]), int(p[4]) if len(p) > 4 else 0))
        else:
            sys.stderr.write(line)
    os.unlink(trace_path)
    for r in results:
        t0, tf, t1 = r.pop("t0"), r.pop("t_first"), r.pop("t1")
        for key, hi in (("library_until_first_window", tf), ("library_first_pass", t1)):
            by = collections.defaultdict(lambda: [0, 0.0, 0])
            for t, nm, ms, b in lib:
                if t0 <= t <= hi:
                    by[nm][0] += 1
                    by[nm][1] += ms
                    by[nm][2] += b
            r[key] = {nm: [c, round(ms, 2), b] for nm, (c, ms, b) in sorted(by.items(), key=lambda kv: -kv[1][1])}
        print(json.dumps(r))
    if a.json:
        json.dump(results, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
