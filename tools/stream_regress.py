#!/usr/bin/env python3
"""Why is the later pass of a plain 1280x720 stream 48 k frames/s in round 5's bench line and 57-58 k in round 4's?  The same
steady-state measurement (bench.py: one pass over 8 fresh windows, then the best of two passes over 16) in a fresh child process
per condition:

  plain        nothing before it
  legs         bench.py's sequence before it: process() frames, process_batch calls, then four fresh trackers streaming fresh windows
  legs128      the same with LT_DEVICE_CACHE_GB=128 (round 4's cache: nothing goes back to the driver)
  early        the windows of the measurement allocated BEFORE the four trackers run (same memory traffic, older pages)
  noprocess    legs without the process() / process_batch part

Each child prints its rate, AnonHugePages of the process and what share of the measured windows sits in huge pages.

    python tools/stream_regress.py [--size 1280x720] [--repeat 2]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("LT_PKG_ROOT"):               # another tree of the package first (a round-4 export: only the mode "plain" exists there)
    sys.path.insert(0, os.environ["LT_PKG_ROOT"])


def huge_share(arrs):
    """Share of the arrays' bytes that /proc/self/smaps reports as AnonHugePages in the mappings that hold them."""
    spans = sorted((a.ctypes.data, a.ctypes.data + a.nbytes) for a in arrs)
    total = huge = 0
    cur = None
    for line in open("/proc/self/smaps"):
        p = line.split()
        if "-" in p[0] and len(p) >= 5 and ":" not in p[0]:
            try:
                lo, hi = (int(v, 16) for v in p[0].split("-"))
            except ValueError:
                continue
            cur = (lo, hi) if any(s < hi and e > lo for s, e in spans) else None
        elif cur and p[0] == "Rss:":
            total += int(p[1])
        elif cur and p[0] == "AnonHugePages:":
            huge += int(p[1])
    return round(huge / total, 3) if total else None


def anon_huge_mb():
    for line in open("/proc/self/smaps_rollup"):
        if line.startswith("AnonHugePages:"):
            return int(line.split()[1]) // 1024
    return None


def child(a):
    import bench
    from lane_tracker_amd import calib
    from lane_tracker_amd.lane_tracker import LaneTracker
    window, nwin = 256, 8
    base = bench.render_streams(96)[a.size]
    cal = calib.reference_calibration() if a.size == "1280x720" else calib.scaled_calibration(1.5)
    frames = bench.stream_windows(base, window, 1)[0]
    lt = LaneTracker(**cal)

    def rate(ws, tracker=lt):
        t0 = time.perf_counter()
        for _ in tracker.process_stream(ws, annotate=False):
            pass
        return len(ws) * window / (time.perf_counter() - t0)
    cold = bench.stream_windows(base, window, nwin) if a.mode == "early" else None
    if a.mode in ("legs", "legs128", "early"):
        for f in frames[:64]:
            lt.process(f)
        for ann in (False, True):
            for _ in range(3):
                lt.process_batch(frames, annotate=ann)
    if a.mode in ("legs", "legs128", "early", "noprocess"):
        for k in range(4):
            ws = bench.stream_windows(base, window, nwin)
            fresh = LaneTracker(**cal)
            try:
                if k < 3:
                    fresh.warm(window, False)
                rate(ws, fresh)
            finally:
                fresh.close()
            del ws
    if cold is None:
        cold = bench.stream_windows(base, window, nwin)
    first = rate(cold)
    best = max(rate(cold + cold) for _ in range(3))
    print("RESULT " + json.dumps({"mode": a.mode, "first_pass": round(first), "later_pass_best_of_3": round(best),
                                  "anon_huge_mb": anon_huge_mb(), "huge_share_of_windows": huge_share(cold),
                                  "slots": lt._ctx.capacity if hasattr(lt._ctx, "capacity") else None}), flush=True)
    lt.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--mode", default="plain")
    ap.add_argument("--size", default="1280x720")
    ap.add_argument("--repeat", type=int, default=2)
    ap.add_argument("--modes", default="plain,legs,legs128,early,noprocess")
    a = ap.parse_args()
    if a.child:
        return child(a)
    for r in range(a.repeat):
        for mode in a.modes.split(","):
            env = dict(os.environ)
            if mode == "legs128":
                env["LT_DEVICE_CACHE_GB"] = "128"
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--mode", mode, "--size", a.size], env=env,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            got = [l for l in p.stdout.split("\n") if l.startswith("RESULT ")]
            print(got[0] if got else "FAILED %s rc %d: %s" % (mode, p.returncode, p.stderr[-400:]), flush=True)


if __name__ == "__main__":
    main()
