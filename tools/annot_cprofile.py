#!/usr/bin/env python3
"""Where the driving thread spends an annotated stream's time: cProfile over a later pass of process_stream(annotate=...) on
bench.py's windows (8 x 256 frames), top functions by own time and by cumulative time, in us per frame.
  python tools/annot_cprofile.py [1280x720|1920x1080] [True|inplace|False]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from lane_tracker_amd import calib
from lane_tracker_amd.lane_tracker import LaneTracker
size = sys.argv[1] if len(sys.argv) > 1 else "1280x720"
mode = sys.argv[2] if len(sys.argv) > 2 else "True"
ann = {"True": True, "False": False}.get(mode, mode)
base = bench.render_streams(96)[size]
cal = calib.reference_calibration() if size == "1280x720" else calib.scaled_calibration(1.5)
wins = bench.stream_windows(base, 256, 8)
lt = LaneTracker(**cal)
def run():
    n = 0
    for out in lt.process_stream(wins, annotate=ann):
        n += len(out)
    return n
run(); 
t0 = time.perf_counter(); n = run(); dt = time.perf_counter() - t0
print("plain timing: %.1f frames/s, %.1f us per frame" % (n / dt, dt / n * 1e6))
pr = cProfile.Profile()
pr.enable(); n = run(); pr.disable()
st = pstats.Stats(pr)
rows = []
for (fn, line, name), (cc, nc, tt, ct, callers) in st.stats.items():
    rows.append((tt, ct, nc, "%s:%d %s" % (os.path.basename(fn), line, name)))
print("by own time (us per frame, calls per frame):")
for tt, ct, nc, nm in sorted(rows, reverse=True)[:28]:
    print("  %7.2f own %7.2f cum %6.2f calls  %s" % (tt / n * 1e6, ct / n * 1e6, nc / n, nm))
print("by cumulative time:")
for tt, ct, nc, nm in sorted(rows, key=lambda r: -r[1])[:22]:
    print("  %7.2f own %7.2f cum %6.2f calls  %s" % (tt / n * 1e6, ct / n * 1e6, nc / n, nm))
lt.close()
