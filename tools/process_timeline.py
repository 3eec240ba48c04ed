#!/usr/bin/env python3
"""Device timeline of one process() frame from a rocprofv3 --kernel-trace --memory-copy-trace run of tools/process_loop.py:
frames are cut at the dispatches of k_undistort_rows; per kernel / copy (in dispatch order) the median start and end in us from
the start of the frame's undistortion, over the last frames of the run.  usage: process_timeline.py <dir> [undistortions per frame]
(2 for a frame that takes both tries: tools/config1_loop.py)"""
import collections, csv, glob, statistics, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][-56:], r.get("Stream_Id", r.get("Queue_Id", "?"))))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy " + r.get("Direction", "?"), "-"))
ev.sort()
starts = [i for i, e in enumerate(ev) if "k_undistort_rows" in e[2]][::int(sys.argv[2]) if len(sys.argv) > 2 else 1]
frames = []
for a, b in zip(starts[:-1], starts[1:]):
    # the upload of a frame precedes its undistortion: attach the copies between the previous frame's last kernel and this start
    frames.append(ev[a:b])
frames = frames[-120:]
shape = collections.Counter(tuple(e[2] for e in fr) for fr in frames).most_common(1)[0][0]
sel = [fr for fr in frames if tuple(e[2] for e in fr) == shape]
print("frames: %d of %d with the common sequence of %d events; frame period median %.1f us" % (
    len(sel), len(frames), len(shape), statistics.median((b[0][0] - a[0][0]) / 1e3 for a, b in zip(frames[:-1], frames[1:]))))
for k, name in enumerate(shape):
    s = statistics.median((fr[k][0] - fr[0][0]) / 1e3 for fr in sel)
    e = statistics.median((fr[k][1] - fr[0][0]) / 1e3 for fr in sel)
    q = collections.Counter(fr[k][3] for fr in sel).most_common(1)[0][0]
    print("%8.1f %8.1f  %6.1f us  stream %-4s %s" % (s, e, e - s, q, name))
