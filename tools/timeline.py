#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --memory-copy-trace CSV pair into a timeline of the LAST process_batch call:
per stream/queue, merged busy intervals and the gaps between them.  usage: timeline.py <dir>"""
import csv, glob, sys, collections
d = sys.argv[1]
rows = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r.get("Stream_Id", r.get("Queue_Id", "?")), r["Kernel_Name"][:40]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", r.get("Direction", "?"), r.get("Direction", "copy")))
rows.sort()
# the last window: everything after the last gap > 3 ms
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - max(r[1] for r in rows[max(0, i - 50):i]) > 3_000_000:
        cut = i
rows = rows[cut:]
t0 = rows[0][0]
print("events in the last window:", len(rows), "span %.3f ms" % ((max(r[1] for r in rows) - t0) / 1e6))
for r in rows:
    print("%8.3f %8.3f %7.1f us  %s %-6s %s" % ((r[0] - t0) / 1e6, (r[1] - t0) / 1e6, (r[1] - r[0]) / 1e3, r[2], r[3], r[4]))
