#!/bin/bash
# Per-kernel durations inside LaneTracker.process() (one frame per call): rocprofv3 kernel trace of tools/process_trace.py
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/proc_k; rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 $root/tools/process_trace.py > $out/run.log 2>&1
echo rc=$?
python3 - $out <<'PY'
import csv, glob, sys
for p in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = sorted(csv.DictReader(open(p)), key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:22]:
        print("%-90s calls %4s avg %8.1f us" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
find $out -name "*.csv" -size +5M -delete
