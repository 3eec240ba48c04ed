#!/bin/bash
# Kernel trace of the bench on one stream (run on the GPU box from the repo root): tools/trace.sh <tag> [bench args]
# -> gpurun_out/<tag>/t_kernel_stats.csv ; prints the per-kernel averages
set -u
tag=${1:-trace}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-fed --streams 1 "$@" > $out/trace.log 2>&1
echo "trace rc=$?"
python3 - "$out" <<'PY'
import csv, glob, sys
for p in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(p)))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    for r in rows[:24]:
        print("%-100s calls %4s avg %9.1f us  %5s%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
