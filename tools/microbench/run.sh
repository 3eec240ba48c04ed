#!/bin/bash
# Calibration run on the GPU box (from the repo root):  bash tools/microbench/run.sh
# Builds the two microbenchmarks, writes gpurun_out/mb/valu_issue.json and the FETCH_SIZE / WRITE_SIZE counter passes of
# fetch_calib (separate --pmc passes, no tracing options beside them), then the per-pattern correction factors.
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/mb
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 $root/tools/microbench/valu_issue.hip -o /tmp/valu_issue || exit 1
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 $root/tools/microbench/fetch_calib.hip -o /tmp/fetch_calib || exit 1
timeout 240 /tmp/valu_issue 2000 > $out/valu_issue.json; echo "valu_issue rc=$?"
python3 $root/tools/microbench/valu_table.py $out/valu_issue.json $out/valu_issue_summary.json > $out/valu_issue_table.txt; cat $out/valu_issue_table.txt
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o fetch -- /tmp/fetch_calib > $out/fetch.log 2>&1; echo "fetch rc=$?"
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o write -- /tmp/fetch_calib > $out/write.log 2>&1; echo "write rc=$?"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o trace -- /tmp/fetch_calib > $out/trace.log 2>&1; echo "trace rc=$?"
python3 $root/tools/microbench/calib_summary.py $out > $out/fetch_calib.json; cat $out/fetch_calib.json
