#!/bin/bash
# rocprofv3 kernel + copy timeline of process() (run on the GPU box from the repo root): tools/process_timeline.sh <tag> [env assignments]
root=${GRAFT_REPO_ROOT:-$(pwd)}
tag=$1; shift
out=$root/gpurun_out/ptl_$tag
rm -rf $out; mkdir -p $out
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -o t -- python3 $root/tools/process_loop.py > $out/run.log 2>&1
echo rc=$?; tail -1 $out/run.log
python3 $root/tools/process_timeline.py $out > $out/timeline.txt 2>&1
find $out -name "*.csv" -size +20M -delete
