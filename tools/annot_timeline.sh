#!/bin/bash
# rocprofv3 kernel + memory-copy timeline of the annotated stream (run on the GPU box from the repo root)
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/annot_tl
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out -o t -- python3 $root/tools/annot_trace.py > $out/run.log 2>&1
echo rc=$?; tail -2 $out/run.log
python3 $root/tools/timeline.py $out > $out/timeline.txt 2>&1
head -3 $out/timeline.txt
find $out -name "*.csv" -size +20M -delete
