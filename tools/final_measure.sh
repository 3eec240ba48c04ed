#!/bin/bash
# Final measurement set of a round (run on the GPU box from the repo root): tests, bench, kernel trace, PMC passes.
# usage: tools/final_measure.sh <tag>      outputs under gpurun_out/<tag>/
set -u
tag=${1:-final}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -q -m gpu 2>&1 | tail -3 > $out/pytest_gpu.txt
timeout 600 python bench.py --steps 10 --warmup 3 > $out/bench.json 2> $out/bench.err
timeout 300 python tools/stream_bench.py > $out/stream_bench.json 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --streams 1 > $out/trace.log 2>&1
echo "trace rc=$?"
cd $GRAFT_REPO_ROOT
bash tools/prof_pmc.sh $tag/pmc > $out/pmc.log 2>&1
python3 tools/pmc_agg.py $out/pmc > $out/pmc_summary.txt 2>&1
python3 tools/make_traffic.py $out/pmc $out/traffic.json 256 >> $out/pmc.log 2>&1
cat $out/pytest_gpu.txt; cut -c1-400 $out/bench.json; tail -2 $out/pmc.log
