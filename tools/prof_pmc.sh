#!/bin/bash
# usage: tools/prof_pmc.sh <tag>   (run on the GPU box from the repo root) -- separate passes per counter set
set -u
tag=${1:-pmc}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { # name, counters...
  name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o $name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --streams 1 > $out/$name.log 2>&1
  echo "== $name rc=$?"
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $out/$name > $out/$name.summary.txt 2>&1
  cat $out/$name.summary.txt
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
