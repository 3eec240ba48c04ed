#!/bin/bash
# A/B of a kernel switch with HBM-side counters: per-kernel ms (bench serial pass) and FETCH_SIZE / WRITE_SIZE per kernel.
# usage (GPU box, repo root): tools/ab_fetch.sh VAR [tag]
V=${1:-LT_MORPH_XCD}; tag=${2:-ab}
root=${GRAFT_REPO_ROOT:-$(pwd)}
for r in 0 1; do
  echo "== $V=$r"
  env $V=$r python3 $root/tools/bench_kernels.py --no-host-fed --no-stream
  out=$root/gpurun_out/${tag}_$r; mkdir -p $out
  ( cd /tmp && export TMPDIR=/tmp && export $V=$r && timeout 300 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d $out -o f -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-fed --no-stream --streams 1 > $out/log.txt 2>&1 )
  python3 $root/tools/pmc_kernels.py $out k_morph k_warp k_bilateral k_undist k_merge | grep -v "^   FETCH\|per wave"
  rm -rf $out
done
