#!/bin/bash
# A/B of a kernel switch with the HBM-side read counter: per-kernel ms (bench serial pass) and FETCH_SIZE per kernel.
# usage (GPU box, repo root): tools/ab_fetch.sh VAR [kernel-name substrings ...]
V=${1:-LT_MORPH_XCD}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
for r in 0 1; do
  echo "== $V=$r"
  env $V=$r python3 $root/bench.py --only-settings --streams 1 | python3 -c "import json,sys; print(json.load(sys.stdin)[\"process_defaults\"][\"kernels_ms\"])"
  out=$root/gpurun_out/ab_$r; mkdir -p $out
  ( cd /tmp && export TMPDIR=/tmp && export $V=$r && timeout 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -o f -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-host-fed --no-stream --no-settings --streams 1 > $out/log.txt 2>&1; echo "pmc rc=$?" )
  python3 $root/tools/pmc_kernels.py $out "$@" | grep -v "per wave"
  tail -2 $out/log.txt | cut -c1-200
  rm -rf $out
done
