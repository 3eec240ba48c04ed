#!/bin/bash
# A/B of a kernel switch: per-kernel ms of bench.py's serial timing pass with the environment variable off / on.
# usage: tools/ab_probe.sh VAR [steps]
V=${1:-LT_XCD_REMAP}; k=${2:-20}
for r in 0 1 0 1; do
  echo "$V=$r"
  env $V=$r timeout 200 python bench.py --steps $k --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print(d['value'], {k: round(v, 3) for k, v in d['kernels_ms_per_step'].items()})"
done
