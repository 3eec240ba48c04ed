#!/bin/bash
# A/B on the GPU box of the issue-priority switch of the 55x55 top-hat kernels (k_tophat.hip: LT_MORPH_PRIO / LT_MORPH_PRIO_LEVEL):
# builds variant libraries under /tmp and runs bench.py's one-stream timing pass on each.  usage: tools/ab_prio.sh
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root/lane_tracker_amd/csrc
F="-O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fvisibility-inlines-hidden"
build() { hipcc $F $2 --offload-arch=gfx950 -c k_tophat.hip -o /tmp/k_tophat_$1.o && hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o /tmp/libprio_$1.so lt_api.o lt_memory.o lt_present.o lt_chain.o lt_gather.o lt_tables.o k_frontend.o k_filter.o /tmp/k_tophat_$1.o k_threshold.o k_threshold_walk.o k_search.o k_overlay.o; }
build off "-DLT_MORPH_PRIO=0"; build l2 "-DLT_MORPH_PRIO=1 -DLT_MORPH_PRIO_LEVEL=2"; build rev "-DLT_MORPH_PRIO=2"; build all "-DLT_MORPH_PRIO=5"
cd $root
run() { python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-host-fed --no-stream 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernels_ms_per_step', {})
print('%-9s value %8.0f  stage %.4f ms ' % ('$1', d['value'], d['roofline']['stage_ms_per_launch']), {a: round(b, 4) for a, b in k.items() if 'erode' in a or 'tophat' in a})"; }
for r in 1 2; do
  run product
  for v in off l2 rev all; do LANE_TRACKER_AMD_LIB=/tmp/libprio_$v.so run $v; done
done
