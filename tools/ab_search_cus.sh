#!/bin/bash
# A/B on the GPU box: the batch step with the searches on the slots' streams (default) and on CUs set aside for them
# (LT_BENCH_SEARCH_CUS=n LT_SEARCH_ON_RESERVED=1).  Prints value / ms_per_step per variant, three rounds interleaved.
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
run() { python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-host-fed --no-stream 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['value'], d['ms_per_step'], d.get('single_copy_frames_per_s'))"; }
for r in 1 2 3; do
  run default
  LT_BENCH_SEARCH_CUS=8 LT_SEARCH_ON_RESERVED=1 run cus8_routed
  LT_BENCH_SEARCH_CUS=2 LT_SEARCH_ON_RESERVED=1 run cus2_routed
  LT_BENCH_SEARCH_CUS=8 run cus8_not_routed
done
