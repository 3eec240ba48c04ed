#!/bin/bash
# sweep the band count of each top-hat kernel (single stream, 256 frames); prints kernel ms per setting
for k in 29E 29D 55E 55D; do
  for nb in 2 3 4 5 6 7 8 10; do
    r=$(env LT_MORPH_NB_$k=$nb timeout 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --streams 1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())['kernels_ms_per_step']; print(d['erode_r29'], d['tophat_r29'], d['erode_b55'], d['tophat_b55'])")
    echo "$k nb=$nb : $r"
  done
done
