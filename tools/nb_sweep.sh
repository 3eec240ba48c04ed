#!/bin/bash
# Band-count sweep of the top-hat launches on the bench shape (GPU box, repo root): kernel ms per band count.
# usage: tools/nb_sweep.sh [kernels: 29E 29D 55E 55D ...]   (default: all four together)
ks=${*:-"29E 29D 55E 55D"}
for nb in 2 3 4 5 6 8 10; do
  env=""
  for k in $ks; do env="$env LT_MORPH_NB_$k=$nb"; done
  r=$(env $env timeout 120 python3 bench.py --only-settings --streams 1 | python3 -c "
import sys, json; d = json.load(sys.stdin)['process_defaults']['kernels_ms']; print(d['erode_r29'], d['tophat_r29'], d['erode_b55'], d['tophat_b55'])")
  echo "bands=$nb ($ks): $r"
done
