#!/bin/bash
# Band-count sweep of the four top-hat launches on the bench shape (run on the GPU box): stage times per band count
for nb in 2 3 4 5 6 8; do
  echo "bands=$nb $(LT_MORPH_NB_29E=$nb LT_MORPH_NB_29D=$nb LT_MORPH_NB_55E=$nb LT_MORPH_NB_55D=$nb python tools/bench_kernels.py | cut -d' ' -f1-2,9-)"
done
