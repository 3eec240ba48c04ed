#!/bin/bash
# Band count of the top-hat kernels against the THROUGHPUT of the three-stream step (85-frame launches per slice):
# the per-kernel times of the single-stream pass (nb_sweep.sh) do not see how concurrent launches share the chip.
for nb in 0 2 3 4 5 6 8; do
  if [ $nb = 0 ]; then E=""; else E="LT_MORPH_NB_29E=$nb LT_MORPH_NB_29D=$nb LT_MORPH_NB_55E=$nb LT_MORPH_NB_55D=$nb"; fi
  for rep in 1 2; do
    v=$(env $E timeout 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline | python -c "import sys,json; print(json.loads(sys.stdin.read())['value'])")
    echo "nb=$nb : $v"
  done
done
