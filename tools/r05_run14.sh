O=gpurun_out/r05n; mkdir -p $O
bash tools/front_probe.sh > $O/front_probe.log 2>&1
bash tools/prof_round.sh r05 38d1f81 > $O/prof_round.log 2>&1
