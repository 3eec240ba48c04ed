O=gpurun_out/r05C; mkdir -p $O
for k in 1 2 3; do timeout 600 python tools/annot_modes.py 1280x720 15 >> $O/modes_$k.log 2>&1; done
