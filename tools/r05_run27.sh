O=gpurun_out/r05A; mkdir -p $O
for s in 1280x720 1920x1080; do
  timeout 1200 python tools/annot_ab.py $s True "" "LT_COPY_SPINNERS=1" "LT_STRIP_PIECE=256" "LT_COPY_SPINNERS=1 LT_STRIP_PIECE=256" "LT_COPY_SPIN_US=0" "LT_COPY_THREADS=12" >> $O/ab.log 2>&1
done
timeout 600 python tools/annot_ab.py 1280x720 inplace "" "LT_COPY_SPINNERS=1" >> $O/ab.log 2>&1
