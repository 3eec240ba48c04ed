O=gpurun_out/r05D; mkdir -p $O
for rep in 1 2; do
for cfg in "X=1" "LT_UPLOAD_ENQUEUE=1"; do
  echo "$cfg" >> $O/process.log
  env $cfg timeout 120 python tools/process_trace.py >> $O/process.log 2>&1
  env $cfg timeout 120 python tools/process_trace.py 1.5 >> $O/process.log 2>&1
done; done
LT_OVERLAY_TIMING=1 timeout 120 python tools/process_trace.py 2> $O/overlay_timing.txt > /dev/null
tail -40 $O/overlay_timing.txt > $O/overlay_timing_tail.txt; rm $O/overlay_timing.txt
