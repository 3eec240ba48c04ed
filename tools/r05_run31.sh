O=gpurun_out/r05E; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "enqueued or bad_filter or one_and_two or source_row" > $O/tests_a.log 2>&1; echo rc $? >> $O/tests_a.log
timeout 900 python -m pytest tests/test_gpu_tracker.py tests/test_gpu_chain.py tests/test_gpu_overlay.py -x -q -m gpu > $O/tests_b.log 2>&1; echo rc $? >> $O/tests_b.log
for rep in 1 2; do
for cfg in "X=1" "LT_UPLOAD_ENQUEUE=0"; do
  echo "$cfg" >> $O/process.log
  env $cfg timeout 120 python tools/process_trace.py >> $O/process.log 2>&1
  env $cfg timeout 120 python tools/process_trace.py 1.5 >> $O/process.log 2>&1
done; done
timeout 200 python tools/process_throttle_probe.py 1280x720 3.0 >> $O/probe.log 2>&1
timeout 200 python tools/process_throttle_probe.py 1920x1080 3.0 >> $O/probe.log 2>&1
bash tools/process_timeline.sh final > $O/tl.log 2>&1
cp gpurun_out/ptl_final/timeline.txt $O/timeline_final.txt
find gpurun_out -name "*.csv" -path "*ptl_*" -delete
