O=gpurun_out/r05H; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_overlay.py tests/test_gpu_tracker.py -x -q -m gpu > $O/tests_a.log 2>&1; echo rc $? >> $O/tests_a.log
timeout 900 python -m pytest tests/test_gpu_chain.py -x -q -m gpu > $O/tests_b.log 2>&1; echo rc $? >> $O/tests_b.log
for rep in 1 2; do
for cfg in "X=1" "LT_LANE_DEVICE=0"; do
  echo "$cfg" >> $O/process.log
  env $cfg timeout 120 python tools/process_trace.py >> $O/process.log 2>&1
  env $cfg timeout 120 python tools/process_trace.py 1.5 >> $O/process.log 2>&1
done; done
timeout 200 python tools/process_throttle_probe.py 1280x720 3.0 >> $O/probe.log 2>&1
LT_LANE_DEVICE=0 timeout 200 python tools/process_throttle_probe.py 1280x720 3.0 >> $O/probe.log 2>&1
timeout 200 python tools/process_throttle_probe.py 1920x1080 3.0 >> $O/probe.log 2>&1
LT_LANE_DEVICE=0 timeout 200 python tools/process_throttle_probe.py 1920x1080 3.0 >> $O/probe.log 2>&1
bash tools/process_timeline.sh devlane > $O/tl.log 2>&1
cp gpurun_out/ptl_devlane/timeline.txt $O/timeline_devlane.txt
find gpurun_out -name "*.csv" -path "*ptl_*" -delete
