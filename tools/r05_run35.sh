O=gpurun_out/r05I; mkdir -p $O
for rep in 1 2; do
for cfg in "X=1" "LT_UPLOAD_SYNC=1"; do
  echo "$cfg" >> $O/process.log
  env $cfg timeout 120 python tools/process_trace.py >> $O/process.log 2>&1
  env $cfg timeout 120 python tools/process_trace.py 1.5 >> $O/process.log 2>&1
done; done
timeout 200 python tools/process_throttle_probe.py 1280x720 3.0 >> $O/probe.log 2>&1
LT_UPLOAD_SYNC=1 timeout 200 python tools/process_throttle_probe.py 1280x720 3.0 >> $O/probe.log 2>&1
timeout 200 python tools/process_throttle_probe.py 1920x1080 3.0 >> $O/probe.log 2>&1
LT_UPLOAD_SYNC=1 timeout 200 python tools/process_throttle_probe.py 1920x1080 3.0 >> $O/probe.log 2>&1
timeout 3000 python -m pytest tests -x -q -m gpu > $O/tests_full.log 2>&1; echo rc $? >> $O/tests_full.log
