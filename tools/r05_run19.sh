O=gpurun_out/r05s; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_and_two or mask_chain or one_frame or filter_lane_points or alternative" > $O/tests_a.log 2>&1; echo rc $? >> $O/tests_a.log
timeout 900 python -m pytest tests/test_gpu_tracker.py tests/test_gpu_chain.py -x -q -m gpu > $O/tests_b.log 2>&1; echo rc $? >> $O/tests_b.log
for cfg in "X=1" "LT_SIDE_FIRST=1" "LT_OPEN_SMALL=0" "LT_THRESHOLD_PHASES=0" "LT_MORPH_ONE=0" "LT_MORPH_ONE=8 LT_MORPH_ONE_WGS=256"; do
  echo "$cfg" >> $O/process.log
  env $cfg timeout 120 python tools/process_trace.py >> $O/process.log 2>&1
  echo "$cfg" >> $O/kernels.log
  env $cfg timeout 120 python tools/process_kernels.py >> $O/kernels.log 2>&1
done
timeout 120 python tools/process_trace.py 1.5 >> $O/process.log 2>&1
bash tools/process_timeline.sh now > $O/tl.log 2>&1
cp gpurun_out/ptl_now/timeline.txt $O/timeline_now.txt
find gpurun_out -name "*.csv" -path "*ptl_*" -delete
