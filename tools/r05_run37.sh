O=gpurun_out/r05K; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_tracker.py tests/test_gpu_overlay.py -x -q -m gpu > $O/tests_a.log 2>&1; echo rc $? >> $O/tests_a.log
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_memory.py -x -q -m gpu > $O/tests_b.log 2>&1; echo rc $? >> $O/tests_b.log
timeout 200 python tools/process_throttle_probe.py 1280x720 3.0 >> $O/probe.log 2>&1
timeout 200 python tools/process_throttle_probe.py 1920x1080 3.0 >> $O/probe.log 2>&1
bash tools/process_timeline.sh final > $O/tl.log 2>&1
cp gpurun_out/ptl_final/timeline.txt $O/timeline_final.txt
find gpurun_out -name "*.csv" -path "*ptl_*" -delete
