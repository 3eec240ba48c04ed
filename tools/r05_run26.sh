O=gpurun_out/r05z; mkdir -p $O
timeout 200 python tools/process_throttle_probe.py 1920x1080 3.0 >> $O/probe.log 2>&1
timeout 200 python tools/process_throttle_probe.py 1280x720 3.0 >> $O/probe.log 2>&1
for k in 1 2; do timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench$k.json 2> $O/bench$k.err; echo rc $? >> $O/bench$k.err; done
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_tracker.py tests/test_gpu_memory.py tests/test_gpu_overlay.py -x -q -m gpu > $O/tests_a.log 2>&1; echo rc $? >> $O/tests_a.log
