O=gpurun_out/r05c; mkdir -p $O
python -m pytest tests/test_gpu_tracker.py tests/test_gpu_overlay.py tests/test_gpu_chain.py tests/test_gpu_memory.py tests/test_gpu_streams.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
python tools/process_trace.py > $O/process_trace.log 2>&1
python tools/process_trace.py x >> $O/process_trace.log 2>&1
for args in "--size 1280x720" "--size 1280x720 --warm" "--size 1280x720 --annotate" "--size 1280x720 --annotate --warm" "--size 1920x1080 --annotate" "--size 1920x1080 --annotate --warm"; do
  echo "== $args" >> $O/cold.log
  timeout 300 python tools/cold_start.py $args --events 70 >> $O/cold.log 2>> $O/cold.err
done
timeout 900 python tools/numa_probe.py 2 > $O/numa.log 2>&1
