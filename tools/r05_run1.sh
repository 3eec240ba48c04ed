mkdir -p gpurun_out/r05a
O=gpurun_out/r05a
python -m pytest tests/test_gpu_tracker.py tests/test_gpu_memory.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
for args in "--size 1280x720" "--size 1280x720 --prefix bench" "--size 1280x720 --annotate" "--size 1920x1080" "--size 1920x1080 --annotate" "--size 1280x720 --annotate --prefix bench"; do
  echo "== $args" >> $O/cold.log
  timeout 300 python tools/cold_start.py $args --events 60 >> $O/cold.log 2>> $O/cold.err
done
timeout 500 python tools/close_hang.py --cache-gb 32 --limit 60 > $O/close_hang.log 2>&1
cp gpurun_out/close_hang_stderr.log $O/ 2>/dev/null
python tools/process_kernels.py > $O/process_kernels.log 2>&1
python tools/process_kernels.py x >> $O/process_kernels.log 2>&1
python tools/process_trace.py > $O/process_trace.log 2>&1
python tools/process_trace.py x >> $O/process_trace.log 2>&1
