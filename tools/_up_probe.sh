timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tracker.py tests/test_gpu_walk.py -x -q --timeout 600 2>&1 | tail -4
for i in 1 2; do
echo "default"; timeout 120 python tools/process_trace.py 2>&1 | tail -1 | cut -c1-420
echo "LT_THRESHOLD_SPLIT=0"; LT_THRESHOLD_SPLIT=0 timeout 120 python tools/process_trace.py 2>&1 | tail -1 | cut -c1-420
done
timeout 120 python tools/process_kernels.py 2>&1 | tail -3
