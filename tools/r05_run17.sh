O=gpurun_out/r05q; mkdir -p $O
hipcc -O2 --offload-arch=gfx950 tools/microbench/sync_latency.hip -o /tmp/sync_latency 2>/dev/null
for w in 5 20 100; do timeout 120 /tmp/sync_latency $w 2000 >> $O/sync_latency.log 2>&1; done
for cfg in "LT_RECORD_POLL=0" "LT_RECORD_POLL=1" "LT_MORPH_ONE=8 LT_MORPH_ONE_WGS=256" "LT_MORPH_ONE=0"; do
  echo "$cfg" >> $O/process.log
  env $cfg timeout 120 python tools/process_trace.py >> $O/process.log 2>&1
  env $cfg timeout 120 python tools/process_trace.py 1.5 >> $O/process.log 2>&1
done
echo "LT_WALK_MIN_FRAMES=0" >> $O/kernels.log; LT_WALK_MIN_FRAMES=0 timeout 120 python tools/process_kernels.py >> $O/kernels.log 2>&1
timeout 900 python -m pytest tests/test_gpu_tracker.py tests/test_gpu_chain.py tests/test_gpu_overlay.py -x -q -m gpu > $O/tests_b.log 2>&1; echo rc $? >> $O/tests_b.log
