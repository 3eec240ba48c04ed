O=gpurun_out/r05p; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "morph or one_frame or mask_chain_bit_exact or filter_lane_points" > $O/tests_a.log 2>&1; echo rc $? >> $O/tests_a.log
for v in 0 4 8; do for wgs in 0 128 256 384 768 1024; do
  echo "LT_MORPH_ONE=$v WGS=$wgs" >> $O/kernels.log
  LT_MORPH_ONE=$v LT_MORPH_ONE_WGS=$wgs timeout 120 python tools/process_kernels.py >> $O/kernels.log 2>&1
  [ $v = 0 ] && break
done; done
for v in 0 4; do echo "LT_MORPH_ONE=$v" >> $O/process.log; LT_MORPH_ONE=$v timeout 120 python tools/process_trace.py >> $O/process.log 2>&1; LT_MORPH_ONE=$v timeout 120 python tools/process_trace.py 1.5 >> $O/process.log 2>&1; done
timeout 900 python -m pytest tests/test_gpu_tracker.py tests/test_gpu_chain.py -x -q -m gpu > $O/tests_b.log 2>&1; echo rc $? >> $O/tests_b.log
timeout 300 python tools/stream_regress.py --repeat 2 --modes plain > $O/regress_now.log 2>&1
LT_PKG_ROOT=$PWD/tools/_r4 timeout 300 python tools/stream_regress.py --repeat 2 --modes plain > $O/regress_r4.log 2>&1
