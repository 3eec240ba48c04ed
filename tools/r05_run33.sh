O=gpurun_out/r05G; mkdir -p $O
LT_R_THRESHOLD_MAIN=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "one_and_two or mask_chain_bit_exact" > $O/tests_a.log 2>&1; echo rc $? >> $O/tests_a.log
for rep in 1 2 3; do
for cfg in "X=1" "LT_R_THRESHOLD_MAIN=1"; do
  echo "$cfg" >> $O/process.log
  env $cfg timeout 120 python tools/process_trace.py >> $O/process.log 2>&1
  echo "$cfg" >> $O/kernels.log
  env $cfg timeout 120 python tools/process_kernels.py >> $O/kernels.log 2>&1
done; done
bash tools/process_timeline.sh rmain LT_R_THRESHOLD_MAIN=1 > $O/tl.log 2>&1
cp gpurun_out/ptl_rmain/timeline.txt $O/timeline_rmain.txt
find gpurun_out -name "*.csv" -path "*ptl_*" -delete
