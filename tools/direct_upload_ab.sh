mkdir -p gpurun_out/r06t
python -m pytest tests/test_gpu_direct_upload.py -x -q > gpurun_out/r06t/test.log 2>&1; tail -15 gpurun_out/r06t/test.log
for rep in 1 2; do
for sz in 1280x720 1920x1080; do
python tools/process_throttle_probe.py $sz 2 engine | cut -c1-260
python tools/process_throttle_probe.py $sz 2 | cut -c1-260
done; done 2>&1 | tee gpurun_out/r06t/ab.txt
