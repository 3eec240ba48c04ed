O=gpurun_out/r05y; mkdir -p $O
s=1920x1080
timeout 200 python tools/process_throttle_probe.py $s 3.0 >> $O/probe.log 2>&1
OPENBLAS_NUM_THREADS=16 timeout 200 python tools/process_throttle_probe.py $s 3.0 >> $O/probe.log 2>&1
OPENBLAS_NUM_THREADS=1 timeout 200 python tools/process_throttle_probe.py $s 3.0 >> $O/probe.log 2>&1
LT_COPY_SPINNERS=1 timeout 200 python tools/process_throttle_probe.py $s 3.0 >> $O/probe.log 2>&1
OPENBLAS_NUM_THREADS=16 LT_COPY_SPINNERS=1 timeout 200 python tools/process_throttle_probe.py $s 3.0 >> $O/probe.log 2>&1
OPENBLAS_NUM_THREADS=16 timeout 200 python tools/process_throttle_probe.py 1280x720 3.0 >> $O/probe.log 2>&1
OPENBLAS_NUM_THREADS=16 LT_COPY_SPINNERS=1 timeout 200 python tools/process_throttle_probe.py 1280x720 3.0 >> $O/probe.log 2>&1
