O=gpurun_out/r05x; mkdir -p $O
cat /sys/fs/cgroup/cpu.max > $O/cpu_max.txt 2>&1; nproc >> $O/cpu_max.txt
for s in 1280x720 1920x1080; do
  timeout 200 python tools/process_throttle_probe.py $s 1.0 >> $O/probe.log 2>&1
  LT_COPY_THREADS=4 timeout 200 python tools/process_throttle_probe.py $s 1.0 >> $O/probe.log 2>&1
  LT_COPY_THREADS=2 timeout 200 python tools/process_throttle_probe.py $s 1.0 >> $O/probe.log 2>&1
  LT_COPY_SPIN_US=0 timeout 200 python tools/process_throttle_probe.py $s 1.0 >> $O/probe.log 2>&1
  LT_COPY_SPIN_US=2000 timeout 200 python tools/process_throttle_probe.py $s 1.0 >> $O/probe.log 2>&1
done
