O=gpurun_out/r05M; mkdir -p $O
for k in 1 2 3; do timeout 900 python bench.py > $O/bench$k.json 2> $O/bench$k.err; echo rc $? >> $O/bench$k.err; done
timeout 200 python tools/process_throttle_probe.py 1280x720 3.0 >> $O/probe.log 2>&1
timeout 200 python tools/process_throttle_probe.py 1920x1080 3.0 >> $O/probe.log 2>&1
