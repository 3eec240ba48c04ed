O=gpurun_out/r05u; mkdir -p $O
timeout 900 python bench.py --steps 10 --warmup 2 --no-settings --no-cpu-baseline --no-host-fed > $O/bench.json 2> $O/bench.err; echo rc $? >> $O/bench.err
for s in "" 1.5; do timeout 100 python tools/process_trace.py $s >> $O/process.log 2>&1; done
