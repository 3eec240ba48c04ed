O=gpurun_out/r05d; mkdir -p $O
/opt/rocm/bin/hipcc -O2 --offload-arch=gfx950 tools/microbench/alloc_probe.hip -o /tmp/alloc_probe 2> /dev/null
/tmp/alloc_probe 1.5 10 > $O/alloc_probe_1.jsonl 2>&1
/tmp/alloc_probe 1.5 10 > $O/alloc_probe_2.jsonl 2>&1
for args in "--size 1280x720" "--size 1280x720 --annotate" "--size 1280x720 --annotate --warm" "--size 1920x1080 --annotate --warm"; do
  echo "== $args" >> $O/cold.log
  timeout 300 python tools/cold_start.py $args >> $O/cold.log 2>> $O/cold.err
done
/tmp/alloc_probe 1.5 10 > $O/alloc_probe_3.jsonl 2>&1
python tools/process_trace.py > $O/process_trace.log 2>&1
python tools/process_trace.py x >> $O/process_trace.log 2>&1
HSA_ENABLE_INTERRUPT=0 python tools/process_trace.py >> $O/process_trace.log 2>&1
HSA_ENABLE_INTERRUPT=0 python tools/process_trace.py x >> $O/process_trace.log 2>&1
LT_COPY_SPIN_US=0 python tools/process_trace.py x >> $O/process_trace.log 2>&1
python tools/process_trace.py x >> $O/process_trace.log 2>&1
python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
python bench.py > $O/bench.json 2> $O/bench.err
