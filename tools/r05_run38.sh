O=gpurun_out/r05L; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
for k in 1 2 3; do timeout 900 python bench.py > $O/bench$k.json 2> $O/bench$k.err; echo rc $? >> $O/bench$k.err; done
timeout 120 python tools/process_trace.py > $O/process.jsonl 2>&1
timeout 120 python tools/process_trace.py 1.5 >> $O/process.jsonl 2>&1
timeout 120 python tools/process_kernels.py >> $O/process.jsonl 2>&1
timeout 3000 python -m pytest tests -x -q -m gpu > $O/tests_full.log 2>&1; echo rc $? >> $O/tests_full.log
