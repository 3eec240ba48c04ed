O=gpurun_out/r05J; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
for k in 1 2 3; do timeout 900 python bench.py > $O/bench$k.json 2> $O/bench$k.err; echo rc $? >> $O/bench$k.err; done
bash tools/process_timeline.sh final > $O/tl.log 2>&1
cp gpurun_out/ptl_final/timeline.txt $O/timeline_final.txt
find gpurun_out -name "*.csv" -path "*ptl_*" -delete
timeout 120 python tools/process_trace.py > $O/process.jsonl 2>&1
timeout 120 python tools/process_trace.py 1.5 >> $O/process.jsonl 2>&1
timeout 120 python tools/process_kernels.py >> $O/process.jsonl 2>&1
