O=gpurun_out/r05t; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q -m gpu -k "inplace or process_stream_equals or annotated" > $O/tests_a.log 2>&1; echo rc $? >> $O/tests_a.log
timeout 900 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; echo rc $? >> $O/bench.err
timeout 3000 python -m pytest tests -x -q -m gpu > $O/tests_full.log 2>&1; echo rc $? >> $O/tests_full.log
