O=gpurun_out/r05w; mkdir -p $O
timeout 300 python tools/annot_cprofile.py 1280x720 True > $O/cprof_720_true.txt 2>&1
timeout 300 python tools/annot_cprofile.py 1280x720 inplace > $O/cprof_720_inplace.txt 2>&1
timeout 300 python tools/annot_cprofile.py 1920x1080 True > $O/cprof_1080_true.txt 2>&1
LT_STRIP_PIECE=16 timeout 300 python tools/annot_cprofile.py 1280x720 True > $O/cprof_720_true_p16.txt 2>&1
LT_STRIP_PIECE=64 timeout 300 python tools/annot_cprofile.py 1280x720 True > $O/cprof_720_true_p64.txt 2>&1
LT_STRIP_PIECE=256 timeout 300 python tools/annot_cprofile.py 1280x720 True > $O/cprof_720_true_p256.txt 2>&1
timeout 100 python tools/process_trace.py 1.0 bench > $O/process_bench.log 2>&1
timeout 100 python tools/process_trace.py 1.5 bench >> $O/process_bench.log 2>&1
timeout 900 python -m pytest tests/test_gpu_chain.py -x -q -m gpu -k "inplace or process_stream_equals or annotated or process_batch_chained" > $O/tests_a.log 2>&1; echo rc $? >> $O/tests_a.log
