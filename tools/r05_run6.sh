O=gpurun_out/r05f; mkdir -p $O
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag > $O/thp.txt 2>&1
python -m pytest tests/test_gpu_chain.py tests/test_gpu_tracker.py tests/test_gpu_overlay.py tests/test_gpu_streams.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
python tools/annot_probe.py 1280x720 3 > $O/annot_720.log 2>&1
python tools/annot_probe.py 1920x1080 3 > $O/annot_1080.log 2>&1
LT_COPY_NT=0 python tools/annot_probe.py 1920x1080 3 > $O/annot_1080_nont.log 2>&1
python bench.py > $O/bench.json 2> $O/bench.err
