O=gpurun_out/r05e; mkdir -p $O
python -m pytest tests/test_gpu_chain.py tests/test_gpu_tracker.py tests/test_gpu_overlay.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
python tools/annot_probe.py 1280x720 3 > $O/annot_720.log 2>&1
python tools/annot_probe.py 1920x1080 3 > $O/annot_1080.log 2>&1
LT_COPY_THREADS=12 python tools/annot_probe.py 1920x1080 3 > $O/annot_1080_t12.log 2>&1
LT_COPY_THREADS=4 python tools/annot_probe.py 1920x1080 3 > $O/annot_1080_t4.log 2>&1
LT_HOST_TEXT=0 python tools/annot_probe.py 1920x1080 3 > $O/annot_1080_rowruns.log 2>&1
LT_HOST_TEXT=0 python tools/annot_probe.py 1280x720 3 > $O/annot_720_rowruns.log 2>&1
