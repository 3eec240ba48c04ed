O=gpurun_out/r05g; mkdir -p $O
python -m pytest tests/test_gpu_tracker.py tests/test_gpu_overlay.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
for v in 1 0 1 0; do LT_OVERLAY_DIRECT=$v python tools/process_trace.py >> $O/process_direct_$v.log 2>&1; LT_OVERLAY_DIRECT=$v python tools/process_trace.py x >> $O/process_direct_$v.log 2>&1; done
bash tools/tophat_probe.sh > $O/tophat_probe.log 2>&1
python tools/annot_probe.py 1280x720 3 > $O/annot_720.log 2>&1
python tools/annot_probe.py 1920x1080 3 > $O/annot_1080.log 2>&1
