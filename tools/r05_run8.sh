O=gpurun_out/r05h; mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_tracker.py tests/test_gpu_chain.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
for v in 1 0 1 0; do echo "LT_MORPH_PF=$v" >> $O/process_kernels.log; LT_MORPH_PF=$v python tools/process_kernels.py >> $O/process_kernels.log 2>&1; LT_MORPH_PF=$v python tools/process_kernels.py x >> $O/process_kernels.log 2>&1; done
for v in 1 0 1 0; do echo "LT_MORPH_PF=$v" >> $O/process_trace.log; LT_MORPH_PF=$v python tools/process_trace.py >> $O/process_trace.log 2>&1; LT_MORPH_PF=$v python tools/process_trace.py x >> $O/process_trace.log 2>&1; done
python tools/annot_probe.py 1280x720 3 > $O/annot_720.log 2>&1
python tools/annot_probe.py 1920x1080 3 > $O/annot_1080.log 2>&1
make -C tools/_r4/lane_tracker_amd/csrc -s -j8 > $O/r4_build.log 2>&1
for i in 1 2 3; do LT_PKG_ROOT=$PWD/tools/_r4 timeout 300 python tools/close_hang.py --cache-gb 32 --limit 60 > $O/close_hang_r4_$i.log 2>&1; done
