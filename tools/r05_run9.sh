O=gpurun_out/r05i; mkdir -p $O
for m in 0 1 2 3 0 2; do echo "LT_UPLOAD1=$m" >> $O/process_upload.log; LT_UPLOAD1=$m python tools/process_trace.py >> $O/process_upload.log 2>&1; LT_UPLOAD1=$m python tools/process_trace.py x >> $O/process_upload.log 2>&1; done
python tools/annot_probe.py 1280x720 3 > $O/annot_720.log 2>&1
python tools/annot_probe.py 1920x1080 3 > $O/annot_1080.log 2>&1
make -C tools/_r4/lane_tracker_amd/csrc -s -j8 > $O/r4_build.log 2>&1
for i in 1 2 3 4 5 6; do LT_PKG_ROOT=$PWD/tools/_r4 timeout 300 python tools/close_hang.py --cache-gb 32 --limit 45 > $O/close_hang_r4_$i.log 2>&1; cp gpurun_out/close_hang_stderr.log $O/close_hang_r4_stderr_$i.log; done
