O=gpurun_out/r05r; mkdir -p $O
bash tools/process_timeline.sh one4 > $O/tl.log 2>&1
bash tools/process_timeline.sh one0 LT_MORPH_ONE=0 >> $O/tl.log 2>&1
bash tools/process_timeline.sh one8 LT_MORPH_ONE=8 LT_MORPH_ONE_WGS=256 >> $O/tl.log 2>&1
cp gpurun_out/ptl_one4/timeline.txt $O/timeline_one4.txt; cp gpurun_out/ptl_one0/timeline.txt $O/timeline_one0.txt; cp gpurun_out/ptl_one8/timeline.txt $O/timeline_one8.txt
find gpurun_out -name "*.csv" -path "*ptl_*" -size +8M -delete
timeout 120 python tools/process_trace.py > $O/process.log 2>&1
