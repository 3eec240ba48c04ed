O=gpurun_out/r05o; mkdir -p $O
timeout 900 python tools/stream_regress.py --repeat 2 > $O/stream_regress.log 2>&1
echo rc $? >> $O/stream_regress.log
