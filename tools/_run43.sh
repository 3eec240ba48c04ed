bash tools/prof_round.sh r05 f093c64 > gpurun_out/prof_round_final.log 2>&1
O=gpurun_out/r05Q; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05Q/ptrace -o t -- python3 $GRAFT_REPO_ROOT/tools/process_loop.py > $GRAFT_REPO_ROOT/gpurun_out/r05Q/ptrace.log 2>&1
cd $GRAFT_REPO_ROOT
cp $(find gpurun_out/r05Q/ptrace -name "*kernel_stats.csv" | head -1) gpurun_out/r05Q/process_kernel_stats.csv
find gpurun_out/r05Q/ptrace -name "*.csv" -size +1M -delete
find gpurun_out/r05 -name "*.csv" -size +4M -delete
