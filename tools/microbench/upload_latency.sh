#!/bin/bash
# bash tools/microbench/upload_latency.sh  (on the GPU box, from the repo root) -> gpurun_out/upload_latency.txt
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/upload_latency.txt
/opt/rocm/bin/hipcc -O3 -mavx2 --offload-arch=gfx950 $root/tools/microbench/upload_latency.hip -o /tmp/upload_latency -lpthread || exit 1
: > $out
for cold in "" 1; do
for b in 913920 2073600; do
for m in engine2d engine2d1 engine1d pinned; do COLD=$cold timeout 60 /tmp/upload_latency $m 1 $b >> $out 2>&1 || echo "rc=$? $m" >> $out; done
for t in 1 2 4 6; do COLD=$cold timeout 60 /tmp/upload_latency staged $t $b >> $out 2>&1 || echo "rc=$? staged $t" >> $out; done
for t in 1 2 3 4 6 8; do COLD=$cold timeout 60 /tmp/upload_latency bar $t $b >> $out 2>&1 || echo "rc=$? bar $t" >> $out; done
done; done
cat $out
