#!/bin/bash
# Round profile set (GPU box, repo root): tools/prof_round.sh <tag, e.g. r04> <commit>
#   kernel trace + stats of bench.py on one stream, PMC passes (one counter group per pass), traffic table, bench line.
set -u
tag=${1:-r06}
commit=${2:-unknown}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag; mkdir -p $out
B="$root/bench.py --no-cpu-baseline --no-host-fed --no-stream --no-settings --streams 1"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $B --steps 5 --warmup 2 > $out/trace.log 2>&1; echo "trace rc=$?"
run() { name=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o $name -- python3 $B --steps 1 --warmup 1 > $out/$name.log 2>&1; echo "== $name rc=$?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES
cd $root
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/${tag}_kernel_stats.csv
python3 - $out/${tag}_kernel_stats.csv $out/${tag}_kernel_stats_summary.json $commit <<'PY'
import csv, json, re, sys
MASK = ("k_undistort_rows", "k_warp_split", "k_morph_runs", "k_bilateral_walk", "k_merge_open5")
rows = {}
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", r["Name"])
    if m:
        rows[m.group(1) + (m.group(2) or "").replace("lt::(anonymous namespace)::", "")] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6}
stage = sum(v["avg_ms"] for k, v in rows.items() if k.startswith(MASK))
json.dump({"commit": sys.argv[3], "what": "rocprofv3 --kernel-trace --stats of bench.py --steps 5 --warmup 2 --streams 1 (every dispatch 256 frames)",
           "mask_stage_ms": round(stage, 4), "kernels": rows}, open(sys.argv[2], "w"), indent=1)
print("rocprof mask stage %.4f ms" % stage)
PY
python3 tools/pmc_kernels.py $out k_ > $out/${tag}_pmc_kernels.txt
python3 tools/make_traffic.py $out $out/${tag}_traffic.json 256 $commit
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err; tail -c 400 $out/${tag}_bench.json
# the parameter sets of bench.py's settings leg (defaults, demo 1-3, second try), every dispatch 256 frames: the kernels only they use
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/strace -o t -- python3 $root/bench.py --only-settings --streams 1 > $out/strace.log 2>&1; echo "settings trace rc=$?"
srun() { name=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --output-format csv -d $out/settings/$name -o $name -- python3 $root/bench.py --only-settings --streams 1 > $out/s_$name.log 2>&1; echo "== settings $name rc=$?"; }
srun fetch FETCH_SIZE
srun write WRITE_SIZE
srun sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
srun sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES
cd $root
cp $(find $out/strace -name "*kernel_stats.csv" | head -1) $out/${tag}_settings_kernel_stats.csv
python3 tools/pmc_kernels.py $out/settings "walk_hv<65" k_adaptive_box_walk "k_merge_open5<6" "k_merge_open5<2" "true, true, true, true" > $out/${tag}_settings_pmc_kernels.txt
python3 bench.py --only-settings > $out/${tag}_settings.json 2>/dev/null
python3 tools/stream_profile.py > $out/${tag}_stream_profile.json 2>&1
# process() one frame at a time: host time per call, device time per stage
{ timeout 120 python3 tools/process_trace.py 2>/dev/null | tail -1; timeout 120 python3 tools/process_trace.py x 2>/dev/null | tail -1;
  timeout 120 python3 tools/process_kernels.py 2>/dev/null | tail -1; } > $out/${tag}_process.jsonl
# process() one frame per call: the device timeline of a frame (kernel + copy trace condensed) and the --stats summary of the same loop
bash tools/process_timeline.sh $tag > $out/ptl.log 2>&1; cp $root/gpurun_out/ptl_$tag/timeline.txt $out/${tag}_process_timeline.txt
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ptrace -o t -- python3 $root/tools/process_loop.py > $out/ptrace.log 2>&1; echo "process trace rc=$?"
cd $root
cp $(find $out/ptrace -name "*kernel_stats.csv" | head -1) $out/${tag}_process_kernel_stats.csv
rm -rf $out/ptrace
rm -rf $out/strace $out/settings
rm -rf $out/trace $out/fetch $out/write $out/sq1 $out/sq2
ls -la $out
