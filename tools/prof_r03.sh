#!/bin/bash
# Round-3 profile set (GPU box, repo root): tools/prof_r03.sh <commit>
#   kernel trace + stats of bench.py on one stream, PMC passes (one counter group per pass), traffic table, bench line.
set -u
commit=${1:-unknown}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/r03; mkdir -p $out
B="$root/bench.py --no-cpu-baseline --no-host-fed --no-stream --streams 1"
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python3 $B --steps 5 --warmup 2 > $out/trace.log 2>&1; echo "trace rc=$?"
run() { name=$1; shift
  timeout 200 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o $name -- python3 $B --steps 1 --warmup 1 > $out/$name.log 2>&1; echo "== $name rc=$?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES
cd $root
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/r03_kernel_stats.csv
python3 - $out/r03_kernel_stats.csv $out/r03_kernel_stats_summary.json $commit <<'PY'
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
python3 tools/pmc_kernels.py $out k_ > $out/r03_pmc_kernels.txt
python3 tools/make_traffic.py $out $out/r03_traffic.json 256 $commit
python3 bench.py > $out/r03_bench.json 2> $out/r03_bench.err; tail -c 400 $out/r03_bench.json
python3 tools/stream_profile.py > $out/r03_stream_profile.json 2>&1
rm -rf $out/trace $out/fetch $out/write $out/sq1 $out/sq2
ls -la $out
