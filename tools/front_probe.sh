#!/bin/bash
# Bounds for a warp that stages its undistorted footprint in LDS instead of reading the undistorted rows from HBM:
# variant builds of k_frontend.hip (results WRONG, timing only) through bench.py's serial timing pass, 256 frames per launch.
#   bash tools/front_probe.sh        (on the GPU box, from the repo root)
# the probe blocks (kernels with WRONG results) live in tools/probes/*.patch, not in the product sources: a patched copy of the
# one file is compiled here
mkdir -p /tmp/lt_probe_src && cp lane_tracker_amd/csrc/k_frontend.hip /tmp/lt_probe_src/ && patch -s /tmp/lt_probe_src/k_frontend.hip tools/probes/k_frontend_probes.patch || exit 1
cd lane_tracker_amd/csrc
F="-O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fvisibility-inlines-hidden --offload-arch=gfx950"
OBJS="lt_api.o lt_memory.o lt_present.o lt_chain.o lt_gather.o lt_tables.o k_filter.o k_tophat.o k_threshold.o k_threshold_walk.o k_adaptive_walk.o k_search.o k_overlay.o"
for v in base WARP_NO_TAPS UND_NO_STORE; do
  D=""; [ "$v" != base ] && D="-DLT_PROBE_$v"
  /opt/rocm/bin/hipcc $F $D -I. -c /tmp/lt_probe_src/k_frontend.hip -o /tmp/kf_$v.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o /tmp/libfront_$v.so $OBJS /tmp/kf_$v.o || exit 1
  for r in 1 2; do
    echo "== $v"
    (cd ../.. && LANE_TRACKER_AMD_LIB=/tmp/libfront_$v.so timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-fed --no-stream --no-settings 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernels_ms_per_step']
print(json.dumps({'value': d['value'], 'undistort_rows': k['undistort_rows'], 'warp_split': k['warp_split'], 'stage_ms': d['roofline']['stage_ms_per_launch']}))")
  done
done
