#!/bin/bash
# Are the two partial planes the vertical tasks of the bilateral walks write what the walks wait for?  Variant
# build of k_threshold_walk.hip without those stores (results WRONG, timing only) through bench.py's serial timing pass, 256 frames
# per launch.      bash tools/walk_probe.sh        (on the GPU box, from the repo root)
# the probe blocks (kernels with WRONG results) live in tools/probes/*.patch, not in the product sources: a patched copy of the
# one file is compiled here
mkdir -p /tmp/lt_probe_src && cp lane_tracker_amd/csrc/k_threshold_walk.hip /tmp/lt_probe_src/ && patch -s /tmp/lt_probe_src/k_threshold_walk.hip tools/probes/k_threshold_walk_probes.patch || exit 1
cd lane_tracker_amd/csrc
F="-O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fvisibility-inlines-hidden --offload-arch=gfx950"
OBJS="lt_api.o lt_memory.o lt_present.o lt_chain.o lt_gather.o lt_tables.o k_frontend.o k_filter.o k_tophat.o k_threshold.o k_adaptive_walk.o k_search.o k_overlay.o"
for v in base WALK_NO_VSTORE; do
  D=""; [ "$v" != base ] && D="-DLT_PROBE_$v"
  /opt/rocm/bin/hipcc $F $D -I. -c /tmp/lt_probe_src/k_threshold_walk.hip -o /tmp/kw_$v.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o /tmp/libwalk_$v.so $OBJS /tmp/kw_$v.o || exit 1
  for r in 1 2 3; do
    echo "== $v"
    (cd ../.. && LANE_TRACKER_AMD_LIB=/tmp/libwalk_$v.so timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-fed --no-stream --no-settings 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
k = d['kernels_ms_per_step']
print(json.dumps({'value': d['value'], 'threshold': k['threshold'], 'open5': k['open5'], 'stage_ms': d['roofline']['stage_ms_per_launch']}))")
  done
done
