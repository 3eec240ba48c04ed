#!/bin/bash
# What bounds a step of the one-frame 55x55 top-hat walk (a single wave per SIMD, 32 dependent row pairs): variant builds of
# k_tophat.hip with one ingredient of the step removed (results WRONG, timing only), one frame through tools/process_kernels.py.
#   bash tools/tophat_probe.sh        (on the GPU box, from the repo root)
# the probe blocks (kernels with WRONG results) live in tools/probes/*.patch, not in the product sources: a patched copy of the
# one file is compiled here
mkdir -p /tmp/lt_probe_src && cp lane_tracker_amd/csrc/k_tophat.hip /tmp/lt_probe_src/ && patch -s /tmp/lt_probe_src/k_tophat.hip tools/probes/k_tophat_probes.patch || exit 1
cd lane_tracker_amd/csrc
F="-O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -fvisibility-inlines-hidden --offload-arch=gfx950"
OBJS="lt_api.o lt_memory.o lt_present.o lt_chain.o lt_gather.o lt_tables.o k_frontend.o k_filter.o k_threshold.o k_threshold_walk.o k_adaptive_walk.o k_search.o k_overlay.o"
for v in base NO_ACCUM NO_CHAIN "NO_ACCUM -DLT_PROBE_NO_CHAIN"; do
  tag=$(echo $v | tr -d ' -' )
  D=""; [ "$v" != base ] && D="-DLT_PROBE_$v"
  /opt/rocm/bin/hipcc $F $D -I. -c /tmp/lt_probe_src/k_tophat.hip -o /tmp/kt_$tag.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o /tmp/libprobe_$tag.so $OBJS /tmp/kt_$tag.o || exit 1
  echo "== $v"
  (cd ../.. && LANE_TRACKER_AMD_LIB=/tmp/libprobe_$tag.so python tools/process_kernels.py)
done
