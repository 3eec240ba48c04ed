#!/bin/bash
# One-frame top-hat pair (k_morph_one_pair): waves per task and workgroups per plane, through tools/process_kernels.py's wall time
# of mask chain + band search + record for one frame (experiments build), three runs each.   bash tools/pair_sweep.sh ["w55 list"] ["w29 list"]
export LANE_TRACKER_AMD_LIB=$(pwd)/lane_tracker_amd/liblane_tracker_amd_exp.so
for w55 in ${1:-0 128 192 256 320 384}; do for w29 in ${2:-0 128 256 384}; do
  echo -n "Q=4 wgs55=$w55 wgs29=$w29  "
  for r in 1 2 3; do
  LT_PAIR_Q=4 LT_PAIR_WGS55=$w55 LT_PAIR_WGS29=$w29 timeout 60 python tools/process_kernels.py ${3:-} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['wall_us_mask_plus_band_plus_records'], end=' ')"
  done; echo
done; done
