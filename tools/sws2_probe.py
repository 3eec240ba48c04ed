#!/usr/bin/env python3
"""k_sws_fit2 phase by phase for ONE frame: a build of k_search.hip with -DLT_SWS2_PROBE prints 100 MHz ticks per phase
(A band sums -> LDS, B the recurrence over the levels, C windows -> row masks, fit).
  cd lane_tracker_amd/csrc && make && hipcc <the Makefile's flags> -DLT_SWS2_PROBE --offload-arch=gfx950 -c k_search.hip -o /tmp/k_search.o \
    && hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=exports.map -o ../liblane_tracker_amd_sws2probe.so <the other .o files> /tmp/k_search.o
  LANE_TRACKER_AMD_LIB=lane_tracker_amd/liblane_tracker_amd_sws2probe.so python tools/sws2_probe.py
Round 6: A 4.8 us, B 39 -> 28 us (two waves, one per side; one-pass scan), C 7.5 us, fit 4.5 us."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import _native, calib, synth
cal = calib.reference_calibration()
frames = synth.stream_lanes(4, seed=5, cal=cal)
ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], capacity=2)
for k in range(4):
    ctx.upload_frames(frames[k:k + 1], first=0)
    ctx.mask_run(1, first=0)
    ctx.sws_fit_run(1, first=0)
    print(ctx.download_records(1, first=0)["detected"])
ctx.close()
