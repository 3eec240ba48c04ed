#!/usr/bin/env python3
"""Is the 2.25-round grid of the top-hat kernels a tail problem?  Stage times per frame for frame counts that give
2.0, 2.25, 2.5, 3.0 ... rounds of the chip (9 strips x 4 bands x n frames tasks on 4096 wave slots)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import _native, calib, synth
cal = calib.reference_calibration()
r = synth.SceneRenderer(cal)
base = np.stack([r.render(900 + i)[0] for i in range(8)], 0)
N = 342
ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=N)
for c0 in range(0, N, 114):
    ctx.upload_frames(base[np.arange(min(114, N - c0)) % 8], first=c0)
fp = _native.filter_params()
for n in (114, 171, 200, 228, 256, 285, 313, 342):
    for _ in range(2):
        ctx.mask_run(n, fp)
    ctx.sync()
    ctx.set_stage_timing(True); ctx.stage_reset()
    for _ in range(6):
        ctx.mask_run(n, fp)
    ctx.sync()
    st = ctx.stage_ms(); ctx.set_stage_timing(False)
    rounds = n * 36 / 4096.0
    print("n=%3d rounds=%.2f  us/frame: " % (n, rounds) + "  ".join("%s %.3f" % (k, st[k][0] / 6 / n * 1e3) for k in ("erode_r29", "tophat_r29", "erode_b55", "tophat_b55", "threshold", "warp_split")))
ctx.close()
