#!/usr/bin/env python3
"""Top-hat kernel time against the number of tasks in a launch (frames per launch, band count pinned with LT_MORPH_NB_*): is a
launch bound by the work in it or by the length of its longest task?"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import _native, calib, synth
cal = calib.reference_calibration()
f = synth.SceneRenderer(cal).render(3)[0]
ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=256)
ctx.upload_frames(np.broadcast_to(f, (256,) + f.shape))
fp = _native.filter_params()
out = {}
for n in (8, 16, 32, 64, 128, 192, 256):
    ctx.mask_run(n, fp); ctx.sync()
    ctx.set_stage_timing(True); ctx.stage_reset()
    for _ in range(3):
        ctx.mask_run(n, fp)
    ctx.sync()
    st = ctx.stage_ms(); ctx.set_stage_timing(False)
    out[n] = {k: round(st[k][0] / 3, 4) for k in ("erode_r29", "tophat_r29", "erode_b55", "tophat_b55", "threshold", "warp_split")}
print(json.dumps(out))
