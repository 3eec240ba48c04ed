#!/usr/bin/env python3
"""The second-try mask ('neighborhood' filter) for launches of 1 .. 256 frames: threshold-stage device time per launch.
LT_ADAPTIVE_TILES=1 selects the per-pixel window kernel for comparison."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import _native, calib, synth
cal = calib.reference_calibration()
f = synth.SceneRenderer(cal).render(3)[0]
ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=256)
ctx.upload_frames(np.broadcast_to(f, (256,) + f.shape))
fp = _native.filter_params(filter_type="neighborhood", ksize_r=15, C_r=5, ksize_b=35, C_b=5)
out = {}
for n in (1, 2, 4, 8, 32, 128, 256):
    ctx.mask_run(n, fp); ctx.sync()
    ctx.set_stage_timing(True); ctx.stage_reset()
    for _ in range(5):
        ctx.mask_run(n, fp)
    ctx.sync()
    st = ctx.stage_ms(); ctx.set_stage_timing(False)
    out[n] = {"threshold_us": round(st["threshold"][0] / 5 * 1e3, 1), "merge_us": round(st["merge"][0] / 5 * 1e3, 1), "open_us": round(st["open5"][0] / 5 * 1e3, 1)}
print(json.dumps({"box_walk": ctx.last_adaptive_path() == 1, "per_launch": out}))
