#!/usr/bin/env python3
"""Threshold stage time against the number of frames per call, walking kernels vs tile kernel (one stream).
usage: python tools/threshold_crossover.py          (LT_BILATERAL_TILES=1 for the tile kernel)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import _native, calib, synth
cal = calib.reference_calibration()
r = synth.SceneRenderer(cal)
base = np.stack([r.render(900 + i)[0] for i in range(8)], 0)
ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=256)
ctx.upload_frames(base[np.arange(256) % 8])
fp = _native.filter_params()
out = {}
for n in (1, 2, 4, 8, 16, 32, 64, 96, 128, 192, 256):
    for _ in range(3):
        ctx.mask_run(n, fp)
    ctx.sync()
    ctx.set_stage_timing(True); ctx.stage_reset()
    for _ in range(10):
        ctx.mask_run(n, fp)
    ctx.sync()
    st = ctx.stage_ms(); ctx.set_stage_timing(False)
    out[n] = (round(st["threshold"][0] / 10 * 1e3, 1), round(sum(v[0] for k, v in st.items() if k != "sws_fit") / 10 * 1e3, 1))
print("path", ctx.last_threshold_path(), {n: v for n, v in out.items()})   # n: (threshold us, whole mask chain us)
ctx.close()
