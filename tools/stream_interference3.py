#!/usr/bin/env python3
"""Mask launches of 128 frames back to back beside chained searches of k frames per launch (k = 0, 16, 32, 64, 128): how the
slow-down of the mask chain scales with the time the chain kernel is running.  Prints the chain's own time per frame too."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import _native, calib, synth
cal = calib.reference_calibration()
n, blocks = 128, 8
base = synth.stream_lanes(32, seed=5)
frames = np.concatenate([base, base[::-1]] * 2, 0)[:n].copy()
ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=3 * n)
sp = _native.search_params()
for r in range(3):
    ctx.upload_frames(frames, first=r * n)
ctx.mask_run(3 * n); ctx.sws_fit_run(1, sp, first=2 * n); ctx.sync()
seed = ctx.download_records(1, first=2 * n)[0]
seed = np.concatenate([seed["left_coeffs"], seed["right_coeffs"]])
def t_chain(k):
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(8):
        ctx.band_fit_chain_run(k, seed, sp, first=2 * n)
    ctx.sync()
    return (time.perf_counter() - t0) / (8 * k) * 1e6
def run(k):
    ctx.sync(); t0 = time.perf_counter()
    for b in range(blocks):
        ctx.mask_run(n, first=(b % 2) * n)
        if k:
            ctx.band_fit_chain_run(k, seed, sp, first=2 * n)
    ctx.sync()
    return (time.perf_counter() - t0) / (blocks * n) * 1e6
out = {"chain_alone_us_per_frame": round(min(t_chain(128) for _ in range(3)), 2)}
for k in (0, 16, 32, 64, 128):
    run(k)
    out["masks_us_per_frame_beside_chains_of_%d" % k] = round(min(run(k) for _ in range(3)), 2)
print(json.dumps(out))
