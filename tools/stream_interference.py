#!/usr/bin/env python3
"""What slows the mask chain inside the stream pipeline: 8 mask launches of 128 frames back to back (a) alone, (b) with the
uploads of the frames running beside them on the copy stream, (c) with chained searches running beside them on the search
stream, (d) with both -- each with no CU, one CU and eight CUs reserved for the search stream (lt_set_search_cus).

Finding (profiles/r03_stream_interference.json): the chain kernel is ONE workgroup, yet the masks run 1.45x slower beside it:
the mask kernels spread their workgroups over the chip once, so the CU they share with the chain's eight busy waves finishes
its share late and every kernel ends with that CU.  Keeping the slots' streams off one CU removes most of it."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import _native, calib, synth
cal = calib.reference_calibration()
n, blocks = 128, 8
base = synth.stream_lanes(32, seed=5)
frames = np.concatenate([base, base[::-1]] * 2, 0)[:n].copy()
pin = _native.pinned_empty(frames.shape); pin[...] = frames
ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=3 * n)
sp = _native.search_params()
for r in range(3):
    ctx.upload_frames(frames, first=r * n)
ctx.mask_run(3 * n)
ctx.sws_fit_run(1, sp, first=0)
ctx.sync()

def run(uploads, chains):
    ctx.sync()
    t0 = time.perf_counter()
    for b in range(blocks):
        region = (b % 2) * n                      # masks alternate between regions 0 and 1; chains walk region 2 (its masks exist)
        if uploads:
            ctx.upload_frame_rows_async(pin, first=region)
        ctx.mask_run(n, first=region)
        if chains:
            ctx.band_fit_chain_run(n - 1, None, sp, first=2 * n + 1) if False else ctx.band_fit_chain_run(n, np.zeros(6) + [0, 0, 440, 0, 0, 640], sp, first=2 * n)
    ctx.sync()
    return (time.perf_counter() - t0) / (blocks * n) * 1e6

out = {}
for cus in (0, 1, 8):
    ctx.set_search_cus(cus)
    row = out["search_cus=%d" % cus] = {}
    for name, u, c in (("masks alone", 0, 0), ("+ uploads", 1, 0), ("+ chains", 0, 1), ("+ uploads + chains", 1, 1)):
        run(u, c)
        row[name] = round(min(run(u, c) for _ in range(3)), 2)
print(json.dumps({"us_per_frame": out}, indent=1))
