#!/usr/bin/env python3
"""One frame through the mask chain + band search, as process() issues them: device time per stage (hipEvents around every kernel,
one stream) against the wall time from the first launch to the records on the host."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import _native, calib, synth
cal = calib.reference_calibration() if len(sys.argv) < 2 else calib.scaled_calibration(1.5)
f = synth.stream_lanes(2, seed=5, cal=cal)
ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=2)
fp, sp = _native.filter_params(), _native.search_params()
ctx.upload_frames(f)
ctx.mask_run(1, fp); ctx.sws_fit_run(1, sp)
seed = ctx.download_records(1)
prev = np.concatenate([seed[0]["left_coeffs"], seed[0]["right_coeffs"]])[None]
def once():
    ctx.mask_run(1, fp)
    ctx.band_fit_run(1, prev, sp)
    return ctx.download_records(1)
for _ in range(20): once()
t0 = time.perf_counter()
for _ in range(200): once()
wall = (time.perf_counter() - t0) / 200 * 1e6
ctx.set_stage_timing(True); ctx.stage_reset()
for _ in range(50): once()
st = ctx.stage_ms()
ctx.set_stage_timing(False)
per = {k: round(v[0] / 50 * 1e3, 1) for k, v in st.items() if v[1]}
print(json.dumps({"wall_us_mask_plus_band_plus_records": round(wall, 1), "device_us_by_stage": per, "device_us_sum": round(sum(per.values()), 1)}))
ctx.close()
