#!/usr/bin/env python3
"""Where the overlapped host-fed rate goes (GPU box): uploads alone, uploads + chain, chain alone, per 256-frame batch."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from lane_tracker_amd import _native, calib, synth

cal = calib.reference_calibration()
B = 256
r = synth.SceneRenderer(cal)
one = np.stack([r.render(i)[0] for i in range(8)], 0)
frames = np.concatenate([one] * (B // 8), 0)
ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=2 * B)
pin = [_native.pinned_empty(frames.shape), _native.pinned_empty(frames.shape)]
pin[0][...] = frames
pin[1][...] = frames[::-1]
fp, sp = _native.filter_params(), _native.search_params()
ctx.set_streams(4)


def run(nb, upload, compute):
    for k in range(nb):
        half = k & 1
        if upload:
            ctx.upload_frame_rows_async(pin[half], first=half * B)
        if compute:
            ctx.mask_run(B, fp, first=half * B)
            ctx.sws_fit_run(B, sp, first=half * B)
    ctx.sync()


for name, up, co in (("uploads only", True, False), ("chain only", False, True), ("uploads + chain", True, True)):
    run(2, True, True)
    t0 = time.perf_counter()
    run(12, up, co)
    dt = (time.perf_counter() - t0) / 12
    print("%-16s %.3f ms per batch of %d  (%.0f frames/s)" % (name, dt * 1e3, B, B / dt))
# enqueue cost on the host: the same loop without waiting for the GPU in between is what run() does; time the calls alone
t0 = time.perf_counter()
for k in range(12):
    half = k & 1
    ctx.mask_run(B, fp, first=half * B)
    ctx.sws_fit_run(B, sp, first=half * B)
t1 = time.perf_counter()
ctx.sync()
print("host time to enqueue one batch's chain: %.3f ms" % ((t1 - t0) / 12 * 1e3))
ctx.close()
