#!/usr/bin/env python3
"""Mask launches of 128 frames back to back, alone and beside a do-nothing kernel (one sleeping workgroup) on another stream."""
import ctypes, json, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import _native, calib, synth
here = os.path.dirname(os.path.abspath(__file__))
subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(here, "microbench", "spin.hip"), "-o", "/tmp/libspin.so"],
                      stderr=subprocess.DEVNULL)
cal = calib.reference_calibration()
n, blocks = 128, 8
base = synth.stream_lanes(32, seed=5)
frames = np.concatenate([base, base[::-1]] * 2, 0)[:n].copy()
ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=2 * n)
spin = ctypes.CDLL("/tmp/libspin.so")
spin.spin_start.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int]
spin.spin_start_lds.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
ctx.upload_frames(frames, first=0); ctx.upload_frames(frames, first=n)
ctx.mask_run(2 * n); ctx.sync()

def run(spin_args):
    ctx.sync(); spin.spin_sync()
    if spin_args:
        assert (spin.spin_start_lds if len(spin_args) == 5 else spin.spin_start)(*spin_args) == 0
    t0 = time.perf_counter()
    for b in range(blocks):
        ctx.mask_run(n, first=(b % 2) * n)
    ctx.sync()
    dt = time.perf_counter() - t0
    spin.spin_sync()
    return dt / (blocks * n) * 1e6

spin.spin_sync(); t0 = time.perf_counter(); spin.spin_start(90.0, 64, 1, 0); spin.spin_sync()
print('a "90 ms" spin takes %.2f ms' % ((time.perf_counter() - t0) * 1e3))
out = {}
for name, a in (("masks alone", None), ("+ 1 workgroup of 512 busy with f64 mul/add/div", (90.0, 512, 1, 2, 0)), ("+ the same at ~10 % duty", (90.0, 512, 1, 6, 0)),
                ("+ f64 mul/add only", (90.0, 512, 1, 7, 0)), ("+ f64 in one lane of each wave only", (90.0, 512, 1, 8, 0)), ("+ one wave of 64 busy with f64", (90.0, 64, 1, 2, 0)),
                ("+ the same arithmetic in f32", (90.0, 512, 1, 9, 0))):
    run(a)
    out[name] = round(min(run(a) for _ in range(3)), 2)
print(json.dumps({"us_per_frame": out}, indent=1))
