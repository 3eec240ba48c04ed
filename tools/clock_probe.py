#!/usr/bin/env python3
"""Shader clock while the mask chain runs alone and beside one workgroup busy with float arithmetic (tools/microbench/spin.hip)."""
import ctypes, os, subprocess, sys, threading, time, glob
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import _native, calib, synth
here = os.path.dirname(os.path.abspath(__file__))
subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(here, "microbench", "spin.hip"), "-o", "/tmp/libspin.so"], stderr=subprocess.DEVNULL)
spin = ctypes.CDLL("/tmp/libspin.so")
spin.spin_start_lds.argtypes = [ctypes.c_double, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
cal = calib.reference_calibration()
n = 128
frames = synth.stream_lanes(8, seed=5); frames = np.concatenate([frames] * 16, 0)[:n].copy()
ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=n)
ctx.upload_frames(frames); ctx.mask_run(n); ctx.sync()

def sclk():
    vals = []
    for p in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for line in open(p):
                if "*" in line:
                    vals.append(line.strip())
        except OSError:
            pass
    return vals

def sample(label, mode):
    seen, stop = [], [False]
    def poll():
        while not stop[0]:
            seen.extend(sclk()); time.sleep(0.02)
    t = threading.Thread(target=poll); t.start()
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < 1.5:
        if mode is not None:
            spin.spin_start_lds(90.0, 512, 1, mode, 0)
        for _ in range(8):
            ctx.mask_run(n)
        ctx.sync(); spin.spin_sync(); k += 8
    dt = time.perf_counter() - t0
    stop[0] = True; t.join()
    from collections import Counter
    print(label, "%.2f us/frame" % (dt / (k * n) * 1e6), Counter(seen).most_common(4))

sample("masks alone", None)
sample("beside float arithmetic", 9)
sample("beside integer arithmetic", 5)
