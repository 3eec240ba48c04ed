"""Where one process() call spends its time (single stream, one frame in flight)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from lane_tracker_amd import _native, calib, synth
cal = calib.reference_calibration()
ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], capacity=1)
f = synth.SceneRenderer(cal).render(3)[0]
fp, sp = _native.filter_params(), _native.search_params()
def t(fn, n=200):
    for _ in range(20): fn()
    ctx.sync(); t0 = time.perf_counter()
    for _ in range(n): fn()
    ctx.sync(); return (time.perf_counter() - t0) / n * 1e6
print("upload_frames          %7.1f us" % t(lambda: ctx.upload_frames(f)))
print("mask_run (async)+sync  %7.1f us" % t(lambda: (ctx.mask_run(1, fp), ctx.sync())))
print("sws_fit_run+sync       %7.1f us" % t(lambda: (ctx.sws_fit_run(1, sp), ctx.sync())))
print("download_records       %7.1f us" % t(lambda: ctx.download_records(1)))
print("mask+sws+record        %7.1f us" % t(lambda: (ctx.mask_run(1, fp), ctx.sws_fit_run(1, sp), ctx.download_records(1))))
print("upload+mask+sws+record %7.1f us" % t(lambda: (ctx.upload_frames(f), ctx.mask_run(1, fp), ctx.sws_fit_run(1, sp), ctx.download_records(1))))
ctx.set_stage_timing(True); ctx.stage_reset()
for _ in range(50): ctx.mask_run(1, fp); ctx.sws_fit_run(1, sp)
ctx.sync()
ms = ctx.stage_ms()
print({k: round(v[0] / max(v[1], 1) * 1e3, 1) for k, v in ms.items() if v[1]}, "us per kernel (GPU time)")
prev = np.array([1e-5, -0.02, 440.0, 1e-5, -0.02, 640.0])
print("band_fit_run+sync      %7.1f us" % t(lambda: (ctx.band_fit_run(1, prev, sp), ctx.sync())))
# host side of one process() call
from lane_tracker_amd.lane_tracker import LaneTracker
lt = LaneTracker(**cal)
frames = synth.stream_lanes(12, seed=5)
for fr in frames[:4]: lt.process(fr)
import cProfile, pstats, io
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    for fr in frames[4:]: lt._step(fr, (15, 8, 35, 5, 'bilateral', False, 140, 65, 10, 30, 40, 20, 0.1, 8, 0.25, 360, 30, 25, 1.0), 2, False, slot=0, have_mask=False, lazy=False, annotate=False)
pr.disable()
st = io.StringIO(); pstats.Stats(pr, stream=st).sort_stats("cumulative").print_stats(18); print(st.getvalue()[:3800])
