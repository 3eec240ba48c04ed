import sys, time, numpy as np
sys.path.insert(0, ".")
from lane_tracker_amd import _native, calib, synth
cal = calib.reference_calibration()
B = 256
r = synth.SceneRenderer(cal)
uniq = np.stack([r.render(i)[0] for i in range(32)], 0)
frames = uniq[np.arange(B) % 32]
ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], capacity=B)
ctx.upload_frames(frames)
fp, sp = _native.filter_params(), _native.search_params()
def run(chunk, streams, steps=8):
    ctx.set_streams(streams)
    def step():
        for lo in range(0, B, chunk):
            ctx.mask_run(chunk, fp, first=lo)
            ctx.sws_fit_run(chunk, sp, first=lo)
    for _ in range(2): step()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    ctx.sync()
    return (time.perf_counter() - t0) / steps * 1e3
for streams in (1, 3, 4):
    for chunk in (256, 128, 64, 32, 16):
        print("streams %d chunk %3d: %.3f ms/step" % (streams, chunk, run(chunk, streams)))
