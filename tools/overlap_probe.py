import sys, time, numpy as np
sys.path.insert(0, ".")
from lane_tracker_amd import _native, calib, synth
cal = calib.reference_calibration()
B = 256
r = synth.SceneRenderer(cal)
uniq = np.stack([r.render(i)[0] for i in range(32)], 0)
frames = uniq[np.arange(B) % 32]
def mk(n):
    c = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], capacity=n)
    return c
fp, sp = _native.filter_params(), _native.search_params()
def run(split, steps=10):
    ctxs = [mk(n) for n in split]
    off = 0
    for c, n in zip(ctxs, split):
        c.upload_frames(frames[off:off + n]); off += n
    def step():
        for c, n in zip(ctxs, split): c.mask_run(n, fp)
        for c, n in zip(ctxs, split): c.sws_fit_run(n, sp)
    for _ in range(3): step()
    for c in ctxs: c.sync()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    for c in ctxs: c.sync()
    dt = (time.perf_counter() - t0) / steps
    recs = np.concatenate([c.download_records(n) for c, n in zip(ctxs, split)])
    for c in ctxs: c.close()
    return dt * 1e3, B / dt, int(recs["detected"].sum())
for split in ([256], [128, 128], [160, 96], [192, 64], [96, 96, 64], [64, 64, 64, 64]):
    print(split, "ms/step %.3f  fps %.0f  detected %d" % run(split))
