#!/usr/bin/env python3
"""Randomised differential run of the HIP path against the CPU oracle (run on the GPU box):
   python tests/fuzz_gpu.py [iterations] [seed]
Mask chain with random filter parameters on random frames (reference calibration), filter_lane_points on random
small images of random sizes, both searches with random parameters.  Prints the first mismatch and exits 1."""
import sys
import numpy as np
sys.path.insert(0, ".")
from lane_tracker_amd import _native, calib, synth
from oracle import oracle as O

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
cal = calib.reference_calibration()
oc = O.make_calib(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0])
ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], capacity=4)
renderer = synth.SceneRenderer(cal)


def random_frame(i):
    kind = rng.integers(0, 5)
    if kind == 0:
        return synth.frame_uniform(int(rng.integers(0, 1 << 30)))
    f = renderer.render(int(rng.integers(0, 1 << 20)))[0].copy()
    if kind == 2:      # saturated / dark patches
        for _ in range(6):
            y, x = rng.integers(0, 650), rng.integers(0, 1200)
            f[y:y + rng.integers(5, 200), x:x + rng.integers(5, 300)] = rng.integers(0, 256, 3)
    if kind == 3:      # strong noise
        f = np.clip(f.astype(int) + rng.integers(-60, 61, f.shape), 0, 255).astype(np.uint8)
    if kind == 4:
        f = (f // 32 * 32).astype(np.uint8)    # posterised: many exact ties
    return f


def random_filter():
    ft = "bilateral" if rng.random() < 0.6 else "neighborhood"
    odd = lambda lo, hi: int(rng.integers(lo, hi) // 2 * 2 + 1)
    kw = dict(filter_type=ft, ksize_r=odd(3, 70) if ft == "neighborhood" else int(rng.integers(1, 80)), C_r=int(rng.integers(0, 25)),
              ksize_b=odd(3, 70) if ft == "neighborhood" else int(rng.integers(1, 80)), C_b=int(rng.integers(0, 25)),
              mask_noise=bool(rng.random() < 0.4), noise_thresh=int(rng.integers(100, 180)), ksize_noise=int(rng.integers(1, 120)),
              C_noise=int(rng.integers(0, 25)))
    if rng.random() < 0.5:       # the parameter classes the batch-size kernels take (with LT_WALK_MIN_FRAMES=0 also for these two frames)
        if ft == "bilateral":
            kw.update(ksize_r=int(rng.choice([15, 20, 35])), ksize_b=int(rng.choice([15, 20, 35])))
            if kw["mask_noise"]:
                kw.update(ksize_noise=65, noise_thresh=int(rng.choice([0, 100, 128, 140, 200, 255, 256, 300])),
                          C_noise=int(rng.choice([0, 10, 40, 249])))
        else:
            kw.update(mask_noise=False, C_r=int(rng.integers(-30, 60)), C_b=int(rng.integers(-30, 60)),
                      ksize_r=odd(1, 64), ksize_b=odd(1, 64))
    return kw


bad = 0
for it in range(iters):
    frames = np.stack([random_frame(it) for _ in range(2)], 0)
    kw = random_filter()
    ctx.upload_frames(frames)
    ctx.mask_run(2, _native.filter_params(**kw))
    got = ctx.download_masks(2)
    for k in range(2):
        want = O.mask_from_frame(oc, frames[k], O.filter_params(**kw))
        if not np.array_equal(got[k], want):
            d = np.argwhere(got[k] != want)
            print("MASK MISMATCH it", it, "frame", k, kw, len(d), d[:5].tolist())
            bad += 1
    # searches on the resident masks
    sp = dict(window_width=int(rng.choice([30, 20, 31, 60, 64, 66, 100])), window_height=int(rng.choice([40, 25, 118, 64, 65])),
              search_range=int(rng.choice([20, 60, 5])), mu=float(rng.choice([0.1, 0.5, 1.0, 0.0])),
              no_success_limit=int(rng.choice([8, 3, 50, 1])), start_slice=float(rng.choice([0.25, 0.1, 1.0])),
              ignore_sides=int(rng.choice([360, 0, 100])), ignore_bottom=int(rng.choice([30, 0, 7])),
              partial=float(rng.choice([1.0, 0.5, 0.3])))
    ctx.sws_fit_run(2, _native.search_params(**sp))
    recs = ctx.download_records(2)
    for k in range(2):
        o = O.sliding_window_search(got[k], O.search_params(**sp))
        ly, lx = ctx.download_pixels(k, 0)
        ry, rx = ctx.download_pixels(k, 1)
        ok = (bool(recs[k]["detected"]) == o["detected"] and np.array_equal(ly, o["left_y"]) and np.array_equal(lx, o["left_x"])
              and np.array_equal(ry, o["right_y"]) and np.array_equal(rx, o["right_x"])
              and ctx.download_centroids(k, 0) == o["left_centroids"] and ctx.download_centroids(k, 1) == o["right_centroids"])
        if not ok:
            print("SWS MISMATCH it", it, "frame", k, sp)
            bad += 1
    bw = int(rng.choice([25, 30, 5, 31, 32, 60, 0]))
    prev = np.array([rng.uniform(-1e-4, 1e-4), rng.uniform(-0.2, 0.1), rng.uniform(300, 560),
                     rng.uniform(-1e-4, 1e-4), rng.uniform(-0.2, 0.1), rng.uniform(560, 800)])
    bp = dict(bandwidth=bw, ignore_bottom=int(rng.choice([30, 0])), partial=float(rng.choice([1.0, 0.5])))
    for k in range(2):
        ctx.band_fit_run(1, prev, _native.search_params(**bp), first=k)
        o = O.band_search(got[k], prev[:3], prev[3:], O.search_params(**bp))
        rec = ctx.download_records(1, first=k)[0]
        ly, lx = ctx.download_pixels(k, 0)
        ry, rx = ctx.download_pixels(k, 1)
        ok = (bool(rec["detected"]) == o["detected"] and np.array_equal(ly, o["left_y"]) and np.array_equal(lx, o["left_x"])
              and np.array_equal(ry, o["right_y"]) and np.array_equal(rx, o["right_x"]))
        if not ok:
            print("BAND MISMATCH it", it, "frame", k, bp, prev.tolist())
            bad += 1
    # filter_lane_points on a small random image of random size (generic kernel paths)
    h, w = int(rng.integers(60, 260)), int(rng.integers(60, 300))
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if rng.random() < 0.5:
        img = (img // 64 * 64).astype(np.uint8)
    fctx = _native.Context((2, 2), (w, h), np.eye(3), np.zeros(5), np.eye(3), capacity=1)
    kw2 = random_filter()
    got2 = fctx.filter_lane_points(img, _native.filter_params(**kw2))
    want2 = O.filter_lane_points(img, O.filter_params(**kw2))
    if not np.array_equal(got2, want2):
        d = np.argwhere(got2 != want2)
        print("FILTER MISMATCH it", it, (h, w), kw2, len(d), d[:5].tolist())
        bad += 1
    fctx.close()
    if it % 10 == 9:
        print("iteration", it + 1, "mismatches", bad, flush=True)
ctx.close()
print("done:", iters, "iterations,", bad, "mismatches")
sys.exit(1 if bad else 0)
