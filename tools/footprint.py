#!/usr/bin/env python3
"""Camera-row footprint of bird's-eye rows under the reference calibration (CPU only): which undistorted camera rows a tile of
the warp reads, and the redundancy a warp that undistorts its own footprint would pay per tile shape.
NOTES_r05 D.6 quotes this output.

    python tools/footprint.py [--scale 1.5]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import calib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    a = ap.parse_args()
    cal = calib.reference_calibration() if a.scale == 1.0 else calib.scaled_calibration(a.scale)
    W, H = cal["warped_size"]
    Minv = np.asarray(cal["warp_matrices"][1], np.float64)       # bird's-eye -> camera
    ys, xs = np.mgrid[0:H, 0:W]
    p = Minv @ np.stack([xs.ravel(), ys.ravel(), np.ones(xs.size)], 0).astype(np.float64)
    sy = (p[1] / p[2]).reshape(H, W)
    sx = (p[0] / p[2]).reshape(H, W)
    spread = sy.max(1) - sy.min(1)
    print("bird's-eye %dx%d; camera row of a bird's-eye row varies by at most %.3f rows along the row" % (W, H, spread.max()))
    r = sy.mean(1)
    for lo, hi in ((0, 700 * H // 1100), (700 * H // 1100, H)):
        print("bird's-eye rows %4d-%4d <- camera rows %.1f-%.1f (%.1f rows)" % (lo, hi - 1, r[lo:hi].min(), r[lo:hi].max(), r[lo:hi].max() - r[lo:hi].min()))
    print("camera columns read: %.1f-%.1f" % (sx.min(), sx.max()))
    # tiles of KH whole bird's-eye rows: staged camera rows = floor(min)..floor(max)+1
    total_rows = np.floor(r.max()) + 2 - np.floor(r.min())
    print("camera rows read at all: %d" % total_rows)
    for KH in (4, 8, 16, 32, 64):
        staged = 0
        for t in range(0, H, KH):
            seg = sy[t:t + KH]
            staged += np.floor(seg.max()) + 2 - np.floor(seg.min())
        print("tiles of %2d bird's-eye rows: %5d staged camera rows = %.2fx" % (KH, staged, staged / total_rows))
    for KH in (4, 6, 8, 12):
        print("bands of %2d staged camera rows serve %2d rows of taps: %.2fx, %d KB of LDS per frame" % (KH, KH - 1, KH / (KH - 1), KH * cal["img_size"][0] * 4 // 1024))


if __name__ == "__main__":
    main()
