"""Cross-check of two cv2-backed stages of the oracle against scikit-image, an independent third-party implementation
(floating point, so the comparison is "within one grey level", not bit-exact -- the pin that only a real OpenCV can give
stays open, tests/test_opencv_crosscheck.py).

What it pins: the colour science of the 8-bit RGB2LAB b channel (sRGB gamma, D65 white point, matrix, cube root, the
+128 offset: SURVEY App. A.4) and the geometry of warpPerspective (direction of the matrix, pixel-centre convention,
bilinear taps, zero border: App. A.2).  scikit-image lives in a second interpreter of the build image
(/opt/conda/bin/python3.9); where that is missing the tests skip (the GPU box).  CPU only."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from lane_tracker_amd import calib
from oracle import oracle as O

SIDE_PY = os.environ.get("LT_SKIMAGE_PYTHON", "/opt/conda/bin/python3.9")
HERE = os.path.dirname(os.path.abspath(__file__))


def _have_side():
    if not os.path.exists(SIDE_PY):
        return False
    r = subprocess.run([SIDE_PY, "-c", "import skimage, numpy"], capture_output=True)
    return r.returncode == 0


pytestmark = pytest.mark.skipif(not _have_side(), reason="no interpreter with scikit-image here")


def _smooth_image(h, w, seed):
    """low-frequency colour field: at most ~1.5 grey levels per pixel, so that OpenCV's 1/32-pixel coordinate grid
    and a float resampler agree within rounding"""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.zeros((h, w, 3))
    for ch in range(3):
        for _ in range(3):
            fx, fy = rng.uniform(0.002, 0.012, 2)
            img[..., ch] += rng.uniform(20, 40) * np.sin(fx * x + fy * y + rng.uniform(0, 6.28))
        img[..., ch] += 128
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


@pytest.fixture(scope="module")
def side():
    cal = calib.reference_calibration()
    rng = np.random.default_rng(11)
    grays = np.repeat(np.arange(256, dtype=np.uint8)[:, None], 3, 1)
    corners = np.array([[r, g, b] for r in (0, 255) for g in (0, 255) for b in (0, 255)], np.uint8)
    colors = np.concatenate([grays, corners, rng.integers(0, 256, (40000, 3), dtype=np.uint8)], 0)
    w, h = cal["img_size"]
    image = _smooth_image(h, w, 5)
    ww, wh = cal["warped_size"]
    with tempfile.TemporaryDirectory() as td:
        np.savez(os.path.join(td, "in.npz"), colors=colors, image=image, M=np.asarray(cal["warp_matrices"][0], np.float64),
                 out_hw=np.array([wh, ww]))
        r = subprocess.run([SIDE_PY, os.path.join(HERE, "skimage_side.py"), os.path.join(td, "in.npz"), os.path.join(td, "out.npz")],
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out = dict(np.load(os.path.join(td, "out.npz")))
    print("\ncross-checking the oracle against scikit-image", out["version"])
    return cal, colors, image, out


def test_lab_b_channel_matches_skimage_within_one_level(side):
    _, colors, _, out = side
    got = O.lab_b(colors[None])[0].astype(np.int64)                      # OpenCV's 8-bit integer path, restated
    ref = out["lab_b"] + 128.0                                            # 8-bit Lab: b + 128
    want = np.clip(np.rint(ref), 0, 255).astype(np.int64)
    d = got - want
    print("Lab b: %d colours, equal %.2f %%, |diff| <= 1: %.3f %%, max |diff| %d, max |got - float| %.3f"
          % (len(d), 100 * np.mean(d == 0), 100 * np.mean(np.abs(d) <= 1), np.abs(d).max(), np.abs(got - ref).max()))
    assert np.abs(d).max() <= 1
    assert np.mean(d == 0) > 0.90
    assert np.abs(got - ref).max() < 1.5          # the integer path (gamma on 1/8 steps, 15-bit cube-root table) stays near the real-valued b*
    assert np.array_equal(got[:256][[0, 255]], want[:256][[0, 255]])   # black and white are exactly neutral (128)


def test_warp_perspective_matches_skimage_within_one_level(side):
    cal, _, image, out = side
    oc = O.make_calib(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0])
    got = O.warp(oc, image).astype(np.float64)
    ref = out["warped"]
    assert got.shape == ref.shape
    # pixels whose four taps lie inside the frame (the float resampler treats the border row / column differently only there)
    xy, _ = O.warp_map(oc)
    sx, sy = xy[..., 0].astype(np.int64), xy[..., 1].astype(np.int64)
    w, h = cal["img_size"]
    inside = (sx >= 1) & (sx + 2 < w) & (sy >= 1) & (sy + 2 < h)
    d = np.abs(got - ref)[inside]
    outside_all = (sx < -1) | (sx > w) | (sy < -1) | (sy > h)
    print("warp: %.1f %% of the bird's-eye pixels sample inside the frame; there max |diff| %.3f, mean %.4f levels; "
          "%.1f %% sample outside and are 0 in both" % (100 * inside.mean(), d.max(), d.mean(), 100 * outside_all.mean()))
    assert inside.mean() > 0.8
    assert d.max() <= 1.0 and d.mean() < 0.3
    assert np.all(got[outside_all] == 0) and np.all(ref[outside_all] == 0)
