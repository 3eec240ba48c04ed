"""Container-only fuzz: the CPU oracle against the reference's own NumPy methods, imported from
/root/reference (tests/gen_golden.py harness).  Skipped wherever the reference is absent (it never
travels to the GPU box); the committed fixtures in tests/golden/ carry the same pin there."""
import os
import sys

import numpy as np
import pytest

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "lane_tracker.py")),
                                reason="reference not present (expected on the GPU box)")


@pytest.fixture(scope="module")
def ref():
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import gen_golden
    saved = np.linspace
    mod = gen_golden.import_reference(REF)
    yield gen_golden, mod
    np.linspace = saved
    sys.modules.pop("cv2", None)
    sys.modules.pop("lane_tracker", None)
    sys.modules.pop("utils", None)


def _random_case(rng):
    from lane_tracker_amd import synth
    kind = rng.integers(0, 6)
    seed = int(rng.integers(0, 1 << 30))
    if kind == 0:
        m = synth.random_mask(seed, density=float(10 ** rng.uniform(-4, -0.3)))
    elif kind == 1:
        m = synth.synth_mask(seed, noise=float(10 ** rng.uniform(-4, -1.3)), curv=3e-4, slope=0.25)[0]
    elif kind == 2:
        m = synth.synth_mask(seed, noise=0.0, left_base=(0, 40), sep=(200, 1000))[0]
    elif kind == 3:
        m = synth.synth_mask(seed, noise=1e-3)[0]
        y0, y1 = sorted(rng.integers(0, 1100, 2))
        m[y0:y1, : int(rng.integers(0, 1080))] = 0
    elif kind == 4:
        m = synth.synth_mask(seed, noise=1e-4, drop_left=bool(rng.integers(0, 2)), drop_right=bool(rng.integers(0, 2)))[0]
    else:
        m = synth.synth_mask(seed, noise=float(10 ** rng.uniform(-4, -2)))[0]
    p = dict(window_width=int(rng.choice([30, 30, 20, 31, 60])), window_height=int(rng.choice([40, 40, 25, 118])),
             search_range=int(rng.choice([20, 20, 60, 5])), mu=float(rng.choice([0.1, 0.1, 0.5, 1.0, 0.0])),
             no_success_limit=int(rng.choice([8, 8, 3, 50, 1])), start_slice=float(rng.choice([0.25, 0.25, 0.1, 1.0])),
             ignore_sides=int(rng.choice([360, 360, 0, 100])), ignore_bottom=int(rng.choice([30, 30, 0, 7])),
             partial=rng.choice([1, 1, 0.5, 0.3]).item())
    return m, p


def test_sliding_window_search_fuzz(oracle, ref):
    gen, mod = ref
    rng = np.random.default_rng(2024)
    n_det = 0
    for _ in range(120):
        m, p = _random_case(rng)
        lt = gen.new_tracker(mod)
        lt.sliding_window_search(m, **p)
        r = oracle.sliding_window_search(m, oracle.search_params(**p))
        assert r["detected"] == bool(lt.detected_pixels), p
        if r["detected"]:
            n_det += 1
            assert np.array_equal(r["left_y"], lt.left_y) and np.array_equal(r["left_x"], lt.left_x), p
            assert np.array_equal(r["right_y"], lt.right_y) and np.array_equal(r["right_x"], lt.right_x), p
            assert r["left_centroids"] == [int(v) for v in lt.left_window_centroids], p
            assert r["right_centroids"] == [int(v) for v in lt.right_window_centroids], p
    assert n_det > 40


def test_band_search_fuzz(oracle, ref):
    gen, mod = ref
    from lane_tracker_amd import synth
    rng = np.random.default_rng(77)
    for _ in range(40):
        seed = int(rng.integers(0, 1 << 30))
        m, lc, rc = synth.synth_mask(seed, noise=float(10 ** rng.uniform(-4, -0.5)))
        lc = lc + rng.normal(0, [1e-6, 1e-3, 3.0])
        rc = rc + rng.normal(0, [1e-6, 1e-3, 3.0])
        p = dict(bandwidth=int(rng.choice([25, 30, 5, 80])), ignore_bottom=int(rng.choice([30, 0, 11])), partial=1)
        lt = gen.new_tracker(mod)
        lt.last_left_coeffs, lt.last_right_coeffs = lc, rc
        lt.band_search(m, **p)
        r = oracle.band_search(m, lc, rc, oracle.search_params(**p))
        assert r["detected"] == bool(lt.detected_pixels)
        if r["detected"]:
            assert np.array_equal(r["left_y"], lt.left_y) and np.array_equal(r["left_x"], lt.left_x)
            assert np.array_equal(r["right_y"], lt.right_y) and np.array_equal(r["right_x"], lt.right_x)
