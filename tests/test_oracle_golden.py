"""The CPU oracle against the fixtures generated from the reference itself (tests/gen_golden.py):
sliding_window_search / band_search (bit-exact index arrays, centroid lists), fit_poly,
check_validity, get_poly_points.  Reference: lane_tracker.py:242-528, 561-627."""
import os

import numpy as np
import pytest

from helpers import coeff_close, golden_files, params_of, unpack_mask


@pytest.mark.parametrize("path", golden_files("sws_"), ids=os.path.basename)
def test_sliding_window_search_matches_reference(oracle, path):
    d = np.load(path)
    mask, p = unpack_mask(d), params_of(d)
    r = oracle.sliding_window_search(mask, oracle.search_params(**p))
    assert r["detected"] == bool(d["detected"])
    if not r["detected"]:
        return
    for k in ("left_y", "left_x", "right_y", "right_x"):
        assert np.array_equal(r[k], d[k]), k
    assert r["left_centroids"] == d["left_centroids"].tolist()
    assert r["right_centroids"] == d["right_centroids"].tolist()
    h = mask.shape[0]
    lf, rf = oracle.polyfit2(r["left_y"], r["left_x"]), oracle.polyfit2(r["right_y"], r["right_x"])
    assert coeff_close(lf, d["left_coeffs"], h) and coeff_close(rf, d["right_coeffs"], h)


@pytest.mark.parametrize("path", golden_files("band"), ids=os.path.basename)
def test_band_search_matches_reference(oracle, path):
    d = np.load(path)
    mask, p = unpack_mask(d), params_of(d)
    r = oracle.band_search(mask, d["prev_left"], d["prev_right"], oracle.search_params(**p))
    assert r["detected"] == bool(d["detected"])
    if not r["detected"]:
        return
    for k in ("left_y", "left_x", "right_y", "right_x"):
        assert np.array_equal(r[k], d[k]), k
    lf, rf = oracle.polyfit2(r["left_y"], r["left_x"]), oracle.polyfit2(r["right_y"], r["right_x"])
    assert coeff_close(lf, d["left_coeffs"]) and coeff_close(rf, d["right_coeffs"])


@pytest.mark.parametrize("path", golden_files("sws_") + golden_files("band"), ids=os.path.basename)
def test_validity_and_poly_points_match_reference(oracle, path):
    d = np.load(path)
    if not bool(d["detected"]):
        pytest.skip("nothing detected in this fixture")
    h, w = [int(v) for v in d["mask_shape"]]
    if (h, w) != (1100, 1080):
        pytest.skip("validity fixtures are generated on the reference BEV size only")
    lf, rf = d["left_coeffs"], d["right_coeffs"]
    assert oracle.check_validity((w, h), lf, rf) == bool(d["valid"])
    partial = params_of(d).get("partial", 1)
    ly, lx, ry, rx = oracle.get_poly_points((w, h), lf, rf, partial)
    assert np.array_equal(lx, d["poly_left_x"]) and np.array_equal(rx, d["poly_right_x"])
    assert np.array_equal(ly, d["poly_left_y"]) and np.array_equal(ry, d["poly_right_y"])


def test_polyfit2_tracks_numpy(oracle):
    rng = np.random.default_rng(5)
    for n in (3, 4, 50, 5000):
        y = rng.integers(0, 1100, n)
        x = np.rint(2e-4 * y * y - 0.3 * y + 500 + rng.normal(0, 3, n)).astype(np.int64)
        if len(np.unique(y)) < 3:
            continue
        got, want = oracle.polyfit2(y, x), np.polyfit(y, x, 2)
        assert np.allclose(got, want, rtol=1e-8, atol=1e-10)


def test_polyfit2_rank_deficient_is_minimum_norm(oracle):
    import warnings
    y = np.array([10, 10, 20, 20, 20])
    x = np.array([1, 3, 7, 8, 9])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = np.polyfit(y, x, 2)
    assert np.allclose(oracle.polyfit2(y, x), want, rtol=1e-7, atol=1e-9)
