"""BASELINE config 1: real camera frames (the reference's test_images/, stored losslessly under
tests/golden/photo_*.png) against what the reference's own process() made of them
(tests/gen_golden.py; cv2 answered by the oracle, so this pins the reference's op order, control flow and
NumPy arithmetic on real data, not OpenCV's numerics)."""
import hashlib
import os

import numpy as np
import pytest

from helpers import GOLDEN, coeff_close

PHOTOS = ("test4", "straight_lines1", "test5")
H, W = 1100, 1080


def load_photo(name):
    from PIL import Image
    frame = np.asarray(Image.open(os.path.join(GOLDEN, f"photo_{name}.png")).convert("RGB"), np.uint8)
    d = np.load(os.path.join(GOLDEN, f"photo_{name}.npz"))
    assert hashlib.sha1(frame.tobytes()).hexdigest() == str(d["frame_sha1"])
    return frame, d


def fixture_mask(d, i):
    return (np.unpackbits(d[f"mask{i}_bits"])[:H * W].reshape(H, W) * 255).astype(np.uint8)


def test_decoded_test4_is_the_surveyed_frame():
    frame, _ = load_photo("test4")
    assert frame.shape == (720, 1280, 3)
    assert hashlib.sha1(frame.tobytes()).hexdigest() == "341f1a5ca7bbf345e799f2b5f32aa0643a0786a3"


@pytest.mark.parametrize("name", PHOTOS)
def test_oracle_reproduces_the_reference_run(oracle, ref_calib, name):
    frame, d = load_photo(name)
    assert int(d["n_tries"]) == 2 and [str(m) for m in d["modes"]] == ["sws", "sws"]
    tries = [(oracle.filter_params(), oracle.search_params()),
             (oracle.filter_params(filter_type="neighborhood", C_r=5), oracle.search_params(no_success_limit=50, bandwidth=30))]
    for i, (fp, sp) in enumerate(tries):
        mask = oracle.mask_from_frame(ref_calib, frame, fp)
        assert hashlib.sha256(mask.tobytes()).hexdigest() == str(d[f"mask{i}_sha256"])
        assert np.array_equal(mask, fixture_mask(d, i))
        r = oracle.sliding_window_search(mask, sp)
        assert (len(r["left_x"]), len(r["right_x"])) == tuple(d["counts"][i])
    lf, rf = oracle.polyfit2(r["left_y"], r["left_x"]), oracle.polyfit2(r["right_y"], r["right_x"])
    # upstream's hard-coded limits are the "Demo 2" set: these project-video frames (lanes ~205 px apart at y3) fail it
    assert bool(d["valid"]) is False and oracle.check_validity((W, H), lf, rf) is False
    from lane_tracker_amd import settings
    assert oracle.check_validity((W, H), lf, rf, settings.DEMO_1["validity"]) is True


def test_test4_geometry_matches_the_readme_figure(oracle, ref_calib):
    """Qualitative anchor for the unpinned cv2 front end: in output_images/test4_warped.png the
    yellow line sits at x ~ 450-465 and the dashed white line at x ~ 660 in the bird's-eye view."""
    frame, _ = load_photo("test4")
    mask = oracle.mask_from_frame(ref_calib, frame)
    r = oracle.sliding_window_search(mask)
    assert r["detected"]
    assert 440 <= np.median(r["left_centroids"][:6]) <= 470
    assert 650 <= np.median(r["right_centroids"][:6]) <= 680
    bev = oracle.front_end(ref_calib, frame)
    yellow = bev[900:1050, 445:470].reshape(-1, 3).astype(int)
    assert (yellow[:, 0] - yellow[:, 2]).max() > 80           # a strongly yellow (R >> B) stripe is there
    assert not bev[1090:, :40].any() and not bev[1090:, -40:].any()   # black bottom-corner triangles


def test_straight_lines_fit_is_straight(oracle, ref_calib):
    frame, _ = load_photo("straight_lines1")
    r = oracle.sliding_window_search(oracle.mask_from_frame(ref_calib, frame))
    lf, rf = oracle.polyfit2(r["left_y"], r["left_x"]), oracle.polyfit2(r["right_y"], r["right_x"])
    assert abs(lf[0]) < 1e-5 and abs(rf[0]) < 1e-5 and abs(lf[1]) < 0.02 and abs(rf[1]) < 0.02
    assert 195 <= rf[2] - lf[2] <= 215


# ---- GPU --------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", PHOTOS)
def test_gpu_process_on_real_frames(oracle, ref_calib, name):
    from lane_tracker_amd import calib, settings
    from lane_tracker_amd.lane_tracker import LaneTracker
    frame, d = load_photo(name)
    cal = calib.reference_calibration()
    lt = LaneTracker(**cal)
    try:
        # the two masks, bit for bit
        binary, mode = lt.find_lane_points(frame, mask_noise=False, partial=1.0)
        assert mode == "sws" and np.array_equal(binary, fixture_mask(d, 0))
        assert (len(lt.left_x), len(lt.right_x)) == tuple(d["counts"][0])
        # process() with upstream's limits: detected, rejected by check_validity, like the reference
        out = lt.process(frame)
        assert out.shape == frame.shape
        assert bool(lt.detected_pixels) == bool(d["detected"]) and bool(lt.valid_lane_lines) == bool(d["valid"])
        assert lt.last_detection == int(d["last_detection"]) + 0 and lt.success == int(d["success"])
        assert (len(lt.left_x), len(lt.right_x)) == tuple(d["counts"][1])
        assert np.array_equal(lt._ctx.download_masks(1, first=lt._slot)[0], fixture_mask(d, 1))
    finally:
        lt.close()
    # the demo-1 parameter set (the video these frames come from) accepts them
    lt = LaneTracker(**cal)
    try:
        kw = settings.apply(lt, settings.DEMO_1)
        lt.process(frame, **kw)
        assert lt.valid_lane_lines and lt.success == 1
        mask = oracle.mask_from_frame(ref_calib, frame, oracle.filter_params(mask_noise=True))
        assert np.array_equal(lt._ctx.download_masks(1, first=lt._slot)[0], mask)
        r = oracle.sliding_window_search(mask, oracle.search_params(no_success_limit=50, bandwidth=30))
        assert coeff_close(lt.last_left_coeffs, oracle.polyfit2(r["left_y"], r["left_x"]))
        assert coeff_close(lt.last_right_coeffs, oracle.polyfit2(r["right_y"], r["right_x"]))
        assert lt.average_curve_radius is not None and abs(lt.eccentricity) < 1.0
    finally:
        lt.close()
