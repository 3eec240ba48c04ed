#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING THE REFERENCE (container-only; /root/reference never
travels to the GPU box, the fixtures do).

What is pinned, and how:
  * sws_*.npz / band_*.npz -- LaneTracker.sliding_window_search / band_search / fit_poly /
    check_validity / get_poly_points of the reference run UNMODIFIED on seeded binary masks
    (inputs stored bit-packed, outputs stored as arrays).  These methods are plain NumPy.
  * viz_*.npz -- visualize_sliding_window_search / visualize_band_search / triple_split_view of the
    reference on the searches above.  Their NumPy indexing is the reference's; the three cv2 calls
    inside (merge, fillPoly, addWeighted, resize) are answered by the oracle, i.e. unpinned.
  * photo_*.png / photo_*.npz -- three frames of the reference's test_images/ (decoded with Pillow, stored
    losslessly) and what the reference's process() makes of them: masks of each try, search mode, pixel
    counts, coefficients, validity (BASELINE config 1; cv2 answered by the oracle as above).
  * process_trace.npz -- the reference's process() state machine run over a synthetic stream.
    process() needs cv2, which does not exist in this image.  The harness registers a stand-in
    `cv2` module whose functions are answered by this repo's CPU oracle (oracle/), so this fixture
    pins ONLY the reference's control flow (two tries, search-mode choice, history/averaging,
    counters) and its NumPy arithmetic -- it says nothing about OpenCV numerics (parity of the
    cv2-backed stages stays unpinned, see oracle/lt_oracle.h).

Harness-side shims (the reference source is not modified): `np.int = int`; `np.linspace` accepts a
float `num` (2017 NumPy truncated it); integral float `partial` is passed as int to band_search /
get_poly_points (2017 NumPy accepted float slice bounds).  SURVEY.md F5.

Usage: python tests/gen_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import hashlib
import os
import sys
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from lane_tracker_amd import calib as lcalib  # noqa: E402
from lane_tracker_amd import synth  # noqa: E402
from oracle import oracle as O  # noqa: E402


# ---- cv2 stand-in answered by the oracle -----------------------------------------------------------
def make_cv2_stub():
    cv2 = types.ModuleType("cv2")
    for i, name in enumerate(["MORPH_ELLIPSE", "COLOR_RGB2LAB", "MORPH_TOPHAT", "MORPH_OPEN", "CV_16S",
                              "BORDER_CONSTANT", "ADAPTIVE_THRESH_MEAN_C", "THRESH_BINARY",
                              "INTER_LINEAR", "FONT_HERSHEY_SIMPLEX", "LINE_AA"]):
        setattr(cv2, name, 1000 + i)

    def generic_calib(src_shape, M, dsize, K=None, D=None):
        return O.make_calib((src_shape[1], src_shape[0]), dsize,
                            np.eye(3) if K is None else K, np.zeros(5) if D is None else D, M)

    def warpPerspective(img, M, dsize, flags=None, borderMode=None):
        return O.warp(generic_calib(img.shape, M, dsize), img)

    def undistort(img, K, D, R, newK):
        return O.undistort(generic_calib(img.shape, np.eye(3), (1, 1), K, D), img)

    def getStructuringElement(shape, ksize):
        return O.ellipse_kernel(ksize[0])

    def cvtColor(img, code):
        out = np.zeros_like(img)
        out[:, :, 2] = O.lab_b(img)
        return out

    def morphologyEx(img, op, strel, iterations=1):
        k = strel.shape[0]
        return O.tophat(img, k) if op == cv2.MORPH_TOPHAT else O.morph_open(img, k)

    def filter2D(img, ddepth, kernel, anchor=None, delta=0, borderType=None):
        kh, kw = kernel.shape
        ax, ay = anchor
        h, w = img.shape
        pad = np.zeros((h + 2 * kh, w + 2 * kw), np.int64)
        pad[kh:kh + h, kw:kw + w] = img
        out = np.full((h, w), int(delta), np.int64)
        for i in range(kh):
            for j in range(kw):
                out += int(kernel[i, j]) * pad[kh + i - ay:kh + i - ay + h, kw + j - ax:kw + j - ax + w]
        return np.clip(out, -32768, 32767).astype(np.int16)

    def adaptiveThreshold(src, maxValue, adaptiveMethod, thresholdType, blockSize, C):
        return O.adaptive_mean_threshold(src, blockSize, -C)

    def inRange(a, lo, hi):
        return (((a >= lo) & (a <= hi)).astype(np.uint8)) * 255

    def fillPoly(img, pts, color):
        for poly in pts:
            O.fill_poly(img, np.asarray(poly).reshape(-1, 2), color)
        return img

    def putText(img, *a, **k):
        return img

    def addWeighted(a, alpha, b, beta, gamma):
        return O.add_weighted(a, alpha, b, beta, gamma)

    for f in (warpPerspective, undistort, getStructuringElement, cvtColor, morphologyEx, filter2D,
              adaptiveThreshold, inRange, fillPoly, putText, addWeighted):
        setattr(cv2, f.__name__, f)
    cv2.merge = lambda chans: np.stack(chans, axis=2)
    cv2.resize = lambda img, dsize: O.resize_linear(img, dsize)
    return cv2


def import_reference(ref_dir):
    sys.modules["cv2"] = make_cv2_stub()
    if not hasattr(np, "int"):
        np.int = int
    _linspace = np.linspace
    np.linspace = lambda start, stop, num=50, *a, **k: _linspace(start, stop, int(num), *a, **k)
    sys.path.insert(0, ref_dir)
    import lane_tracker as ref  # the reference module
    sys.path.remove(ref_dir)
    _band, _gpp = ref.LaneTracker.band_search, ref.LaneTracker.get_poly_points
    as_int = lambda p: int(p) if float(p).is_integer() else p
    ref.LaneTracker.band_search = lambda self, img, bandwidth, ignore_bottom=30, partial=1, diagnostics=False: \
        _band(self, img, bandwidth, ignore_bottom, as_int(partial), diagnostics)
    ref.LaneTracker.get_poly_points = lambda self, l, r, partial=1: _gpp(self, l, r, as_int(partial))
    return ref


def new_tracker(ref, cal=None, **kw):
    cal = cal or lcalib.reference_calibration()
    return ref.LaneTracker(img_size=cal["img_size"], warped_size=cal["warped_size"],
                           cam_matrix=cal["cam_matrix"], dist_coeffs=cal["dist_coeffs"],
                           warp_matrices=cal["warp_matrices"], mpp_conversion=cal["mpp_conversion"], **kw)


SWS_DEFAULT = dict(window_width=30, window_height=40, search_range=20, mu=0.1, no_success_limit=8,
                   start_slice=0.25, ignore_sides=360, ignore_bottom=30, partial=1)


def sws_cases():
    """(name, mask, params) -- SURVEY.md section 8(c) list of conditions."""
    cases = []
    for i, noise in enumerate([1e-4, 1e-3, 1e-2, 5e-2]):
        cases.append((f"lanes_noise{i}", synth.synth_mask(10 + i, noise=noise)[0], {}))
    cases.append(("solid_both", synth.synth_mask(20, dashed_right=False)[0], {}))
    cases.append(("curvy", synth.synth_mask(21, curv=3e-4, slope=0.2)[0], dict(search_range=60)))
    cases.append(("curvy_mu05", synth.synth_mask(22, curv=3e-4, slope=0.2)[0], dict(mu=0.5)))
    cases.append(("curvy_mu10", synth.synth_mask(23, curv=2e-4, slope=0.15)[0], dict(mu=1.0, search_range=60)))
    cases.append(("random50", synth.random_mask(30, density=0.5), {}))
    cases.append(("random1e4", synth.random_mask(31, density=1e-4), {}))
    cases.append(("empty", np.zeros((1100, 1080), np.uint8), {}))
    cases.append(("no_left", synth.synth_mask(32, drop_left=True, noise=0)[0], {}))
    cases.append(("no_right", synth.synth_mask(33, drop_right=True, noise=0)[0], {}))
    cases.append(("limit3", synth.synth_mask(34, noise=2e-4)[0], dict(no_success_limit=3)))
    cases.append(("limit50", synth.synth_mask(35, noise=2e-4)[0], dict(no_success_limit=50)))
    cases.append(("partial05", synth.synth_mask(36)[0], dict(partial=0.5)))
    cases.append(("nine_windows", synth.synth_mask(37)[0], dict(window_height=118)))
    cases.append(("left_edge", synth.synth_mask(38, left_base=(2, 14), sep=(300, 400))[0], dict(ignore_sides=0)))
    cases.append(("right_edge", synth.synth_mask(39, left_base=(500, 520), sep=(545, 575), slope=0.0, curv=0.0)[0],
                  dict(ignore_sides=0)))
    cases.append(("wide_window", synth.synth_mask(40)[0], dict(window_width=60, window_height=20)))
    cases.append(("odd_window", synth.synth_mask(41)[0], dict(window_width=31, window_height=37)))
    m = synth.synth_mask(42, noise=0)[0]
    m[:, :] = np.where(np.arange(1100)[:, None] < 700, 0, m)  # lines end early: abort branches
    cases.append(("short_lines", m, {}))
    m = synth.synth_mask(43, noise=0, dashed_right=True)[0]
    m[500:900, :540] = 0  # left gap: 'follow the other side' branch
    cases.append(("left_gap", m, dict(no_success_limit=50)))
    m = synth.synth_mask(44, noise=0, dashed_right=False)[0]
    m[300:800, 540:] = 0
    cases.append(("right_gap", m, dict(no_success_limit=50)))
    cases.append(("value1", synth.synth_mask(45, value=1)[0], {}))
    cases.append(("small_img", synth.synth_mask(46, h=300, w=400, left_base=(120, 150), sep=(90, 120))[0],
                  dict(ignore_sides=60, ignore_bottom=10, window_height=25)))
    return cases


def arr_or_empty(a):
    return np.asarray(a if a is not None else [], dtype=np.int64)


def run_sws(ref, mask, params):
    lt = new_tracker(ref)
    if mask.shape != (1100, 1080):
        lt.warped_size = (mask.shape[1], mask.shape[0])
    p = dict(SWS_DEFAULT)
    p.update(params)
    lt.sliding_window_search(mask, **p)
    out = dict(detected=np.bool_(lt.detected_pixels))
    if lt.detected_pixels:
        out.update(left_y=arr_or_empty(lt.left_y), left_x=arr_or_empty(lt.left_x),
                   right_y=arr_or_empty(lt.right_y), right_x=arr_or_empty(lt.right_x),
                   left_centroids=np.asarray(lt.left_window_centroids, np.int64),
                   right_centroids=np.asarray(lt.right_window_centroids, np.int64))
        out.update(run_fit_validity(lt, float(p["partial"])))
    return p, out


def run_fit_validity(lt, partial):
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        lf, rf = lt.fit_poly()
    lt.valid_lane_lines = False
    lt.check_validity(lf, rf)
    pts = lt.get_poly_points(lf, rf, partial)
    return dict(left_coeffs=np.asarray(lf, np.float64), right_coeffs=np.asarray(rf, np.float64),
                valid=np.bool_(lt.valid_lane_lines),
                poly_points_len=np.asarray([len(pts[0]), len(pts[2])], np.int64),
                poly_left_x=np.asarray(pts[1], np.int64), poly_right_x=np.asarray(pts[3], np.int64),
                poly_left_y=np.asarray(pts[0], np.int64), poly_right_y=np.asarray(pts[2], np.int64))


def band_cases():
    cases = []
    for i, (noise, bw, dl, dr) in enumerate([(1e-3, 25, 0.0, 0.0), (1e-2, 30, 6.0, -4.0), (0.0, 25, 40.0, 0.0),
                                             (5e-2, 25, 1.5, 2.5), (1e-3, 60, 0.0, 0.0), (1e-3, 5, 0.3, 0.7)]):
        m, lc, rc = synth.synth_mask(60 + i, noise=noise)
        lc, rc = lc.copy(), rc.copy()
        lc[2] += dl
        rc[2] += dr
        cases.append((f"band{i}", m, lc, rc, dict(bandwidth=bw, ignore_bottom=30, partial=1)))
    m, lc, rc = synth.synth_mask(70, noise=1e-3)
    cases.append(("band_ignore0", m, lc, rc, dict(bandwidth=25, ignore_bottom=0, partial=1)))
    cases.append(("band_random50", synth.random_mask(71, density=0.5), lc, rc, dict(bandwidth=25, ignore_bottom=30, partial=1)))
    cases.append(("band_offimage", m, lc + np.array([0, 0, -600.0]), rc + np.array([0, 0, 500.0]),
                  dict(bandwidth=25, ignore_bottom=30, partial=1)))
    # half-integer band edges: strict inequalities matter
    cases.append(("band_halfint", m, np.array([0.0, 0.0, 440.0]), np.array([0.0, 0.0, 640.5]),
                  dict(bandwidth=25, ignore_bottom=30, partial=1)))
    return cases


def run_band(ref, mask, lc, rc, p):
    lt = new_tracker(ref)
    lt.last_left_coeffs, lt.last_right_coeffs = lc, rc
    lt.band_search(mask, **p)
    out = dict(detected=np.bool_(lt.detected_pixels))
    if lt.detected_pixels:
        out.update(left_y=arr_or_empty(lt.left_y), left_x=arr_or_empty(lt.left_x),
                   right_y=arr_or_empty(lt.right_y), right_x=arr_or_empty(lt.right_x))
        out.update(run_fit_validity(lt, float(p["partial"])))
    return out


def trace_plan():
    """Frame schedule of the process() trace: (kind, index).  Blank/noise frames force failures so
    that the fallback, n_reset and n_fail branches are exercised (lane_tracker.py:1142-1173)."""
    plan = []
    for i in range(36):
        if i in (6, 7) or 12 <= i <= 21:
            plan.append(("blank", i))      # nothing detectable: both tries fail
        elif i == 28:
            plan.append(("noise", i))      # iid-uniform frame: dense mask, band search still "succeeds"
        else:
            plan.append(("lane", i))
    return plan


def trace_frames(plan, seed=7):
    lanes = synth.stream_lanes(len(plan), seed=seed)
    frames = []
    for (kind, i) in plan:
        if kind == "lane":
            frames.append(lanes[i])
        elif kind == "noise":
            frames.append(synth.frame_uniform(9000 + i))
        else:
            frames.append(np.full_like(lanes[i], 128))
    return frames


def run_trace(ref):
    plan = trace_plan()
    frames = trace_frames(plan)
    lt = new_tracker(ref)
    rec = {k: [] for k in ("detected", "valid", "last_detection", "success", "counter", "left_avg", "right_avg",
                           "last_left", "last_right", "curve_radius", "eccentricity", "hist_len", "frame_sha")}
    nan3 = np.full(3, np.nan)
    for f in frames:
        rec["frame_sha"].append(hashlib.sha256(f.tobytes()).hexdigest())
        lt.process(f.copy(), partial=1)
        rec["detected"].append(bool(lt.detected_pixels))
        rec["valid"].append(bool(lt.valid_lane_lines))
        rec["last_detection"].append(int(lt.last_detection))
        rec["success"].append(int(lt.success))
        rec["counter"].append(int(lt.counter))
        rec["left_avg"].append(nan3 if lt.left_avg_coeffs is None else np.asarray(lt.left_avg_coeffs, np.float64))
        rec["right_avg"].append(nan3 if lt.right_avg_coeffs is None else np.asarray(lt.right_avg_coeffs, np.float64))
        rec["last_left"].append(nan3 if lt.last_left_coeffs is None else np.asarray(lt.last_left_coeffs, np.float64))
        rec["last_right"].append(nan3 if lt.last_right_coeffs is None else np.asarray(lt.last_right_coeffs, np.float64))
        rec["curve_radius"].append(-12345 if lt.average_curve_radius is None else int(lt.average_curve_radius))
        rec["eccentricity"].append(np.nan if lt.eccentricity is None else float(lt.eccentricity))
        rec["hist_len"].append(len(lt.left_fit_coeffs))
    out = {k: np.asarray(v) for k, v in rec.items()}
    out["plan_kind"] = np.asarray([k for k, _ in plan])
    out["success_ratio"] = np.asarray(lt.get_success_ratio(), np.float64)
    return out


def run_viz(ref, out_dir):
    """Search visualisations and the split view (lane_tracker.py:687-793)."""
    names = []
    for name, seed, params in [("sws_a", 11, {}), ("sws_curvy", 21, dict(search_range=60)),
                               ("sws_overlap", 39, dict(ignore_sides=0, window_width=400))]:
        kw = dict(curv=3e-4, slope=0.2) if name == "sws_curvy" else {}
        if name == "sws_overlap":
            kw = dict(left_base=(500, 520), sep=(545, 575), slope=0.0, curv=0.0)
        mask = synth.synth_mask(seed, noise=1e-3, **kw)[0]
        lt = new_tracker(ref)
        p = dict(SWS_DEFAULT)
        p.update(params)
        lt.sliding_window_search(mask, **p)
        assert lt.detected_pixels, name
        lf, rf = lt.fit_poly()
        vis = lt.visualize_sliding_window_search(mask, lf, rf, p["window_width"], p["window_height"], p["ignore_bottom"])
        d = pack(mask)
        d.update({"param_" + k: np.asarray(v) for k, v in p.items()})
        d.update(kind=np.asarray("sws"), vis=vis, left_coeffs=np.asarray(lf), right_coeffs=np.asarray(rf))
        np.savez_compressed(os.path.join(out_dir, f"viz_{name}.npz"), **d)
        names.append(name)
    for name, seed, bw, shift, partial in [("band_a", 61, 25, (0.0, 0.0), 1), ("band_wide", 62, 60, (5.0, -3.0), 1),
                                           ("band_edge", 63, 40, (-430.0, 380.0), 1)]:
        mask, lc, rc = synth.synth_mask(seed, noise=1e-3)
        lc, rc = lc.copy(), rc.copy()
        lc[2] += shift[0]
        rc[2] += shift[1]
        lt = new_tracker(ref)
        lt.last_left_coeffs, lt.last_right_coeffs = lc, rc
        lt.band_search(mask, bandwidth=bw, ignore_bottom=30, partial=partial)
        d = pack(mask)
        d.update(kind=np.asarray("band"), prev_left=lc, prev_right=rc, param_bandwidth=np.asarray(bw),
                 param_partial=np.asarray(partial), detected=np.bool_(lt.detected_pixels))
        if lt.detected_pixels:
            lf, rf = lt.fit_poly()
            vis = lt.visualize_band_search(mask, lf, rf, bw, partial)
            d.update(vis=vis, left_coeffs=np.asarray(lf), right_coeffs=np.asarray(rf))
        np.savez_compressed(os.path.join(out_dir, f"viz_{name}.npz"), **d)
        names.append((name, bool(lt.detected_pixels)))
    rng = np.random.default_rng(5)
    imgs = [rng.integers(0, 256, (72, 128, 3), dtype=np.uint8), rng.integers(0, 256, (110, 108, 3), dtype=np.uint8),
            rng.integers(0, 256, (110, 108, 3), dtype=np.uint8)]
    lt = new_tracker(ref)
    np.savez_compressed(os.path.join(out_dir, "viz_split.npz"), img0=imgs[0], img1=imgs[1], img2=imgs[2],
                        out=lt.triple_split_view(imgs))
    print("viz:", names)


def run_photos(ref, ref_dir, out_dir):
    """BASELINE config 1: real camera frames from the reference's test_images/ through the reference's
    process() (cv2 answered by the oracle).  The decoded frame is stored losslessly next to the results."""
    from PIL import Image
    names = []
    for name in ("test4", "straight_lines1", "test5"):
        frame = np.asarray(Image.open(os.path.join(ref_dir, "test_images", name + ".jpg")).convert("RGB"), np.uint8)
        Image.fromarray(frame).save(os.path.join(out_dir, f"photo_{name}.png"), optimize=True)
        lt = new_tracker(ref)
        captured = {}
        _flp = lt.find_lane_points

        def spy(img, **kw):
            binary, mode = _flp(img, **kw)
            captured.setdefault("masks", []).append(binary.copy())
            captured.setdefault("modes", []).append(mode)
            captured.setdefault("counts", []).append((len(lt.left_x), len(lt.right_x)) if lt.detected_pixels else (0, 0))
            return binary, mode
        lt.find_lane_points = spy
        lt.process(frame.copy(), partial=1)
        d = dict(frame_sha1=np.asarray(hashlib.sha1(frame.tobytes()).hexdigest()),
                 n_tries=np.asarray(len(captured["masks"])), modes=np.asarray(captured["modes"]),
                 counts=np.asarray(captured["counts"], np.int64),
                 detected=np.bool_(lt.detected_pixels), valid=np.bool_(lt.valid_lane_lines),
                 last_detection=np.asarray(lt.last_detection), success=np.asarray(lt.success))
        for i, m in enumerate(captured["masks"]):
            d[f"mask{i}_bits"] = np.packbits(m != 0)
            d[f"mask{i}_sha256"] = np.asarray(hashlib.sha256(m.tobytes()).hexdigest())
        if lt.valid_lane_lines:
            d.update(left_coeffs=np.asarray(lt.last_left_coeffs, np.float64), right_coeffs=np.asarray(lt.last_right_coeffs, np.float64),
                     curve_radius=np.asarray(int(lt.average_curve_radius)), eccentricity=np.asarray(float(lt.eccentricity)))
        np.savez_compressed(os.path.join(out_dir, f"photo_{name}.npz"), **d)
        names.append((name, bool(lt.detected_pixels), bool(lt.valid_lane_lines), captured["modes"], captured["counts"]))
    print("photos:", names)


def pack(mask):
    return dict(mask_bits=np.packbits(mask != 0), mask_shape=np.asarray(mask.shape, np.int64),
                mask_value=np.asarray(int(mask.max()) if mask.any() else 255, np.int64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--skip-trace", action="store_true")
    a = ap.parse_args()
    ref = import_reference(a.ref)
    os.makedirs(a.out, exist_ok=True)
    names = []
    for name, mask, params in sws_cases():
        p, out = run_sws(ref, mask, params)
        d = pack(mask)
        d.update({"param_" + k: np.asarray(v) for k, v in p.items()})
        d.update(out)
        np.savez_compressed(os.path.join(a.out, f"sws_{name}.npz"), **d)
        names.append((name, bool(out["detected"]), int(out.get("left_x", np.zeros(0)).size), bool(out.get("valid", False))))
    print("sws:", names)
    names = []
    for name, mask, lc, rc, p in band_cases():
        out = run_band(ref, mask, lc, rc, p)
        d = pack(mask)
        d.update({"param_" + k: np.asarray(v) for k, v in p.items()})
        d.update(prev_left=np.asarray(lc, np.float64), prev_right=np.asarray(rc, np.float64))
        d.update(out)
        np.savez_compressed(os.path.join(a.out, f"{name}.npz"), **d)
        names.append((name, bool(out["detected"]), int(out.get("left_x", np.zeros(0)).size)))
    print("band:", names)
    run_viz(ref, a.out)
    run_photos(ref, a.ref, a.out)
    if not a.skip_trace:
        tr = run_trace(ref)
        np.savez_compressed(os.path.join(a.out, "process_trace.npz"), **tr)
        print("trace valid:", tr["valid"].astype(int).tolist())
        print("trace last_detection:", tr["last_detection"].tolist())


if __name__ == "__main__":
    main()
