"""The only real-OpenCV outputs that exist for the cv2-backed front end: two figures the reference's author published
(README.md:114 `output_images/test4_warped.png` = cv2.undistort + cv2.warpPerspective of test_images/test4.jpg;
`output_images/calib_img_undist.png` = cv2.undistort of camera_calib/calibration02.jpg).  They are matplotlib renderings
(the 1080x1100 bird's-eye view drawn into 1090x1110 screen pixels, the 1280x720 frame into 1100x619), so the comparison is
"after the same resampling, within a grey level or a few" -- not bit-exact, but against pixels that OpenCV itself produced
with this calibration.  Kept as data under tests/golden/ref_figures/ (the figures and the decoded calibration photo).
CPU only."""
import os

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")

from lane_tracker_amd import calib
from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
FIG = os.path.join(HERE, "golden", "ref_figures")


def _rgb(path):
    return np.asarray(PIL.open(path).convert("RGB"))


def _resized(img, size, how):
    return np.asarray(PIL.fromarray(img).resize(size, how)).astype(np.float64)


@pytest.fixture(scope="module")
def oc():
    c = calib.reference_calibration()
    return O.make_calib(c["img_size"], c["warped_size"], c["cam_matrix"], c["dist_coeffs"], c["warp_matrices"][0])


def test_birds_eye_view_of_test4_matches_the_authors_opencv_figure(oc):
    src = _rgb(os.path.join(HERE, "golden", "photo_test4.png"))
    fig = _rgb(os.path.join(FIG, "test4_warped_figure.png")).astype(np.float64)
    axes = fig[10:10 + 1110, 40:40 + 1090]                       # the image area inside the axes frame
    inner = (slice(20, -20), slice(20, -20))                       # away from the anti-aliased frame line
    got = _resized(O.front_end(oc, src), (1090, 1110), PIL.BILINEAR)
    d = np.abs(axes - got)[inner]
    no_undistort = np.abs(axes - _resized(O.warp(oc, src), (1090, 1110), PIL.BILINEAR))[inner]
    print("\nbird's-eye view vs the author's figure: mean |diff| %.3f levels, median %.1f, within 3 levels %.1f %%; "
          "without the undistortion step %.2f" % (d.mean(), np.median(d), 100 * np.mean(d <= 3), no_undistort.mean()))
    assert d.mean() < 1.0 and np.median(d) <= 1.0 and np.mean(d <= 3) > 0.94
    assert no_undistort.mean() > 8 * d.mean()                     # the check can tell a 1-2 pixel geometric error


def test_undistorted_calibration_photo_matches_the_authors_opencv_figure(oc):
    src = _rgb(os.path.join(FIG, "calibration02.png"))
    assert src.shape == (720, 1280, 3)
    fig = _rgb(os.path.join(FIG, "calib_img_undist_figure.png")).astype(np.float64)
    axes = fig[10:10 + 619, 33:33 + 1100]
    inner = (slice(10, -10), slice(10, -10))
    got = _resized(O.undistort(oc, src), (1100, 619), PIL.BOX)
    d = np.abs(axes - got)[inner]
    raw = np.abs(axes - _resized(src, (1100, 619), PIL.BOX))[inner]
    print("\nundistorted chessboard vs the author's figure (drawn at 0.86x): mean |diff| %.2f levels, median %.1f; "
          "the distorted photo itself %.1f" % (d.mean(), np.median(d), raw.mean()))
    assert d.mean() < 4.5 and np.median(d) <= 4.0
    assert raw.mean() > 6 * d.mean()                              # the whole frame, where the lens moves pixels by tens of pixels


# ---- the thresholded bird's-eye views of test4 (README.md:120) -------------------------------------------------------
# The author published the bilateral and the cv2.adaptiveThreshold masks of "the raw warped colour channels with no prior
# tophat" of test4, without the parameters.  Each figure is a {0,255} image drawn 1080x1100 -> 1090x1110; the oracle's
# chain undistort -> warp -> R / Lab-b -> threshold -> OR -> open 5x5 is run over a parameter grid and compared by IoU.
# These BOUND the cv2-backed stages a1, a2, a3.1, a3.3 / a3.4, a3.6, a3.7 with pixels real OpenCV produced; they do not pin
# them bit for bit (unknown parameters, re-rendered figure), and say nothing about the top-hats (a3.2).
def _iou(mask, fig_white):
    up = np.asarray(PIL.fromarray(mask).resize((1090, 1110), PIL.NEAREST)) > 127
    return float((up & fig_white).sum()) / float(max(1, (up | fig_white).sum()))


@pytest.fixture(scope="module")
def test4_planes(oc):
    bev = O.front_end(oc, _rgb(os.path.join(HERE, "golden", "photo_test4.png")))
    return np.ascontiguousarray(bev[:, :, 0]), O.lab_b(bev)


def _merged(tr, tb):
    return np.where((tr > 0) | (tb > 0), 255, 0).astype(np.uint8)


def test_bilateral_threshold_of_test4_matches_the_authors_figure(test4_planes):
    R, b = test4_planes
    fig = np.asarray(PIL.open(os.path.join(FIG, "test4_thresh_bilat_axes.png")).convert("L")) > 127
    assert fig.shape == (1110, 1090) and 0.02 < fig.mean() < 0.035
    grid = {}
    tb = {(k, c): O.bilateral_adaptive_threshold(b, k, c) for k in (25, 35, 45) for c in (3, 5, 8)}
    for kr in (10, 15, 20, 25):
        for cr in (8, 15, 19, 20, 21, 25):
            tr = O.bilateral_adaptive_threshold(R, kr, cr)
            for (kb, cb), t in tb.items():
                grid[(kr, cr, kb, cb)] = _iou(O.morph_open(_merged(tr, t), 5), fig)
    best = max(grid, key=grid.get)
    tr, t = O.bilateral_adaptive_threshold(R, *best[:2]), tb[best[2:]]
    merged = _merged(tr, t)
    opened = O.morph_open(merged, 5)
    flipped = O.morph_open(_merged(O.bilateral_adaptive_threshold(R, best[0], best[1], mode="ceil"), O.bilateral_adaptive_threshold(b, best[2], best[3], mode="ceil")), 5)
    report = dict(best=best, iou=round(grid[best], 4), defaults_15_8_35_5=round(grid[(15, 8, 35, 5)], 4), no_open=round(_iou(merged, fig), 4),
                  shifted_1px_x=round(_iou(np.roll(opened, 1, 1), fig), 4), shifted_1px_y=round(_iou(np.roll(opened, 1, 0), fig), 4),
                  ceil_mode=round(_iou(flipped, fig), 4), r_only=round(_iou(O.morph_open(tr, 5), fig), 4))
    print("\nbilateral threshold of test4 vs the author's OpenCV figure:", report)
    assert best == (15, 20, 35, 5)                              # a sharp maximum at round numbers: the author's parameters
    assert grid[best] > 0.94
    runner_up = max(v for k, v in grid.items() if k != best)
    assert grid[best] - runner_up > 0.01
    # the comparison can tell a one-pixel shift, a missing open, a flipped inequality and a missing channel
    assert report["shifted_1px_x"] < grid[best] - 0.12 and report["shifted_1px_y"] < grid[best] - 0.05
    assert report["no_open"] < grid[best] - 0.15 and report["ceil_mode"] < 0.05 and report["r_only"] < 0.7


def test_adaptive_mean_threshold_of_test4_matches_the_authors_figure(test4_planes):
    R, b = test4_planes
    fig = np.asarray(PIL.open(os.path.join(FIG, "test4_thresh_cv2adapt_axes.png")).convert("L")) > 127
    assert fig.shape == (1110, 1090) and 0.06 < fig.mean() < 0.09
    grid = {}
    tb = {(k, c): O.adaptive_mean_threshold(b, k, c) for k in (25, 35, 45) for c in (3, 5, 8)}
    for kr in (15, 25, 35):
        for cr in (5, 8, 10, 11, 12, 13, 15):
            tr = O.adaptive_mean_threshold(R, kr, cr)
            for (kb, cb), t in tb.items():
                grid[(kr, cr, kb, cb)] = _iou(O.morph_open(_merged(tr, t), 5), fig)
    best = max(grid, key=grid.get)
    tr, t = O.adaptive_mean_threshold(R, *best[:2]), tb[best[2:]]
    merged = _merged(tr, t)
    opened = O.morph_open(merged, 5)
    report = dict(best=best, iou=round(grid[best], 4), no_open=round(_iou(merged, fig), 4),
                  shifted_1px_x=round(_iou(np.roll(opened, 1, 1), fig), 4), shifted_1px_y=round(_iou(np.roll(opened, 1, 0), fig), 4),
                  block_size_plus_2=round(_iou(O.morph_open(_merged(O.adaptive_mean_threshold(R, best[0] + 2, best[1]), t), 5), fig), 4))
    print("\ncv2.adaptiveThreshold of test4 vs the author's OpenCV figure:", report)
    assert best == (25, 12, 35, 5) and grid[best] > 0.975
    assert report["shifted_1px_x"] < grid[best] - 0.12 and report["no_open"] < grid[best] - 0.1
    assert report["block_size_plus_2"] < grid[best] - 0.01


def test_lab_b_plane_correlates_with_the_authors_lab_b_panel():
    """Weak anchor for a3.1: the panel is an autoscaled grey map at 0.3x of a frame close to (not identical with) test4, so
    only the affine-invariant correlation is meaningful -- it separates Lab-b from every RGB plane by a wide margin."""
    src = _rgb(os.path.join(HERE, "golden", "photo_test4.png"))
    panel = _rgb(os.path.join(FIG, "color_channels10_test4_lab_b_panel.png"))[:, :, 0].astype(np.float64)
    h, w = panel.shape

    def corr(plane):
        small = _resized(np.ascontiguousarray(plane), (w, h), PIL.BILINEAR)
        x, y = small[2:-2, 2:-2].ravel() - small[2:-2, 2:-2].mean(), panel[2:-2, 2:-2].ravel() - panel[2:-2, 2:-2].mean()
        return float((x * y).sum() / np.sqrt((x * x).sum() * (y * y).sum()))
    got = {"lab_b": corr(O.lab_b(src)), "r": corr(src[:, :, 0]), "g": corr(src[:, :, 1]), "b": corr(src[:, :, 2])}
    print("\ncorrelation with the author's LAB B-Channel panel:", {k: round(v, 3) for k, v in got.items()})
    assert got["lab_b"] > 0.95 and max(got["r"], got["g"], got["b"]) < 0.0


def _demo3_pair_scores(oc, pair):
    """IoU of the oracle's mask of the camera frame recovered from `search_lane_result0<pair>` with the mask pixels of
    `search_lane_vis0<pair>`, for the full chain and for the chain with one stage changed."""
    from scipy import ndimage as ndi
    from lane_tracker_amd import settings
    crop = _rgb(os.path.join(FIG, "search_lane_result%02d_axes.png" % pair))
    frame = np.asarray(PIL.fromarray(crop).resize((1280, 720), PIL.BICUBIC)).astype(np.int32)
    R, G = frame[..., 0], frame[..., 1]
    poly = G - R > 38                                    # grey asphalt, yellow and white paint all have G - R near 0 outside the polygon
    poly[:430] = False
    poly = ndi.binary_fill_holes(ndi.binary_closing(ndi.binary_opening(poly, iterations=2), iterations=6))
    lab, nl = ndi.label(poly)
    poly = lab == (1 + int(np.argmax(ndi.sum(poly, lab, range(1, nl + 1)))))
    assert 90000 < poly.sum() < 130000                   # the lane polygon of the figure
    frame[..., 1] = np.where(poly, np.clip(G - 76, 0, 255), G)
    bev = O.front_end(oc, frame.astype(np.uint8))
    ax = _rgb(os.path.join(FIG, "search_lane_vis%02d_axes.png" % pair)).astype(np.int32)
    near = lambda c: np.abs(ax - np.array(c)).sum(-1) < 90
    want = near((255, 255, 255)) | near((255, 0, 0)) | near((0, 0, 255))        # mask pixels: plain, found left, found right
    visible = ~near((255, 255, 0))                                              # (the fitted curves are drawn over the mask)

    def iou(mask):
        got = np.asarray(PIL.fromarray(mask).resize((ax.shape[1], ax.shape[0]), PIL.BOX)) > 127
        return float((got & want & visible).sum()) / float(((got | want) & visible).sum())
    P = settings.DEMO_3["process"]
    kw = dict(ksize_r=P["ksize_r"], C_r=P["C_r"], ksize_b=P["ksize_b"], C_b=P["C_b"], mask_noise=True, noise_thresh=P["noise_thresh"],
              ksize_noise=P["ksize_noise"], C_noise=P["C_noise"])
    Rp, bp = np.ascontiguousarray(bev[..., 0]), O.lab_b(bev)

    def chain(kR, kB, open5=True):                        # filter_lane_points with other (or no) top-hats
        thR, thB = (O.tophat(Rp, kR) if kR else Rp), (O.tophat(bp, kB) if kB else bp)
        m = (O.bilateral_adaptive_threshold(thR, 15, 8) > 0) | (O.bilateral_adaptive_threshold(thB, 35, 5) > 0)
        m &= (~(bp >= 140)) | (O.bilateral_adaptive_threshold(bp, 65, 10) > 0)
        m = (m * 255).astype(np.uint8)
        return O.morph_open(m, 5) if open5 else m
    full = O.filter_lane_points(bev, O.filter_params(**kw))
    assert np.array_equal(chain(29, 55), full)
    s_full = iou(full)
    s_no_noise = iou(O.filter_lane_points(bev, O.filter_params(**dict(kw, mask_noise=False))))
    s_try2 = iou(O.filter_lane_points(bev, O.filter_params(filter_type="neighborhood", ksize_r=15, C_r=5, ksize_b=35, C_b=5)))
    s_no_open = iou(chain(29, 55, open5=False))
    s_no_tophat = iou(chain(0, 0))
    sweep = {k: round(iou(chain(*k)), 3) for k in ((27, 55), (31, 55), (29, 53), (29, 57))}
    return dict(full=s_full, no_noise=s_no_noise, try2=s_try2, no_open=s_no_open, no_tophat=s_no_tophat, sweep=sweep)


def test_demo3_mask_bounds_the_greenery_mask_the_open_and_the_filter_type(oc):
    """The one published figure pair that went through the WHOLE `filter_lane_points` with real OpenCV -- top-hats, bilateral
    thresholds, greenery mask (Demo 3 settings, tracker_settings.md:74-111), open: `search_lane_result01.png` (the annotated
    camera frame, drawn at 0.70x) and `search_lane_vis01.png` (its bird's-eye mask under the search visualisation).  The
    camera frame is recovered from the figure -- resampled back to 1280 x 720, the lane polygon's addWeighted(.., 0.3) taken
    out of the green channel -- and sent through the oracle; the mask is compared with the figure's white / red / blue
    pixels at figure resolution.

    What this bounds (IoU, figure resolution): the greenery mask (0.53 with, 0.29 without), the 5x5 open (0.43 without),
    the filter type (second-try set 0.23).  What it does NOT resolve: the top-hats.  With them 0.528, without 0.504, and the
    structuring-element sizes 25..33 / 51..59 all give 0.526-0.529 -- a frame that has been through a 0.70x resampling and back
    has lost the contrast detail the top-hats act on.  So a3.2 stays without an OpenCV pixel; this is a bound on a3.5-a3.7,
    not a pin of anything."""
    sc = _demo3_pair_scores(oc, 1)
    s_full, s_no_noise, s_try2, s_no_open, s_no_tophat, sweep = (sc[k] for k in ("full", "no_noise", "try2", "no_open", "no_tophat", "sweep"))
    print("\ndemo-3 mask vs the author's figure: IoU %.3f; without the greenery mask %.3f, without the open %.3f, second-try filter %.3f; "
          "without the top-hats %.3f, SE sizes +-2: %s -- the top-hats are not resolved by this figure" % (s_full, s_no_noise, s_no_open, s_try2, s_no_tophat, sweep))
    assert s_full > 0.45
    assert s_no_noise < s_full - 0.15 and s_try2 < s_full - 0.2 and s_no_open < s_full - 0.05
    assert abs(s_no_tophat - s_full) < 0.06               # documented: this comparison cannot tell (keep the claim honest)


def test_demo3_second_frame_pair_does_not_resolve_the_tophats_either(oc):
    """The last unused figure pair (README.md:146-148): the SECOND frame of the same video (`search_lane_result02.png`) and its
    band-search visualisation (`search_lane_vis02.png`), through the same machinery.  Same kind of data (an annotated frame drawn
    at 0.70x, resampled back), same verdict: it bounds the greenery mask, the open and the filter type once more and cannot tell
    the chain with top-hats from the chain without, nor the structuring-element sizes apart.  With this the subject is closed:
    no published OpenCV pixel resolves a3.2."""
    sc = _demo3_pair_scores(oc, 2)
    print("\ndemo-3 second frame vs the author's figure: IoU %.3f; without the greenery mask %.3f, without the open %.3f, second-try filter "
          "%.3f; without the top-hats %.3f, SE sizes +-2: %s" % (sc["full"], sc["no_noise"], sc["no_open"], sc["try2"], sc["no_tophat"], sc["sweep"]))
    assert sc["full"] > 0.40
    assert sc["no_noise"] < sc["full"] - 0.10 and sc["try2"] < sc["full"] - 0.15
    assert abs(sc["no_tophat"] - sc["full"]) < 0.08           # documented: cannot tell
    assert max(sc["sweep"].values()) - min(sc["sweep"].values()) < 0.02
