"""Presentation stage, host side: the polygon row intervals the overlay kernel consumes,
the NumPy visualisations and the split view, checked against the oracle's generic fillPoly / addWeighted /
resize restatements and against fixtures produced by the reference's own visualisation methods
(tests/gen_golden.py; their cv2 calls answered by the oracle, so those three calls stay unpinned)."""
import os

import numpy as np
import pytest

from helpers import golden_files, params_of, unpack_mask
from lane_tracker_amd import _native, overlay, utils
from oracle import oracle as O

H, W = 1100, 1080


def _spans_image(spans, w):
    xs = np.arange(w)
    return (((xs[None, :] >= spans[:, 0:1]) & (xs[None, :] <= spans[:, 1:2])) * 255).astype(np.uint8)


def _oracle_polygon(h, w, ly, lx, ry, rx):
    img = np.zeros((h, w), np.uint8)
    if len(lx) + len(rx):
        O.fill_poly(img, np.concatenate([np.stack([lx, ly], 1), np.stack([rx, ry], 1)[::-1]]), 255)
    return img


@pytest.mark.parametrize("seed", range(6))
def test_lane_polygon_spans_match_generic_fillpoly(seed):
    rng = np.random.default_rng(seed)
    for t in range(40):
        partial = 1 if t % 2 else 0.5
        lf = np.array([rng.uniform(-6e-4, 6e-4), rng.uniform(-1.2, 1.2), rng.uniform(100, 700)])
        rf = lf + np.array([rng.uniform(-1e-4, 1e-4), rng.uniform(-0.2, 0.2), rng.uniform(-60, 400)])
        ly, lx, ry, rx = O.get_poly_points((W, H), lf, rf, partial)
        if t % 7 == 0:
            ry, rx = ry[:0], rx[:0]
        if t % 11 == 0:
            ly, lx = ly[:0], lx[:0]
        spans = _native.lane_polygon_spans(H, ly, lx, ry, rx)
        assert np.array_equal(_spans_image(spans, W), _oracle_polygon(H, W, ly, lx, ry, rx)), (seed, t)


def test_polygon_spans_general_chains():
    """Chains with row gaps, repeated rows and points outside the image (generic callers of draw_lane)."""
    rng = np.random.default_rng(99)
    for t in range(60):
        n1, n2 = rng.integers(1, 40, 2)
        ly = np.sort(rng.choice(np.arange(-5, 130), n1, replace=False))
        ry = np.sort(rng.choice(np.arange(-5, 130), n2, replace=False))
        lx = rng.integers(-10, 60, n1)
        rx = lx.mean().astype(int) + rng.integers(30, 90, n2)
        spans = _native.lane_polygon_spans(120, ly, lx, ry, rx)
        got = _spans_image(spans, 140)
        want = _oracle_polygon(120, 140, ly, lx, ry, rx)
        assert np.array_equal(got, want), t


def test_polygon_spans_empty_and_errors():
    e = np.zeros(0, np.int64)
    spans = _native.lane_polygon_spans(50, e, e, e, e)
    assert (spans[:, 0] > spans[:, 1]).all()
    with pytest.raises(ValueError):
        _native.lane_polygon_spans(0, e, e, e, e)


def test_oracle_fill_poly_against_point_in_polygon():
    """The oracle's generic rasteriser on convex polygons: every pixel strictly inside is painted and
    nothing farther than one pixel from the polygon is."""
    rng = np.random.default_rng(3)
    for _ in range(20):
        ang = np.sort(rng.uniform(0, 2 * np.pi, 7))
        pts = np.stack([40 + 30 * np.cos(ang), 40 + 30 * np.sin(ang)], 1).round().astype(np.int32)
        img = np.zeros((80, 80), np.uint8)
        O.fill_poly(img, pts, 255)
        yy, xx = np.mgrid[0:80, 0:80]
        inside = np.ones((80, 80), bool)
        near = np.ones((80, 80), bool)
        for i in range(len(pts)):
            a, b = pts[i], pts[(i + 1) % len(pts)]
            cross = (b[0] - a[0]) * (yy - a[1]) - (b[1] - a[1]) * (xx - a[0])
            nrm = max(np.hypot(*(b - a)), 1e-9)
            inside &= cross > 0
            near &= cross / nrm > -1.0
        assert (img[inside] == 255).all()
        assert (img[~near] == 0).all()


def test_add_weighted_matches_oracle():
    rng = np.random.default_rng(4)
    a = rng.integers(0, 256, (64, 65, 3), dtype=np.uint8)
    b = rng.integers(0, 256, (64, 65, 3), dtype=np.uint8)
    for alpha, beta, gamma in [(1, 0.3, 0), (1, 0.5, 0.0), (0.7, 0.3, 2.5), (1, 1, 0)]:
        assert np.array_equal(overlay.add_weighted(a, alpha, b, beta, gamma), O.add_weighted(a, alpha, b, beta, gamma))
    # the ties that separate round-half-even from round-half-up
    assert O.add_weighted(np.array([0, 1], np.uint8), 1, np.array([255, 255], np.uint8), 0.5, 0).tolist() == [128, 128]
    assert O.add_weighted(np.array([10], np.uint8), 1, np.array([255], np.uint8), 0.3, 0).tolist() == [86]


@pytest.mark.parametrize("shape,dsize", [((1100, 1080, 3), (640, 652)), ((37, 53), (20, 11)), ((20, 30, 3), (61, 47)),
                                         ((9, 9), (9, 9)), ((2, 2, 3), (7, 3))])
def test_resize_linear_matches_oracle(shape, dsize):
    img = np.random.default_rng(5).integers(0, 256, shape, dtype=np.uint8)
    assert np.array_equal(utils.resize_linear(img, dsize), O.resize_linear(img, dsize))


def test_resize_linear_known_values():
    img = (np.arange(16, dtype=np.uint8).reshape(4, 4) * 10)
    assert utils.resize_linear(img, (2, 2)).tolist() == [[25, 45], [105, 125]]      # exact 2x2 box means
    assert np.array_equal(utils.resize_linear(img, (4, 4)), img)
    up = utils.resize_linear(np.array([[0, 100]], np.uint8), (4, 1))
    assert up.tolist() == [[0, 25, 75, 100]]


def test_window_mask_slicing():
    img = np.zeros((100, 80), np.uint8)
    m = overlay.window_mask(img, 30, 40, 10, 0, 10)
    assert m.sum() == 40 * 25 and m[50:90, 0:25].all()
    m = overlay.window_mask(img, 31, 37, 70.5, 1, 10)
    assert m[16:53, 55:80].all() and m.sum() == 37 * 25


@pytest.mark.parametrize("path", golden_files("viz_sws_"))
def test_visualize_sliding_window_search_golden(path):
    d = np.load(path)
    mask, p = unpack_mask(d), params_of(d)
    r = O.sliding_window_search(mask, O.search_params(**{k: v for k, v in p.items()}))
    pts = O.get_poly_points((mask.shape[1], mask.shape[0]), d["left_coeffs"], d["right_coeffs"])
    vis = overlay.visualize_sliding_window_search(mask, r["left_centroids"], r["right_centroids"],
                                                  (r["left_y"], r["left_x"]), (r["right_y"], r["right_x"]), pts,
                                                  p["window_width"], p["window_height"], p["ignore_bottom"])
    assert vis.shape == d["vis"].shape and np.array_equal(vis, d["vis"])


@pytest.mark.parametrize("path", golden_files("viz_band_"))
def test_visualize_band_search_golden(path):
    d = np.load(path)
    mask = unpack_mask(d)
    bw, partial = int(d["param_bandwidth"]), d["param_partial"].item()
    r = O.band_search(mask, d["prev_left"], d["prev_right"], O.search_params(bandwidth=bw, ignore_bottom=30, partial=partial))
    assert r["detected"] == bool(d["detected"])
    size = (mask.shape[1], mask.shape[0])
    band = O.get_poly_points(size, d["prev_left"], d["prev_right"], partial)
    fit = O.get_poly_points(size, d["left_coeffs"], d["right_coeffs"])
    vis = overlay.visualize_band_search(mask, (r["left_y"], r["left_x"]), (r["right_y"], r["right_x"]), band, fit, bw)
    assert np.array_equal(vis, d["vis"])


def test_triple_split_view_golden():
    d = np.load(golden_files("viz_split")[0])
    imgs = [d["img0"], d["img1"], d["img2"]]
    # the method needs no device state: call it unbound
    from lane_tracker_amd.lane_tracker import LaneTracker
    out = LaneTracker.triple_split_view(None, imgs)
    assert out.shape == d["out"].shape and np.array_equal(out, d["out"])
    # a one-channel third image (no pixels detected -> the bare mask) fills all three channels
    out2 = LaneTracker.triple_split_view(None, [imgs[0], imgs[1], imgs[2][:, :, 0]])
    assert np.array_equal(out2[72:, 64:, 0], out2[72:, 64:, 2])


def _atlas_blend(frame, font, lines, origin=(20, 8), step=35):
    """k_overlay_text / lt_text_blend_host in NumPy: every glyph cell up to its advance, white over the frame."""
    atlas, adv, first_char = font
    want = frame.astype(np.int32)
    for j, line in enumerate(lines):
        x = origin[0]
        for ch in line:
            g = ord(ch) - first_char
            cell = atlas[g][:, :adv[g]].astype(np.int32)
            y = origin[1] + step * j
            reg = want[y:y + cell.shape[0], x:x + cell.shape[1]]
            reg += ((255 - reg) * cell[:reg.shape[0], :reg.shape[1], None] + 127) // 255
            x += int(adv[g])
    return want.astype(np.uint8)


def test_text_on_the_host_equals_the_atlas_blend():
    """lt_text_blend_host (the calling thread) and lt_host_text_async_group (the copy threads: rows copied from the source frames,
    then the lines) against the NumPy statement the GPU text kernel is tested with -- no GPU needed."""
    import pytest
    from lane_tracker_amd import _native, overlay
    font = overlay.font_atlas()
    if font is None:
        pytest.skip("Pillow is not installed: no glyph atlas")
    rng = np.random.default_rng(11)
    n, H, W = 5, 144, 420
    src = rng.integers(0, 256, (n, H, W, 3), dtype=np.uint8)
    src[1] = 0
    texts = [["Curve Radius: %d m" % (1000 + 7 * i), "Eccentricity: %.2f m" % (-0.3 + 0.1 * i), "Frame: %d" % i][:1 + i % 3] for i in range(n)]
    texts[3] = ["Lane Line Detection Failed", "Frame: 3"]
    buf, nl = _native.text_bytes(texts)
    assert nl == 3 and len(buf) == n * 3 * 40
    # in place, on the calling thread
    got = src.copy()
    _native.text_blend(got, font, buf, nl)
    for i in range(n):
        assert np.array_equal(got[i], _atlas_blend(src[i], font, texts[i])), i
    assert (got[1] == 255).any()
    # on the copy threads: rows [t0, t1) from the source frames, then the text; the other rows of dst are left alone
    gh = font[0].shape[1]
    t0, t1 = 8, min(8 + 2 * 35 + gh, H)
    dst = np.full_like(src, 77)
    g = _native.host_copy_group()
    _native.host_text_async(g, dst, src, (t0, t1), font, buf, nl)
    assert _native.load().lt_host_copy_wait_group(g) == 0
    for i in range(n):
        want = _atlas_blend(src[i], font, texts[i])
        assert np.array_equal(dst[i, t0:t1], want[t0:t1]), i
        assert (dst[i, :t0] == 77).all() and (dst[i, t1:] == 77).all()
    # rows only (no font): a plain copy of the run
    dst2 = np.zeros_like(src)
    _native.host_text_async(g, dst2, src, (10, 20), None, None, 0)
    assert _native.load().lt_host_copy_wait_group(g) == 0
    assert np.array_equal(dst2[:, 10:20], src[:, 10:20]) and not dst2[:, :10].any() and not dst2[:, 20:].any()
    # two runs (everything but a middle run: what a window's strips leave to the host), text over the first
    dst3 = np.full_like(src, 9)
    _native.host_text_async(g, dst3, src, (0, 125, 135, H), font, buf, nl)
    assert _native.load().lt_host_copy_wait_group(g) == 0
    for i in range(n):
        want = _atlas_blend(src[i], font, texts[i])
        assert np.array_equal(dst3[i, :125], want[:125]) and np.array_equal(dst3[i, 135:], want[135:]) and (dst3[i, 125:135] == 9).all(), i
    # text that runs off the right edge and a line origin near the bottom are clipped, not wrapped
    small = np.zeros((1, 40, 64, 3), np.uint8)
    b2, n2 = _native.text_bytes([["WWWWWWWWWWWWWWWW"]])
    _native.text_blend(small, font, b2, n2, origin=(20, 30))
    assert small.any()
    _native.host_copy_group_release(g)


def test_vector_text_blend_is_the_scalar_blend_for_every_value_and_alpha():
    """The 16-pixels-a-step form of lt_text_blend_host (AVX2; t / 255 as (t + 1 + (t >> 8)) >> 8) against the NumPy statement for
    every frame value 0..255 under every alpha 0..255, at cell widths that leave ragged last steps, flush against the right edge
    (the scalar form takes over where a step would leave the row) and against the scalar build switch."""
    import subprocess, sys, hashlib
    from lane_tracker_amd import _native
    rng = np.random.default_rng(3)
    for gw, adv_w in ((16, 16), (21, 19), (33, 33), (12, 7)):
        ng, gh = 16, 16
        atlas = rng.permutation(np.arange(ng * gh * gw) % 256).astype(np.uint8).reshape(ng, gh, gw)
        atlas[0, 0, :min(gw, 8)] = [0, 1, 2, 127, 128, 254, 255, 255][:min(gw, 8)]
        adv = np.full(ng, adv_w, np.uint8)
        font = (atlas, adv, 65)
        text = "".join(chr(65 + g) for g in range(ng))
        for W in (adv_w * ng + 20 + 40, adv_w * ng + 20, adv_w * ng + 20 - 5):      # room to spare / flush / clipped
            frames = np.empty((256, gh + 12, W, 3), np.uint8)
            frames[:] = np.arange(256, dtype=np.uint8)[:, None, None, None]
            frames[:, :, :, 1] = 255 - frames[:, :, :, 0]
            frames[:, :, :, 2] = (frames[:, :, :, 0].astype(np.int32) * 7 + 3).astype(np.uint8)
            buf, nl = _native.text_bytes([[text]] * 256)
            got = frames.copy()
            _native.text_blend(got, font, buf, nl, origin=(20, 8))
            for v in (0, 1, 2, 100, 127, 128, 200, 254, 255) if W != adv_w * ng + 60 else range(256):
                want = frames[v].astype(np.int32)
                x = 20
                for g in range(ng):
                    cell = atlas[g][:, :adv_w].astype(np.int32)
                    reg = want[8:8 + gh, x:x + adv_w]
                    reg += ((255 - reg) * cell[:, :reg.shape[1], None] + 127) // 255
                    x += adv_w
                assert np.array_equal(got[v], want.astype(np.uint8)), (gw, adv_w, W, v)
    # the same text through the scalar switch, in a process of its own (the library reads the switch once)
    code = ("import numpy as np, hashlib; from lane_tracker_amd import _native, overlay; f = overlay.font_atlas();\n"
            "rng = np.random.default_rng(5); a = rng.integers(0, 256, (3, 144, 420, 3), dtype=np.uint8)\n"
            "b, n = _native.text_bytes([['Curve Radius: 1234 m', 'Eccentricity: -0.25 m', 'Frame: 77']] * 3)\n"
            "_native.text_blend(a, f, b, n) if f is not None else None; print(hashlib.sha256(a.tobytes()).hexdigest())")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for sw in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=dict(os.environ, LT_TEXT_SCALAR=sw), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-800:]
        outs.append(r.stdout.strip())
    assert outs[0] == outs[1]


def test_line_iterator_in_closed_form_is_the_walk():
    """k_lane_spans_from_fit (csrc/k_overlay.hip) lets a whole workgroup take the pixels of a long polygon edge, pixel i of
    OpenCV's LineIterator in closed form: with len = max(|dx|, |dy|), across = min(|dx|, |dy|), the minor coordinate has advanced
    T_i = max(0, ceil((2 across i - len) / (2 len))) times before step i.  Held here against the walk itself -- lt_lane_polygon_spans
    of a two-vertex polygon is the hull per row of exactly that walk's pixels -- for random segments, no GPU needed."""
    rng = np.random.default_rng(17)
    for _ in range(400):
        xa, ya, xb, yb = (int(v) for v in rng.integers(0, 120, 4))
        if (xa, ya) == (xb, yb):
            continue
        x0, y0, x1, y1 = (xa, ya, xb, yb) if xa <= xb else (xb, yb, xa, ya)      # the iterator starts at the left end point
        adx, ady, ystep = x1 - x0, abs(y1 - y0), (-1 if y1 < y0 else 1)
        tall = ady > adx
        ln, across = (ady, adx) if tall else (adx, ady)
        lo, hi = np.full(120, 32767), np.full(120, -32768)
        for i in range(ln + 1):
            num = 2 * across * i - ln
            T = 0 if num <= 0 else (num + 2 * ln - 1) // (2 * ln)
            x, y = (x0 + T, y0 + ystep * i) if tall else (x0 + i, y0 + ystep * T)
            lo[y], hi[y] = min(lo[y], x), max(hi[y], x)
        spans = _native.lane_polygon_spans(120, [ya], [xa], [yb], [xb])          # closed polygon of two vertices: the edge, there and back
        assert np.array_equal(spans[:, 0], lo) and np.array_equal(spans[:, 1], hi), (xa, ya, xb, yb)
