"""GPU parity tests proper: the HIP path, called through the C ABI (ctypes), against the CPU oracle on
the same seeded inputs and against the fixtures generated from the reference (tests/golden/).
Bit-exact for every integer/byte/index result; polynomial coefficients within the BASELINE
tolerance (1e-4 relative with the absolute floor of SURVEY.md 8(a), see helpers.coeff_close)."""
import os

import numpy as np
import pytest

from helpers import coeff_close, golden_files, params_of, unpack_mask

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from lane_tracker_amd import _native
    _native.load()
    return _native


@pytest.fixture(scope="module")
def cal():
    from lane_tracker_amd import calib
    return calib.reference_calibration()


@pytest.fixture(scope="module")
def ctx(nat, cal):
    c = nat.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"],
                    cal["warp_matrices"][0], device=0, capacity=8)
    yield c
    c.close()


@pytest.fixture(scope="module")
def frames():
    from lane_tracker_amd import synth
    r = synth.SceneRenderer()
    fr = [synth.frame_uniform(1), synth.frame_uniform(2), r.render(11)[0], r.render(12)[0],
          np.zeros((720, 1280, 3), np.uint8), np.full((720, 1280, 3), 255, np.uint8)]
    return np.stack(fr, 0)


def assert_same(got, want, what):
    got, want = np.asarray(got), np.asarray(want)
    assert got.shape == want.shape, f"{what}: shape {got.shape} != {want.shape}"
    bad = np.argwhere(got != want)
    assert bad.shape[0] == 0, (f"{what}: {bad.shape[0]} of {got.size} differ; first at {bad[:5].tolist()} "
                               f"got {got[tuple(bad[0])]} want {want[tuple(bad[0])]}")


def test_info_and_row_window(ctx, oracle, ref_calib):
    i = ctx.info()
    assert (i.src_row0, i.src_row1) == oracle.warp_source_rows(ref_calib) == (457, 695)
    assert i.alg_bytes_mask == 238 * 1280 * 3 + 1100 * 1080
    assert i.cu_count > 0


def test_front_end_bit_exact(ctx, oracle, ref_calib, frames):
    n = frames.shape[0]
    ctx.upload_frames(frames)
    ctx.mask_run(n)
    und = ctx.download_undistorted(n)
    R, B = ctx.download_plane(0, n), ctx.download_plane(1, n)
    r0, r1 = oracle.warp_source_rows(ref_calib)
    for k in range(n):
        assert_same(und[k], oracle.undistort(ref_calib, frames[k])[r0:r1], f"undistorted rows, frame {k}")
        bev = oracle.front_end(ref_calib, frames[k])
        assert_same(R[k], bev[:, :, 0], f"R plane, frame {k}")
        assert_same(B[k], oracle.lab_b(bev), f"Lab-b plane, frame {k}")


@pytest.mark.parametrize("kw", [dict(), dict(filter_type="neighborhood", C_r=5),
                                dict(mask_noise=True), dict(ksize_r=20, C_r=5),
                                dict(filter_type="neighborhood", ksize_r=3, ksize_b=7, C_r=0, C_b=1),
                                dict(filter_type="neighborhood", mask_noise=True, noise_thresh=120),
                                dict(ksize_r=100, C_r=2, ksize_b=128, C_b=1), dict(ksize_r=1, C_r=0, ksize_b=2, C_b=0)],
                         ids=["bilateral_default", "neighborhood_try2", "mask_noise", "demo2", "neighborhood_small",
                              "neighborhood_noise", "huge_k_fallback", "tiny_k"])
def test_mask_chain_bit_exact(ctx, nat, oracle, ref_calib, frames, kw):
    n = frames.shape[0]
    ctx.upload_frames(frames)
    ctx.mask_run(n, nat.filter_params(**kw))
    masks = ctx.download_masks(n)
    # ... and the same call through the kernels a batch of >= 80 frames takes (lt_set_walk_min_frames): the walking thresholds
    # (windows 15 / 20 / 35, the greenery mask included), the running box sums of 'neighborhood'
    ctx.set_walk_min_frames(0)
    try:
        ctx.mask_run(n, nat.filter_params(**kw))
        if os.environ.get("LT_BILATERAL_TILES") != "1" and kw.get("filter_type", "bilateral") == "bilateral" and \
                kw.get("ksize_r", 15) in (15, 20, 35) and kw.get("ksize_b", 35) in (15, 20, 35):
            assert ctx.last_threshold_path() == 1
        walked = ctx.download_masks(n)
    finally:
        ctx.set_walk_min_frames(-1)
    assert np.array_equal(walked, masks), "the batch-size kernels and the few-frame kernels disagree"
    thr, thb = ctx.download_plane(2, n), ctx.download_plane(3, n)
    merged = ctx.download_plane(4, n)
    for k in range(n):
        bev = oracle.front_end(ref_calib, frames[k])
        want, planes = oracle.filter_lane_points(bev, oracle.filter_params(**kw), want_planes=True)
        assert_same(oracle.morph_open(merged[k], 5), want, f"open(merged plane), frame {k}")
        if kw.get("filter_type", "bilateral") == "bilateral":
            assert_same(thr[k], planes[2], f"tophat R, frame {k}")
            assert_same(thb[k], planes[3], f"tophat b, frame {k}")
        assert_same(masks[k], want, f"mask, frame {k}, {kw}")


@pytest.mark.parametrize("kw", [dict(), dict(ksize_r=20, C_r=5), dict(mask_noise=True), dict(filter_type="neighborhood", C_r=5),
                                dict(ksize_r=100, C_r=2, ksize_b=128, C_b=1)],
                         ids=["default", "demo2", "mask_noise", "neighborhood", "huge_k"])
def test_one_and_two_frame_chains_bit_exact(ctx, nat, oracle, ref_calib, frames, kw):
    """The chain as process() issues it -- ONE frame per launch (and two): the one-frame top-hat kernels, the R plane's chain on
    the side stream, the thresholds' H and V phases in workgroups of their own, the OR + open in one launch -- against the oracle,
    in both slots process() alternates between."""
    fp = nat.filter_params(**kw)
    for n, start, first in ((1, 2, 0), (1, 3, 1), (2, 2, 0), (1, 0, 1), (2, 4, 2)):
        ctx.upload_frames(frames[start:start + n], first=first)
        ctx.mask_run(n, fp, first=first)
        masks = ctx.download_masks(n, first=first)
        merged = ctx.download_plane(4, n, first=first)
        for k in range(n):
            bev = oracle.front_end(ref_calib, frames[start + k])
            want = oracle.filter_lane_points(bev, oracle.filter_params(**kw))
            assert_same(masks[k], want, f"mask of frame {start + k} in slot {first + k}, {n} per launch, {kw}")
            assert_same(oracle.morph_open(merged[k], 5), want, f"open(merged plane) of frame {start + k}, {kw}")


def test_enqueued_row_upload_is_the_waited_for_one(ctx, nat, oracle, ref_calib, frames):
    """lt_upload_frame_rows_enqueue (what process() uses: the copy is not waited for, the mask chain is launched behind it on the
    slots' own streams) against lt_upload_frame_rows: same masks, for one frame and for several slots' streams at once; the array
    handed over may be dropped once a record / a sync has been waited for."""
    for first, n in ((0, 1), (1, 1), (0, 4), (3, 2)):
        ctx.upload_frame_rows(frames[1:1 + n], first=first)          # what the slots held before: other frames
        ctx.mask_run(n, first=first)
        keep = ctx.upload_frame_rows(frames[2:2 + n].copy(), first=first, enqueue=True)
        ctx.mask_run(n, first=first)
        ctx.sync()
        keep[...] = 0
        del keep
        masks = ctx.download_masks(n, first=first)
        for k in range(n):
            want = oracle.filter_lane_points(oracle.front_end(ref_calib, frames[2 + k]), oracle.filter_params())
            assert_same(masks[k], want, f"frame {2 + k} in slot {first + k}, enqueued upload of {n}")


def test_bad_filter_type_raises_value_error(ctx, nat, frames):
    ctx.upload_frames(frames[:1])
    with pytest.raises(ValueError):
        ctx.mask_run(1, nat.filter_params(filter_type="median"))


def test_slots_are_independent(ctx, frames):
    """A frame's mask must not depend on the slot it sits in or on its batch neighbours."""
    ctx.upload_frames(frames[[2, 0, 3, 2]])
    ctx.mask_run(4)
    m = ctx.download_masks(4)
    ctx.upload_frames(frames[2:3])
    ctx.mask_run(1)
    single = ctx.download_masks(1)[0]
    assert_same(m[0], single, "slot 0 vs single")
    assert_same(m[3], single, "slot 3 vs single")


def _ctx_for(nat, cache, shape):
    if shape not in cache:
        h, w = shape
        cache[shape] = nat.Context((2, 2), (w, h), np.eye(3), np.zeros(5), np.eye(3), device=0, capacity=2)
    return cache[shape]


@pytest.fixture(scope="module")
def plane_ctxs():
    cache = {}
    yield cache
    for c in cache.values():
        c.close()


@pytest.mark.parametrize("path", golden_files("sws_"), ids=os.path.basename)
def test_sliding_window_search_vs_reference_fixture(nat, plane_ctxs, oracle, path):
    d = np.load(path)
    mask, p = unpack_mask(d), params_of(d)
    c = _ctx_for(nat, plane_ctxs, mask.shape)
    c.upload_masks(mask)
    c.sws_fit_run(1, nat.search_params(**p))
    rec = c.download_records(1)[0]
    assert bool(rec["detected"]) == bool(d["detected"])
    o = oracle.sliding_window_search(mask, oracle.search_params(**p))
    ly, lx = c.download_pixels(0, 0)
    ry, rx = c.download_pixels(0, 1)
    # always comparable with the oracle, even when the reference did not store anything
    assert_same(ly, o["left_y"], "left_y vs oracle"); assert_same(lx, o["left_x"], "left_x vs oracle")
    assert_same(ry, o["right_y"], "right_y vs oracle"); assert_same(rx, o["right_x"], "right_x vs oracle")
    assert c.download_centroids(0, 0) == o["left_centroids"]
    assert c.download_centroids(0, 1) == o["right_centroids"]
    if not bool(d["detected"]):
        return
    assert_same(ly, d["left_y"], "left_y"); assert_same(lx, d["left_x"], "left_x")
    assert_same(ry, d["right_y"], "right_y"); assert_same(rx, d["right_x"], "right_x")
    assert c.download_centroids(0, 0) == d["left_centroids"].tolist()
    assert c.download_centroids(0, 1) == d["right_centroids"].tolist()
    assert (int(rec["n_left"]), int(rec["n_right"])) == (len(lx), len(rx))
    h = mask.shape[0]
    assert int(rec["fit_flags"]) == 0
    assert coeff_close(rec["left_coeffs"], d["left_coeffs"], h), (rec["left_coeffs"], d["left_coeffs"])
    assert coeff_close(rec["right_coeffs"], d["right_coeffs"], h), (rec["right_coeffs"], d["right_coeffs"])
    # far tighter than the contract in practice: document the actual agreement
    assert np.allclose(rec["left_coeffs"], d["left_coeffs"], rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("path", golden_files("band"), ids=os.path.basename)
def test_band_search_vs_reference_fixture(nat, plane_ctxs, path):
    d = np.load(path)
    mask, p = unpack_mask(d), params_of(d)
    c = _ctx_for(nat, plane_ctxs, mask.shape)
    c.upload_masks(mask)
    prev = np.concatenate([d["prev_left"], d["prev_right"]])
    c.band_fit_run(1, prev, nat.search_params(**p))
    rec = c.download_records(1)[0]
    assert bool(rec["detected"]) == bool(d["detected"])
    assert int(rec["mode"]) == 1
    if not bool(d["detected"]):
        return
    ly, lx = c.download_pixels(0, 0)
    ry, rx = c.download_pixels(0, 1)
    assert_same(ly, d["left_y"], "left_y"); assert_same(lx, d["left_x"], "left_x")
    assert_same(ry, d["right_y"], "right_y"); assert_same(rx, d["right_x"], "right_x")
    assert coeff_close(rec["left_coeffs"], d["left_coeffs"]) and coeff_close(rec["right_coeffs"], d["right_coeffs"])
    assert np.allclose(rec["right_coeffs"], d["right_coeffs"], rtol=1e-8, atol=1e-9)


def test_search_fuzz_vs_oracle(nat, plane_ctxs, oracle):
    """Random masks/parameters (abort branches, borders, odd windows): GPU == oracle, bit for bit."""
    from lane_tracker_amd import synth
    rng = np.random.default_rng(99)
    c = _ctx_for(nat, plane_ctxs, (1100, 1080))
    for it in range(40):
        seed = int(rng.integers(0, 1 << 30))
        kind = it % 4
        if kind == 0:
            m = synth.random_mask(seed, density=float(10 ** rng.uniform(-4, -0.3)))
        elif kind == 1:
            m = synth.synth_mask(seed, noise=float(10 ** rng.uniform(-4, -1.3)), curv=3e-4, slope=0.25)[0]
        elif kind == 2:
            m = synth.synth_mask(seed, noise=0.0, left_base=(0, 40), sep=(200, 1000))[0]
        else:
            m = synth.synth_mask(seed, noise=1e-3, drop_left=bool(rng.integers(0, 2)))[0]
            y0, y1 = sorted(rng.integers(0, 1100, 2))
            m[y0:y1, : int(rng.integers(0, 1080))] = 0
        p = dict(window_width=int(rng.choice([30, 20, 31, 60, 100])), window_height=int(rng.choice([40, 25, 118])),
                 search_range=int(rng.choice([20, 60, 5])), mu=float(rng.choice([0.1, 0.5, 1.0, 0.0])),
                 no_success_limit=int(rng.choice([8, 3, 50, 1])), start_slice=float(rng.choice([0.25, 0.1, 1.0])),
                 ignore_sides=int(rng.choice([360, 0, 100])), ignore_bottom=int(rng.choice([30, 0, 7])),
                 partial=float(rng.choice([1.0, 0.5, 0.3])))
        c.upload_masks(m)
        c.sws_fit_run(1, nat.search_params(**p))
        o = oracle.sliding_window_search(m, oracle.search_params(**p))
        rec = c.download_records(1)[0]
        assert bool(rec["detected"]) == o["detected"], p
        for side, (ky, kx) in enumerate((("left_y", "left_x"), ("right_y", "right_x"))):
            y, x = c.download_pixels(0, side)
            assert_same(y, o[ky], f"{ky} {p}"); assert_same(x, o[kx], f"{kx} {p}")
        assert c.download_centroids(0, 0) == o["left_centroids"], p
        assert c.download_centroids(0, 1) == o["right_centroids"], p
        if o["detected"] and int(rec["fit_flags"]) == 0:
            for key, yy, xx in (("left_coeffs", o["left_y"], o["left_x"]), ("right_coeffs", o["right_y"], o["right_x"])):
                assert coeff_close(rec[key], np.polyfit(yy, xx, 2)), (key, p)
        # band search with the oracle's own fit as the previous lanes
        lc, rc = oracle.polyfit2(o["left_y"], o["left_x"]) if len(o["left_y"]) else np.zeros(3), \
                 oracle.polyfit2(o["right_y"], o["right_x"]) if len(o["right_y"]) else np.zeros(3)
        bp = dict(bandwidth=int(rng.choice([25, 30, 5, 80])), ignore_bottom=int(rng.choice([30, 0, 11])),
                  partial=float(rng.choice([1.0, 0.5])))
        c.band_fit_run(1, np.concatenate([lc, rc]), nat.search_params(**bp))
        ob = oracle.band_search(m, lc, rc, oracle.search_params(**bp))
        rec = c.download_records(1)[0]
        assert bool(rec["detected"]) == ob["detected"], bp
        if ob["detected"]:
            for side, (ky, kx) in enumerate((("left_y", "left_x"), ("right_y", "right_x"))):
                y, x = c.download_pixels(0, side)
                assert_same(y, ob[ky], f"band {ky} {bp}"); assert_same(x, ob[kx], f"band {kx} {bp}")


def test_fit_rank_deficient_flag_and_host_answer(nat, plane_ctxs):
    import warnings
    c = _ctx_for(nat, plane_ctxs, (1100, 1080))
    m = np.zeros((1100, 1080), np.uint8)
    m[1060:1062, 440:452] = 255          # two distinct rows only, both lanes
    m[1060:1062, 640:652] = 255
    c.upload_masks(m)
    c.sws_fit_run(1, nat.search_params())
    rec = c.download_records(1)[0]
    assert bool(rec["detected"]) and int(rec["fit_flags"]) == 3
    ys, xs = c.download_pixels(0, 0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        want = np.polyfit(ys, xs, 2)
    got = c.fit_poly2(ys, xs)
    assert np.allclose(got, want, rtol=1e-7, atol=1e-7)
    y = np.arange(100, 600)
    x = np.rint(1e-4 * y * y - 0.1 * y + 500).astype(np.int64)
    assert coeff_close(c.fit_poly2(y, x), np.polyfit(y, x, 2))


@pytest.mark.parametrize("shape,k,C", [((97, 131), 15, 8), ((64, 64), 35, 5), ((300, 17), 65, 10), ((1, 1), 3, 0),
                                       ((33, 500), 30, 0)])
def test_bilateral_adaptive_threshold_function(nat, oracle, shape, k, C):
    from lane_tracker_amd.lane_tracker import bilateral_adaptive_threshold
    rng = np.random.default_rng(shape[0] * 7 + k)
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    for mode in ("floor", "ceil"):
        got = bilateral_adaptive_threshold(img, ksize=k, C=C, mode=mode, true_value=200, false_value=3)
        assert_same(got, oracle.bilateral_adaptive_threshold(img, k, C, mode, 200, 3), f"{shape} {k} {mode}")
    with pytest.raises(ValueError):
        bilateral_adaptive_threshold(img, ksize=k, C=C, mode="round")


@pytest.mark.parametrize("shape", [(120, 150), (61, 333)])
def test_filter_lane_points_any_size(ctx, nat, oracle, shape):
    rng = np.random.default_rng(shape[1])
    bev = rng.integers(0, 256, shape + (3,), dtype=np.uint8)
    for kw in (dict(), dict(filter_type="neighborhood", C_r=5), dict(mask_noise=True, noise_thresh=135, ksize_r=25)):
        got = ctx.filter_lane_points(bev, nat.filter_params(**kw))
        assert_same(got, oracle.filter_lane_points(bev, oracle.filter_params(**kw)), f"{shape} {kw}")


BATCH_SETS = {
    "process_defaults": (dict(), dict()),
    # the author's Demo 1 / Demo 3 filter (tracker_settings.md:10-13, 86-89: the greenery mask) with Demo 3's half look-ahead
    "demo1_demo3": (dict(mask_noise=True, noise_thresh=140, ksize_noise=65, C_noise=10), dict(no_success_limit=50, bandwidth=30, partial=0.5)),
    # process()'s hard-coded second try (lane_tracker.py:1081-1099)
    "second_try": (dict(filter_type="neighborhood", ksize_r=15, C_r=5, ksize_b=35, C_b=5), dict(no_success_limit=50, bandwidth=30)),
}


@pytest.mark.parametrize("which", list(BATCH_SETS))
def test_full_batch_256_unique_frames_bit_exact(nat, cal, oracle, ref_calib, which):
    """BASELINE configs 2 / 3: a batch of 256 DIFFERENT synthetic frames resident in HBM, every mask compared bit for bit
    with the oracle and every record with its search + fit (the oracle runs one frame per host thread) -- for process()'s
    defaults, the greenery-mask filter of the author's demos and the second try, each through its batch-size kernels."""
    from concurrent.futures import ThreadPoolExecutor
    from lane_tracker_amd import synth
    fkw, skw = BATCH_SETS[which]
    fp, sp = nat.filter_params(**fkw), nat.search_params(**skw)
    ofp, osp = oracle.filter_params(**fkw), oracle.search_params(**skw)
    r = synth.SceneRenderer()
    n = 256
    batch = np.stack([r.render(500 + i)[0] if i % 16 else synth.frame_uniform(500 + i) for i in range(n)], 0)
    c = nat.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"],
                    cal["warp_matrices"][0], device=0, capacity=n)
    try:
        c.upload_frames(batch)
        c.set_frame_base(n, 1000)
        c.mask_run(n, fp)
        if fkw.get("filter_type") == "neighborhood":
            assert c.last_adaptive_path() == 1           # the running box sums, not the per-pixel windows
        else:
            assert c.last_threshold_path() == 1          # the walking kernels, the greenery mask included
        c.sws_fit_run(n, sp)
        rec = c.download_records(n)
        masks = c.download_masks(n)
        assert rec["frame"].tolist() == list(range(1000, 1000 + n))
        threads = min(64, len(os.sched_getaffinity(0))) if hasattr(os, "sched_getaffinity") else 8
        oracle.frame_sws_fit(ref_calib, batch[0])        # builds the oracle's per-calibration tables before the threads start
        with ThreadPoolExecutor(threads) as ex:          # ctypes releases the GIL
            want = list(ex.map(lambda i: oracle.frame_sws_fit(ref_calib, batch[i], ofp, osp, want_mask=True), range(n)))
        bad_masks = [i for i in range(n) if not np.array_equal(masks[i], want[i]["mask"])]
        assert not bad_masks, f"{len(bad_masks)} of {n} masks differ from the oracle, first: frame {bad_masks[0]}"
        for i, o in enumerate(want):
            assert (int(rec[i]["n_left"]), int(rec[i]["n_right"]), bool(rec[i]["detected"])) == (o["n_left"], o["n_right"], o["detected"]), i
            if o["detected"]:
                assert coeff_close(rec[i]["left_coeffs"], o["coeffs"][0]) and coeff_close(rec[i]["right_coeffs"], o["coeffs"][1]), i
        assert int(rec["detected"].sum()) >= n - n // 16 - 2     # the lane-like frames are found, the noise frames need not be
        for i in (3, 77, 255):
            ys, xs = c.download_pixels(i, 0)
            assert len(ys) == int(rec[i]["n_left"])
    finally:
        c.close()


@pytest.mark.parametrize("k", [5, 29, 55])
@pytest.mark.parametrize("shape", [(200, 300), (57, 129), (1, 1), (130, 64), (64, 1081)])
def test_morph_ellipse_operators(ctx, oracle, k, shape):
    """Single-pass erode/dilate/top-hat/open, run-decomposed kernel vs the oracle (and vs direct taps)."""
    rng = np.random.default_rng(k * 31 + shape[1])
    img = rng.integers(0, 256, shape, dtype=np.uint8)
    img[rng.random(shape) < 0.02] = 0
    img[rng.random(shape) < 0.02] = 255
    assert_same(ctx.morph_ellipse(img, k, "erode"), oracle.erode(img, k), f"erode {k} {shape}")
    assert_same(ctx.morph_ellipse(img, k, "dilate"), oracle.dilate(img, k), f"dilate {k} {shape}")
    assert_same(ctx.morph_ellipse(img, k, "tophat"), oracle.tophat(img, k), f"tophat {k} {shape}")
    assert_same(ctx.morph_ellipse(img, k, "open"), oracle.morph_open(img, k), f"open {k} {shape}")
    assert_same(ctx.morph_ellipse(img, k, "erode", direct=True), oracle.erode(img, k), f"direct erode {k} {shape}")


@pytest.mark.parametrize("k", [29, 55])
def test_one_frame_morphology_at_full_size_against_direct_taps(ctx, k):
    """The one-frame kernels (k_morph_one: the walk of a band split over the waves of a workgroup) on a whole bird's-eye plane,
    and on sizes whose last band / last strip are ragged, against the direct evaluation of the footprint on the same device
    (itself held against the oracle by test_morph_ellipse_operators)."""
    for shape in ((1100, 1080), (1650, 1620), (333, 516), (61, 132)):
        rng = np.random.default_rng(k + shape[0])
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        img[rng.random(shape) < 0.01] = 0
        img[rng.random(shape) < 0.01] = 255
        for op in ("erode", "dilate", "tophat"):
            assert_same(ctx.morph_ellipse(img, k, op), ctx.morph_ellipse(img, k, op, direct=True), f"{op} {k} {shape}")


def test_morph_ellipse_footprint_probe(ctx, oracle):
    """Delta images expose every tap of the footprint, including across the 64-column lane seams."""
    for k in (29, 55):
        for x in (0, 63, 64, 127, 128, 200, 299):
            img = np.full((140, 300), 255, np.uint8)
            img[70, x] = 0
            assert_same(ctx.morph_ellipse(img, k, "erode"), oracle.erode(img, k), f"erode probe {k} x={x}")
            assert_same(ctx.morph_ellipse(255 - img, k, "dilate"), oracle.dilate(255 - img, k), f"dilate probe {k} x={x}")


def test_multi_stream_context_gives_identical_results(nat, cal, frames):
    """lt_set_streams: slot slices on different HIP streams must not change a single bit, whatever
    the slice boundaries do to the batch (also partial ranges and a band search on top)."""
    from lane_tracker_amd import synth
    r = synth.SceneRenderer()
    batch = np.concatenate([frames, np.stack([r.render(700 + i)[0] for i in range(5)], 0)], 0)   # 11 frames
    n = batch.shape[0]
    ref = None
    for k in (1, 2, 3, 8):
        c = nat.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"],
                        cal["warp_matrices"][0], device=0, capacity=n)
        try:
            c.set_streams(k)
            c.upload_frames(batch)
            c.mask_run(n)
            c.sws_fit_run(n)
            c.mask_run(4, nat.filter_params(filter_type="neighborhood", C_r=5), first=3)     # partial range, other params
            c.sws_fit_run(4, nat.search_params(no_success_limit=50), first=3)
            masks, rec = c.download_masks(n), c.download_records(n)
            prev = np.tile(np.concatenate([rec[8]["left_coeffs"], rec[8]["right_coeffs"]]), (n, 1))
            c.band_fit_run(n, prev)
            rec_b = c.download_records(n)
            pix = [c.download_pixels(i, s) for i in (0, 5, 10) for s in (0, 1)]
            got = (masks, rec, rec_b, pix)
        finally:
            c.close()
        if ref is None:
            ref = got
            continue
        assert np.array_equal(got[0], ref[0]), k
        assert got[1].tobytes() == ref[1].tobytes() and got[2].tobytes() == ref[2].tobytes(), k
        for (y, x), (ry, rx) in zip(got[3], ref[3]):
            assert np.array_equal(y, ry) and np.array_equal(x, rx), k
    with pytest.raises(ValueError):
        c2 = nat.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0])
        try:
            c2.set_streams(0)
        finally:
            c2.close()


def test_odd_sizes_take_the_generic_paths(nat, oracle):
    """Camera 641x361 and bird's-eye 541x551: nothing is a multiple of 4, so the dword fast paths
    (4 px/thread warp, aligned band staging, vectorised band sums / window extraction, dword mask
    expansion) all fall back to their generic variants.  Same bar: bit-exact against the oracle."""
    from lane_tracker_amd import calib
    S = np.diag([0.5, 0.5, 1.0])
    K = S @ calib.CAM_MATRIX
    M = S @ calib.M @ np.diag([2.0, 2.0, 1.0])
    img_size, warped = (641, 361), (541, 551)
    oc = oracle.make_calib(img_size, warped, K, calib.DIST_COEFFS, M)
    c = nat.Context(img_size, warped, K, calib.DIST_COEFFS, M, device=0, capacity=3)
    try:
        rng = np.random.default_rng(12)
        frames = rng.integers(0, 256, (3, 361, 641, 3), dtype=np.uint8)
        frames[1] = np.clip(90 + rng.integers(-10, 11, (361, 641, 1)), 0, 255).astype(np.uint8)
        frames[1, 200:, 250:262] = (235, 235, 235)                      # two bright stripes on grey
        frames[1, 200:, 380:392] = (220, 190, 60)
        c.upload_frames(frames)
        i = c.info()
        assert (i.src_row0, i.src_row1) == oracle.warp_source_rows(oc)
        for kw in (dict(), dict(filter_type="neighborhood", C_r=5), dict(mask_noise=True)):
            c.mask_run(3, nat.filter_params(**kw))
            masks = c.download_masks(3)
            R, B = c.download_plane(0, 3), c.download_plane(1, 3)
            for k in range(3):
                bev = oracle.front_end(oc, frames[k])
                assert_same(R[k], bev[:, :, 0], f"odd R {k}")
                assert_same(B[k], oracle.lab_b(bev), f"odd Lab-b {k}")
                assert_same(masks[k], oracle.filter_lane_points(bev, oracle.filter_params(**kw)), f"odd mask {k} {kw}")
        sp = dict(window_width=14, window_height=20, search_range=10, ignore_sides=180, ignore_bottom=15)
        c.mask_run(3)
        c.sws_fit_run(3, nat.search_params(**sp))
        masks = c.download_masks(3)
        rec = c.download_records(3)
        for k in range(3):
            o = oracle.sliding_window_search(masks[k], oracle.search_params(**sp))
            assert bool(rec[k]["detected"]) == o["detected"]
            for side, (ky, kx) in enumerate((("left_y", "left_x"), ("right_y", "right_x"))):
                y, x = c.download_pixels(k, side)
                assert_same(y, o[ky], f"odd {ky} {k}"); assert_same(x, o[kx], f"odd {kx} {k}")
            assert c.download_centroids(k, 0) == o["left_centroids"] and c.download_centroids(k, 1) == o["right_centroids"]
        prev = np.tile(np.array([0.0, 0.0, 200.0, 0.0, 0.0, 330.0]), (3, 1))
        bp = dict(bandwidth=12, ignore_bottom=15)
        c.band_fit_run(3, prev, nat.search_params(**bp))
        rec = c.download_records(3)
        for k in range(3):
            o = oracle.band_search(masks[k], prev[k, :3], prev[k, 3:], oracle.search_params(**bp))
            assert bool(rec[k]["detected"]) == o["detected"]
            if o["detected"]:
                y, x = c.download_pixels(k, 1)
                assert_same(y, o["right_y"], f"odd band right_y {k}"); assert_same(x, o["right_x"], f"odd band right_x {k}")
    finally:
        c.close()


def test_full_size_morphology_properties(ctx):
    """Size-independent properties at the bird's-eye size: opening is idempotent and anti-extensive,
    top-hat + opening reassemble the image, erosion <= image <= dilation."""
    rng = np.random.default_rng(77)
    img = rng.integers(0, 256, (1100, 1080), dtype=np.uint8)
    for k in (29, 55):
        op = ctx.morph_ellipse(img, k, "open")
        assert np.array_equal(ctx.morph_ellipse(op, k, "open"), op)
        assert (op <= img).all()
        th = ctx.morph_ellipse(img, k, "tophat")
        assert np.array_equal(th.astype(np.int32) + op, img)
        er, di = ctx.morph_ellipse(img, k, "erode"), ctx.morph_ellipse(img, k, "dilate")
        assert (er <= img).all() and (img <= di).all()
        assert np.array_equal(255 - ctx.morph_ellipse(255 - img, k, "dilate"), er)     # duality


def test_source_row_upload_is_enough_for_the_path(nat, cal, frames):
    """lt_upload_frame_rows moves only the camera rows undistort + warp read; masks and fits are unchanged."""
    a = nat.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0],
                    device=0, capacity=len(frames))
    b = nat.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0],
                    device=0, capacity=len(frames))
    try:
        r0, r1 = a.source_rows()
        assert 0 <= r0 < r1 <= 720 and r1 - r0 < 400       # about a third of the frame (rows 450..689 here)
        a.upload_frames(frames)
        b.upload_frames(255 - frames)                     # stale content everywhere ...
        b.upload_frame_rows(frames)                       # ... except the rows that matter
        for c in (a, b):
            c.mask_run(len(frames))
            c.sws_fit_run(len(frames))
        assert_same(b.download_masks(len(frames)), a.download_masks(len(frames)), "masks after a source-row upload")
        ra, rb = a.download_records(len(frames)), b.download_records(len(frames))
        assert ra.tobytes() == rb.tobytes()
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("ww,wh", [(64, 40), (65, 40), (66, 40), (32, 64), (33, 65), (20, 130), (4, 7)])
def test_search_kernel_limits_both_versions(nat, plane_ctxs, oracle, ww, wh):
    """Window sizes around the limits of k_sws_fit2 (2 * int(ww / 2) <= 64 columns; any height): whichever kernel the
    launcher picks, the result is the oracle's."""
    from lane_tracker_amd import synth
    c = _ctx_for(nat, plane_ctxs, (1100, 1080))
    for seed in (5, 6):
        m = synth.synth_mask(seed, noise=2e-3, curv=2e-4, slope=0.1)[0]
        p = dict(window_width=ww, window_height=wh, search_range=25, ignore_sides=200)
        c.upload_masks(m)
        c.sws_fit_run(1, nat.search_params(**p))
        o = oracle.sliding_window_search(m, oracle.search_params(**p))
        rec = c.download_records(1)[0]
        assert bool(rec["detected"]) == o["detected"], p
        for side, (ky, kx) in enumerate((("left_y", "left_x"), ("right_y", "right_x"))):
            y, x = c.download_pixels(0, side)
            assert_same(y, o[ky], f"{ky} {p}"); assert_same(x, o[kx], f"{kx} {p}")
        assert c.download_centroids(0, 0) == o["left_centroids"] and c.download_centroids(0, 1) == o["right_centroids"], p
        if o["detected"] and len(set(o["left_y"].tolist())) >= 3 and len(set(o["right_y"].tolist())) >= 3:
            assert coeff_close(rec["left_coeffs"], oracle.polyfit2(o["left_y"], o["left_x"]))
            assert coeff_close(rec["right_coeffs"], oracle.polyfit2(o["right_y"], o["right_x"]))


@pytest.mark.parametrize("bandwidth", [0, 1, 30, 31, 32, 200])
def test_band_kernel_limits_both_versions(nat, plane_ctxs, oracle, bandwidth):
    """Band widths around the 64-column limit of k_band_fit2 (2 * bandwidth + 2 <= 64)."""
    from lane_tracker_amd import synth
    c = _ctx_for(nat, plane_ctxs, (1100, 1080))
    m, lc, rc = synth.synth_mask(77, noise=5e-3)
    lc, rc = lc + np.array([0, 0, 0.4]), rc + np.array([0, 0, -0.6])
    for partial, ib in ((1.0, 30), (0.5, 0)):
        p = dict(bandwidth=bandwidth, ignore_bottom=ib, partial=partial)
        c.upload_masks(m)
        c.band_fit_run(1, np.concatenate([lc, rc]), nat.search_params(**p))
        o = oracle.band_search(m, lc, rc, oracle.search_params(**p))
        rec = c.download_records(1)[0]
        assert bool(rec["detected"]) == o["detected"], p
        for side, (ky, kx) in enumerate((("left_y", "left_x"), ("right_y", "right_x"))):
            y, x = c.download_pixels(0, side)
            assert_same(y, o[ky], f"{ky} {p}"); assert_same(x, o[kx], f"{kx} {p}")
        if o["detected"] and bandwidth > 0:
            assert coeff_close(rec["left_coeffs"], oracle.polyfit2(o["left_y"], o["left_x"]))


def test_randomised_differential_run():
    """A short run of tests/fuzz_gpu.py (random filter parameters, frames, image sizes, search parameters; the
    700-iteration campaign of the round is recorded in DESIGN.md)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz_gpu.py"), "8", "123"], cwd=root, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "0 mismatches" in r.stdout


EXP_LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lane_tracker_amd", "liblane_tracker_amd_exp.so")


@pytest.mark.gpu
@pytest.mark.parametrize("switch", ["LT_SEARCH_U8=1,LT_BILATERAL_TILES=1,LT_UNDISTORT_UNALIGNED=1,LT_OPEN5_SEPARATE=1,LT_OPEN_SMALL=0,LT_MORPH_ONE_ROW=1,LT_MORPH_WIDE=0",
                                    "LT_SWS_V1=1,LT_BAND_V1=1"])      # (one fallback per stage in the first run: they do not interact)
def test_fallback_kernel_paths_keep_parity_at_the_reference_geometry(switch):
    """The release library has one path per stage plus FALLBACKS it takes by itself for what the main kernels do not cover: u8
    masks handed in by the caller, search windows beyond the bit-plane kernels' limits, image widths that are not a multiple of
    four / strips that are not aligned, a few frames per call, rows wider than 4096 px.  The tests of those geometries reach them
    in the release build (kernel limits, odd sizes, small calls); here the EXPERIMENTS build (`make EXPERIMENTS=1`,
    liblane_tracker_amd_exp.so: the same sources with the measurement switches alive) forces each of them at the reference
    geometry, through the mask-chain, operator and search parity tests of this file, in a process of its own."""
    import subprocess
    import sys
    if not os.path.exists(EXP_LIB):
        subprocess.check_call(["make", "-C", os.path.join(os.path.dirname(EXP_LIB), "csrc"), "-s", "-j8", "EXPERIMENTS=1"])
    env = dict(os.environ, LANE_TRACKER_AMD_LIB=EXP_LIB)
    for item in switch.split(","):
        name, value = item.split("=")
        env[name] = value
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-x", "-p", "no:cacheprovider",
                        "-k", "mask_chain_bit_exact or one_and_two_frame or morph_ellipse_operators or sliding_window_search_vs_reference or "
                              "band_search_vs_reference or front_end_bit_exact or odd_slot_ranges"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_odd_slot_ranges_and_the_batch_paths_of_the_open_stage(nat, cal, oracle, ref_calib):
    """Slot pairs share the interleaved undistorted rows and the front-end kernels walk pairs: ranges that start or end on
    an odd slot, and a batch large enough (>= 16 frames) for the fused merge + open pass on an already merged plane
    (tile threshold kernel: a window size the walking kernels do not have) must give the oracle's planes slot by slot."""
    from lane_tracker_amd import synth
    r = synth.SceneRenderer(cal)
    n = 21
    batch = np.stack([r.render(900 + i)[0] if i % 4 else synth.frame_uniform(50 + i) for i in range(n)], 0)
    c = nat.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0],
                    device=0, capacity=n)
    try:
        c.set_streams(3)                                   # slices 0-6, 6-14, 14-21: even boundaries inside an odd capacity
        c.upload_frames(batch)
        kw = dict(ksize_r=25, C_r=6, ksize_b=31, C_b=4)    # not 15 / 20 / 35: tile kernel, then the fused open of 21 frames
        c.mask_run(n, nat.filter_params(**kw))
        assert c.last_threshold_path() == 0
        masks, merged = c.download_masks(n), c.download_plane(4, n)
        und = c.download_undistorted(n)
        r0, r1 = oracle.warp_source_rows(ref_calib)
        for k in (0, 1, 6, 13, 14, 20):
            bev = oracle.front_end(ref_calib, batch[k])
            want, planes = oracle.filter_lane_points(bev, oracle.filter_params(**kw), want_planes=True)
            assert_same(und[k], oracle.undistort(ref_calib, batch[k])[r0:r1], f"undistorted rows, slot {k}")
            assert_same(oracle.morph_open(merged[k], 5), want, f"open(merged), slot {k}")
            assert_same(masks[k], want, f"mask, slot {k}")
        # an odd range on top: slots 3..7 with the default parameters, everything else untouched
        c.mask_run(5, first=3)
        after = c.download_masks(n)
        for k in range(n):
            if 3 <= k < 8:
                assert_same(after[k], oracle.filter_lane_points(oracle.front_end(ref_calib, batch[k])), f"odd range, slot {k}")
            else:
                assert np.array_equal(after[k], masks[k]), k
        # single odd slot through the presentation path that reads the interleaved rows
        bev7 = c.download_bev(1, first=7)[0] if hasattr(c, "download_bev") else None
        if bev7 is not None:
            assert_same(bev7, oracle.front_end(ref_calib, batch[7]), "bird's-eye RGB of slot 7")
    finally:
        c.close()


def test_config5_1080p_batch_masks_bit_exact(nat, oracle):
    """BASELINE config 5 geometry through the BATCH path: a 1920x1080 camera (6.2 MB frames, 2.25x the undistorted rows)
    with an odd number of frames -- the pair-interleaved undistorted rows, the aligned-dword undistortion and the pair strips
    of the top-hats at a second calibration."""
    from lane_tracker_amd import calib, synth
    cal = calib.scaled_calibration(1.5)
    frames = synth.stream_lanes(7, seed=9, cal=cal)
    assert frames.shape[1:] == (1080, 1920, 3)
    oc = oracle.make_calib(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0])
    c = nat.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0],
                    device=0, capacity=7)
    try:
        c.set_streams(2)
        c.upload_frames(frames)
        c.mask_run(7)
        c.sws_fit_run(7)
        masks, rec = c.download_masks(7), c.download_records(7)
        r0, r1 = oracle.warp_source_rows(oc)
        und = c.download_undistorted(7)
        for k in range(7):
            assert_same(und[k], oracle.undistort(oc, frames[k])[r0:r1], f"undistorted rows, frame {k}")
            want = oracle.mask_from_frame(oc, frames[k])
            assert_same(masks[k], want, f"mask, frame {k}")
            o = oracle.sliding_window_search(want)
            assert bool(rec[k]["detected"]) == bool(o["detected"]) and int(rec[k]["n_left"]) == len(o["left_y"]), k
    finally:
        c.close()


def test_mask_rerun_skips_the_front_end_only_while_the_planes_are_the_frames(nat, cal, oracle, ref_calib, frames):
    """lt_mask_rerun (the second try of a frame, lane_tracker.py:1081-1101): another filter over the bird's-eye planes the first
    lt_mask_run left -- the oracle's mask for the second parameter set; after an upload of new camera rows into a slot the planes
    are stale and the re-run computes them again (the oracle's mask of the NEW frame, not of the old planes)."""
    n = min(3, frames.shape[0])
    c = nat.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=n)
    try2 = dict(filter_type="neighborhood", C_r=5)
    try:
        c.upload_frames(frames[:n])
        c.mask_run(n)
        first = c.download_masks(n)
        c.mask_run(n, nat.filter_params(**try2), reuse_front=True)
        for k in range(n):
            assert_same(c.download_masks(1, first=k)[0], oracle.mask_from_frame(ref_calib, frames[k], oracle.filter_params(**try2)), f"second try, slot {k}")
        c.mask_run(n, reuse_front=True)                        # ... and back: the first try's masks again
        assert np.array_equal(c.download_masks(n), first)
        c.upload_frame_rows(frames[n - 1:n], first=0)          # slot 0 gets another frame's rows: its planes are stale
        c.mask_run(1, nat.filter_params(**try2), first=0, reuse_front=True)
        assert_same(c.download_masks(1, first=0)[0], oracle.mask_from_frame(ref_calib, frames[n - 1], oracle.filter_params(**try2)), "stale planes recomputed")
        c.mask_run(1, first=1, reuse_front=True)               # slot 1 was not touched: still served from its planes
        assert np.array_equal(c.download_masks(1, first=1)[0], first[1])
    finally:
        c.close()


def test_lane_lists_in_one_round_trip_equal_the_piecewise_downloads(nat, cal, frames):
    """lt_download_lane_lists against lt_download_pixels + lt_download_centroids: after a sliding-window search (column masks per
    window row, centroids) and after a band search (column masks per band row) over the same slot."""
    c = nat.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=2)
    try:
        c.upload_frames(frames[:2])
        c.mask_run(2)
        c.sws_fit_run(2)
        rec = c.download_records(2)
        for slot in (0, 1):
            ly, lx, ry, rx, cl, cr = c.download_lane_lists(slot, True)
            py0, px0 = c.download_pixels(slot, 0)
            py1, px1 = c.download_pixels(slot, 1)
            assert np.array_equal(ly, py0) and np.array_equal(lx, px0) and np.array_equal(ry, py1) and np.array_equal(rx, px1)
            assert cl == c.download_centroids(slot, 0) and cr == c.download_centroids(slot, 1)
            assert (len(ly), len(ry)) == (int(rec[slot]["n_left"]), int(rec[slot]["n_right"]))
        if rec[0]["detected"]:
            prev = np.concatenate([rec[0]["left_coeffs"], rec[0]["right_coeffs"]])
            c.band_fit_run(1, prev[None], first=0)
            ly, lx, ry, rx, cl, cr = c.download_lane_lists(0, False)
            py0, px0 = c.download_pixels(0, 0)
            py1, px1 = c.download_pixels(0, 1)
            assert np.array_equal(ly, py0) and np.array_equal(lx, px0) and np.array_equal(ry, py1) and np.array_equal(rx, px1)
            assert cl is None and cr is None and len(ly) > 0
    finally:
        c.close()
