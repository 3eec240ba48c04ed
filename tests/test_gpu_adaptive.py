"""The running-box-sum kernel of the 'neighborhood' filter (csrc/k_adaptive_walk.hip; cv2.adaptiveThreshold,
lane_tracker.py:217-218) against the oracle and against a NumPy restatement from prefix sums: bird's-eye images of many
sizes (one strip, several strips, a last strip of one word, heights below one window), every odd window up to 63,
negative / zero / large C.  `last_adaptive_path` tells that the walk -- not the per-pixel window kernel -- ran."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [(1080, 1100), (1084, 300), (256, 128), (132, 70), (8, 5), (64, 64), (4, 3), (192, 40), (196, 33), (388, 20), (1280, 90),
         (72, 300)]
PARAMS = [(15, 5, 35, 5), (3, 0, 63, 1), (1, 0, 5, -3), (25, 12, 35, 5), (61, 40, 7, 255), (35, -300, 15, 300)]


def _img(rng, h, w, kind):
    if kind == 0:
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if kind == 1:      # smooth blocks + little noise: differences to the mean around C
        base = rng.integers(40, 200, ((h + 15) // 16, (w + 15) // 16, 3)).repeat(16, 0).repeat(16, 1)[:h, :w]
        return np.clip(base + rng.integers(-7, 8, (h, w, 3)), 0, 255).astype(np.uint8)
    img = np.full((h, w, 3), 255, np.int64)                 # saturated plane with dark structures: the largest sums
    for _ in range(5):
        x = int(rng.integers(0, w))
        img[:, max(x - 2, 0):x + 3] = rng.integers(0, 60)
    return img.astype(np.uint8)


def adaptive_np(p, bs, C):
    """255 iff p - round(boxmean) > C, replicated border."""
    r = bs // 2
    q = np.pad(p.astype(np.int64), r, mode="edge")
    P = np.zeros((q.shape[0] + 1, q.shape[1] + 1), np.int64)
    P[1:, 1:] = q.cumsum(0).cumsum(1)
    h, w = p.shape
    S = P[bs:bs + h, bs:bs + w] - P[:h, bs:bs + w] - P[bs:bs + h, :w] + P[:h, :w]
    mean = (2 * S + bs * bs) // (2 * bs * bs)
    return p.astype(np.int64) - mean > C


@pytest.mark.parametrize("size", SIZES)
def test_box_walk_matches_the_oracle_on_many_sizes(size):
    from lane_tracker_amd import _native, calib
    from oracle import oracle as O
    w, h = size
    cal = calib.reference_calibration()
    ctx = _native.Context(cal["img_size"], (w, h), cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=3)
    rng = np.random.default_rng(w * 131 + h)
    try:
        for pi, (kr, cr, kb, cb) in enumerate(PARAMS):
            if (w * h > 500000) and pi > 2:
                continue
            bev = np.stack([_img(rng, h, w, k) for k in range(3)], 0)
            kw = dict(filter_type="neighborhood", ksize_r=kr, C_r=cr, ksize_b=kb, C_b=cb)
            ctx.upload_bev(bev)
            ctx.filter_run(3, _native.filter_params(**kw))
            assert ctx.last_adaptive_path() == 1, "the box walk did not take these parameters"
            got = ctx.download_masks(3)
            merged = ctx.download_plane(_native.PLANE_MERGED, 3)
            for i in range(3):
                want, planes = O.filter_lane_points(bev[i], O.filter_params(**kw), want_planes=True)
                expect = adaptive_np(planes[0], kr, cr) | adaptive_np(planes[1], kb, cb)
                tag = (size, (kr, cr, kb, cb), i)
                assert np.array_equal(merged[i] > 0, expect), (tag, int(((merged[i] > 0) != expect).sum()))
                assert np.array_equal(got[i], want), tag
    finally:
        ctx.close()


def test_what_the_box_walk_does_not_take_falls_back_and_agrees():
    """Windows above 63, a width that is not a multiple of four and the greenery mask take the per-pixel window kernel."""
    from lane_tracker_amd import _native, calib
    from oracle import oracle as O
    cal = calib.reference_calibration()
    rng = np.random.default_rng(11)
    for (w, h), kw in (((260, 150), dict(ksize_r=65, C_r=3, ksize_b=35, C_b=5)), ((262, 90), dict(ksize_r=15, C_r=5, ksize_b=35, C_b=5)),
                       ((260, 150), dict(ksize_r=15, C_r=5, ksize_b=35, C_b=5, mask_noise=True, noise_thresh=120))):
        ctx = _native.Context(cal["img_size"], (w, h), cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=1)
        try:
            bev = _img(rng, h, w, 1)[None]
            kw = dict(filter_type="neighborhood", **kw)
            ctx.upload_bev(bev)
            ctx.filter_run(1, _native.filter_params(**kw))
            assert ctx.last_adaptive_path() == 0
            assert np.array_equal(ctx.download_masks(1)[0], O.filter_lane_points(bev[0], O.filter_params(**kw)))
        finally:
            ctx.close()


def test_second_try_on_a_batch_and_both_kernels_agree(monkeypatch):
    """process()'s second-try set (lane_tracker.py:1081-1099) on 40 rendered camera frames: the walk's masks equal the
    per-pixel kernel's (LT_ADAPTIVE_TILES=1 in a second context is not possible -- the switch is read once per process --
    so the comparison is against the oracle on a subset and against a one-frame call, which takes more, shorter bands)."""
    from lane_tracker_amd import _native, calib, synth
    from oracle import oracle as O
    cal = calib.reference_calibration()
    r = synth.SceneRenderer(cal)
    n = 40
    frames = np.stack([r.render(700 + i)[0] if i % 8 else synth.frame_uniform(700 + i) for i in range(n)], 0)
    fp = _native.filter_params(filter_type="neighborhood", ksize_r=15, C_r=5, ksize_b=35, C_b=5)
    oc = O.make_calib(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0])
    ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=n)
    try:
        ctx.set_streams(3)
        ctx.upload_frames(frames)
        ctx.mask_run(n, fp)
        assert ctx.last_adaptive_path() == 1
        masks = ctx.download_masks(n)
        for k in (0, 1, 7, 8, 23, 39):
            want = O.filter_lane_points(O.front_end(oc, frames[k]), O.filter_params(filter_type="neighborhood", ksize_r=15, C_r=5, ksize_b=35, C_b=5))
            assert np.array_equal(masks[k], want), k
        for k in (5, 16):
            ctx.mask_run(1, fp, first=k)
            assert np.array_equal(ctx.download_masks(1, first=k)[0], masks[k]), k
    finally:
        ctx.close()
