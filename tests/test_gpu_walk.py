"""The long-walk threshold kernels (csrc/k_threshold_walk.hip) against the oracle on bird's-eye images of many sizes:
widths that are and are not multiples of 64 / 128, heights below and above one 128-row group, images smaller than a
window, every supported pair of window sizes.  They run through the context's own filter chain (lt_upload_bev +
lt_filter_run); `last_threshold_path` tells that the walking kernels -- not the tile kernel -- produced the masks."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SIZES = [(1080, 1100), (1084, 300), (256, 128), (132, 70), (8, 5), (64, 64), (1280, 200), (644, 130), (128, 129),
         (1152, 64), (72, 300)]
PARAMS = [(15, 8, 35, 5), (20, 5, 35, 5), (35, 0, 15, 12), (15, 0, 15, 0), (20, 3, 20, 7), (35, 9, 35, 1)]


def _bev(rng, h, w, kind):
    if kind == 0:
        return rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if kind == 1:      # smooth blocks + noise: many pixels close to the threshold
        base = rng.integers(40, 200, ((h + 15) // 16, (w + 15) // 16, 3)).repeat(16, 0).repeat(16, 1)[:h, :w]
        return np.clip(base + rng.integers(-12, 13, (h, w, 3)), 0, 255).astype(np.uint8)
    img = rng.integers(0, 30, (h, w, 3))                    # dark image with bright thin structures and saturated patches
    for _ in range(6):
        x = int(rng.integers(0, w))
        img[:, max(x - 3, 0):x + 4] = rng.integers(180, 256)
        y = int(rng.integers(0, h))
        img[max(y - 2, 0):y + 3, :] = 255
    return img.astype(np.uint8)


def _bilateral_np(p, k, C):
    """bilateral_adaptive_threshold (lane_tracker.py:14-83) from prefix sums, zero outside the image."""
    out = np.zeros(p.shape, bool)
    t = k * p.astype(np.int64) - C * k
    for axis in (0, 1):
        pad = [(k + 1, k + 1) if a == axis else (0, 0) for a in range(2)]
        P = np.cumsum(np.pad(p.astype(np.int64), pad), axis=axis)
        n = p.shape[axis]
        take = lambda lo: np.take(P, np.arange(lo, lo + n), axis=axis)
        before, after = take(k) - take(0), take(2 * k + 1) - take(k + 1)
        out |= (before < t) & (after < t)
    return out


@pytest.mark.parametrize("size", SIZES)
def test_walk_kernels_match_the_oracle_on_many_sizes(size, monkeypatch):
    from lane_tracker_amd import _native, calib
    from oracle import oracle as O
    w, h = size
    cal = calib.reference_calibration()
    ctx = _native.Context(cal["img_size"], (w, h), cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=3)
    ctx.set_walk_min_frames(0)                             # these small calls would take the tile kernel by default
    rng = np.random.default_rng(w * 7919 + h)
    try:
        for pi, (kr, cr, kb, cb) in enumerate(PARAMS):
            if (w * h > 500000) and pi > 1:
                continue                                    # the oracle's 55x55 top-hat on a full-size plane takes a while
            bev = np.stack([_bev(rng, h, w, k) for k in range(3)], 0)
            fp = _native.filter_params(ksize_r=kr, C_r=cr, ksize_b=kb, C_b=cb)
            ctx.upload_bev(bev)
            ctx.filter_run(3, fp)
            assert ctx.last_threshold_path() == 1, "the walking kernels did not take these parameters"
            got = ctx.download_masks(3)
            merged = ctx.download_plane(_native.PLANE_MERGED, 3)
            th_r = ctx.download_plane(_native.PLANE_TOPHAT_R, 3)
            for i in range(3):
                want, planes = O.filter_lane_points(bev[i], O.filter_params(ksize_r=kr, C_r=cr, ksize_b=kb, C_b=cb), want_planes=True)
                tag = (size, (kr, cr, kb, cb), i)
                assert np.array_equal(th_r[i], planes[2]), (tag, "top-hat plane (stored with a padded pitch)")
                # the merged plane is what the two thresholds OR into: brute-force NumPy of lane_tracker.py:14-83 on the oracle's top-hats
                expect = _bilateral_np(planes[2], kr, cr) | _bilateral_np(planes[3], kb, cb)
                assert np.array_equal(merged[i] > 0, expect), (tag, int(((merged[i] > 0) != expect).sum()))
                assert np.array_equal(got[i], want), tag
    finally:
        ctx.close()


NOISE = [dict(), dict(noise_thresh=120, C_noise=0), dict(noise_thresh=0), dict(noise_thresh=256), dict(noise_thresh=300, C_noise=40),
         dict(noise_thresh=-5, C_noise=3), dict(noise_thresh=255, C_noise=249)]


@pytest.mark.parametrize("size", [(1080, 1100), (1084, 300), (256, 128), (132, 70), (8, 5), (64, 64), (644, 130), (72, 300)])
def test_greenery_mask_through_the_walking_kernels(size, monkeypatch):
    """mask_noise (lane_tracker.py:221-231; the author's Demo 1 / Demo 3 settings): the third walk with window 65 over the raw
    Lab-b plane -- which the 55x55 top-hat launch leaves in the padded layout -- with the inRange term folded in, AND-ed
    into the merged plane on the way into the 5x5 open."""
    from lane_tracker_amd import _native, calib
    from oracle import oracle as O
    w, h = size
    cal = calib.reference_calibration()
    ctx = _native.Context(cal["img_size"], (w, h), cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=3)
    ctx.set_walk_min_frames(0)                             # these small calls would take the tile kernel by default
    rng = np.random.default_rng(w * 31 + h)
    try:
        for pi, nz in enumerate(NOISE):
            if (w * h > 500000) and pi > 1:
                continue
            kw = dict(ksize_r=15, C_r=8, ksize_b=35, C_b=5, mask_noise=True, **nz)
            bev = np.stack([_bev(rng, h, w, k) for k in range(3)], 0)
            if pi % 2:   # Lab-b values around the threshold: shades of grey with a yellow / blue tint
                tint = rng.integers(-40, 41, (3, h, w))
                grey = rng.integers(60, 200, (3, h, w))
                bev = np.clip(np.stack([grey + tint, grey + tint, grey - tint], -1), 0, 255).astype(np.uint8)
            ctx.upload_bev(bev)
            ctx.filter_run(3, _native.filter_params(**kw))
            assert ctx.last_threshold_path() == 1, "the walking kernels did not take the greenery mask"
            got = ctx.download_masks(3)
            merged = ctx.download_plane(_native.PLANE_MERGED, 3)
            for i in range(3):
                want, planes = O.filter_lane_points(bev[i], O.filter_params(**kw), want_planes=True)
                b = planes[1]
                noise = ~(b >= kw.get("noise_thresh", 140)) | _bilateral_np(b, 65, kw.get("C_noise", 10))
                expect = (_bilateral_np(planes[2], 15, 8) | _bilateral_np(planes[3], 35, 5)) & noise
                tag = (size, nz, i)
                assert np.array_equal(merged[i] > 0, expect), (tag, int(((merged[i] > 0) != expect).sum()))
                assert np.array_equal(got[i], want), tag
            # the same slots again without the mask: nothing of it may linger
            fp = _native.filter_params(ksize_r=15, C_r=8, ksize_b=35, C_b=5)
            ctx.filter_run(3, fp)
            assert np.array_equal(ctx.download_masks(3)[1], O.filter_lane_points(bev[1], O.filter_params(ksize_r=15, C_r=8, ksize_b=35, C_b=5)))
    finally:
        ctx.close()


def test_other_parameters_take_the_tile_kernel_and_agree(monkeypatch):
    """Window sizes outside {15, 20, 35}, a greenery mask with another window than 65 and LT-internal limits fall back to
    k_bilateral_tile2."""
    from lane_tracker_amd import _native, calib
    from oracle import oracle as O
    cal = calib.reference_calibration()
    w, h = 260, 150
    ctx = _native.Context(cal["img_size"], (w, h), cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=1)
    ctx.set_walk_min_frames(0)                             # these small calls would take the tile kernel by default
    rng = np.random.default_rng(5)
    try:
        bev = _bev(rng, h, w, 1)[None]
        for kw in (dict(ksize_r=17, C_r=8, ksize_b=35, C_b=5), dict(ksize_r=15, C_r=8, ksize_b=35, C_b=5, mask_noise=True, ksize_noise=45),
                   dict(ksize_r=15, C_r=8, ksize_b=35, C_b=5, mask_noise=True, C_noise=250)):
            ctx.upload_bev(bev)
            ctx.filter_run(1, _native.filter_params(**kw))
            assert ctx.last_threshold_path() == 0
            assert np.array_equal(ctx.download_masks(1)[0], O.filter_lane_points(bev[0], O.filter_params(**kw)))
            assert np.array_equal(ctx.download_plane(_native.PLANE_TOPHAT_R, 1)[0],
                                  O.filter_lane_points(bev[0], O.filter_params(**kw), want_planes=True)[1][2])
    finally:
        ctx.close()


def test_small_calls_take_the_tile_kernel_by_default(monkeypatch):
    """The policy: a call with few frames cannot fill the chip with long walks and takes the tile kernel."""
    from lane_tracker_amd import _native, calib, synth
    cal = calib.reference_calibration()
    ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=96)
    try:
        f = synth.SceneRenderer(cal).render(3)[0]
        ctx.upload_frames(np.broadcast_to(f, (96,) + f.shape))
        ctx.mask_run(2)
        assert ctx.last_threshold_path() == 0
        small = ctx.download_masks(2)
        ctx.mask_run(96)
        assert ctx.last_threshold_path() == 1
        big = ctx.download_masks(96)
        assert np.array_equal(big[0], small[0]) and np.array_equal(big[95], small[1])
        ctx.set_walk_min_frames(0)                           # the threshold is the caller's to move (lt_set_walk_min_frames) ...
        ctx.mask_run(2)
        assert ctx.last_threshold_path() == 1 and np.array_equal(ctx.download_masks(2), small)
        ctx.set_walk_min_frames(-1)                          # ... and to put back
        ctx.mask_run(2)
        assert ctx.last_threshold_path() == 0
    finally:
        ctx.close()
