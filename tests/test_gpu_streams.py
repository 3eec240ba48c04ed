"""The context's stream arrangements must never change a result: CUs reserved for the chained search (lt_set_search_cus),
the urgent stream of a frame whose first try failed (lt_set_urgent), the presentation stream and the targeted wait for
annotated frames (lt_download_overlay_wait).  Everything is compared with the same calls on a plain context."""
import numpy as np
import pytest

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
def frames():
    from lane_tracker_amd import synth
    return synth.stream_lanes(48, seed=12)


def _ctx(nat, cal, capacity):
    return nat.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0,
                       capacity=capacity)


def _reference_run(nat, cal, frames):
    c = _ctx(nat, cal, len(frames))
    try:
        sp = nat.search_params()
        c.upload_frames(frames)
        c.mask_run(len(frames))
        c.sws_fit_run(len(frames), sp)
        rec = c.download_records(len(frames)).copy()
        masks = c.download_masks(len(frames)).copy()
        seed = np.concatenate([rec["left_coeffs"][0], rec["right_coeffs"][0]])
        c.band_fit_chain_run(len(frames) - 1, seed, sp, first=1)
        chain = c.band_fit_chain_collect(len(frames) - 1, first=1).copy()
        return rec, masks, chain
    finally:
        c.close()


@pytest.mark.parametrize("cus", [1, 8])
def test_reserved_cus_change_no_result(nat, cal, frames, cus):
    rec0, masks0, chain0 = _reference_run(nat, cal, frames)
    c = _ctx(nat, cal, len(frames))
    try:
        sp = nat.search_params()
        c.set_search_cus(cus)
        c.set_streams(3)                                       # streams created after the reservation carry the mask too
        c.upload_frames(frames)
        c.mask_run(len(frames))
        c.sws_fit_run(len(frames), sp)
        rec = c.download_records(len(frames))
        assert rec.tobytes() == rec0.tobytes() and np.array_equal(c.download_masks(len(frames)), masks0)
        seed = np.concatenate([rec["left_coeffs"][0], rec["right_coeffs"][0]])
        c.band_fit_chain_run(len(frames) - 1, seed, sp, first=1)
        c.mask_run(len(frames) // 2, first=0)                  # the mask chain beside the chain on its own CUs
        assert c.band_fit_chain_collect(len(frames) - 1, first=1).tobytes() == chain0.tobytes()
        c.set_search_cus(0)                                    # and back
        c.mask_run(len(frames))
        c.sws_fit_run(len(frames), sp)
        assert c.download_records(len(frames)).tobytes() == rec0.tobytes()
        with pytest.raises(ValueError):
            c.set_search_cus(65)
    finally:
        c.close()


def test_urgent_calls_equal_ordinary_calls_and_stay_ordered(nat, cal, frames):
    rec0, masks0, _ = _reference_run(nat, cal, frames)
    n = len(frames)
    c = _ctx(nat, cal, 2 * n)
    try:
        sp = nat.search_params()
        fp2 = nat.filter_params('neighborhood', 15, 5, 35, 5)
        c.upload_frames(frames, first=0)
        c.upload_frames(frames, first=n)
        c.mask_run(n, first=0)
        c.sync()
        # a long queue on the slots' streams (masks of the second half, twice), then urgent work on slots 4..11 of the first half
        c.mask_run(n, first=n)
        c.mask_run(n, first=n)
        with c.urgent():
            c.sws_fit_run(8, sp, first=4)
            rec = c.download_records(8, first=4)
            assert rec.tobytes() == rec0[4:12].tobytes()
            c.mask_run(8, fp2, first=4)                        # second-try masks ...
            c.sws_fit_run(8, sp, first=4)
            second = c.download_records(8, first=4).copy()
            c.mask_run(8, first=4)                             # ... and the first-try masks again, left in flight
        seed = np.concatenate([rec0["left_coeffs"][3], rec0["right_coeffs"][3]])
        c.band_fit_run(8, np.tile(seed, (8, 1)), sp, first=4)  # ordinary call on the same slots: ordered behind the urgent masks
        band = c.download_records(8, first=4).copy()
        assert np.array_equal(c.download_masks(n, first=0), masks0) and np.array_equal(c.download_masks(n, first=n), masks0)
        # the same second-try and band results on a fresh context, nothing urgent
        d = _ctx(nat, cal, n)
        try:
            d.upload_frames(frames)
            d.mask_run(8, fp2, first=4)
            d.sws_fit_run(8, sp, first=4)
            assert d.download_records(8, first=4).tobytes() == second.tobytes()
            d.mask_run(n)
            d.band_fit_run(8, np.tile(seed, (8, 1)), sp, first=4)
            assert d.download_records(8, first=4).tobytes() == band.tobytes()
        finally:
            d.close()
    finally:
        c.close()


def test_overlay_pieces_with_targeted_wait_equal_one_blocking_download(nat, cal, frames):
    from lane_tracker_amd.lane_tracker import LaneTracker
    n = 24
    t = LaneTracker.__new__(LaneTracker)
    t.warped_size = cal["warped_size"]
    ploty, ploty2 = t._plot_rows(1)
    rng = np.random.default_rng(2)
    coeffs = np.stack([[rng.uniform(-1e-4, 1e-4), rng.uniform(-0.2, 0.1), rng.uniform(380, 470), 0, 0, 0] for _ in range(n)])
    coeffs[:, 3:] = coeffs[:, :3] + [0.0, 0.0, 200.0]
    packed = nat.poly_points(cal["warped_size"], coeffs, ploty, ploty2)
    texts = [["Curve Radius: %d m" % (100 + i), "Eccentricity: 0.%02d m" % i] for i in range(n)]
    c = _ctx(nat, cal, 2 * n)
    try:
        import os
        if os.environ.get("LT_TEST_SEARCH_CUS"):
            c.set_search_cus(int(os.environ["LT_TEST_SEARCH_CUS"]))
        c.overlay_configure(cal["warp_matrices"][1])
        from lane_tracker_amd import overlay as ov
        font = ov.font_atlas()
        if font is not None:
            c.overlay_set_font(font[0], font[1], font[2])
        c.upload_frames(frames[:n], first=0)
        c.overlay_run_packed(*packed, first=0)
        if font is not None:
            c.overlay_text(texts, first=0)
        want = c.download_overlay(n, first=0).copy()
        # the same frames in the other half of the context, in three pieces, while a mask chain is queued behind an upload
        c.upload_frame_rows_async(frames[:n], first=n)
        c.mask_run(n, first=n)
        c.upload_frame_rest(frames[:n], first=n)
        out = nat.pinned_empty((n,) + frames.shape[1:])
        le, re = np.cumsum(packed[0]), np.cumsum(packed[1])
        for a, b in ((0, 5), (5, 16), (16, n)):
            la, ra = (le[a - 1] if a else 0), (re[a - 1] if a else 0)
            c.overlay_run_packed(packed[0][a:b], packed[1][a:b], packed[2][la:le[b - 1]], packed[3][ra:re[b - 1]], first=n + a)
            if font is not None:
                c.overlay_text(texts[a:b], first=n + a)
            c.download_overlay_async(out[a:b], first=n + a)
        c.download_overlay_wait()
        assert np.array_equal(out, want)
    finally:
        c.close()


def test_kernel_download_path_gives_the_same_frames():
    """Annotated frames copied back by a kernel on the CUs set aside for it, instead of the copy engine -- what the measured choice
    (lt_set_download_method, next test) takes when the engine is the slower way -- forced for every copy of a process: the
    pieces test above and an annotated stream against process().  Forcing it process-wide is a measurement switch
    (LT_DL_KERNEL=1), so this runs on the experiments build in a process of its own."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exp = os.path.join(root, "lane_tracker_amd", "liblane_tracker_amd_exp.so")
    if not os.path.exists(exp):
        subprocess.check_call(["make", "-C", os.path.join(root, "lane_tracker_amd", "csrc"), "-s", "-j8", "EXPERIMENTS=1"])
    env = dict(os.environ, LT_DL_KERNEL="1", LT_TEST_SEARCH_CUS="3", LANE_TRACKER_AMD_LIB=exp)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu",
                        "tests/test_gpu_streams.py::test_overlay_pieces_with_targeted_wait_equal_one_blocking_download",
                        "tests/test_gpu_chain.py::test_process_stream_equals_process_frame_by_frame[True-sizes1-9-2-1]"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_download_method_is_chosen_by_measurement_and_every_choice_gives_the_same_frames(nat, cal):
    """lt_download_overlay_async times every copy and moves the frames by the copy engine or by a kernel, whichever the
    measurements favour (lt_set_download_method / lt_download_stats).  The same annotated frames come back whichever way:
    forced engine, forced kernel, and the measured choice over enough copies for it to have samples of its own."""
    from lane_tracker_amd import synth
    n = 24
    frames = synth.stream_lanes(n, seed=21, cal=cal)
    e = np.zeros(0, np.int64)
    c = nat.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0],
                    device=0, capacity=n)
    try:
        c.overlay_configure(cal["warp_matrices"][1])
        c.upload_frames(frames)
        ys = np.arange(300, 1100, dtype=np.int64)
        poly = (ys, np.full_like(ys, 430), ys, np.full_like(ys, 640))
        c.overlay_run([poly if i % 3 else (e, e, e, e) for i in range(n)])
        want = c.download_overlay(n)                       # the blocking download
        assert c.download_stats()["engine_copies"] == 0 and c.download_stats()["kernel_copies"] == 0
        outs = {}
        for method in (0, 1, -1):
            c.set_download_method(method)
            out = nat.pinned_empty(want.shape)
            out[...] = 0
            for rep in range(8 if method < 0 else 1):      # 8 x 3 copies of 22 MB: enough samples for the rule to look at
                for a, b in ((0, 8), (8, 16), (16, n)):
                    c.download_overlay_async(out[a:b], first=a)
            c.download_overlay_wait()
            outs[method] = out
            assert np.array_equal(out, want), method
        st = c.download_stats()
        assert st["engine_copies"] >= 3 and st["kernel_copies"] >= 3, st      # both were really used and timed
        assert st["method"] in ("engine", "kernel") and st["engine_GBs"] > 1.0 and st["kernel_GBs"] > 1.0, st
        with pytest.raises(ValueError):
            c.set_download_method(2)
    finally:
        c.close()
