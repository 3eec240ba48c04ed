"""One frame's rows stored by the calling thread through the PCIe aperture (`lt_upload_frame_rows_enqueue` of a small call on a
large-BAR box; `lt_set_direct_upload`): the same bytes by another way.  What could go wrong is not arithmetic but visibility -- a
kernel reading what an XCD's L2 still holds of the slot's PREVIOUS frame -- so every case re-uses its slots many times with
different frames and compares with the oracle, and with the copy engine's run of the same calls."""
import numpy as np
import pytest

from lane_tracker_amd import _native, calib, synth

pytestmark = pytest.mark.gpu


def _frames(n, seed):
    r = synth.SceneRenderer()
    rng = np.random.default_rng(seed)
    out = []
    for k in range(n):
        out.append(synth.frame_uniform(seed * 1000 + k) if k % 3 == 2 else r.render(int(rng.integers(1 << 30)))[0])
    return np.stack(out)


def _ctx(cal, capacity=2):
    return _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0],
                           capacity=capacity)


def test_rows_through_the_aperture_are_the_rows_the_engine_brings(oracle, ref_calib):
    cal = calib.reference_calibration()
    frames = _frames(24, 5)
    r0, r1 = oracle.warp_source_rows(ref_calib)
    ctx = _ctx(cal)
    try:
        if not ctx.set_direct_upload(True):
            pytest.skip("device memory is not mapped into the process (no large BAR): the engine's path is the only one")
        got = {}
        for direct in (True, False):
            assert ctx.set_direct_upload(direct) == direct
            before = ctx.direct_upload_count()
            und, masks, recs = [], [], []
            for k, f in enumerate(frames):                       # two slots, twelve frames each: every upload lands on a slot
                slot = k & 1                                     # whose previous rows the device has just read
                keep = ctx.upload_frame_rows(f[None], first=slot, enqueue=True)
                ctx.mask_run(1, first=slot)
                ctx.sws_fit_run(1, first=slot)
                und.append(ctx.download_undistorted(1, first=slot)[0])
                masks.append(ctx.download_masks(1, first=slot)[0])
                recs.append(ctx.download_records(1, first=slot).tobytes())
                del keep
            assert ctx.direct_upload_count() - before == (len(frames) if direct else 0)
            got[direct] = (und, masks, recs)
        for k, f in enumerate(frames):
            want = oracle.undistort(ref_calib, f)[r0:r1]
            assert np.array_equal(got[True][0][k], want), ("undistorted rows", k)
            assert np.array_equal(got[True][0][k], got[False][0][k]), k
            assert np.array_equal(got[True][1][k], got[False][1][k]), k
            assert got[True][2][k] == got[False][2][k], k
        for k in (0, 7, 23):
            assert np.array_equal(got[True][1][k], oracle.mask_from_frame(ref_calib, frames[k])), ("mask", k)
    finally:
        ctx.close()


def test_back_to_back_uploads_without_a_download_in_between(oracle, ref_calib):
    """The host-side wait in front of the stores: the second upload into a slot arrives while the first frame's kernels are
    still queued (nothing was downloaded in between); it must not overtake their reads."""
    cal = calib.reference_calibration()
    frames = _frames(8, 9)
    ctx = _ctx(cal)
    try:
        if not ctx.set_direct_upload(True):
            pytest.skip("no large BAR")
        for rounds in range(6):
            a, b = frames[rounds], frames[rounds + 2]
            ctx.upload_frame_rows(a[None], first=0, enqueue=True)
            ctx.mask_run(1, first=0)
            ctx.upload_frame_rows(b[None], first=1, enqueue=True)
            ctx.mask_run(1, first=1)
            ctx.upload_frame_rows(b[None], first=0, enqueue=True)        # slot 0 again: its mask chain may still be running
            m1 = ctx.download_masks(1, first=1)[0]
            ctx.mask_run(1, first=0)
            m0 = ctx.download_masks(1, first=0)[0]
            assert np.array_equal(m0, m1), rounds
            assert np.array_equal(m0, oracle.mask_from_frame(ref_calib, b)), rounds
    finally:
        ctx.close()


def test_process_takes_the_aperture_and_a_large_call_does_not():
    from lane_tracker_amd.lane_tracker import LaneTracker
    cal = calib.reference_calibration()
    frames = synth.stream_lanes(12, seed=3)
    lt = LaneTracker(**cal)
    try:
        able = lt._ctx.set_direct_upload(-1)
        outs = [lt.process(f).copy() for f in frames]
        # (the aperture is taken when the host has SEEN the slot's previous readers finish -- the completion word of the previous
        # frame's lane -- and the engine otherwise: the first frames of a video, frames behind a failure)
        took = lt._ctx.direct_upload_count()
        assert (len(frames) // 2 < took <= len(frames)) if able else took == 0, took
        lt2 = LaneTracker(**cal)
        try:
            lt2._ctx.set_direct_upload(False)
            outs2 = [lt2.process(f).copy() for f in frames]
            assert lt2._ctx.direct_upload_count() == 0
            assert lt2.success == lt.success and lt.success > 0
        finally:
            lt2.close()
        for k in range(len(frames)):
            assert np.array_equal(outs[k], outs2[k]), k
    finally:
        lt.close()
    ctx = _ctx(cal, capacity=16)
    try:
        keep = ctx.upload_frame_rows(np.stack([frames[k % 12] for k in range(16)]), enqueue=True)     # 16 x 914 KB: the engine's business
        ctx.sync()
        del keep
        assert ctx.direct_upload_count() == 0
    finally:
        ctx.close()
