"""Presentation stage on the GPU: lt_overlay_run / lt_download_bev against the oracle's
draw_lane / front_end restatements, bit for bit, and LaneTracker.process() with every return shape of
the reference (annotated frame, (frame, search visualisation), split view)."""
import numpy as np
import pytest

from test_gpu_parity import assert_same

pytestmark = pytest.mark.gpu

H, W = 1100, 1080


@pytest.fixture(scope="module")
def nat():
    from lane_tracker_amd import _native
    _native.load()
    return _native


@pytest.fixture(scope="module")
def cal():
    from lane_tracker_amd import calib
    return calib.reference_calibration()


def _polygons(oracle):
    rng = np.random.default_rng(8)
    polys = []
    for t in range(6):
        lf = np.array([rng.uniform(-1e-4, 1e-4), rng.uniform(-0.2, 0.1), rng.uniform(380, 470)])
        rf = lf + np.array([rng.uniform(-3e-5, 3e-5), rng.uniform(-0.05, 0.05), rng.uniform(170, 215)])
        polys.append(oracle.get_poly_points((W, H), lf, rf, 1 if t % 2 == 0 else 0.5))
    # curves that leave the image (chains of different length -> slanted closing edge), one side only, nothing
    polys.append(oracle.get_poly_points((W, H), np.array([4e-4, -0.9, 500.0]), np.array([3e-4, -0.2, 620.0]), 1))
    ly, lx, ry, rx = polys[0]
    polys.append((ly, lx, ry[:0], rx[:0]))
    polys.append((ly[:0], lx[:0], ry[:0], rx[:0]))
    return polys


def test_overlay_matches_oracle_draw_lane(nat, cal, oracle, ref_calib):
    from lane_tracker_amd import synth
    polys = _polygons(oracle)
    n = len(polys)
    r = synth.SceneRenderer()
    frames = np.stack([r.render(40 + i)[0] if i % 3 else synth.frame_uniform(70 + i) for i in range(n)], 0)
    c = nat.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0],
                    device=0, capacity=n)
    try:
        with pytest.raises(nat.NativeError):
            c.overlay_run(polys)                       # before lt_overlay_configure
        c.overlay_configure(cal["warp_matrices"][1])
        c.upload_frames(frames)
        c.overlay_run(polys)
        out = c.download_overlay(n)
        painted = 0
        for i, p in enumerate(polys):
            want = oracle.draw_lane(ref_calib, cal["warp_matrices"][1], frames[i], *p)
            assert_same(out[i], want, f"overlay {i}")
            painted += int((want != frames[i]).any())
        assert painted >= n - 2                       # the test really exercises the blend
        assert_same(out[-1], frames[-1], "empty polygon = plain copy")
        # one slot at a time, in a different slot, with another alpha
        c.overlay_run([polys[1]], first=4, alpha=0.5)
        one = c.download_overlay(1, first=4)[0]
        lane = oracle.draw_lane(ref_calib, cal["warp_matrices"][1], np.zeros_like(frames[4]), *polys[1])   # 0.3 * lane
        g = np.where(lane[:, :, 1] > 0)
        assert (one[:, :, 0] == frames[4][:, :, 0]).all() and (one[:, :, 2] == frames[4][:, :, 2]).all()
        assert (one[g][:, 1] >= frames[4][g][:, 1]).all() and (one[:, :, 1] != frames[4][:, :, 1]).sum() >= len(g[0]) * 0.9
        # the input frames were not modified
        c.mask_run(n)
        c2 = nat.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"],
                         cal["warp_matrices"][0], device=0, capacity=n)
        try:
            c2.upload_frames(frames)
            c2.mask_run(n)
            assert_same(c.download_masks(n), c2.download_masks(n), "masks after overlay")
        finally:
            c2.close()
    finally:
        c.close()


def test_download_bev_matches_oracle_front_end(nat, cal, oracle, ref_calib):
    from lane_tracker_amd import synth
    frames = np.stack([synth.frame_uniform(5), synth.SceneRenderer().render(6)[0]], 0)
    c = nat.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0],
                    device=0, capacity=2)
    try:
        c.upload_frames(frames)
        with pytest.raises(nat.NativeError):
            c.download_bev(2)                          # needs the undistorted rows of lt_mask_run
        c.mask_run(2)
        bev = c.download_bev(2)
        for k in range(2):
            assert_same(bev[k], oracle.front_end(ref_calib, frames[k]), f"bev {k}")
    finally:
        c.close()


def test_overlay_odd_sizes_generic_kernel(nat, oracle):
    """641x361 camera: the row length is not a multiple of 4, so the one-pixel-per-thread kernel runs."""
    from lane_tracker_amd import calib
    S = np.diag([0.5, 0.5, 1.0])
    K = S @ calib.CAM_MATRIX
    M = S @ calib.M @ np.diag([2.0, 2.0, 1.0])
    Minv = np.diag([0.5, 0.5, 1.0]) @ calib.MINV @ np.diag([2.0, 2.0, 1.0])
    img_size, warped = (641, 361), (541, 551)
    oc = oracle.make_calib(img_size, warped, K, calib.DIST_COEFFS, M)
    c = nat.Context(img_size, warped, K, calib.DIST_COEFFS, M, device=0, capacity=2)
    try:
        frames = np.random.default_rng(3).integers(0, 256, (2, 361, 641, 3), dtype=np.uint8)
        polys = [oracle.get_poly_points(warped, np.array([1e-4, -0.1, 215.0]), np.array([1e-4, -0.1, 310.0]), 1),
                 oracle.get_poly_points(warped, np.array([-2e-4, 0.2, 190.0]), np.array([-1e-4, 0.1, 330.0]), 0.5)]
        c.overlay_configure(Minv)
        c.upload_frames(frames)
        c.overlay_run(polys)
        out = c.download_overlay(2)
        for k in range(2):
            want = oracle.draw_lane(oc, Minv, frames[k], *polys[k])
            assert (want != frames[k]).any()
            assert_same(out[k], want, f"odd overlay {k}")
    finally:
        c.close()


def _tracker(cal, **kw):
    from lane_tracker_amd.lane_tracker import LaneTracker
    return LaneTracker(**cal, **kw)


TEXT_ROWS = 125      # rows the three text lines can touch (baselines at 35, 70, 105)


def test_process_returns_the_oracle_overlay_below_the_text(cal, oracle, ref_calib):
    from lane_tracker_amd import synth
    frames = synth.stream_lanes(4, seed=3)
    lt = _tracker(cal, print_frame_count=True)
    try:
        for i, f in enumerate(frames):
            keep = f.copy()
            out = lt.process(f)
            assert_same(f, keep, "input frame untouched")
            assert lt.valid_lane_lines, i
            want = oracle.draw_lane(ref_calib, cal["warp_matrices"][1], f, lt.left_avg_y, lt.left_avg_x, lt.right_avg_y,
                                    lt.right_avg_x)
            assert_same(out[TEXT_ROWS:], want[TEXT_ROWS:], f"annotated frame {i}")
            assert (out[:TEXT_ROWS] != want[:TEXT_ROWS]).any()          # the text is there
        # draw_lane on a frame that is not resident on the device
        other = synth.frame_uniform(1)
        out = lt.draw_lane(other)
        want = oracle.draw_lane(ref_calib, cal["warp_matrices"][1], other, lt.left_avg_y, lt.left_avg_x, lt.right_avg_y,
                                lt.right_avg_x)
        assert_same(out[TEXT_ROWS:], want[TEXT_ROWS:], "draw_lane(foreign frame)")
        # failure frames: text only
        blank = np.full_like(frames[0], 128)
        for _ in range(lt.n_fail + 1):
            out = lt.process(blank)
        assert not lt.valid_lane_lines and lt.last_detection > lt.n_fail
        assert_same(out[TEXT_ROWS:], blank[TEXT_ROWS:], "print_failure leaves the image alone")
    finally:
        lt.close()


def test_process_visualize_search_and_split_view(cal, oracle, ref_calib):
    from lane_tracker_amd import overlay, synth
    frames = synth.stream_lanes(3, seed=5)
    lt = _tracker(cal)
    try:
        # frame 0: sliding-window search -> windows visualisation
        out, vis = lt.process(frames[0], visualize_search=True)
        assert out.shape == frames[0].shape and vis.shape == (H, W, 3) and vis.dtype == np.uint8
        mask = oracle.mask_from_frame(ref_calib, frames[0])
        r = oracle.sliding_window_search(mask)
        lf, rf = lt.last_left_coeffs, lt.last_right_coeffs
        want = overlay.visualize_sliding_window_search(mask, r["left_centroids"], r["right_centroids"],
                                                       (r["left_y"], r["left_x"]), (r["right_y"], r["right_x"]),
                                                       oracle.get_poly_points((W, H), lf, rf), 30, 40, 30)
        assert_same(vis, want, "sliding-window visualisation")
        assert (vis[:, :, 0] != vis[:, :, 2]).any()
        # frame 1: band search -> band visualisation
        prev = (lt.last_left_coeffs.copy(), lt.last_right_coeffs.copy())
        out, vis = lt.process(frames[1], visualize_search=True)
        mask = oracle.mask_from_frame(ref_calib, frames[1])
        r = oracle.band_search(mask, prev[0], prev[1], oracle.search_params(bandwidth=25))
        want = overlay.visualize_band_search(mask, (r["left_y"], r["left_x"]), (r["right_y"], r["right_x"]),
                                             oracle.get_poly_points((W, H), prev[0], prev[1], 1.0),
                                             oracle.get_poly_points((W, H), lt.last_left_coeffs, lt.last_right_coeffs), 25)
        assert_same(vis, want, "band visualisation")
        # split view: camera frame on top, bird's-eye image and visualisation below
        view = lt.process(frames[2], split_view=True)
        assert view.shape == (720 + 652, 1280, 3)
        from lane_tracker_amd import utils
        bev = oracle.front_end(ref_calib, frames[2])
        assert_same(view[720:, :640], utils.resize_linear(bev, (640, 652)), "bird's-eye pane")
        # a frame with nothing on it: the visualisation is the bare mask
        out, vis = lt.process(np.full_like(frames[0], 128), visualize_search=True)
        assert vis.ndim == 2 and not lt.detected_pixels
        assert_same(vis, oracle.mask_from_frame(ref_calib, np.full_like(frames[0], 128), oracle.filter_params(
            filter_type='neighborhood', C_r=5)), "bare second-try mask")
        view = lt.process(np.full_like(frames[0], 128), split_view=True)
        assert view.shape == (720 + 652, 1280, 3)
    finally:
        lt.close()


def test_process_batch_annotations_equal_process(cal):
    from lane_tracker_amd import synth
    lanes = synth.stream_lanes(10, seed=9)
    frames = np.stack([lanes[i] if i not in (4, 5) else np.full_like(lanes[0], 128) for i in range(10)], 0)
    a, b = _tracker(cal, print_frame_count=True), _tracker(cal, print_frame_count=True)
    try:
        outs_a = [a.process(f) for f in frames]
        outs_b = b.process_batch(frames)
        assert len(outs_b) == 10
        for i in range(10):
            assert_same(outs_b[i], outs_a[i], f"annotated frame {i}")
        assert a.get_success_ratio() == b.get_success_ratio()
    finally:
        a.close()
        b.close()


def test_overlay_text_matches_the_atlas_blend(nat, cal):
    """lt_overlay_text: every glyph cell of the build's atlas blended in white at its advance position."""
    from lane_tracker_amd import overlay, synth
    font = overlay.font_atlas()
    if font is None:
        pytest.skip("Pillow is not installed: no glyph atlas")
    atlas, adv, first_char = font
    frames = np.stack([synth.frame_uniform(3), np.zeros((720, 1280, 3), np.uint8)], 0)
    texts = [["Curve Radius: 1234 m", "Eccentricity: -0.25 m", "Frame: 17"], ["Lane Line Detection Failed"]]
    e = np.zeros(0, np.int64)
    c = nat.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0],
                    device=0, capacity=2)
    try:
        c.overlay_configure(cal["warp_matrices"][1])
        c.upload_frames(frames)
        with pytest.raises(nat.NativeError):
            c.overlay_text(texts)                              # no font yet
        c.overlay_set_font(atlas, adv, first_char)
        c.overlay_run([(e, e, e, e)] * 2)
        c.overlay_text(texts, origin=(20, 8), step=35)
        out = c.download_overlay(2)
        for k in range(2):
            want = frames[k].astype(np.int32)
            for j, line in enumerate(texts[k]):
                x = 20
                for ch in line:
                    g = ord(ch) - first_char
                    cell = atlas[g][:, :adv[g]].astype(np.int32)
                    y = 8 + 35 * j
                    reg = want[y:y + cell.shape[0], x:x + cell.shape[1]]
                    reg += ((255 - reg) * cell[:, :, None] + 127) // 255
                    x += int(adv[g])
            assert_same(out[k], want.astype(np.uint8), f"text frame {k}")
        assert (out[1] == 255).any()                           # fully opaque glyph cores on the black frame
    finally:
        c.close()


@pytest.mark.gpu
def test_overlay_text_pieces_with_different_line_counts_do_not_share_staging(nat, cal):
    """Several lt_overlay_text calls in flight over disjoint slot ranges with alternating line counts (a run of failed frames
    has one line per slot, a lane piece two or three): each slot's lines sit at the buffers' fixed per-slot stride, so a
    later call's host copy cannot land in the bytes an earlier call's queued copy kernel still reads."""
    from lane_tracker_amd import overlay
    font = overlay.font_atlas()
    if font is None:
        pytest.skip("Pillow is not installed: no glyph atlas")
    atlas, adv, first_char = font
    n = 64
    frames = np.zeros((n, 720, 1280, 3), np.uint8)
    e = np.zeros(0, np.int64)
    c = nat.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0],
                    device=0, capacity=n)

    def lines_of(i):
        return ["Curve Radius: %d m" % (100 + i), "Eccentricity: %.2f m" % (i / 64.0), "Frame: %d" % i][:1 + (i // 8) % 3]

    try:
        c.overlay_configure(cal["warp_matrices"][1])
        c.upload_frames(frames)
        c.overlay_set_font(atlas, adv, first_char)
        c.overlay_run([(e, e, e, e)] * n)
        c.overlay_text([lines_of(i) for i in range(16, 24)], first=16)        # sizes the buffers for three lines first
        for rep in range(6):                                                  # pieces of 8 slots, 1 / 2 / 3 lines, back to back
            for p0 in range(0, n, 8):
                c.overlay_text([lines_of(i) for i in range(p0, p0 + 8)], first=p0)
        out = c.download_overlay(n)
        for i in (0, 7, 8, 9, 23, 24, 40, 63):
            want = np.zeros((720, 1280), np.int32)
            for j, line in enumerate(lines_of(i)):
                x = 20
                for ch in line:
                    g = ord(ch) - first_char
                    cell = atlas[g][:, :adv[g]].astype(np.int32)
                    y = 8 + 35 * j
                    reg = want[y:y + cell.shape[0], x:x + cell.shape[1]]
                    for _ in range(6 + (1 if 16 <= i < 24 else 0)):            # blended once per call
                        reg += ((255 - reg) * cell + 127) // 255
                    x += int(adv[g])
            assert_same(out[i][:, :, 0], want.astype(np.uint8), f"text of slot {i}")
    finally:
        c.close()


def test_pinned_pool_hands_out_writable_arrays_and_reuses_blocks():
    """lt_host_alloc behind NumPy arrays: writable, correctly shaped, the block returns to the pool when the
    array and its views are gone, and download_overlay results live in such arrays."""
    import gc
    from lane_tracker_amd import _native
    pool = _native._PinnedPool(limit=64 << 20, keep_per_size=2)
    a = pool.empty((3, 5, 7, 3))
    a[...] = 7
    view = a[1]
    assert a.shape == (3, 5, 7, 3) and a.dtype == np.uint8 and a.flags.writeable and int(view.sum()) == 7 * 5 * 7 * 3
    nbytes = a.nbytes
    assert pool.outstanding == nbytes
    del a
    gc.collect()
    assert pool.outstanding == nbytes                  # the view still holds the block
    del view
    gc.collect()
    assert pool.outstanding == 0 and len(pool.free[nbytes]) == 1
    b = pool.empty((3, 5, 7, 3))
    assert pool.free[nbytes] == [] and pool.outstanding == nbytes       # reused, not reallocated
    big = pool.empty((65 << 20,))                      # over the limit: a plain array, not an error
    assert big.shape == (65 << 20,) and pool.outstanding == nbytes
    del b, big
    gc.collect()
    assert _native.load().lt_host_free(None) == 0


def test_device_lane_from_fit_equals_the_hosts_average_points_and_polygon(nat, cal):
    """k_lane_spans_from_fit (lt_present_lane_from_fit_async: the lane of a process() frame drawn by the device behind its
    search) against what the host forms from the same fit: the running average (_mean_of_rows: the sum in order, the new fit
    last, one division), get_poly_points (lt_poly_points) and the polygon's row intervals (lt_lane_polygon_spans) -- random
    lanes, curves that leave the image on either side (the kept points are compacted towards the bottom), steep curves whose
    edges span several columns per row, one side or both outside the image, every `partial`, histories of 0 to 3 older fits;
    a record without a usable fit leaves empty intervals."""
    from lane_tracker_amd.lane_tracker import _mean_of_rows
    ctx = nat.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=2)
    try:
        rng = np.random.default_rng(21)
        empty = np.tile(np.array([32767, -32768], np.int16), (H, 1))
        for t in range(160):
            partial = (1.0, 1.0, 0.5, 0.25, 0.7)[t % 5]
            ploty = np.linspace(H * (1 - partial), H - 1, int(H * partial))
            ploty2 = ploty ** 2
            kind = t % 8
            lf = np.array([rng.uniform(-1e-4, 1e-4), rng.uniform(-0.2, 0.1), rng.uniform(380, 470)])
            rf = lf + np.array([rng.uniform(-3e-5, 3e-5), rng.uniform(-0.05, 0.05), rng.uniform(170, 215)])
            if kind == 1:
                lf, rf = np.array([4e-4, -0.9, 500.0]), np.array([3e-4, -0.2, 620.0])          # leave the image: chains of different length
            elif kind == 2:
                lf = np.array([rng.uniform(1e-3, 3e-3), rng.uniform(-4, -2), rng.uniform(900, 1400)])    # steep: several columns per row
            elif kind == 3:
                rf = np.array([0.0, 0.0, 5000.0])                                                # the right curve wholly outside
            elif kind == 4:
                lf, rf = np.array([0.0, 0.0, -50.0]), np.array([0.0, 0.0, 5000.0])               # both outside: no polygon
            elif kind == 5:
                lf, rf = np.array([0.0, 0.0, 0.0]), np.array([0.0, 0.0, float(W - 1)])           # exactly on the borders (<=, >=)
            hist = int(rng.integers(0, 4))
            older_l = [lf + rng.normal(0, [1e-6, 1e-3, 2.0]) for _ in range(hist)]
            older_r = [rf + rng.normal(0, [1e-6, 1e-3, 2.0]) for _ in range(hist)]
            prev = None
            if hist:
                accl, accr = older_l[0], older_r[0]
                for k in range(1, hist):
                    accl, accr = accl + older_l[k], accr + older_r[k]
                prev = np.concatenate([accl, accr])
            la, ra = _mean_of_rows(older_l + [lf]), _mean_of_rows(older_r + [rf])
            ln, rn, lyx, ryx = nat.poly_points((W, H), np.concatenate([la, ra])[None], ploty, ploty2)
            want = nat.lane_polygon_spans(H, lyx[:, 0], lyx[:, 1], ryx[:, 0], ryx[:, 1])
            got = ctx.lane_spans_from_fit(np.concatenate([lf, rf]), prev, hist + 1, ploty, ploty2)
            assert np.array_equal(got, want), (t, kind, partial, hist, np.argwhere(got != want)[:4].tolist())
            if t < 4:
                assert np.array_equal(ctx.lane_spans_from_fit(np.concatenate([lf, rf]), prev, hist + 1, ploty, ploty2, detected=False), empty)
                assert np.array_equal(ctx.lane_spans_from_fit(np.concatenate([lf, rf]), prev, hist + 1, ploty, ploty2, fit_flags=2), empty)
    finally:
        ctx.close()
