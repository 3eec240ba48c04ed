"""The device-chained band search of one stateful stream (lt_band_fit_chain_run, SURVEY 8(f) N2 / BASELINE config 5) and the
stream pipeline built on it (LaneTracker.process_batch).  The reference's behaviour here is `process()` frame by frame
(lane_tracker.py:851-872, 1064-1128, 1178-1199): every comparison below is against exactly that -- the chained run must
leave the same records, the same lane pixels and the same tracker state, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ctx(cal, n):
    from lane_tracker_amd import _native
    return _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0],
                           device=0, capacity=n)


def _coeffs(rec):
    return np.concatenate([rec["left_coeffs"], rec["right_coeffs"]])


@pytest.mark.parametrize("streams,batch_kernel", [(1, True), (3, True), (1, False)])
def test_chain_records_and_pixels_equal_frame_by_frame_band_search(streams, batch_kernel):
    """batch_kernel: the frame-by-frame side runs k_band_fit2 (the kernel of lt_band_fit_run for two and more frames: frame i is
    searched together with frame i + 1, whose result under the wrong seed is overwritten by the next step); otherwise one frame
    per call, which is itself a chain of one (launch_band_fit_one)."""
    from lane_tracker_amd import _native, calib, synth
    cal = calib.reference_calibration()
    n = 24
    frames = synth.stream_lanes(n, seed=21)
    sp = _native.search_params()
    a, b = _ctx(cal, n), _ctx(cal, n)
    try:
        for c in (a, b):
            c.set_streams(streams)                          # 3 streams: the chain crosses slice boundaries
            c.upload_frames(frames)
            c.mask_run(n)
            c.sws_fit_run(1, sp, first=0)
        # frame by frame: each band search seeded by the previous frame's record through the host
        for i in range(1, n):
            prev = _coeffs(a.download_records(1, first=i - 1)[0])
            if batch_kernel and i + 1 < n:
                a.band_fit_run(2, np.stack([prev, prev]), sp, first=i)
            else:
                a.band_fit_run(1, prev[None, :], sp, first=i)
        want = a.download_records(n)
        assert want["detected"].all() and (want["mode"][1:] == 1).all()
        # chained: one launch, seeded by the record of slot 0
        b.band_fit_chain_run(n - 1, None, sp, first=1)
        got = b.download_records(n)
        assert got.tobytes() == want.tobytes()
        for slot in (1, n // 2, n - 1):
            for side in (0, 1):
                gy, gx = b.download_pixels(slot, side)
                wy, wx = a.download_pixels(slot, side)
                assert np.array_equal(gy, wy) and np.array_equal(gx, wx)
        # seeded by value from the middle of the stream: the same records from there on
        b.band_fit_chain_run(n - 10, _coeffs(want[9]), sp, first=10)
        assert b.download_records(n).tobytes() == want.tobytes()
        with pytest.raises(ValueError):
            b.band_fit_chain_run(3, None, sp, first=0)       # nothing to continue from
    finally:
        a.close()
        b.close()


def test_chain_stops_behind_a_frame_without_lanes_and_marks_the_rest():
    from lane_tracker_amd import _native, calib, synth
    cal = calib.reference_calibration()
    n = 10
    frames = synth.stream_lanes(n, seed=22).copy()
    frames[4] = 0                                            # nothing to find in this frame
    sp = _native.search_params()
    c = _ctx(cal, n)
    try:
        c.upload_frames(frames)
        c.mask_run(n)
        c.set_frame_base(n, 700)
        c.sws_fit_run(1, sp, first=0)
        c.band_fit_chain_run(n - 1, None, sp, first=1)
        rec = c.download_records(n)
        assert rec["detected"][:4].all() and not rec["detected"][4] and rec["mode"][4] == 1
        assert (rec["mode"][5:] == 255).all() and not rec["detected"][5:].any()
        assert list(rec["frame"]) == list(range(700, 700 + n))      # the caller's tags survive
        # a wide band is outside the chain kernel's limits: an error, not a wrong answer
        with pytest.raises(_native.NativeError):
            c.band_fit_chain_run(2, _coeffs(rec[0]), _native.search_params(bandwidth=40), first=1)
    finally:
        c.close()


def test_cancelled_chain_stops_and_later_chains_run():
    from lane_tracker_amd import _native, calib, synth
    cal = calib.reference_calibration()
    n = 200
    base = synth.stream_lanes(20, seed=23)
    frames = np.concatenate([base, base[::-1]] * 5, 0)
    sp = _native.search_params()
    c = _ctx(cal, n)
    try:
        c.upload_frames(frames)
        c.mask_run(n)
        c.sws_fit_run(1, sp, first=0)
        c.sync()
        c.mask_run(n, first=0)                                 # the chain waits for the masks of its slots: ~4 ms of mask
        c.mask_run(n, first=0)                                 # work are queued in front of it (same masks again) ...
        c.sws_fit_run(1, sp, first=0)
        c.band_fit_chain_run(n - 1, None, sp, first=1)         # ... then ~2.5 ms of chain ...
        c.band_fit_chain_cancel()                              # ... told to stop right away: long before it can finish
        rec = c.band_fit_chain_collect(n, first=0)
        stopped = int(np.argmax(rec["mode"] == 255))
        assert rec["mode"][-1] == 255 and 1 <= stopped < n and (rec["mode"][stopped:] == 255).all()
        assert rec["detected"][:stopped].all()
        c.band_fit_chain_run(n - 1, None, sp, first=1)         # a chain enqueued after the cancel is a new speculation
        full = c.band_fit_chain_collect(n, first=0)
        assert full["detected"].all() and (full["mode"][1:] == 1).all()
        assert full[:stopped].tobytes() == rec[:stopped].tobytes()
        with pytest.raises(_native.NativeError):
            c.band_fit_chain_collect(4, first=0)               # collected tickets are gone
    finally:
        c.close()


def _state(lt):
    return dict(detected=lt.detected_pixels, valid=lt.valid_lane_lines, last_detection=lt.last_detection, success=lt.success,
                counter=lt.counter, left_avg=None if lt.left_avg_coeffs is None else lt.left_avg_coeffs.tobytes(),
                right_avg=None if lt.right_avg_coeffs is None else lt.right_avg_coeffs.tobytes(),
                last_left=None if lt.last_left_coeffs is None else np.asarray(lt.last_left_coeffs).tobytes(),
                last_right=None if lt.last_right_coeffs is None else np.asarray(lt.last_right_coeffs).tobytes(),
                hist=[c.tobytes() for c in lt.left_fit_coeffs] + [c.tobytes() for c in lt.right_fit_coeffs],
                radii=list(lt.average_curve_radii), radius=lt.average_curve_radius, lr=(lt.left_curve_radius, lt.right_curve_radius),
                ecc=lt.eccentricity, avg_pts=(lt.left_avg_y.tobytes(), lt.left_avg_x.tobytes(), lt.right_avg_y.tobytes(),
                                              lt.right_avg_x.tobytes()))


def _stream_with_failures(n, every, seed, cal=None):
    """A drifting lane; every `every`-th frame is replaced in turn by noise, a flat grey frame, or black."""
    from lane_tracker_amd import synth
    frames = synth.stream_lanes(n, seed=seed, cal=cal).copy()
    for k, i in enumerate(range(every - 1, n, every)):
        if k % 3 == 0:
            frames[i] = synth.frame_uniform(4000 + i, img_size=(frames.shape[2], frames.shape[1]))
        elif k % 3 == 1:
            frames[i] = 128
        else:
            frames[i] = 0
    return frames


@pytest.mark.parametrize("every,windows,annotate", [(10, (64,), False), (10, (7, 33, 20, 4), False), (4, (64,), False),
                                                     (10, (40, 24), True), (1000, (64,), False)])
def test_process_batch_chained_equals_process_frame_by_frame(every, windows, annotate):
    """Failures every ~10 (and every 4) frames, several window splits, with and without annotation: the state after every
    window and the annotated frames equal frame-by-frame process()."""
    from lane_tracker_amd import calib
    from lane_tracker_amd.lane_tracker import LaneTracker
    cal = calib.reference_calibration()
    n = sum(windows)
    frames = _stream_with_failures(n, every, seed=31)
    seq, bat = LaneTracker(**cal), LaneTracker(**cal)
    seq.host_copies_rows = False         # whole frames from the device on the frame-by-frame side, row runs in the batch
    try:
        assert bat.chain_searches and bat.host_copies_rows
        lo = 0
        for w in windows:
            outs_seq = [seq.process(f) for f in frames[lo:lo + w]]
            outs = bat.process_batch(frames[lo:lo + w], annotate=annotate)
            assert _state(bat) == _state(seq), (lo, w)
            assert np.array_equal(bat.left_x, seq.left_x) and np.array_equal(bat.left_y, seq.left_y)
            assert np.array_equal(bat.right_x, seq.right_x) and np.array_equal(bat.right_y, seq.right_y)
            assert bat.left_window_centroids == seq.left_window_centroids
            if annotate:
                for k, (g, s) in enumerate(zip(outs, outs_seq)):
                    assert np.array_equal(g, s), (lo, k)
            lo += w
        if every < 100:
            assert 0 < bat.success < bat.counter == n        # the stream really had failures and recoveries
        else:
            assert bat.success == n
    finally:
        seq.close()
        bat.close()


def test_process_batch_chained_equals_unchained_per_frame_state():
    """Frame-by-frame state: windows of one frame through the chained driver against the unchained driver on a stream with
    failures (every state of the machine is visited: sws start, band, one-off failure, reset to sws after n_reset misses)."""
    from lane_tracker_amd import calib
    from lane_tracker_amd.lane_tracker import LaneTracker
    cal = calib.reference_calibration()
    frames = _stream_with_failures(48, 6, seed=33)
    frames[20:27] = 0                                        # a long outage: falls back to sliding windows (:851)
    a, b = LaneTracker(**cal), LaneTracker(**cal)
    b.chain_searches = False
    try:
        for i in range(0, 48, 3):
            a.process_batch(frames[i:i + 3], annotate=False)
            b.process_batch(frames[i:i + 3], annotate=False)
            assert _state(a) == _state(b), i
        assert 0 < a.success < a.counter and a.last_detection <= 1
    finally:
        a.close()
        b.close()


def test_config5_1080p_stream_chained():
    """BASELINE config 5 through the chained pipeline: 1920x1080 camera, band-search warm start, validity on the host."""
    from lane_tracker_amd import calib
    from lane_tracker_amd.lane_tracker import LaneTracker
    cal = calib.scaled_calibration(1.5)
    frames = _stream_with_failures(30, 9, seed=35, cal=cal)
    seq, bat = LaneTracker(**cal), LaneTracker(**cal)
    try:
        for f in frames:
            seq.process(f)
        bat.process_batch(frames, annotate=False)
        assert _state(bat) == _state(seq)
        assert np.array_equal(bat.left_x, seq.left_x) and np.array_equal(bat.right_y, seq.right_y)
        assert bat.success >= 24
    finally:
        seq.close()
        bat.close()


def test_config5_1080p_annotated_frames_as_row_runs():
    """1920x1080 (BASELINE config 5): the annotated frames of a batch and of a stream of windows travel as row runs (text
    lines + the rows the lane can reach; the rest copied from the caller's window by the library's copy threads) and equal
    the frames `process()` brings back whole, failures included."""
    from lane_tracker_amd import calib
    from lane_tracker_amd.lane_tracker import LaneTracker
    cal = calib.scaled_calibration(1.5)
    frames = _stream_with_failures(36, 9, seed=39, cal=cal)
    seq, bat, stm = LaneTracker(**cal), LaneTracker(**cal), LaneTracker(**cal)
    seq.host_copies_rows = False
    try:
        rows = bat._present_rows()
        assert rows is not None and rows[2][1] <= rows[2][2] and rows[2][3] <= 1080
        want = [seq.process(f) for f in frames]
        got = bat.process_batch(frames, annotate=True)
        assert _state(bat) == _state(seq)
        assert all(np.array_equal(g, w) for g, w in zip(got, want))
        outs = [o for win in stm.process_stream([frames[:20], frames[20:21], frames[21:]], annotate=True) for o in win]
        assert _state(stm) == _state(seq)
        assert len(outs) == len(want) and all(np.array_equal(g, w) for g, w in zip(outs, want))
        assert 0 < seq.success < seq.counter
    finally:
        seq.close()
        bat.close()
        stm.close()


def test_chain_fuzz_short():
    """A short run of tests/fuzz_chain.py (random streams with jumps and outages, random tracker / chain parameters, random
    window splits): process_batch == process() after every window."""
    import fuzz_chain
    assert fuzz_chain.main(iters=6, seed=11, verbose=False) == 0


@pytest.mark.parametrize("annotate,sizes,every,n_average,band_one",
                         [(False, (24, 24, 10, 40, 24, 1, 24), 9, 2, "1"), (True, (24, 24, 10, 40, 24, 1, 24), 9, 2, "1"),
                          (True, (70, 0, 50, 31), 40, 3, "1"), (True, (64, 64), 1000, 1, "1")])
def test_process_stream_equals_process_frame_by_frame(annotate, sizes, every, n_average, band_one):
    """Windows of one video through process_stream (the next window's uploads and masks run while the current one's searches
    drain; windows resident side by side) -- state after every window and the annotated frames equal process();
    a longer window in the middle forces the context to grow; process() inside an active stream is refused.  The long
    runs of valid frames of the last two cases go through the all-at-once bookkeeping (`_record_successes`), with an empty
    window in between; an annotated window is handed out while the next one's first searches are already in flight."""
    from lane_tracker_amd import calib
    from lane_tracker_amd.lane_tracker import LaneTracker
    cal = calib.reference_calibration()
    frames = _stream_with_failures(sum(sizes), every, seed=37)
    wins, lo = [], 0
    for w in sizes:
        wins.append(frames[lo:lo + w])
        lo += w
    seq, bat = LaneTracker(n_average=n_average, **cal), LaneTracker(n_average=n_average, **cal)
    seq.host_copies_rows = False         # the frame-by-frame side takes whole frames from the device, the stream row runs
    try:
        assert bat.host_copies_rows
        gen = bat.process_stream(wins, annotate=annotate)
        for k, (win, outs) in enumerate(zip(wins, gen)):
            outs_seq = [seq.process(f) for f in win]
            assert _state(bat) == _state(seq), k
            assert np.array_equal(bat.left_x, seq.left_x) and np.array_equal(bat.right_y, seq.right_y)
            assert bat.left_window_centroids == seq.left_window_centroids
            if annotate:
                assert all(np.array_equal(g, s) for g, s in zip(outs, outs_seq)), k
            else:
                assert outs == [None] * len(win)
            if k == 1:
                with pytest.raises(RuntimeError):
                    bat.process(frames[0])
        assert next(gen, None) is None and not bat._in_stream
        assert 0 < bat.success <= bat.counter == sum(sizes) and (bat.success < bat.counter or every > sum(sizes))
        bat.process(frames[0])                                 # usable again once the generator is exhausted
    finally:
        seq.close()
        bat.close()


def test_inplace_annotation_draws_into_the_callers_windows():
    """annotate="inplace": the annotated frames ARE the caller's arrays -- same bytes as the frames annotate=True returns for a
    copy of the same stream, the rows outside the lane's run and the text lines untouched, state equal; a read-only window
    comes back as new frames; process_batch likewise."""
    from lane_tracker_amd import calib
    from lane_tracker_amd.lane_tracker import LaneTracker
    cal = calib.reference_calibration()
    sizes = (24, 40, 1, 31)
    frames = _stream_with_failures(sum(sizes), 13, seed=41)
    pristine = frames.copy()
    wins_a, wins_b, lo = [], [], 0
    for w in sizes:
        wins_a.append(frames[lo:lo + w].copy())          # each window an array of its own (a decoder's buffers)
        wins_b.append(pristine[lo:lo + w])
        lo += w
    wins_a[2].flags.writeable = False
    a, b = LaneTracker(**cal), LaneTracker(**cal)
    try:
        for k, (oa, ob) in enumerate(zip(a.process_stream(wins_a, annotate="inplace"), b.process_stream(wins_b, annotate=True))):
            assert _state(a) == _state(b), k
            assert all(np.array_equal(x, y) for x, y in zip(oa, ob)), k
            if k != 2:
                assert all(np.shares_memory(x, wins_a[k]) for x in oa), k          # the caller's own memory
                assert not np.array_equal(wins_a[k], wins_b[k])                    # ... drawn over
            else:
                assert not any(np.shares_memory(x, wins_a[k]) for x in oa) and np.array_equal(wins_a[k], wins_b[k])
            assert all(not np.shares_memory(y, wins_b[k]) for y in ob) and np.array_equal(wins_b[k], pristine[sum(sizes[:k]):sum(sizes[:k + 1])])
        win = pristine[:32].copy()
        want = b.process_batch(pristine[:32], annotate=True)
        got = a.process_batch(win, annotate="inplace")
        assert all(np.array_equal(x, y) for x, y in zip(got, want)) and all(np.shares_memory(x, win) for x in got)
        assert _state(a) == _state(b)
    finally:
        a.close()
        b.close()


def test_annotated_stream_closed_early_leaves_a_usable_tracker():
    """A consumer that stops after the first window: the generator's clean-up cancels the searches in flight and waits for
    the copies still writing into the page-locked frame arrays; the tracker takes frames again and its first window came
    out right."""
    from lane_tracker_amd import calib, synth
    from lane_tracker_amd.lane_tracker import LaneTracker
    cal = calib.reference_calibration()
    frames = synth.stream_lanes(96, seed=21)
    seq, bat = LaneTracker(**cal), LaneTracker(**cal)
    try:
        gen = bat.process_stream([frames[:32], frames[32:64], frames[64:]], annotate=True)
        first = next(gen)
        want = [seq.process(f) for f in frames[:32]]
        assert all(np.array_equal(a, b) for a, b in zip(first, want))
        gen.close()
        assert not bat._in_stream
        out = bat.process_batch(frames[:8])
        assert len(out) == 8 and all(o.shape == frames[0].shape for o in out)
    finally:
        seq.close()
        bat.close()


def test_one_frame_chain_with_a_chain_behind_it_keeps_both_tickets():
    """A chain of ONE frame followed by a chain seeded by its record: collecting the first must take the first chain's
    ticket (the second one also holds that slot -- as its seed record), so that the second can still be collected."""
    from lane_tracker_amd import _native, calib, synth
    cal = calib.reference_calibration()
    frames = synth.stream_lanes(12, seed=3)
    c = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0,
                        capacity=12)
    try:
        sp = _native.search_params()
        c.upload_frames(frames)
        c.mask_run(12)
        c.sws_fit_run(1, sp, first=0)
        seed = c.download_records(1)[0]
        c.band_fit_chain_run(1, np.concatenate([seed["left_coeffs"], seed["right_coeffs"]]), sp, first=1)   # one frame, seed by value
        c.band_fit_chain_run(1, None, sp, first=2)             # one frame, seeded by slot 1 ...
        c.band_fit_chain_run(5, None, sp, first=3)             # ... and a longer one behind it
        a = c.band_fit_chain_collect(1, first=1)
        b = c.band_fit_chain_collect(1, first=2)
        d = c.band_fit_chain_collect(5, first=3)
        assert a["detected"].all() and b["detected"].all() and d["detected"].all() and (d["mode"] == 1).all()
        c.sws_fit_run(1, sp, first=0)                          # the seed slot itself is collectable when nothing searched it in a chain
        c.band_fit_chain_run(3, None, sp, first=1)
        assert c.band_fit_chain_collect(4, first=0)["detected"].all()
    finally:
        c.close()


@pytest.mark.parametrize("annotate,n_tries,n_reset", [(False, 2, 4), (True, 2, 4), (False, 1, 2), (True, -1, 0)])
def test_long_outages_in_groups_equal_process_frame_by_frame(annotate, n_tries, n_reset):
    """Outages of 3, 11 and 37 frames (noise / grey / black; shorter and longer than n_reset and than the largest group of
    `_fail_group`), an isolated failure and a lane jump: the windows of process_batch leave the state, the pixel lists, the
    centroids and (annotated) the frames of process() frame by frame -- with one try, two, or 'as many as it takes' (-1)."""
    from lane_tracker_amd import calib, synth
    from lane_tracker_amd.lane_tracker import LaneTracker
    cal = calib.reference_calibration()
    a, b = synth.stream_lanes(60, seed=41), synth.stream_lanes(30, seed=42)
    frames = np.concatenate([a[:20], b[:12], a[20:]], 0).copy()          # a jump to another lane and back
    for start, length, kind in ((8, 3, 0), (25, 1, 1), (40, 11, 2), (55, 37, 0)):
        for i in range(start, min(len(frames), start + length)):
            frames[i] = synth.frame_uniform(500 + i) if (kind + i) % 3 == 0 else (128 if (kind + i) % 3 == 1 else 0)
    seq, bat = LaneTracker(n_reset=n_reset, **cal), LaneTracker(n_reset=n_reset, **cal)
    try:
        lo = 0
        for w in (30, 45, len(frames) - 75):
            outs = bat.process_batch(frames[lo:lo + w], annotate=annotate, n_tries=n_tries)
            outs_seq = [seq.process(f, n_tries=n_tries) for f in frames[lo:lo + w]]
            assert _state(bat) == _state(seq), lo
            assert np.array_equal(bat.left_x, seq.left_x) and np.array_equal(bat.right_y, seq.right_y)
            assert bat.left_window_centroids == seq.left_window_centroids and bat.right_window_centroids == seq.right_window_centroids
            if annotate:
                assert all(np.array_equal(g, s) for g, s in zip(outs, outs_seq)), lo
            lo += w
        assert 0 < bat.success < bat.counter == len(frames)
    finally:
        seq.close()
        bat.close()
