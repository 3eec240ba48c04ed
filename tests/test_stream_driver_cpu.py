"""The host logic of the stream pipeline without a GPU: `LaneTracker` on `tests/fake_context.py` (an oracle-backed stand-in
for the device context).  What is checked is the DRIVER -- `_run_window_chained`, `process_stream`'s hand-over between
windows, the speculation / cancel / restart logic -- against the plain frame-by-frame state machine (`chain_searches =
False`), which the GPU suite in turn holds against the reference's own `process()` trace.  Reference: lane_tracker.py:851-872,
1064-1128, 1142-1209."""
import numpy as np
import pytest

import fake_context
from lane_tracker_amd import _native, calib, synth
from lane_tracker_amd.lane_tracker import LaneTracker


@pytest.fixture()
def fake(monkeypatch):
    monkeypatch.setattr(_native, "Context", fake_context.FakeContext)
    fake_context.FakeContext.calls = []
    yield fake_context.FakeContext
    fake_context.FakeContext.calls = None


def _state(lt):
    b = lambda a: None if a is None else np.asarray(a).tobytes()
    return dict(detected=lt.detected_pixels, valid=lt.valid_lane_lines, last_detection=lt.last_detection, success=lt.success,
                counter=lt.counter, left_avg=b(lt.left_avg_coeffs), right_avg=b(lt.right_avg_coeffs), last_left=b(lt.last_left_coeffs),
                last_right=b(lt.last_right_coeffs), hist=[b(c) for c in lt.left_fit_coeffs] + [b(c) for c in lt.right_fit_coeffs],
                radii=list(lt.average_curve_radii), radius=lt.average_curve_radius, ecc=lt.eccentricity,
                pix=(b(lt.left_y), b(lt.left_x), b(lt.right_y), b(lt.right_x)), cent=(lt.left_window_centroids, lt.right_window_centroids))


@pytest.fixture(scope="module")
def pool():
    """Eight distinct frames: a drifting lane, the same lane jumped sideways, noise-free grey and black."""
    a = synth.stream_lanes(3, seed=101)
    b = synth.stream_lanes(3, seed=202)
    grey, black = np.full_like(a[0], 128), np.zeros_like(a[0])
    return [a[0], a[1], a[2], b[0], b[1], b[2], grey, black]


def _stream(pool, plan):
    return np.stack([pool[k] for k in plan], 0)


# a: 0 1 2, b: 3 4 5, grey 6, black 7 -- lanes, a one-off failure, a jump, an outage beyond n_reset, recovery
PLAN = [0, 1, 2, 1, 0, 7, 1, 2, 3, 4, 5, 4, 6, 7, 7, 7, 7, 7, 0, 1, 2, 1, 0, 1, 6, 2, 1, 0, 1, 2]


@pytest.mark.parametrize("chunk,depth,windows", [(2, 1, (30,)), (8, 3, (30,)), (4, 2, (7, 11, 12)), (None, 3, (13, 17))])
def test_chained_driver_equals_frame_by_frame_state_machine(fake, pool, chunk, depth, windows):
    cal = calib.reference_calibration()
    frames = _stream(pool, PLAN)
    seq, bat = LaneTracker(**cal), LaneTracker(**cal)
    seq.chain_searches = False
    bat.chain_chunk, bat.chain_depth = chunk, depth
    lo = 0
    for w in windows:
        for f in frames[lo:lo + w]:
            seq.process_batch(f[None], annotate=False)            # one frame at a time through the plain state machine
        bat.process_batch(frames[lo:lo + w], annotate=False)
        assert _state(bat) == _state(seq), (lo, w)
        lo += w
    assert 0 < bat.success < bat.counter == len(PLAN)
    assert bat._ctx.cancels >= 1                                   # speculation was rejected at least once ...
    assert any(not by_value for _, _, by_value in fake.calls)      # ... and chains were continued on the "device"


def test_process_stream_hands_windows_over_and_refuses_interleaving(fake, pool):
    cal = calib.reference_calibration()
    frames = _stream(pool, PLAN)
    sizes = (8, 8, 3, 11)                                          # the last window is longer than the halves: the context grows
    wins, lo = [], 0
    for w in sizes:
        wins.append(frames[lo:lo + w])
        lo += w
    seq, bat = LaneTracker(**cal), LaneTracker(**cal)
    seq.chain_searches = False
    bat.chain_chunk = 4
    gen = bat.process_stream(wins, annotate=False)
    for k, (win, out) in enumerate(zip(wins, gen)):
        for f in win:
            seq.process_batch(f[None], annotate=False)
        assert out == [None] * len(win)
        assert _state(bat) == _state(seq), k
        if k == 0:
            with pytest.raises(RuntimeError):
                bat.process_batch(win, annotate=False)
    assert next(gen, None) is None and not bat._in_stream
    assert bat.counter == sum(sizes)


def test_wide_band_falls_back_to_frame_by_frame(fake, pool):
    """bandwidth 40 is outside the chain kernel's limits (2 * 40 + 2 > 64): the driver must run frame by frame, same state."""
    cal = calib.reference_calibration()
    frames = _stream(pool, PLAN[:12])
    seq, bat = LaneTracker(**cal), LaneTracker(**cal)
    seq.chain_searches = False
    seq.process_batch(frames, annotate=False, bandwidth=40)
    bat.process_batch(frames, annotate=False, bandwidth=40)
    assert _state(bat) == _state(seq) and bat._ctx.tickets == []


@pytest.mark.parametrize("seed,n_reset,n_fail,n_tries,groups", [(1, 4, 8, 2, True), (2, 1, 2, 2, True), (3, 4, 8, 1, True), (4, 0, 8, 2, True),
                                                                   (5, 6, 3, 2, True), (6, 4, 8, 2, False), (7, 4, 8, 2, True), (8, 2, 8, 2, True),
                                                                   (10, 4, 8, 2, True), (11, 3, 8, 2, True), (13, 4, 8, 2, True)])
def test_outages_handled_in_groups_equal_frame_by_frame(fake, pool, seed, n_reset, n_fail, n_tries, groups):
    """Runs of failing frames go through `_fail_group` (all first tries of a group at once, then the second tries of the frames in
    front of the first success, speculating that the outage lasts): the state after every window -- pixel lists and window
    centroids included -- must be the frame-by-frame state machine's, for random streams of lanes, jumps and outages of
    random length (shorter and longer than n_reset and than the group sizes), with one try or two, and windows that end
    inside an outage."""
    rng = np.random.default_rng(seed)
    plan = []
    while len(plan) < 90:
        kind = rng.integers(0, 4)
        if kind == 0:
            plan += [int(rng.integers(0, 3)) for _ in range(rng.integers(1, 7))]          # lane a
        elif kind == 1:
            plan += [int(rng.integers(3, 6)) for _ in range(rng.integers(1, 5))]          # lane b (a jump)
        else:
            plan += [int(rng.integers(6, 8)) for _ in range(rng.integers(1, 14))]         # an outage
    plan = plan[:90]
    frames = _stream(pool, plan)
    cal = calib.reference_calibration()
    seq = LaneTracker(n_reset=n_reset, n_fail=n_fail, **cal)
    bat = LaneTracker(n_reset=n_reset, n_fail=n_fail, **cal)
    seq.chain_searches = False
    bat.chain_chunk, bat.chain_depth, bat.outage_groups = (6, 2, 8)[seed % 3], 1 + seed % 3, groups
    lo = 0
    for w in (17, 30, 1, 42):
        for f in frames[lo:lo + w]:
            seq.process_batch(f[None], annotate=False, n_tries=n_tries)
        bat.process_batch(frames[lo:lo + w], annotate=False, n_tries=n_tries)
        assert _state(bat) == _state(seq), (lo, w)
        lo += w
    assert 0 < bat.success < bat.counter == 90


@pytest.mark.parametrize("cut", [1, 5, 6, 13, 15, 18, 19, 25])
def test_state_round_trip_through_json_continues_the_stream(fake, pool, cut):
    """`get_state()` -> JSON -> `set_state()` on ANOTHER tracker at frame `cut` of PLAN (clean runs, behind a one-off failure,
    inside and right behind the long outage): the second tracker leaves the uncut run's state after every further frame
    (reference lane_tracker.py:139-176; the GPU form with annotated frames: tests/test_gpu_state.py)."""
    import json
    cal = calib.reference_calibration()
    frames = _stream(pool, PLAN)
    whole = LaneTracker(**cal)
    want = []
    for f in frames:
        whole.process_batch(f[None], annotate=False)
        want.append(_state(whole))
    a = LaneTracker(**cal)
    a.process_batch(frames[:cut], annotate=False)
    text = json.dumps(a.get_state())
    b = LaneTracker(**cal)
    b.set_state(json.loads(text))
    assert _state(b) == want[cut - 1]
    for i in range(cut, len(frames)):
        b.process_batch(frames[i][None], annotate=False)
        assert _state(b) == want[i], (cut, i)
