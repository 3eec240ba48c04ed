"""Tracker state as data (`LaneTracker.get_state()` / `set_state()`; reference lane_tracker.py:139-176, SURVEY.md section 5:
"state is ~20 scalars + two <= n_average lists; trivially serialisable"): a stream cut at an arbitrary frame and continued by
ANOTHER tracker from the JSON text of the first one's state leaves the records, lane pixels, annotated frames and attributes of
the uncut run, bit for bit."""
import json

import numpy as np
import pytest

from test_gpu_chain import _state, _stream_with_failures

pytestmark = pytest.mark.gpu


def _full(lt):
    s = _state(lt)
    s["pix"] = tuple(None if a is None else np.asarray(a).tobytes() for a in (lt.left_y, lt.left_x, lt.right_y, lt.right_x))
    s["cent"] = (lt.left_window_centroids, lt.right_window_centroids)
    s["fit"] = tuple(np.asarray(c).tobytes() for c in lt.fit_poly()) if lt.left_y is not None and len(lt.left_y) and len(lt.right_y) else None
    return s


def test_a_1080p_stream_split_across_two_trackers_equals_the_unsplit_run():
    """BASELINE config 5 geometry (1920x1080), windows through `process_stream`, outages included; the cut falls inside an
    outage for one split, right behind a recovery for another, in the middle of a clean run for the third."""
    from lane_tracker_amd import calib
    from lane_tracker_amd.lane_tracker import LaneTracker
    cal = calib.scaled_calibration(1.5)
    n, w = 96, 16
    frames = _stream_with_failures(n, 9, seed=61, cal=cal)
    frames[40:47] = 0                                        # a long outage: back to sliding windows (:851)
    wins = [frames[i:i + w] for i in range(0, n, w)]
    whole = LaneTracker(**cal)
    try:
        want_frames, want_states = [], []
        for out in whole.process_stream(wins, annotate=True):
            want_frames.append(np.stack(out))
        # (states per window: the same stream window by window through process_batch on a second uncut tracker)
        ref = LaneTracker(**cal)
        try:
            for k, wnd in enumerate(wins):
                outs = ref.process_batch(wnd, annotate=True)
                assert np.array_equal(np.stack(outs), want_frames[k]), k
                want_states.append(_full(ref))
        finally:
            ref.close()
        assert _full(whole) == want_states[-1]
        assert 0 < whole.success < whole.counter == n
    finally:
        whole.close()
    for cut in (2, 3, 5):                                    # windows handled by the first tracker: 32 (clean), 48 (the outage has just ended), 80
        a = LaneTracker(**cal)
        try:
            got = [np.stack(o) for o in a.process_stream(wins[:cut], annotate=True)]
            text = json.dumps(a.get_state())                 # what would be written to disk
        finally:
            a.close()
        b = LaneTracker(**cal)
        try:
            b.set_state(json.loads(text))
            assert _full(b) == want_states[cut - 1], cut     # the restored tracker IS the uncut one at that frame
            for k, wnd in enumerate(wins[cut:], start=cut):
                outs = b.process_batch(wnd, annotate=True)
                got.append(np.stack(outs))
                assert _full(b) == want_states[k], (cut, k)
        finally:
            b.close()
        for k in range(len(wins)):
            assert np.array_equal(got[k], want_frames[k]), (cut, k)


@pytest.mark.parametrize("cut", [1, 7, 12, 21, 23, 30])
def test_process_frame_by_frame_continues_from_a_restored_state(cut):
    """`process()` one frame per call (process_video.py:43) at 1280x720: cut behind the first frame, inside and right behind an
    outage (the restored tracker redraws the last lane: draw_lane on the rebuilt polygon of the averages), and in clean runs."""
    from lane_tracker_amd import calib
    from lane_tracker_amd.lane_tracker import LaneTracker
    cal = calib.reference_calibration()
    n = 36
    frames = _stream_with_failures(n, 8, seed=62)
    frames[18:24] = 0
    whole = LaneTracker(**cal)
    try:
        want = []
        for f in frames:
            img = whole.process(f)
            want.append((img.copy(), _full(whole)))
    finally:
        whole.close()
    a = LaneTracker(**cal)
    try:
        for f in frames[:cut]:
            a.process(f)
        text = json.dumps(a.get_state())
    finally:
        a.close()
    b = LaneTracker(**cal)
    try:
        b.set_state(json.loads(text))
        assert _full(b) == want[cut - 1][1]
        for i in range(cut, n):
            img = b.process(frames[i])
            assert np.array_equal(img, want[i][0]), (cut, i)
            assert _full(b) == want[i][1], (cut, i)
    finally:
        b.close()


def test_set_state_refuses_a_state_of_another_tracker():
    from lane_tracker_amd import calib
    from lane_tracker_amd.lane_tracker import LaneTracker
    cal = calib.reference_calibration()
    a, b = LaneTracker(**cal), LaneTracker(n_average=3, **cal)
    try:
        st = a.get_state()
        assert json.loads(json.dumps(st)) == st
        with pytest.raises(ValueError, match="n_average"):
            b.set_state(st)
        with pytest.raises(ValueError, match="version"):
            a.set_state(dict(st, version=99))
        a.set_state(st)                                      # a fresh tracker's own state: a no-op
        assert a.counter == 0 and a.last_detection == a.n_reset + 1 and a.left_fit_coeffs == []
    finally:
        a.close()
        b.close()
