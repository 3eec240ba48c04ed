"""LaneTracker.process(), the valid first try's bookkeeping in one host call (`_tail_fast` -> lt_frame_tail) and the text lines
drawn at once (`text_now` -> lt_host_text_now_group): frame by frame the annotated frames and the whole tracker state equal those of
the Python functions one after the other (fit_poly, _lane_ahead, check_validity, _record_success, get_curve_radius, the text as a
job for a copy thread) -- over streams with failures and an outage, at both sizes, with the demo settings (greenery mask, other
validity limits, half the look-ahead) and a frame counter in the text."""
import numpy as np
import pytest

from test_gpu_chain import _state, _stream_with_failures

pytestmark = pytest.mark.gpu


def _full(lt):
    s = _state(lt)
    s["pix"] = tuple(None if a is None else np.asarray(a).tobytes() for a in (lt.left_y, lt.left_x, lt.right_y, lt.right_x))
    s["types"] = (type(lt.eccentricity).__name__, type(lt.average_curve_radius).__name__, type(lt.left_curve_radius).__name__,
                  lt.left_avg_coeffs.dtype.str, lt.left_avg_x.dtype.str, lt.left_avg_y.dtype.str)
    return s


@pytest.mark.parametrize("scale,demo,n_average,frame_count", [(1.0, None, 2, False), (1.5, None, 3, True), (1.0, "demo1", 2, False),
                                                              (1.0, "demo3", 1, True)])
def test_one_call_tail_leaves_what_the_python_functions_leave(scale, demo, n_average, frame_count):
    from lane_tracker_amd import calib, settings
    from lane_tracker_amd.lane_tracker import LaneTracker
    cal = calib.reference_calibration() if scale == 1.0 else calib.scaled_calibration(scale)
    n = 72
    frames = _stream_with_failures(n, 7, seed=17 + int(scale * 10), cal=cal)
    frames[40:46] = 0                                             # an outage: sliding windows again (:851)
    fast = LaneTracker(n_average=n_average, print_frame_count=frame_count, **cal)
    slow = LaneTracker(n_average=n_average, print_frame_count=frame_count, **cal)
    slow.fast_tail = slow.text_now = False
    kw = {}
    if demo:
        kw = settings.apply(fast, settings.DEMOS[demo])
        settings.apply(slow, settings.DEMOS[demo])
    took = []
    inner = fast._tail_fast
    fast._tail_fast = lambda partial: took.append(inner(partial)) or took[-1]
    try:
        for k, f in enumerate(frames):
            a, b = fast.process(f, **kw), slow.process(f, **kw)
            assert np.array_equal(a, b), k
            assert _full(fast) == _full(slow), k
        assert fast.success == slow.success and 0 < fast.success < n
        # the one-call path is the usual one: most valid frames took it (not the first frames of a video and those behind a failure,
        # whose lane the device has not drawn)
        assert took.count(True) > fast.success // 2, (took.count(True), took.count(False), took.count(None), fast.success)
    finally:
        fast.close()
        slow.close()


def test_two_trackers_on_two_threads_share_the_copy_threads_without_mixing_anything():
    """Two trackers, one per thread, process() at the same time: the aperture stores and the text lines of both offer pieces to the
    same polling copy threads (work stealing, group 0), each context keeps its own account of its slots' readers.  Frames and final
    states equal those of each tracker run alone."""
    import threading
    from lane_tracker_amd import calib
    from lane_tracker_amd.lane_tracker import LaneTracker
    cal = calib.reference_calibration()
    streams = [_stream_with_failures(120, 11, seed=5), _stream_with_failures(120, 13, seed=6)]

    def run(frames, out):
        lt = LaneTracker(**cal)
        try:
            out.append([lt.process(f).copy() for f in frames])
            out.append(_full(lt))
        finally:
            lt.close()
    alone = []
    for s in streams:
        o = []
        run(s, o)
        alone.append(o)
    together = [[], []]
    threads = [threading.Thread(target=run, args=(streams[i], together[i])) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i in range(2):
        assert len(together[i]) == 2, "a thread died"
        assert together[i][1] == alone[i][1], i
        for k, (a, b) in enumerate(zip(together[i][0], alone[i][0])):
            assert np.array_equal(a, b), (i, k)
