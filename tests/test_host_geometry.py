"""Host-side geometry of the tracker (get_poly_points :511-528, check_validity :561-627): the product's cached /
count-only formulation against the oracle's literal NumPy restatement of the reference (itself pinned by the
sws_/band golden fixtures, which store the reference's own get_poly_points / check_validity outputs)."""
import numpy as np
import pytest

from helpers import golden_files
from lane_tracker_amd.lane_tracker import LaneTracker
from oracle import oracle as O


class HostOnlyTracker(LaneTracker):
    """The geometry methods need no device: skip the constructor."""

    def __init__(self, warped_size=(1080, 1100)):
        self.warped_size = warped_size


def test_poly_points_and_validity_match_the_literal_formulation():
    t = HostOnlyTracker()
    rng = np.random.default_rng(0)
    for i in range(1500):
        lf = np.array([rng.uniform(-6e-4, 6e-4), rng.uniform(-1.2, 1.2), rng.uniform(-200, 1300)])
        rf = lf + np.array([rng.uniform(-1e-4, 1e-4), rng.uniform(-0.2, 0.2), rng.uniform(-60, 400)])
        for partial in (1, 1.0, 0.5, 0.3):
            got, want = t.get_poly_points(lf, rf, partial), O.get_poly_points((1080, 1100), lf, rf, partial)
            assert all(np.array_equal(g, w) and g.dtype == w.dtype for g, w in zip(got, want)), (i, partial)
        t.valid_lane_lines = None
        t.check_validity(lf, rf)
        assert t.valid_lane_lines == O.check_validity((1080, 1100), lf, rf), i


@pytest.mark.parametrize("path", golden_files("sws_")[:12])
def test_poly_points_against_the_reference_fixture(path):
    d = np.load(path)
    if not bool(d["detected"]):
        pytest.skip("nothing detected in this case")
    h, w = [int(v) for v in d["mask_shape"]]
    t = HostOnlyTracker((w, h))
    partial = d["param_partial"].item()
    ly, lx, ry, rx = t.get_poly_points(d["left_coeffs"], d["right_coeffs"], partial)
    assert np.array_equal(lx, d["poly_left_x"]) and np.array_equal(rx, d["poly_right_x"])
    assert np.array_equal(ly, d["poly_left_y"]) and np.array_equal(ry, d["poly_right_y"])
    t.valid_lane_lines = None
    t.check_validity(d["left_coeffs"], d["right_coeffs"])
    assert t.valid_lane_lines == bool(d["valid"])


def test_vectorised_validity_equals_check_validity_entry_by_entry():
    """`_valid_many` (the stream pipeline checks a whole chain of fits at once) must store exactly what `check_validity` stores
    for each fit: same f64 operations, same order -- including fits on the decision boundaries, with custom limits, and the
    reference-generated fixtures."""
    t = HostOnlyTracker()
    rng = np.random.default_rng(3)
    lfs, rfs = [], []
    for i in range(4000):
        lf = np.array([rng.uniform(-3e-4, 3e-4), rng.uniform(-0.6, 0.6), rng.uniform(200, 700)])
        sep = rng.choice([80.0, 110.0, 150.0, 200.0, 230.0, rng.uniform(60, 260)])       # on and around the distance limits
        d_slope = rng.choice([0.25, -0.25, 0.2499999999999, rng.uniform(-0.4, 0.4)])     # ... and the tangent threshold
        rf = lf + np.array([rng.uniform(-2e-5, 2e-5), d_slope if i % 3 == 0 else rng.uniform(-0.05, 0.05), sep])
        lfs.append(lf)
        rfs.append(rf)
    for path in golden_files("sws_") + golden_files("band"):
        d = np.load(path)
        if bool(d["detected"]) and tuple(int(v) for v in d["mask_shape"]) == (1100, 1080):
            lfs.append(np.array(d["left_coeffs"], np.float64))
            rfs.append(np.array(d["right_coeffs"], np.float64))
    LF, RF = np.stack(lfs), np.stack(rfs)
    for limits in (None, dict(min_dist_y1=100, max_dist_y1=300, min_dist_y2=90, max_dist_y2=260, min_dist_y3=60, max_dist_y3=250, thresh=0.3)):
        if limits:
            t.validity_limits = limits
        got = t._valid_many(LF, RF)
        want = []
        for lf, rf in zip(LF, RF):
            t.valid_lane_lines = None
            t.check_validity(lf, rf)
            want.append(t.valid_lane_lines)
        assert got.dtype == bool and got.tolist() == want
        assert 0.02 < np.mean(want) < 0.98                    # both outcomes are exercised
    assert t._valid_many(LF[:1], RF[:1]).shape == (1,)


def test_rendezvous_path_is_new_for_every_elastic_attempt(monkeypatch):
    """The file through which rank 0 publishes the RCCL id carries the launcher's pid, the port, and the elastic run id and
    restart count: a restarted attempt never reads the id of the attempt that died."""
    from lane_tracker_amd import distributed
    monkeypatch.delenv("LT_GATHER_ID", raising=False)
    monkeypatch.setenv("MASTER_PORT", "29512")
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "job/7")
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "0")
    a = distributed.rendezvous_path()
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "1")
    b = distributed.rendezvous_path()
    assert a != b and "29512" in a and "/" not in a.rsplit("/", 1)[1].replace(".id", "").replace("lt_gather_", "")
    monkeypatch.setenv("LT_GATHER_ID", "/tmp/explicit.id")
    assert distributed.rendezvous_path() == "/tmp/explicit.id"
    assert distributed.shares_devices() is False
    monkeypatch.setenv("LT_DEVICE_MODULO", "1")
    assert distributed.shares_devices() is True


def _history_tracker(n_average, print_frame_count=False):
    t = HostOnlyTracker()
    t.n_average, t.print_frame_count = n_average, print_frame_count
    t.mppv, t.mpph = 0.02002, 0.01851            # metres per pixel of the reference calibration's order
    t.left_fit_coeffs, t.right_fit_coeffs, t.average_curve_radii = [], [], []
    t.counter = t.success = 0
    t.last_detection = 3
    return t


def _four(polygon):
    """A deferred polygon as upstream's four int64 arrays (the run recorder defers lt_poly_points' packed int32 pairs)."""
    return polygon.as_tuple() if hasattr(polygon, "as_tuple") else polygon


def _scalar_commit(t, lf, rf, partial, deferred):
    t.counter += 1
    t._fit = ("pending", None, lf, rf)
    t._record_success(lf, rf, partial)
    deferred.append(('lane', (t.left_avg_y, t.left_avg_x, t.right_avg_y, t.right_avg_x), t._lane_text()))


@pytest.mark.parametrize("leaves_the_image", [True, False])
@pytest.mark.parametrize("n_average,partial,frame_count", [(1, 1.0, False), (2, 1.0, True), (3, 0.5, False), (5, 1.0, True), (8, 0.3, False)])
def test_run_of_successes_recorded_at_once_equals_frame_by_frame(n_average, partial, frame_count, leaves_the_image):
    """`_record_successes` (the annotated stream pipeline commits a whole run of valid frames in one go) against
    `_record_success` frame by frame: every picture (polygon points, text lines) and the state left behind, bit for bit --
    with a history that holds a failure and older fits when the run starts.  Without a curve that leaves the image at the bottom
    the pictures are deferred as averaged coefficients (the device forms their points; here their host form is compared), with
    one as lt_poly_points' points."""
    rng = np.random.default_rng(10 + n_average)
    g = 70
    LF = np.stack([[rng.uniform(-3e-4, 3e-4), rng.uniform(-0.5, 0.3), rng.uniform(350, 520)] for _ in range(g)])
    RF = LF + np.stack([[rng.uniform(-2e-5, 2e-5), rng.uniform(-0.05, 0.05), rng.uniform(150, 230)] for _ in range(g)])
    if leaves_the_image:
        LF[20] = [2e-4, -1.4, 1400.0]            # leaves the image at the bottom: fewer points than rows
    else:                                        # tame lanes: every bottom point inside the image
        LF[:, :2] *= 0.2
        RF = LF + (RF - LF) * np.array([0.2, 0.2, 1.0])
    a, b = _history_tracker(n_average, frame_count), _history_tracker(n_average, frame_count)
    for t in (a, b):                             # what happened before the run
        d0 = []
        _scalar_commit(t, LF[0] * 1.01, RF[0] * 1.01, partial, d0)
        t.counter += 1
        t._record_failure()
    want, got = [], []
    for j in range(g):
        _scalar_commit(a, np.array(LF[j]), np.array(RF[j]), partial, want)
    for j in range(n_average - 1):
        _scalar_commit(b, np.array(LF[j]), np.array(RF[j]), partial, got)
    assert b._record_successes(LF, RF, n_average - 1, g - 1, partial, got)
    _scalar_commit(b, np.array(LF[g - 1]), np.array(RF[g - 1]), partial, got)
    assert len(got) == len(want) == g
    from lane_tracker_amd.stream import _CoeffPoly
    assert leaves_the_image or any(isinstance(x[1], _CoeffPoly) for x in got)
    for j, (x, y) in enumerate(zip(got, want)):
        assert x[0] == y[0] and x[2] == y[2], (j, x[2], y[2])
        for p, q in zip(_four(x[1]), _four(y[1])):
            assert p.dtype == q.dtype and np.array_equal(p, q), j
    for name in ("counter", "success", "last_detection", "average_curve_radii", "average_curve_radius", "eccentricity",
                 "left_curve_radius", "right_curve_radius"):
        assert getattr(a, name) == getattr(b, name), name
    for name in ("left_fit_coeffs", "right_fit_coeffs"):
        assert len(getattr(a, name)) == len(getattr(b, name)) and all(np.array_equal(p, q) for p, q in zip(getattr(a, name), getattr(b, name)))
    for name in ("left_avg_coeffs", "right_avg_coeffs", "left_avg_x", "left_avg_y", "right_avg_x", "right_avg_y", "last_left_coeffs"):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name


def test_run_of_successes_hands_the_delicate_frames_back():
    """A radius on an integer (the exact refit decides), a straight lane (infinite radius) or a parabola outside the image:
    nothing is recorded and the caller goes frame by frame."""
    t = _history_tracker(2)
    rng = np.random.default_rng(4)
    LF = np.stack([[rng.uniform(1e-4, 3e-4), rng.uniform(-0.3, 0.1), 420.0] for _ in range(12)])
    RF = LF + [0.0, 0.0, 190.0]
    for case in ("straight", "outside", "integer"):
        L2 = LF.copy()
        if case == "straight":
            L2[5, 0] = 0.0
        elif case == "outside":
            L2[5] = [0.0001, 0.0, 5000.0]
            L2[6] = [0.0001, 0.0, 5000.0]
        else:
            # choose `a` so that the radius is an integer to ~1e-12: solve by bisection on the closed form
            f = lambda a: ((1 + (2 * (a * t.mpph / t.mppv ** 2) * 1100 * t.mppv + L2[5, 1] * t.mpph / t.mppv) ** 2) ** 1.5) / abs(2 * a * t.mpph / t.mppv ** 2)
            target = float(np.floor(f(2e-4)))
            lo, hi = 2e-4, 2.2e-4
            assert (f(lo) - target) * (f(hi) - target) < 0
            for _ in range(200):
                mid = 0.5 * (lo + hi)
                lo, hi = (mid, hi) if (f(mid) - target) * (f(hi) - target) < 0 else (lo, mid)
            L2[5, 0] = lo
            assert abs(f(lo) - target) < 1e-6
        before = (t.counter, t.success, list(t.left_fit_coeffs))
        d = []
        assert t._record_successes(L2, RF, 1, 11, 1.0, d) is False
        assert d == [] and (t.counter, t.success, list(t.left_fit_coeffs)) == before, case


def test_packed_poly_points_equal_get_poly_points():
    """lt_poly_points (C, host-only) = get_poly_points for each pair of parabolas, including ones that leave the image, for
    integer and fractional `partial` (the plot rows come from the tracker's own NumPy linspace)."""
    from lane_tracker_amd import _native
    t = HostOnlyTracker()
    rng = np.random.default_rng(8)
    C = np.stack([[rng.uniform(-6e-4, 6e-4), rng.uniform(-1.2, 1.2), rng.uniform(-200, 1300),
                   rng.uniform(-6e-4, 6e-4), rng.uniform(-1.2, 1.2), rng.uniform(-200, 1300)] for _ in range(300)])
    C[7] = [0, 0, 1079.0, 0, 0, 0.0]             # exactly on both borders
    C[8] = [0, 0, 1079.0000001, 0, 0, -1e-9]     # just outside
    for partial in (1, 1.0, 0.5, 0.3):
        ploty, ploty2 = t._plot_rows(partial)
        ln, rn, lyx, ryx = _native.poly_points(t.warped_size, C, ploty, ploty2)
        le, re = np.cumsum(ln), np.cumsum(rn)
        for i in range(len(C)):
            ly, lx, ry, rx = t.get_poly_points(C[i, :3], C[i, 3:], partial)
            a, b = lyx[le[i] - ln[i]:le[i]], ryx[re[i] - rn[i]:re[i]]
            assert np.array_equal(a[:, 0], ly) and np.array_equal(a[:, 1], lx) and np.array_equal(b[:, 0], ry) and np.array_equal(b[:, 1], rx), (i, partial)
    assert ln[7] == 1100 * 0 + len(t._plot_rows(0.3)[0]) and rn[7] == ln[7] and ln[8] == 0 and rn[8] == 0


def test_mean_of_rows_is_numpy_average():
    """The averaged lane coefficients (reference :1191-1192, np.average over the history) through the cheaper helper: bit for
    bit, for every history length and magnitudes over ten decades."""
    from lane_tracker_amd.lane_tracker import _mean_of_rows
    rng = np.random.default_rng(0)
    for k in range(1, 12):
        for _ in range(500):
            rows = [rng.uniform(-1, 1, 3) * 10.0 ** rng.integers(-6, 4) for _ in range(k)]
            a, b = np.average(rows, axis=0), _mean_of_rows(rows)
            assert a.tobytes() == b.tobytes() and b is not rows[0]


@pytest.mark.parametrize("path", golden_files("sws_")[:12])
def test_packed_poly_points_against_the_reference_fixture(path):
    """lt_poly_points against what the reference's own get_poly_points returned for these fits (tests/gen_golden.py)."""
    from lane_tracker_amd import _native
    d = np.load(path)
    if not bool(d["detected"]):
        pytest.skip("nothing detected in this case")
    h, w = [int(v) for v in d["mask_shape"]]
    t = HostOnlyTracker((w, h))
    partial = d["param_partial"].item()
    ploty, ploty2 = t._plot_rows(partial)
    ln, rn, lyx, ryx = _native.poly_points((w, h), np.concatenate([d["left_coeffs"], d["right_coeffs"]])[None], ploty, ploty2)
    assert np.array_equal(lyx[:, 1], d["poly_left_x"]) and np.array_equal(ryx[:, 1], d["poly_right_x"])
    assert np.array_equal(lyx[:, 0], d["poly_left_y"]) and np.array_equal(ryx[:, 0], d["poly_right_y"])
    assert ln[0] == len(d["poly_left_x"]) and rn[0] == len(d["poly_right_x"])


def test_runs_of_valid_frames_are_recorded_at_once_around_near_straight_frames():
    """`_commit_valid_run`: a run of valid first tries recorded all at once (`_record_successes`) except around frames whose
    curve radius needs the scalar route (`_delicate_radii`: near-straight lanes, a radius above 2.5e7 m or next to an integer)
    -- state, radii, eccentricity and the deferred pictures equal those of the frame-by-frame route, frame by frame."""
    import fake_context
    from lane_tracker_amd import _native, calib
    from lane_tracker_amd.lane_tracker import LaneTracker
    cal = calib.reference_calibration()
    real = _native.Context
    _native.Context = fake_context.FakeContext
    try:
        rng = np.random.default_rng(77)
        for n_average, g in ((2, 40), (3, 64), (1, 23), (2, 9)):
            ys = np.arange(100, 1070, 7, dtype=np.int64)

            def fits(n):
                a = rng.uniform(-8e-5, 8e-5, n)
                a[rng.random(n) < 0.2] *= 1e-6                             # near-straight: radius beyond 2.5e7 m (a dead-straight
                                                                           # fit, radius inf, fails in int() upstream too)
                b = rng.uniform(-0.12, 0.04, n)
                cl = rng.uniform(420, 450, n)
                LF = np.stack([a, b, cl], 1)
                RF = np.stack([a * rng.uniform(0.9, 1.1, n), b + rng.uniform(-0.01, 0.01, n), cl + rng.uniform(175, 195, n)], 1)
                return LF, RF
            LF, RF = fits(g)
            results = []
            for at_once in (False, True):
                lt = LaneTracker(n_average=n_average, **cal)
                deferred = []

                def commit(j, lt=lt, deferred=deferred):
                    lt.counter += 1
                    lf, rf = np.array(LF[j]), np.array(RF[j])
                    # the pixel lists the exact refit of get_curve_radius reads: points on the two parabolas
                    lt._lp['left_y'], lt._lp['right_y'] = ys, ys
                    lt._lp['left_x'] = np.rint(lf[0] * ys ** 2 + lf[1] * ys + lf[2]).astype(np.int64)
                    lt._lp['right_x'] = np.rint(rf[0] * ys ** 2 + rf[1] * ys + rf[2]).astype(np.int64)
                    lt._pending = None
                    lt._fit = (lt._lp['left_y'], lt._lp['right_y'], lf, rf)
                    lt._record_success(lf, rf, 1.0)
                    deferred.append(('lane', (lt.left_avg_y, lt.left_avg_x, lt.right_avg_y, lt.right_avg_x), lt._lane_text()))
                if at_once:
                    lt._commit_valid_run(LF, RF, g, 0, True, 1.0, deferred, commit)
                else:
                    for j in range(g):
                        commit(j)
                results.append((lt, deferred))
            (a, da), (b, db) = results
            assert len(da) == len(db) == g
            for j, (x, y) in enumerate(zip(da, db)):
                assert x[2] == y[2], (n_average, j, x[2], y[2])                                    # the text: radius, eccentricity
                assert all(np.array_equal(p, q) for p, q in zip(_four(x[1]), _four(y[1]))), (n_average, j)       # the polygon
            assert (a.counter, a.success, list(a.average_curve_radii), a.average_curve_radius, a.eccentricity) == \
                   (b.counter, b.success, list(b.average_curve_radii), b.average_curve_radius, b.eccentricity)
            assert all(np.array_equal(p, q) for p, q in zip(a.left_fit_coeffs + a.right_fit_coeffs, b.left_fit_coeffs + b.right_fit_coeffs))
            if g >= 2 * n_average + 4:
                assert b._delicate_radii(LF, RF).any() and not b._delicate_radii(LF, RF).all()       # both kinds of stretch occurred
    finally:
        _native.Context = real


def _bare_tracker(size=(1280, 720)):
    t = HostOnlyTracker(size)
    t.mppv, t.mpph = 30 / 720, 3.7 / 700
    t.n_average, t.left_fit_coeffs, t.right_fit_coeffs, t.average_curve_radii = 2, [], [], []
    t.success, t.counter, t.print_frame_count = 0, 0, False
    return t


def test_curve_radius_on_python_floats_is_the_numpy_scalar_value():
    """get_curve_radius computes on Python floats; upstream (:541-546) on NumPy f64 scalars.  Same doubles, same operations:
    the integers must agree for every lane, nearly straight ones (huge radii) included."""
    t = _bare_tracker()
    rng = np.random.default_rng(11)
    y_eval, mppv, mpph = t.warped_size[1], t.mppv, t.mpph
    done = 0
    for i in range(4000):
        scale = 10.0 ** rng.uniform(-7.5, -3)
        lf = np.array([rng.choice([-1, 1]) * scale, rng.uniform(-1.2, 1.2), rng.uniform(0, 1280)])
        rf = lf + np.array([rng.uniform(-1, 1) * scale * 0.1, rng.uniform(-0.05, 0.05), rng.uniform(100, 300)])
        want = []
        for c in (lf, rf):
            fit_m = (c[0] * mpph / (mppv ** 2), c[1] * mpph / mppv)
            want.append(int(((1 + (2 * fit_m[0] * y_eval * mppv + fit_m[1]) ** 2) ** 1.5) / np.absolute(2 * fit_m[0])))
        t._fit = ("pending", None, lf, rf)
        try:
            t.get_curve_radius()
        except AttributeError:      # within 1e-8 of an integer: the lane pixels are refitted (needs a device), not this test's case
            continue
        done += 1
        assert [t.left_curve_radius, t.right_curve_radius] == want, (i, lf, rf)
    assert done > 3000


def test_record_success_keeps_the_points_of_get_poly_points():
    """_record_success takes the plot points of the averaged parabolas from lt_poly_points (packed, ready for the overlay);
    the public attributes must be what get_poly_points returns for them."""
    t = _bare_tracker((1280, 720))
    rng = np.random.default_rng(12)
    for i in range(300):
        lf = np.array([rng.uniform(-6e-4, 6e-4), rng.uniform(-1.2, 1.2), rng.uniform(-200, 1300)])
        rf = lf + np.array([rng.uniform(-1e-4, 1e-4), rng.uniform(-0.2, 0.2), rng.uniform(-60, 400)])
        partial = (1, 1.0, 0.5, 0.3)[i % 4]
        t._fit = ("pending", None, lf, rf)
        try:
            t._record_success(lf, rf, partial)
        except IndexError:          # no plot point inside the image: get_eccentricity fails upstream as well (:553)
            pass
        want = t.get_poly_points(t.left_avg_coeffs, t.right_avg_coeffs, partial)
        got = (t.left_avg_y, t.left_avg_x, t.right_avg_y, t.right_avg_x)
        assert all(np.array_equal(g, w) and g.dtype == w.dtype for g, w in zip(got, want)), (i, partial)
        b = t._avg_packed[0]
        assert (int(b[1][0]), int(b[1][1])) == (len(want[0]), len(want[2]))
        assert np.array_equal(b[2][:len(want[0])], np.stack([want[0], want[1]], 1))
        assert np.array_equal(b[3][:len(want[2])], np.stack([want[2], want[3]], 1))


def test_frame_tail_is_the_python_functions():
    """lt_frame_tail (one host call between a valid first try's record and its text lines) against the functions it stands for:
    check_validity, the running average (_averages_with), get_poly_points of the average (_points_packed), get_curve_radius on
    the pixel fit, get_eccentricity -- bit for bit, on and around the validity limits, with custom limits, for lanes that are nearly
    straight, leave the image or fail; and the not-reproduced flag exactly where the Python path does something else (a radius
    next to an integer, no plot point inside the image)."""
    import math
    from lane_tracker_amd import _native
    from lane_tracker_amd.lane_tracker import _mean_of_rows
    fn = _native.load().lt_frame_tail
    rng = np.random.default_rng(2024)
    seen = {"valid": 0, "invalid": 0, "flag": 0, "integer": 0}
    for i in range(6000):
        size = [(1080, 1100), (1280, 720), (640, 480)][i % 3]
        t = _bare_tracker(size)
        W, H = size
        if i % 5 == 0:
            t.validity_limits = dict(min_dist_y1=rng.uniform(50, 160), max_dist_y1=rng.uniform(200, 600), min_dist_y2=rng.uniform(40, 120),
                                     max_dist_y2=rng.uniform(200, 600), min_dist_y3=rng.uniform(30, 90), max_dist_y3=rng.uniform(180, 600),
                                     thresh=rng.uniform(0.1, 0.6))
        partial = (1.0, 1, 0.5, 0.3)[(i // 3) % 4]
        scale = 10.0 ** rng.uniform(-7.5, -3.3)
        lf = np.array([rng.choice([-1, 1]) * scale, rng.uniform(-0.3, 0.3), rng.uniform(0.2, 0.45) * W])
        sep = rng.choice([80.0, 110.0, 150.0, 200.0, 230.0, rng.uniform(60, 260), 190.0, 175.0])
        d_slope = rng.choice([0.25, -0.25, 0.2499999999999, rng.uniform(-0.3, 0.3)]) if i % 4 == 0 else rng.uniform(-0.04, 0.04)
        rf = lf + np.array([rng.uniform(-1, 1) * scale * 0.1, d_slope, sep])
        if i % 37 == 0:
            lf[2] -= 3 * W                                               # both curves outside the image: no plot point
            rf[2] -= 3 * W
        if i % 41 == 0:                                                  # a radius that IS an integer: upstream's refit decides
            a_m = lambda c: c[0] * t.mpph / (t.mppv ** 2)
            lf[1] = 0.0 - 2 * lf[0] * H                                  # tangent 0 at y_eval: radius = 1 / |2 a_m|
            lf[0] = np.sign(lf[0]) * t.mppv ** 2 / (2.0 * t.mpph * float(rng.integers(50, 5000)))
            lf[1] = -2 * lf[0] * H
        hist = [np.array([rng.uniform(-1e-4, 1e-4), rng.uniform(-0.2, 0.2), rng.uniform(300, 500)]) for _ in range(int(rng.integers(0, 3)))]
        hist_r = [h + np.array([0.0, 0.0, 180.0]) for h in hist]
        # ---- the Python functions ----
        t.valid_lane_lines = None
        t.check_validity(lf, rf)
        want_valid = t.valid_lane_lines
        la, ra = _mean_of_rows(hist + [lf]), _mean_of_rows(hist_r + [rf])
        wb = t._points_packed(la, ra, partial, 'want')
        want_flag, want_r = False, None
        if want_valid:
            if int(wb[1][0]) < 1 or int(wb[1][1]) < 1:
                want_flag = True
            else:
                t._fit = ("pending", None, lf, rf)
                t.average_curve_radii = []
                try:
                    t.get_curve_radius()
                    want_r = (t.left_curve_radius, t.right_curve_radius)
                except (AttributeError, OverflowError, ValueError):      # next to an integer: the pixel lists are asked for (none here)
                    want_flag = True
                    seen["integer"] += 1
                if not want_flag:
                    mid = int(W / 2)
                    want_ecc = (((mid - np.int64(wb[2][int(wb[1][0]) - 1, 1])) - (np.int64(wb[3][int(wb[1][1]) - 1, 1]) - mid)) / 2) * t.mpph
        # ---- the one call ----
        inp, out = np.zeros(24), np.full(8, -7.0)
        inp[0:3], inp[3:6] = lf, rf
        if hist:
            sl, sr = hist[0], hist_r[0]
            for h, g in zip(hist[1:], hist_r[1:]):
                sl, sr = sl + h, sr + g
            inp[6:9], inp[9:12] = sl, sr
        inp[12] = len(hist) + 1
        lim = t.validity_limits
        inp[13:22] = (lim['min_dist_y1'], lim['max_dist_y1'], lim['min_dist_y2'], lim['max_dist_y2'], lim['min_dist_y3'], lim['max_dist_y3'],
                      lim['thresh'], t.mppv, t.mpph)
        pv, pv2 = t._plot_rows(1)
        gb = t._packed_buffers(partial, 'got')
        pp, pp2 = gb[6]
        gb[1][:] = -1
        rc = fn(W, H, inp.ctypes.data, pv.ctypes.data, pv2.ctypes.data, len(pv), pp.ctypes.data, pp2.ctypes.data, len(pp), gb[0].ctypes.data,
                gb[1].ctypes.data, gb[1].ctypes.data + 4, gb[2].ctypes.data, gb[3].ctypes.data, out.ctypes.data)
        assert rc == 0
        if out[1]:
            assert want_valid and want_flag, (i, lf, rf)
            seen["flag"] += 1
            continue
        assert bool(out[0]) == bool(want_valid), (i, lf, rf)
        if not want_valid:
            seen["invalid"] += 1
            continue
        assert not want_flag, (i, lf, rf)
        seen["valid"] += 1
        assert gb[0][:3].tobytes() == la.tobytes() and gb[0][3:].tobytes() == ra.tobytes(), i
        nl, nr = int(wb[1][0]), int(wb[1][1])
        assert (int(gb[1][0]), int(gb[1][1])) == (nl, nr), i
        assert np.array_equal(gb[2][:nl], wb[2][:nl]) and np.array_equal(gb[3][:nr], wb[3][:nr]), i
        assert (int(out[2]), int(out[3])) == want_r, (i, lf, rf)
        assert np.float64(out[4]).tobytes() == np.float64(want_ecc).tobytes(), i
        assert math.isfinite(out[4])
    assert seen["valid"] > 1200 and seen["invalid"] > 500 and seen["flag"] > 20 and seen["integer"] > 5, seen
