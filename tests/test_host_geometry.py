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
