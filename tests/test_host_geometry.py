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
