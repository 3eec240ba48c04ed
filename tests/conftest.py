import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    if os.environ.get("LT_TEST_SEARCH_CUS"):     # a sub-run of the suite with more CUs set aside (tests/test_gpu_streams.py)
        from lane_tracker_amd.lane_tracker import LaneTracker
        LaneTracker.search_cus = int(os.environ["LT_TEST_SEARCH_CUS"])


def has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def ref_calib(oracle):
    from lane_tracker_amd import calib
    c = calib.reference_calibration()
    return oracle.make_calib(c["img_size"], c["warped_size"], c["cam_matrix"], c["dist_coeffs"],
                             c["warp_matrices"][0])
