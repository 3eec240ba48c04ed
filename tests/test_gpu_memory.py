"""The device memory cache (csrc/lt_memory.cpp: DevCache; lt_device_cache_trim).  Blocks a context gives up are kept and reused,
because memory that goes back to the driver is wiped in the background on an SDMA engine and the process's device-to-host copies
run at half speed meanwhile.  Checked here: a closed context's memory stays with the process and serves the
next context of the same shape; results do not depend on what a reused block held before; the trim returns it to the driver."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat():
    from lane_tracker_amd import _native
    _native.load()
    return _native


@pytest.fixture(scope="module")
def cal():
    from lane_tracker_amd import calib
    return calib.reference_calibration()


def _free_bytes():
    hip = ctypes.CDLL("libamdhip64.so")
    free_b, total_b = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(free_b), ctypes.byref(total_b)) == 0
    return free_b.value


def test_closed_contexts_leave_their_memory_in_the_cache_and_the_next_one_reuses_it(nat, cal, oracle, ref_calib):
    from lane_tracker_amd import synth
    nat.device_cache_trim(0)
    base = _free_bytes()
    frames = np.stack([synth.SceneRenderer(cal).render(40 + i)[0] for i in range(3)], 0)
    want = [oracle.mask_from_frame(ref_calib, f) for f in frames]

    def run():
        c = nat.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0],
                        device=0, capacity=96)
        try:
            used = base - _free_bytes()
            c.upload_frames(frames)
            c.mask_run(3)
            c.sws_fit_run(3)
            masks, rec = c.download_masks(3), c.download_records(3)
            for k in range(3):
                assert np.array_equal(masks[k], want[k]), k
            return used, rec
        finally:
            c.close()
    used1, rec1 = run()
    kept = base - _free_bytes()
    assert used1 > 500e6 and kept > 0.9 * used1              # the closed context's blocks are still the process's
    used2, rec2 = run()                                      # same shape: served from the cache (dirty blocks, same results)
    assert used2 <= used1 * 1.15                             # (what the first run allocated lazily is in the cache as well)
    assert rec1.tobytes() == rec2.tobytes()
    nat.device_cache_trim(0)
    assert base - _free_bytes() < 0.25 * used1               # ... until they are handed back (the runtime keeps some of its own)


def test_cache_can_be_switched_off():
    """LT_DEVICE_CACHE_GB=0 (read once per process): every block goes straight back to the driver."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, ctypes; sys.path.insert(0, %r)\n"
            "from lane_tracker_amd import _native, calib\n"
            "hip = ctypes.CDLL('libamdhip64.so')\n"
            "def free():\n"
            "    a, b = ctypes.c_size_t(), ctypes.c_size_t(); hip.hipMemGetInfo(ctypes.byref(a), ctypes.byref(b)); return a.value\n"
            "cal = calib.reference_calibration()\n"
            "c = _native.Context(cal['img_size'], cal['warped_size'], cal['cam_matrix'], cal['dist_coeffs'], cal['warp_matrices'][0], device=0, capacity=64)\n"
            "before = free(); c.close(); print(free() - before)\n" % root)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LT_DEVICE_CACHE_GB="0"), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-1500:]
    assert int(r.stdout.strip().splitlines()[-1]) > 300e6    # closing gave the memory back at once
