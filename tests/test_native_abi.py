"""CPU-side checks of the boundary: the shared library loads, exports every symbol that
include/lane_tracker_amd.h declares, the record layout is 64 bytes, and -- with no GPU -- the
product fails loudly instead of computing anything on the CPU."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "lane_tracker_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lt_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from lane_tracker_amd import _native
    lib = _native.load()
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert sorted(_native.exported_symbols()) == declared
    assert lib.lt_abi_version() == _native.ABI_VERSION == 5
    assert int(re.search(r"#define LT_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "lane_tracker_amd.h")).read()).group(1)) == 5


def test_library_exports_nothing_but_the_header():
    """Built with -fvisibility=hidden: the dynamic symbol table holds exactly the header's lt_* names (no mangled
    internals, no unprefixed helpers)."""
    import shutil
    import subprocess
    from lane_tracker_amd import _native
    nm = shutil.which("nm") or "/opt/rocm/lib/llvm/bin/llvm-nm"
    out = subprocess.run([nm, "-D", "--defined-only", _native.LIB_PATH], capture_output=True, text=True, check=True).stdout
    defined = sorted({line.split()[-1] for line in out.splitlines() if line.split() and line.split()[-2] in "TtWwBbDdRrVv"})
    # the HIP fat-binary registration objects are emitted by hipcc, not by this source tree
    ours = [n for n in defined if not n.startswith("__hip_")]
    assert ours == _declared_symbols(), sorted(set(ours) ^ set(_declared_symbols()))
    assert _native.load().lt_last_threshold_path(None) == -2**31           # LT_NO_CONTEXT, not the "none yet" -1


def test_struct_layouts():
    import ctypes as C
    from lane_tracker_amd import _native
    assert C.sizeof(_native.LaneRecord) == 64 and _native.RECORD_DTYPE.itemsize == 64
    assert C.sizeof(_native.Calib) == 16 + 8 * 23
    assert C.sizeof(_native.SearchParams) == 8 * 4 + 3 * 8
    assert C.sizeof(_native.FilterParams) == 9 * 4
    names = [_native.load().lt_stage_name(i).decode() for i in range(_native.NUM_STAGES)]
    assert names[0] == "undistort_rows" and names[9] == "sws_fit" and all(names)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_no_gpu_fails_loudly():
    from lane_tracker_amd import _native, calib
    from lane_tracker_amd.lane_tracker import LaneTracker, bilateral_adaptive_threshold
    with pytest.raises(_native.NativeError):
        LaneTracker(**calib.reference_calibration())
    with pytest.raises(_native.NativeError):
        bilateral_adaptive_threshold(np.zeros((8, 8), np.uint8))


def test_argument_validation_needs_no_gpu():
    from lane_tracker_amd.lane_tracker import bilateral_adaptive_threshold
    with pytest.raises(ValueError):
        bilateral_adaptive_threshold(np.zeros((8, 8), np.uint8), mode="round")   # reference :71


def test_dropin_modules_expose_reference_names():
    import importlib.util
    for mod, names in (("lane_tracker", ["LaneTracker", "bilateral_adaptive_threshold"]),
                       ("utils", ["load_camera_calib", "load_warp_params", "create_split_view"])):
        spec = importlib.util.spec_from_file_location("dropin_" + mod, os.path.join(ROOT, "dropin", mod + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        for n in names:
            assert hasattr(m, n)


def test_process_signature_matches_reference_defaults():
    """process() keyword names and defaults, lane_tracker.py:876-900."""
    import inspect
    from lane_tracker_amd.lane_tracker import LaneTracker
    sig = inspect.signature(LaneTracker.process)
    want = dict(ksize_r=15, C_r=8, ksize_b=35, C_b=5, filter_type='bilateral', mask_noise=False, noise_thresh=140,
                ksize_noise=65, C_noise=10, window_width=30, window_height=40, search_range=20, mu=0.1,
                no_success_limit=8, start_slice=0.25, ignore_sides=360, ignore_bottom=30, bandwidth=25, partial=1.0,
                n_tries=2, visualize_search=False, split_view=False, diagnostics=False)
    got = {k: v.default for k, v in sig.parameters.items() if k not in ("self", "img")}
    assert got == want and list(got) == list(want)
    ctor = inspect.signature(LaneTracker.__init__)
    assert list(ctor.parameters)[1:11] == ["img_size", "warped_size", "cam_matrix", "dist_coeffs", "warp_matrices",
                                           "mpp_conversion", "n_fail", "n_reset", "n_average", "print_frame_count"]


def test_utils_loaders_round_trip(tmp_path):
    from lane_tracker_amd import calib, utils
    cam, warp = str(tmp_path / "cam.npz"), str(tmp_path / "warp.npz")
    utils.save_calibration_npz(cam, warp, calib.CAM_MATRIX, calib.DIST_COEFFS, calib.M, calib.MINV,
                               calib.IMAGE_WIDTH_HEIGHT, calib.WARPED_WIDTH_HEIGHT, calib.MPPV, calib.MPPH)
    K, D = utils.load_camera_calib(cam)
    M, Minv, isz, wsz, mppv, mpph = utils.load_warp_params(warp)
    assert np.array_equal(K, calib.CAM_MATRIX) and np.array_equal(D, calib.DIST_COEFFS)
    assert isz == (1280, 720) and wsz == (1080, 1100) and (mppv, mpph) == (calib.MPPV, calib.MPPH)
    assert np.array_equal(M, calib.M) and np.array_equal(Minv, calib.MINV)


def test_calibration_literals_match_reference_pickles():
    ref = "/root/reference/cam_calib.p"
    if not os.path.exists(ref):
        pytest.skip("reference not present")
    import io
    import contextlib
    from lane_tracker_amd import calib, utils
    with contextlib.redirect_stdout(io.StringIO()):
        K, D = utils.load_camera_calib(ref)
        M, Minv, isz, wsz, mppv, mpph = utils.load_warp_params("/root/reference/warp_params.p")
    assert np.array_equal(K, calib.CAM_MATRIX) and np.array_equal(D, calib.DIST_COEFFS)
    assert np.array_equal(M, calib.M) and np.array_equal(Minv, calib.MINV)
    assert (isz, wsz, mppv, mpph) == (calib.IMAGE_WIDTH_HEIGHT, calib.WARPED_WIDTH_HEIGHT, calib.MPPV, calib.MPPH)


def test_header_is_plain_c_and_links(tmp_path):
    """include/lane_tracker_amd.h compiles as C99 (no torch / C++ types in the signatures) and a C program links
    against the shared library and can call the entry points that need no GPU."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("gcc not available")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "lane_tracker_amd")
    src = tmp_path / "abi.c"
    src.write_text('''
#include <stdio.h>
#include <string.h>
#include "lane_tracker_amd.h"
int main(void) {
    lt_calib cal;
    lt_ctx* ctx = 0;
    int16_t spans[8];
    int32_t left[4] = {0, 3, 1, 3}, right[4] = {0, 6, 1, 7};      /* (y, x) pairs */
    memset(&cal, 0, sizeof cal);
    if (lt_abi_version() != LT_ABI_VERSION) return 1;
    if (sizeof(lt_lane_record) != 64) return 2;
    if (lt_create(0, 0, &ctx) == 0) return 3;                      /* null calibration: an error code, no crash */
    if (lt_last_error()[0] == 0) return 4;
    if (lt_lane_polygon_spans(4, left, 2, right, 2, spans) != 0) return 5;
    if (spans[0] != 3 || spans[1] != 6 || spans[2] != 3 || spans[3] != 7) return 6;
    if (spans[4] <= spans[5]) return 7;                            /* untouched rows are empty intervals */
    {   /* lt_frame_tail (host only): a straight, centred lane 180 px wide is valid; its average with nothing is itself */
        double in[24], out[8], avg6[6], ploty[8], ploty2[8];
        int32_t ln = 0, rn = 0, lyx[16], ryx[16];
        int i;
        memset(in, 0, sizeof in);
        in[2] = 450.0; in[5] = 630.0;                              /* left x = 450, right x = 630 */
        in[12] = 1.0;                                              /* the average divides by one fit: this one */
        in[13] = 150; in[14] = 230; in[15] = 110; in[16] = 230; in[17] = 80; in[18] = 200; in[19] = 0.25;
        in[20] = 30.0 / 720.0; in[21] = 3.7 / 700.0;
        for (i = 0; i < 8; ++i) { ploty[i] = 1092.0 + i; ploty2[i] = ploty[i] * ploty[i]; }
        if (lt_frame_tail(1080, 1100, in, ploty, ploty2, 8, ploty, ploty2, 8, avg6, &ln, &rn, lyx, ryx, out) != 0) return 8;
        if (out[0] != 1.0 || ln != 8 || rn != 8 || avg6[2] != 450.0 || avg6[5] != 630.0 || lyx[1] != 450 || ryx[15] != 630) return 9;
        if (out[1] != 1.0) return 10;                              /* a dead-straight lane has no radius: "not reproduced here" */
        in[0] = 1e-4; in[3] = 1e-4;                                /* a curve: radii and eccentricity come back */
        in[2] = 450.0 - 1e-4 * 1099.0 * 1099.0; in[5] = 630.0 - 1e-4 * 1099.0 * 1099.0;
        if (lt_frame_tail(1080, 1100, in, ploty, ploty2, 8, ploty, ploty2, 8, avg6, &ln, &rn, lyx, ryx, out) != 0) return 11;
        if (out[0] != 1.0 || out[1] != 0.0 || !(out[2] > 0.0) || out[2] != out[3]) return 12;
    }
    printf("abi ok\\n");
    return 0;
}
''')
    exe = tmp_path / "abi"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(root, "include"), str(src), "-o", str(exe),
                           "-L", lib_dir, "-llane_tracker_amd", "-Wl,-rpath," + lib_dir])
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = lib_dir + os.pathsep + "/opt/rocm/lib" + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "abi ok" in r.stdout, (r.returncode, r.stdout, r.stderr)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="a GPU is present")
def test_pinned_empty_falls_back_to_pageable_memory_without_gpu():
    """lt_host_alloc needs a device; without one the pool hands out a plain array (same shape and dtype) and
    the library reports the failure instead of crashing."""
    import ctypes as C
    from lane_tracker_amd import _native
    a = _native.pinned_empty((4, 6, 3))
    assert a.shape == (4, 6, 3) and a.dtype == np.uint8 and a.flags.writeable
    out = C.c_void_p()
    assert _native.load().lt_host_alloc(64, C.byref(out)) != 0 and not out.value
    assert _native.load().lt_host_alloc(0, C.byref(out)) != 0
    assert _native.load().lt_host_free(None) == 0


def test_host_copy_thread_copies_and_waits():
    """lt_host_copy_async / lt_host_copy_wait need no GPU: pieces requested from two Python threads all arrive, the wait
    returns only when they have, and a wait with nothing pending returns at once."""
    import threading
    from lane_tracker_amd import _native
    lib = _native.load()
    assert lib.lt_host_copy_wait() == 0
    rng = np.random.default_rng(3)
    src = rng.integers(0, 256, (64, 1 << 16), dtype=np.uint8)
    dst = np.zeros_like(src)

    def feed(rows):
        for r in rows:
            assert lib.lt_host_copy_async(dst[r].ctypes.data, src[r].ctypes.data, src.shape[1]) == 0
    ts = [threading.Thread(target=feed, args=(range(k, 64, 2),)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert lib.lt_host_copy_wait() == 0
    assert np.array_equal(dst, src)
    assert lib.lt_host_copy_async(None, src.ctypes.data, 8) != 0          # a null pointer is refused, nothing is queued
    assert lib.lt_host_copy_async(dst.ctypes.data, src.ctypes.data, 0) == 0
    assert lib.lt_host_copy_wait() == 0


def test_host_copy_groups_complete_independently():
    """Completion per group (ABI 4): a short copy in one group is waited for while another group still has a long queue --
    two trackers on two threads do not wait for each other's copies -- and each group's data is complete after ITS wait."""
    import threading
    import time
    from lane_tracker_amd import _native
    lib = _native.load()
    ga, gb = _native.host_copy_group(), _native.host_copy_group()
    assert ga > 0 and gb > 0 and ga != gb
    rng = np.random.default_rng(5)
    big_src = rng.integers(0, 256, (96, 4 << 20), dtype=np.uint8)          # 384 MB in 96 pieces: tenths of a second of memcpy
    big_dst = np.zeros_like(big_src)
    small_src = rng.integers(0, 256, (1 << 12,), dtype=np.uint8)
    small_dst = np.zeros_like(small_src)
    t = {}

    def long_job():
        assert lib.lt_host_copy2d_async_group(gb, big_dst.ctypes.data, big_src.shape[1], big_src.ctypes.data, big_src.shape[1],
                                              big_src.shape[1], big_src.shape[0]) == 0
        t0 = time.perf_counter()
        assert lib.lt_host_copy_wait_group(gb) == 0
        t["long"] = time.perf_counter() - t0

    th = threading.Thread(target=long_job)
    th.start()
    time.sleep(0.01)                     # the long job is queued and under way
    t0 = time.perf_counter()
    assert lib.lt_host_copy_async_group(ga, small_dst.ctypes.data, small_src.ctypes.data, small_src.size) == 0
    assert lib.lt_host_copy_wait_group(ga) == 0
    t["short"] = time.perf_counter() - t0
    assert np.array_equal(small_dst, small_src)
    th.join()
    assert np.array_equal(big_dst, big_src)
    # the short wait did not sit out the long queue (pieces are taken in order, so it waits for at most the pieces in front of it
    # on the workers -- not for the group as a whole); generous bound for a loaded CI box
    assert t["short"] < max(0.5 * t["long"], 0.05), t
    assert lib.lt_host_copy_wait() == 0                                    # every group
    assert lib.lt_host_copy_async_group(12345678, small_dst.ctypes.data, small_src.ctypes.data, 8) != 0   # unknown group
    assert lib.lt_host_copy_wait_group(12345678) != 0
    _native.host_copy_group_release(ga)
    _native.host_copy_group_release(gb)
    assert lib.lt_host_copy_wait_group(ga) != 0                            # forgotten
    assert lib.lt_host_copy_group_destroy(0) == 0                          # the default group stays


def test_host_copy_threads_survive_shutdown_and_fork():
    """lt_shutdown joins the workers and the next request starts new ones; the child of a fork() in a process that had
    workers gets workers of its own (round 4: its first wait blocked for ever)."""
    import os
    from lane_tracker_amd import _native
    lib = _native.load()
    src = np.arange(1 << 16, dtype=np.uint8)
    dst = np.zeros_like(src)
    assert lib.lt_host_copy_async(dst.ctypes.data, src.ctypes.data, src.size) == 0
    assert lib.lt_shutdown() == 0
    assert np.array_equal(dst, src)      # shutdown finishes what was queued
    dst[:] = 0
    assert lib.lt_host_copy_async(dst.ctypes.data, src.ctypes.data, src.size) == 0
    assert lib.lt_host_copy_wait() == 0
    assert np.array_equal(dst, src)
    r, w = os.pipe()
    pid = os.fork()
    if pid == 0:                         # child: the parent's workers do not exist here
        try:
            os.close(r)
            d2 = np.zeros_like(src)
            import signal
            signal.alarm(20)             # a hang ends the child, the parent sees no "ok"
            ok = lib.lt_host_copy_async(d2.ctypes.data, src.ctypes.data, src.size) == 0 and lib.lt_host_copy_wait() == 0 \
                and np.array_equal(d2, src)
            os.write(w, b"ok" if ok else b"no")
        finally:
            os._exit(0)
    os.close(w)
    got = os.read(r, 2)
    os.close(r)
    os.waitpid(pid, 0)
    assert got == b"ok"


DEPLOYMENT_SWITCHES = {"LT_DEVICE_CACHE_GB", "LT_COPY_THREADS", "LT_COPY_SPINNERS", "LT_COPY_SPIN_US", "LT_STAGING_MB", "LT_TRACE_START",
                       "LT_TRACE_DESTROY", "LT_RCCL_LIB"}
PYTHON_SWITCHES = {"LT_GATHER_ID", "LT_DEVICE_MODULO"}                         # lane_tracker_amd/distributed.py (rank rendezvous, shared-GPU test runs)


def test_release_library_reads_the_deployment_switches_and_no_others():
    """measurement switches exist in the experiments build only (LT_EXP_ENV, csrc/lt_internal.h).  The release
    library's string table holds exactly the documented environment names, the Python package reads two more, and no product
    source carries a wrong-result probe."""
    import glob
    import subprocess
    from lane_tracker_amd import _native
    lib = os.path.join(ROOT, "lane_tracker_amd", "liblane_tracker_amd.so")
    out = subprocess.run(["strings", lib], capture_output=True, text=True, check=True).stdout
    names = {l.strip() for l in out.splitlines() if re.fullmatch(r"LT_[A-Z0-9_]+", l.strip())}
    assert names == DEPLOYMENT_SWITCHES, sorted(names ^ DEPLOYMENT_SWITCHES)
    assert len(names | PYTHON_SWITCHES) <= 12
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for n in DEPLOYMENT_SWITCHES | PYTHON_SWITCHES:
        assert n in doc, n + " is not documented in INTEGRATION.md"
    py = set()
    for f in glob.glob(os.path.join(ROOT, "lane_tracker_amd", "*.py")):
        py |= set(re.findall(r"environ(?:\.get)?\(?\[?\s*[\"'](LT_[A-Z0-9_]+)[\"']", open(f).read()))
    assert py == PYTHON_SWITCHES, sorted(py ^ PYTHON_SWITCHES)
    for f in glob.glob(os.path.join(ROOT, "lane_tracker_amd", "csrc", "*")):
        if f.endswith((".hip", ".cpp", ".h")):
            text = open(f).read()
            assert "LT_PROBE_" not in text, f + " carries a timing probe (they live in tools/probes/*.patch)"
            direct = set(re.findall(r"getenv\(\"(LT_[A-Z0-9_]+)\"\)", re.sub(r"LT_EXP_ENV\([^)]*\)", "", text)))
            assert direct <= DEPLOYMENT_SWITCHES, (f, sorted(direct - DEPLOYMENT_SWITCHES))
    # run_filter_chain reads no switch at all, in either build (its alternatives are gone or are arguments)
    api = open(os.path.join(ROOT, "lane_tracker_amd", "csrc", "lt_api.cpp")).read()
    a = api.index("int run_filter_chain(")
    body = api[a:api.index("\n}\n", a)]
    assert "getenv" not in body and "LT_EXP_ENV" not in body


def test_copy_groups_survive_a_fork_and_two_submitters_do_not_starve_each_other():
    """(1) the child of a fork() gets a fresh copier -- a completion group created before the fork must keep working
    there (its id is taken over on first use) and the child's own groups must not collide with it; (2) pieces submitted by two
    threads at once are all carried out promptly (the wake-up accounting counts what waits in the queue, not one submission)."""
    import subprocess
    import sys
    code = r'''
import ctypes as C, os, sys, threading, time
sys.path.insert(0, %r)
import numpy as np
from lane_tracker_amd import _native
lib = _native.load()
g = _native.host_copy_group()
src, dst = np.arange(1 << 20, dtype=np.uint8), np.zeros(1 << 20, np.uint8)
assert lib.lt_host_copy_async_group(g, dst.ctypes.data, src.ctypes.data, src.size) == 0 and lib.lt_host_copy_wait_group(g) == 0
pid = os.fork()
if pid == 0:
    d2 = np.zeros(1 << 20, np.uint8)
    ok = lib.lt_host_copy_async_group(g, d2.ctypes.data, src.ctypes.data, src.size) == 0 and lib.lt_host_copy_wait_group(g) == 0 and np.array_equal(d2, src)
    g2 = _native.host_copy_group()
    os._exit(0 if ok and g2 != g else 3)
_, status = os.waitpid(pid, 0)
assert os.WEXITSTATUS(status) == 0, status
# two submitters, one piece each, many times: every wait returns quickly
def worker(k, out):
    gg = _native.host_copy_group()
    d = np.zeros(1 << 16, np.uint8)
    worst = 0.0
    for _ in range(300):
        t0 = time.perf_counter()
        assert lib.lt_host_copy_async_group(gg, d.ctypes.data, src.ctypes.data, d.size) == 0 and lib.lt_host_copy_wait_group(gg) == 0
        worst = max(worst, time.perf_counter() - t0)
    out[k] = worst
    _native.host_copy_group_release(gg)
res = {}
ts = [threading.Thread(target=worker, args=(k, res)) for k in range(2)]
[t.start() for t in ts]; [t.join() for t in ts]
assert max(res.values()) < 0.25, res
lib.lt_shutdown()
print("ok")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-800:] + r.stderr[-1500:]
