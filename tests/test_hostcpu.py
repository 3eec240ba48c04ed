"""lane_tracker_amd/hostcpu.py: the CPUs the process may use, and the BLAS pool held to them (no GPU)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(code, **env):
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=dict(os.environ, **env), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-800:]
    return r.stdout.strip()


def test_usable_cpus_is_within_the_affinity_mask_and_the_quota():
    from lane_tracker_amd import hostcpu
    n = hostcpu.usable_cpus()
    assert 1 <= n <= len(os.sched_getaffinity(0))
    q = hostcpu.cpu_quota()
    assert q is None or n <= int(q + 0.999)


def test_blas_pool_is_limited_inside_the_block_only_and_never_raised():
    """the library limits BLAS threads around its own np.polyfit / lstsq calls and leaves the process's
    pool as the application set it."""
    code = ("import numpy as np, threadpoolctl, threading\n"
            "from lane_tracker_amd import hostcpu\n"
            "n = lambda: [p['num_threads'] for p in threadpoolctl.threadpool_info() if p['user_api'] == 'blas']\n"
            "b = n()\n"
            "with hostcpu.blas_limited():\n"
            "    i = n()\n"
            "    with hostcpu.blas_limited():\n"
            "        ii = n()\n"
            "    still = n()\n"
            "    np.polyfit(np.arange(50.0), np.arange(50.0) ** 2, 2)\n"
            "a = n()\n"
            "print(b, i, ii, still, a, hostcpu.usable_cpus())")
    try:
        import threadpoolctl  # noqa: F401
    except ImportError:
        import pytest
        pytest.skip("threadpoolctl is not installed")
    # a pool smaller than the CPUs: left alone, inside and outside
    out = _run(code, OPENBLAS_NUM_THREADS="1")
    assert out.startswith("[1] [1] [1] [1] [1]"), out
    # a pool larger than the CPUs the process may use (an affinity mask of one CPU): lowered inside the block (nested blocks
    # included), and back to the application's value behind it
    cpu = sorted(os.sched_getaffinity(0))[0]
    out = _run("import os; os.sched_setaffinity(0, {%d})\n" % cpu + code, OPENBLAS_NUM_THREADS="4")
    first = out.split("]")[0]
    assert first in ("[4", "[1"), out
    assert out == "%s] [1] [1] [1] %s] 1" % (first, first), out


def test_constructing_the_module_does_not_touch_the_pool():
    code = ("import numpy as np, threadpoolctl\n"
            "n = lambda: [p['num_threads'] for p in threadpoolctl.threadpool_info() if p['user_api'] == 'blas']\n"
            "b = n()\n"
            "import lane_tracker_amd.lane_tracker, lane_tracker_amd.hostcpu as h\n"
            "assert not hasattr(h, 'cap_blas_threads')\n"
            "print(b == n())")
    try:
        import threadpoolctl  # noqa: F401
    except ImportError:
        import pytest
        pytest.skip("threadpoolctl is not installed")
    cpu = sorted(os.sched_getaffinity(0))[0]
    assert _run("import os; os.sched_setaffinity(0, {%d})\n" % cpu + code, OPENBLAS_NUM_THREADS="4") == "True"
