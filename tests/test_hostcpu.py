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


def test_blas_pool_is_lowered_to_the_usable_cpus_and_never_raised():
    code = ("import numpy as np, threadpoolctl\n"
            "from lane_tracker_amd import hostcpu\n"
            "b = [p['num_threads'] for p in threadpoolctl.threadpool_info() if p['user_api'] == 'blas']\n"
            "got = hostcpu.cap_blas_threads()\n"
            "a = [p['num_threads'] for p in threadpoolctl.threadpool_info() if p['user_api'] == 'blas']\n"
            "print(b, got, a, hostcpu.usable_cpus())")
    try:
        import threadpoolctl  # noqa: F401
    except ImportError:
        import pytest
        pytest.skip("threadpoolctl is not installed")
    # a pool smaller than the CPUs: left alone
    out = _run(code, OPENBLAS_NUM_THREADS="1")
    assert out.startswith("[1] 1 [1]"), out
    # a pool larger than the CPUs the process may use (an affinity mask of one CPU): lowered
    cpu = sorted(os.sched_getaffinity(0))[0]
    out = _run("import os; os.sched_setaffinity(0, {%d})\n" % cpu + code, OPENBLAS_NUM_THREADS="4")
    assert out.endswith("[1] 1") and out.split("]")[0] in ("[4", "[1"), out
    # the switch
    out = _run(code, OPENBLAS_NUM_THREADS="4", LT_BLAS_CAP="0")
    assert " None " in out, out
