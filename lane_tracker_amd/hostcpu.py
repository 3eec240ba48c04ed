"""How many CPUs this process may really use, and NumPy's BLAS pool held to that around the library's OWN LAPACK calls.

A container on a large host sees every CPU of the machine (the MI355X boxes: 256) and is granted a few by its cgroup (16).
OpenBLAS starts one thread per VISIBLE CPU.  `np.polyfit` over a lane's pixels -- `LaneTracker.get_curve_radius` refits them
as the reference does (`lane_tracker.py:538-546`) when a radius lies next to an integer, a few times per second of video --
then runs on all of them for a millisecond, the cgroup's CPU quota for the period is gone, and the kernel freezes EVERY thread
of the process until the next period: 40-80 ms per event, `cpu.stat: nr_throttled`, 7-15 % of `process()`'s time at
1920x1080 (`tools/process_throttle_probe.py`, profiles/NOTES_r05.md D.9).

`blas_limited()` is a context manager for exactly those calls: inside it the pool runs on ONE thread -- a lane's pixels are a
13 k x 3 least-squares problem, and OpenBLAS helper threads keep polling for ~0.1 s after a job: 16 of them, the whole quota of a
period, per refit -- and on the way out the pool is what the application had set.  (Round 5 lowered the pool once, for good, from
`LaneTracker.__init__`: a library has no business with its host's global state.)"""
import contextlib
import ctypes
import os
import threading


def cpu_quota():
    """CPUs' worth of time per period the cgroup grants this process (None: unlimited)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return q / per if q > 0 else None
    except Exception:
        return None


def usable_cpus():
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    q = cpu_quota()
    return max(1, min(avail, int(q + 0.999))) if q else avail


_pools = None                    # [(get_num_threads, set_num_threads)] of the BLAS libraries NumPy has loaded, found once
_lock = threading.Lock()
_depth = 0                       # blocks of blas_limited() under way (all threads)
_saved = None                    # the pools' sizes as the first of them found them


def _blas_pools():
    global _pools
    if _pools is not None:
        return _pools
    found = []
    try:
        import threadpoolctl
        for lc in threadpoolctl.ThreadpoolController().lib_controllers:
            if getattr(lc, "user_api", None) == "blas":
                found.append((lc.get_num_threads, lc.set_num_threads))
    except Exception:
        found = []
    if not found:
        try:        # without threadpoolctl: OpenBLAS by name, in the libraries NumPy has loaded
            for line in open("/proc/self/maps"):
                path = line.rsplit(None, 1)[-1]
                if "openblas" in os.path.basename(path).lower():
                    lib = ctypes.CDLL(path)
                    for suffix in ("", "64_", "_64"):
                        get = getattr(lib, "openblas_get_num_threads" + suffix, None)
                        put = getattr(lib, "openblas_set_num_threads" + suffix, None)
                        if get and put:
                            found.append((lambda g=get: int(g()), lambda n, p=put: p(int(n))))
                            break
                    break
        except Exception:
            found = []
    _pools = found
    return found


_warmed = False


def find_blas_pools():
    """Set-up work that must not fall into a frame: (1) the one-time search for the BLAS libraries NumPy has loaded
    (threadpoolctl walks every shared object of the process: ~0.1 s with the HIP runtime loaded); (2) the BLAS pool's own
    start-up.  OpenBLAS creates its helper threads at the first call that touches the pool -- `openblas_set_num_threads`
    included, i.e. the first `blas_limited()` -- and every new thread polls for work for a while before it goes to sleep: 63 of
    them against a cgroup quota of 16 CPUs are one freeze of 40-80 ms for the whole process (tools/process_throttle_probe.py
    names the threads that burned the quota: NOTES_r06 E.3).  `LaneTracker.__init__` calls this: the pool is started here, and
    where it is larger than the CPUs the process may use this call waits the 0.15 s its threads need to fall asleep -- once
    per process, while a tracker is being built, instead of inside one of the first frames.  The pool's size is the application's
    before and after."""
    global _warmed
    n = len(_blas_pools())
    if n and not _warmed:
        _warmed = True
        try:
            import time
            import numpy as np
            size = max(int(get() or 1) for get, _ in _blas_pools())
            with blas_limited():
                np.polyfit(np.arange(8.0), np.arange(8.0) ** 2, 2)
            if size > usable_cpus():
                time.sleep(0.15)
        except Exception:
            pass
    return n


@contextlib.contextmanager
def blas_limited():
    """Inside the block NumPy's BLAS pool runs on one thread; afterwards it is what it was.  Blocks may nest and run on several threads at once: the first one in lowers the pool, the last one out
    restores it.  Without a BLAS pool to be found the block runs as it is."""
    global _depth, _saved
    pools = _blas_pools()
    if pools:
        with _lock:
            if _depth == 0:
                want = 1
                _saved = []
                for get, put in pools:
                    try:
                        have = int(get() or 0)
                        if have > want:
                            put(want)
                            _saved.append((put, have))
                    except Exception:
                        pass
            _depth += 1
    try:
        yield
    finally:
        if pools:
            with _lock:
                _depth -= 1
                if _depth == 0 and _saved:
                    for put, have in _saved:
                        try:
                            put(have)
                        except Exception:
                            pass
                    _saved = None
