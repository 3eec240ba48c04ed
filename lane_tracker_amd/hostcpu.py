"""How many CPUs this process may really use, and NumPy's BLAS pool held to that.

A container on a large host sees every CPU of the machine (the MI355X boxes: 256) and is granted a few by its cgroup (16).
OpenBLAS starts one thread per VISIBLE CPU.  `np.polyfit` over a lane's pixels -- `LaneTracker.get_curve_radius` refits them
as the reference does (`lane_tracker.py:538-546`) when a radius lies next to an integer, a few times per second of video --
then runs on all of them for a millisecond, the cgroup's CPU quota for the period is gone, and the kernel freezes EVERY thread
of the process until the next period: 40-80 ms per event, `cpu.stat: nr_throttled`, 7-15 % of `process()`'s time at
1920x1080 (`tools/process_throttle_probe.py`, profiles/NOTES_r05.md D.9).  `cap_blas_threads()` lowers the pool to the CPUs the
process may use; it never raises it.  `LT_BLAS_CAP=0` leaves the pool alone."""
import ctypes
import os


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


_capped = None


def cap_blas_threads():
    """-> the BLAS thread count in force afterwards (None: no BLAS pool found, or LT_BLAS_CAP=0).  Idempotent."""
    global _capped
    if _capped is not None or os.environ.get("LT_BLAS_CAP") == "0":
        return _capped
    want = usable_cpus()
    try:
        import threadpoolctl
        pools = [p for p in threadpoolctl.threadpool_info() if p.get("user_api") == "blas"]
        if not pools:
            return None
        have = max(p.get("num_threads", 1) for p in pools)
        if have > want:
            threadpoolctl.threadpool_limits(limits=want, user_api="blas")      # (not used as a context manager: stays in force)
            have = want
        _capped = have
        return _capped
    except Exception:
        pass
    try:        # without threadpoolctl: OpenBLAS by name, in the libraries NumPy has loaded
        for line in open("/proc/self/maps"):
            path = line.rsplit(None, 1)[-1]
            if "openblas" in os.path.basename(path).lower():
                lib = ctypes.CDLL(path)
                for suffix in ("", "64_", "_64"):
                    get = getattr(lib, "openblas_get_num_threads" + suffix, None)
                    put = getattr(lib, "openblas_set_num_threads" + suffix, None)
                    if get and put:
                        have = int(get())
                        if have > want:
                            put(int(want))
                            have = want
                        _capped = have
                        return _capped
    except Exception:
        pass
    return None
