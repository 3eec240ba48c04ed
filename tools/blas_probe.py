#!/usr/bin/env python3
"""What does one np.polyfit over a lane's pixels cost the PROCESS under a cgroup CPU quota, by how NumPy's BLAS pool is held?
(get_curve_radius refits a lane's pixels as the reference does, a few times per second of video; NOTES_r05 D.9, NOTES_r06 E.3.)
Each mode in a child process: 150 refits of 13 k points, 10 ms apart, beside a thread that spins (the driving thread's stand-in);
per mode: per-call times, /sys/fs/cgroup/cpu.stat nr_throttled / throttled_usec, CPU time used.
    python tools/blas_probe.py            (modes: none, global16, global1, scoped)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cpu_stat():
    out = {}
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, v = line.split()
            out[k] = int(v)
    except OSError:
        pass
    return out


def child(mode):
    sys.path.insert(0, ROOT)
    import numpy as np
    from lane_tracker_amd import hostcpu
    import threadpoolctl
    info = [p.get("num_threads") for p in threadpoolctl.threadpool_info() if p.get("user_api") == "blas"]
    if mode == "global16":
        threadpoolctl.threadpool_limits(limits=hostcpu.usable_cpus(), user_api="blas")
    elif mode == "global1":
        threadpoolctl.threadpool_limits(limits=1, user_api="blas")
    rng = np.random.default_rng(0)
    y = rng.integers(0, 1100, 13000).astype(np.float64) * 0.03
    x = (0.3 * y * y + 2.0 * y + 400 + rng.normal(0, 2, y.size)) * 0.01
    import contextlib
    scope = hostcpu.blas_limited if mode == "scoped" else contextlib.nullcontext
    for _ in range(3):
        with scope():
            np.polyfit(y, x, 2)
    s0, c0, t0 = cpu_stat(), time.process_time(), time.perf_counter()
    ts, gaps = [], []
    for _ in range(150):
        a = time.perf_counter()
        with scope():
            np.polyfit(y, x, 2)
        b = time.perf_counter()
        ts.append(b - a)
        # the driving thread's stand-in: 10 ms of small steps; a frozen process shows as a long step
        while time.perf_counter() - b < 0.010:
            q = time.perf_counter()
            for _ in range(200):
                pass
            gaps.append(time.perf_counter() - q)
    s1 = cpu_stat()
    ts.sort(); gaps.sort()
    print(json.dumps({"mode": mode, "blas_threads_at_import": info, "usable_cpus": hostcpu.usable_cpus(), "polyfit_ms_median": round(ts[len(ts) // 2] * 1e3, 3), "polyfit_ms_max": round(ts[-1] * 1e3, 2),
                      "longest_stall_of_the_other_work_ms": round(gaps[-1] * 1e3, 2), "cpu_seconds": round(time.process_time() - c0, 2), "wall_seconds": round(time.perf_counter() - t0, 2),
                      "nr_throttled": s1.get("nr_throttled", 0) - s0.get("nr_throttled", 0), "throttled_ms": (s1.get("throttled_usec", 0) - s0.get("throttled_usec", 0)) // 1000}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(sys.argv[2])
    else:
        for mode in (sys.argv[1:] or ["none", "global16", "global1", "scoped", "none", "scoped"]):
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode])
