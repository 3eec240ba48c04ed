#!/usr/bin/env python3
"""Same-box A/B of the annotated stream's later pass under environment settings: each setting in a child process of its own,
round robin, `--repeat` times.   python tools/annot_ab.py 1280x720 "" "LT_COPY_SPINNERS=1" "ATTR_strip_piece=256" ...
(LT_* names: the library's switches -- measurement switches need the experiments build, LANE_TRACKER_AMD_LIB=lane_tracker_amd/liblane_tracker_amd_exp.so;
ATTR_<name>=<int>: a class attribute of LaneTracker, e.g. ATTR_host_text=0, ATTR_stream_lane_on_device=0)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(size, mode):
    node = os.environ.get("AB_NUMA_NODE")          # pin this process (and every thread it starts) to one NUMA node's CPUs
    if node:
        cpus = set()
        for part in open("/sys/devices/system/node/node%s/cpulist" % node).read().strip().split(","):
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
        os.sched_setaffinity(0, cpus & os.sched_getaffinity(0))
    import bench
    from lane_tracker_amd import calib, _native
    from lane_tracker_amd.lane_tracker import LaneTracker
    ann = {"True": True, "False": False}.get(mode, mode)
    base = bench.render_streams(96)[size]
    cal = calib.reference_calibration() if size == "1280x720" else calib.scaled_calibration(1.5)
    wins = bench.stream_windows(base, 256, 8)
    work = [w.copy() for w in wins] if ann == "inplace" else wins
    for k, v in os.environ.items():           # ATTR_<name>=<int>: the tracker's Python-level alternatives are class attributes
        if k.startswith("ATTR_"):
            cur = getattr(LaneTracker, k[5:])
            setattr(LaneTracker, k[5:], type(cur)(int(v)))
    lt = LaneTracker(**cal)

    def run():
        t0 = time.perf_counter()
        n = 0
        for out in lt.process_stream(work, annotate=ann):
            n += len(out)
        dt = time.perf_counter() - t0
        if ann == "inplace":
            for w, c0 in zip(work, wins):
                w[...] = c0
        return n / dt
    run()
    c0 = _native.host_copy_stats()
    t0 = time.perf_counter()
    rates = [run() for _ in range(3)]
    c1 = _native.host_copy_stats()
    print("RESULT " + json.dumps({"fps_best": round(max(rates)), "fps_all": [round(r) for r in rates], "copy_threads": c1["threads"]}))
    lt.close()


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(sys.argv[2], sys.argv[3])
        sys.exit(0)
    size, mode = sys.argv[1], sys.argv[2]
    settings = sys.argv[3:]
    for rep in range(2):
        for st in settings:
            env = dict(os.environ)
            for kv in st.split():
                k, v = kv.split("=")
                env[k] = v
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", size, mode], env=env, capture_output=True, text=True, timeout=600)
            got = [l for l in p.stdout.split("\n") if l.startswith("RESULT ")]
            print(size, mode, repr(st), got[0] if got else "FAILED " + p.stderr[-300:], flush=True)
