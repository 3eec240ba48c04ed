#!/usr/bin/env python3
"""Does process()'s frames/s depend on where the host thread runs?  (2x spread "box to box".)
For every NUMA node of the host, and unpinned: tools/process_trace.py (720p and 1080p) in fresh processes with the CPU affinity
set to the node's CPUs (taskset, before anything touches the GPU); prints the topology the GPU reports and one JSON line per run."""
import glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
nodes = {}
for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    try:
        nodes[int(d.rsplit("node", 1)[1])] = open(os.path.join(d, "cpulist")).read().strip()
    except Exception:
        pass
gpu_nodes = {}
for d in glob.glob("/sys/class/drm/card*/device"):
    try:
        vendor = open(os.path.join(d, "vendor")).read().strip()
        if vendor == "0x1002":
            gpu_nodes[d] = open(os.path.join(d, "numa_node")).read().strip()
    except Exception:
        pass
print(json.dumps({"numa_nodes": nodes, "gpu_numa_node": gpu_nodes, "affinity_now": len(os.sched_getaffinity(0))}))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
for size in ([], ["x"]):
    for name, cpus in [("unpinned", None)] + [("node%d" % k, v) for k, v in nodes.items()]:
        for r in range(reps):
            cmd = [sys.executable, os.path.join(ROOT, "tools", "process_trace.py")] + size
            if cpus:
                cmd = ["taskset", "-c", cpus] + cmd
            try:
                out = subprocess.run(cmd, capture_output=True, text=True, timeout=120).stdout.strip().split("\n")[-1]
                d = json.loads(out)
                print(json.dumps({"size": "1080p" if size else "720p", "where": name, "fps": d["fps"], "us_per_frame": d["us_per_frame"],
                                  "upload_frame_rows": d["us_per_frame_by_call"].get("upload_frame_rows"),
                                  "download_record": d["us_per_frame_by_call"].get("download_record"),
                                  "_present": d["us_per_frame_by_call"].get("_present")}), flush=True)
            except Exception as e:
                print(json.dumps({"size": "1080p" if size else "720p", "where": name, "error": repr(e)}), flush=True)
