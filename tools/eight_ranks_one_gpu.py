#!/usr/bin/env python3
"""bench.py --gpus 8 with all eight rank processes on GPU 0 over the test-only librccl stand-in (tests/fake_rccl.c): the N = 8
code of the launcher, the rendezvous and the rank-major gather, in both scaling forms.  Says nothing about xGMI or throughput."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "tests"))
import fake_rccl
for args in (["--batch", "32", "--steps", "2", "--warmup", "1"], ["--frames", "4096", "--steps", "2", "--warmup", "1"]):
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8"] + args, cwd=root, env=fake_rccl.env(),
                       capture_output=True, text=True, timeout=1200)
    print("rc", r.returncode)
    if r.returncode:
        print(r.stdout[-1500:], r.stderr[-3000:])
        sys.exit(1)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print({k: d[k] for k in ("n_gpus", "value", "scaling", "ms_per_step")}, {k: d["config"].get(k) for k in ("parallelism", "collective", "gathered_records_checked", "ranks_share_devices", "frames_per_step_all_gpus")})
