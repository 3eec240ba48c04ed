#!/usr/bin/env python3
"""The driver's multi-GPU launch line -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port P bench.py --gpus N ... -- with four rank processes on GPU 0 over the test-only librccl stand-in: the launcher
form of the N > 1 path (RANK / LOCAL_RANK / WORLD_SIZE from the environment).  Code path only, not a throughput figure."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(root, "tests"))
import fake_rccl
cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1", "--master-port", "29533",
       os.path.join(root, "bench.py"), "--gpus", "4", "--batch", "64", "--steps", "3", "--warmup", "1"]
r = subprocess.run(cmd, cwd=root, env=fake_rccl.env(), capture_output=True, text=True, timeout=1200)
print("rc", r.returncode)
lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
if r.returncode or len(lines) != 1:
    print(r.stdout[-1500:], r.stderr[-3000:])
    sys.exit(1)
d = json.loads(lines[0])
print({k: d[k] for k in ("n_gpus", "value", "scaling", "ms_per_step")}, {k: d["config"].get(k) for k in ("parallelism", "gathered_records_checked", "ranks_share_devices")})
