"""The sharded path on the GPU: torch.distributed.run with the "nccl" (= RCCL) backend.  The GPU box of the test
tier has one GPU, so this is a world of one rank -- it exercises the device-side record gather and the
process_shard batching; the two-rank partition logic is covered on CPU (tests/test_distributed_cpu.py)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_sharded_run_over_rccl():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "tests", "dist_gpu_worker.py")], cwd=root, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "distributed gpu ok: 40 frames on 1 rank(s)" in r.stdout
