"""The sharded path on the GPU: one process per GPU, records gathered by `lt_gather_*` (RCCL) through the C ABI.
The GPU box of the test tier has one GPU, so the RCCL world there is one rank -- it exercises the communicator
set-up through the id file, the device-side record staging and the gather itself; a two-rank run is added when
two GPUs are visible.  The two-rank partition / padding / re-assembly logic is covered on CPU
(tests/test_distributed_cpu.py).  Launch forms: `spawn_ranks` and torch.distributed.run (which only provides the
environment; the worker never imports torch)."""
import os
import socket
import subprocess
import sys

import pytest

from lane_tracker_amd import distributed

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dist_gpu_worker.py")


def _run_spawned(world, tmp_path):
    # spawn_ranks lets rank 0 inherit stdout: run it in a child interpreter so that the output can be captured
    code = ("import sys; sys.path.insert(0, %r); from lane_tracker_amd import distributed as d; "
            "sys.exit(d.spawn_ranks(%d, [sys.executable, %r], timeout=900))" % (ROOT, world, WORKER))
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=1000)


def test_sharded_run_one_rank_over_rccl(tmp_path):
    r = _run_spawned(1, tmp_path)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "distributed gpu ok: 40 frames on 1 rank(s), shards [40], torch loaded: False" in r.stdout


def test_sharded_run_under_torch_distributed_run():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("LT_GATHER_ID", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), WORKER], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "distributed gpu ok: 40 frames on 1 rank(s)" in r.stdout


def test_sharded_run_two_ranks_when_two_gpus_are_visible(tmp_path):
    if distributed.visible_gpu_count() < 2:
        pytest.skip("one GPU visible: the two-rank RCCL run needs two")
    r = _run_spawned(2, tmp_path)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "distributed gpu ok: 40 frames on 2 rank(s)" in r.stdout


def test_more_ranks_than_gpus_fails_loudly(tmp_path):
    n = distributed.visible_gpu_count()
    r = _run_spawned(n + 1, tmp_path)
    assert r.returncode != 0
