"""BASELINE config 4 at its size (4096-frame synthetic stream, sharded, RCCL gather of the records) and the N > 1 code
of lt_gather.cpp, on the one-GPU test box:

  * one rank over the REAL librccl: 4096 frames in chunks of 512, one gather, bitwise vs a plain run + oracle samples;
  * two rank processes on GPU 0 over tests/fake_rccl.c (LT_RCCL_LIB; the real RCCL refuses two ranks on one device):
    the id-file wait loop, ncclCommInitRank with world 2, the rank-major receive layout, uneven shards (5 + 6 frames),
    the count-agreement check, lt_gather_host / lt_gather_barrier, config 4 split 2048 + 2048, and bench.py --gpus 2.
The reference has no counterpart (process_video.py:41-44 is one process)."""
import json
import os
import subprocess
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fake_rccl  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W4 = os.path.join(ROOT, "tests", "dist_config4_worker.py")
WORKER = os.path.join(ROOT, "tests", "dist_gpu_worker.py")


def _spawn(world, worker, env, timeout=1500):
    code = ("import sys; sys.path.insert(0, %r); from lane_tracker_amd import distributed as d; "
            "sys.exit(d.spawn_ranks(%d, [sys.executable, %r], timeout=%d))" % (ROOT, world, worker, timeout))
    return subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout + 60)


def test_config4_4096_frames_one_rank_real_rccl():
    r = _spawn(1, W4, dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "config4 ok: 4096 frames on 1 rank(s) in chunks of 512" in r.stdout, r.stdout[-500:]
    print(r.stdout.strip().splitlines()[-1])


def test_config4_4096_frames_two_ranks_on_one_gpu_fake_rccl():
    r = _spawn(2, W4, fake_rccl.env(dict(os.environ, LT_FAKE_RCCL_TRACE="1")))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "config4 ok: 4096 frames on 2 rank(s) in chunks of 512, shards [2048, 2048]" in r.stdout, r.stdout[-500:]
    assert "[fake_rccl] rank 0/2 all-gather" in r.stderr        # the traffic really went through the stand-in


def test_two_ranks_uneven_shards_fake_rccl():
    r = _spawn(2, WORKER, fake_rccl.env(dict(os.environ, LT_TEST_FRAMES="11")), timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "distributed gpu ok: 11 frames on 2 rank(s), shards [5, 6]" in r.stdout


def test_two_ranks_disagreeing_record_counts_fail_instead_of_hanging():
    r = _spawn(2, WORKER, fake_rccl.env(dict(os.environ, LT_TEST_FRAMES="11", LT_TEST_MISMATCH="1")), timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "mismatch reported on rank 0: " in r.stdout and "rank 1 passes 4" in r.stdout, r.stdout[-800:]


def test_three_ranks_fake_rccl():
    r = _spawn(3, WORKER, fake_rccl.env(dict(os.environ, LT_TEST_FRAMES="10")), timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert "distributed gpu ok: 10 frames on 3 rank(s), shards [3, 3, 4]" in r.stdout


def test_bench_two_ranks_fake_rccl():
    """bench.py --gpus 2 --frames 512 (config 4's strong-scaling form) self-launches two ranks; the line is labelled as
    a shared-device run, and its gathered records passed bench.py's own rank-major check."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "512", "--steps", "3",
                        "--warmup", "1"], cwd=ROOT, env=fake_rccl.env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["frames_per_step_all_gpus"] == 512
    assert line["config"]["ranks_share_devices"] is True
    assert line["config"]["gathered_records_checked"] == 512


def test_bench_two_ranks_weak_fake_rccl():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "64", "--steps", "2",
                        "--warmup", "1"], cwd=ROOT, env=fake_rccl.env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["config"]["frames_per_step_all_gpus"] == 128


def test_the_drivers_launch_line_four_ranks_on_one_gpu_fake_rccl():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port P bench.py --gpus 4`:
    the launcher form the driver uses for SCALE_rNN.json (RANK / LOCAL_RANK / WORLD_SIZE from torch's agent), here with four
    rank processes sharing GPU 0 over the stand-in.  The environment deliberately does NOT carry HSA_ENABLE_IPC_MODE_LEGACY:
    the ranks must arrive at it by themselves (lane_tracker_amd.distributed.ensure_ipc_env), exactly as under the driver."""
    env = fake_rccl.env()
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "4", "--batch", "64", "--steps", "3", "--warmup", "1"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 4 and line["scaling"] == "weak" and line["config"]["frames_per_step_all_gpus"] == 256
    assert line["config"]["ranks_share_devices"] is True and line["config"]["gathered_records_checked"] == 256
    assert line["config"]["rank_environment"] == {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}


def test_bench_eight_ranks_on_one_gpu_fake_rccl():
    """`bench.py --gpus 8 --frames 4096` (BASELINE config 4 in its 8-way form: 512 frames per rank) with all eight rank processes
    on GPU 0 over the stand-in: the launcher, the rendezvous, the shard arithmetic and the rank-major gather at N = 8 (round 4 had
    this as a tool only).  Says nothing about xGMI or throughput: the ranks share one device and the line says so."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--frames", "4096", "--steps", "2", "--warmup", "1"],
                       cwd=ROOT, env=fake_rccl.env(), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 8 and line["scaling"] == "strong" and line["config"]["frames_per_step_all_gpus"] == 4096
    assert line["config"]["frames_per_gpu_per_step"] == 512
    assert line["config"]["ranks_share_devices"] is True and line["config"]["gathered_records_checked"] == 4096
