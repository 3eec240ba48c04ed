"""The N > 1 path on CPU: world_size-2 gloo processes shard a block of frames, compute their lane
records and all-gather them; the gathered array must equal the single-process result bit for bit
(SURVEY.md section 8(e)).  Record computation here is the oracle (test infrastructure): the
sharding and the collective are what is under test -- the GPU compute is covered by -m gpu."""
import os
import socket

import numpy as np
import pytest

from lane_tracker_amd import _native, distributed


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 256, 4096, 4097):
        for world in (1, 2, 3, 8):
            spans = [distributed.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(distributed.shard_sizes(n, world)) - min(distributed.shard_sizes(n, world)) <= 1
    with pytest.raises(ValueError):
        distributed.shard_range(10, 2, 2)


def _records_for(indices):
    """Deterministic stand-in workload: seeded masks -> oracle search + fit -> lane records."""
    from lane_tracker_amd import synth
    from oracle import oracle as O
    rec = np.zeros(len(indices), _native.RECORD_DTYPE)
    for j, i in enumerate(indices):
        mask = synth.synth_mask(1000 + i, noise=1e-3)[0]
        r = O.sliding_window_search(mask)
        rec[j]["detected"] = r["detected"]
        rec[j]["n_left"], rec[j]["n_right"] = len(r["left_x"]), len(r["right_x"])
        if r["detected"]:
            rec[j]["left_coeffs"] = O.polyfit2(r["left_y"], r["left_x"])
            rec[j]["right_coeffs"] = O.polyfit2(r["right_y"], r["right_x"])
        rec[j]["frame"] = i
    return rec


def _worker(rank, world, port, n_frames, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = distributed.shard_range(n_frames, rank, world)
        local = _records_for(range(lo, hi))
        allrec = distributed.gather_records(local, n_frames, distributed.GlooTransport())
        np.save(os.path.join(out_dir, f"rank{rank}.npy"), allrec)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_frames", [5, 6])
def test_two_rank_gather_equals_single_process(tmp_path, n_frames):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, n_frames, str(tmp_path)), nprocs=2, join=True)
    want = _records_for(range(n_frames))
    for rank in range(2):
        got = np.load(os.path.join(str(tmp_path), f"rank{rank}.npy"))
        assert got.dtype == _native.RECORD_DTYPE and got.shape == want.shape
        assert got.tobytes() == want.tobytes()          # bitwise, frame order preserved
        assert got["frame"].tolist() == list(range(n_frames))


def test_spawn_ranks_environment_and_failure_propagation(tmp_path):
    """The launcher half of `bench.py --gpus N` without a GPU: every rank gets RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* and one shared rendezvous file name; a failing rank ends the job with a non-zero code instead of leaving
    the others waiting in a collective."""
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ok = tmp_path / "rank_ok.py"
    ok.write_text(textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        from lane_tracker_amd import distributed
        rank, local, world = distributed.env_rank()
        assert (rank, world) == (int(os.environ["RANK"]), 3) and local == rank
        assert os.environ["MASTER_ADDR"] == "127.0.0.1" and int(os.environ["MASTER_PORT"]) > 0
        open(os.path.join(%r, "rank%%d.txt" %% rank), "w").write(distributed.rendezvous_path())
    """ % (root, str(tmp_path))))
    code = ("import sys; sys.path.insert(0, %r); from lane_tracker_amd import distributed as d; "
            "sys.exit(d.spawn_ranks(3, [sys.executable, %r], timeout=120))" % (root, str(ok)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    paths = {(tmp_path / ("rank%d.txt" % i)).read_text() for i in range(3)}
    assert len(paths) == 1 and "lt_gather_" in paths.pop()
    bad = tmp_path / "rank_bad.py"
    bad.write_text("import os, sys, time\nif os.environ['RANK'] == '1': sys.exit(7)\ntime.sleep(60)\n")
    code = ("import sys; sys.path.insert(0, %r); from lane_tracker_amd import distributed as d; "
            "sys.exit(d.spawn_ranks(3, [sys.executable, %r], timeout=120))" % (root, str(bad)))
    t0 = __import__("time").time()
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 7 and __import__("time").time() - t0 < 30      # the sleeping ranks were ended, not waited for


def test_bench_refuses_a_rank_count_it_cannot_deliver():
    """`bench.py --gpus N` must never print an n_gpus: 1 line for N > 1: without GPUs it exits 2 with a message, and a
    launcher environment whose WORLD_SIZE differs from --gpus is refused too."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU is visible: covered by tests/test_gpu_distributed.py")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 2 and "refusing" in r.stderr and r.stdout.strip() == ""
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 2 and "refusing" in r.stderr and r.stdout.strip() == ""


def test_every_launch_form_puts_the_ipc_mode_into_a_rank_before_the_library_loads():
    """RCCL's intra-node transport needs HSA_ENABLE_IPC_MODE_LEGACY=0 on these hosts (distributed.ensure_ipc_env says why),
    and the runtime reads it once, at its initialisation.  A rank started the torch.distributed.run way -- RANK /
    LOCAL_RANK / WORLD_SIZE in the environment, nothing else -- has it after importing `lane_tracker_amd.distributed`, i.e.
    before `_native.load()` or any context exists; `spawn_ranks` hands it to its children; a caller's own value survives;
    a single-process run is left alone."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import os, sys; sys.path.insert(0, %r)\n"
            "before = os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')\n"
            "from lane_tracker_amd import distributed\n"
            "at_import = os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')\n"
            "from lane_tracker_amd import _native\n"
            "loaded = _native._LIB is not None if hasattr(_native, '_LIB') else None\n"
            "print(before, at_import, distributed.env_rank(), os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'))\n" % root)
    base = {k: v for k, v in os.environ.items() if k not in ("HSA_ENABLE_IPC_MODE_LEGACY", "RANK", "LOCAL_RANK", "WORLD_SIZE")}

    def run(extra):
        r = subprocess.run([sys.executable, "-c", code], env=dict(base, **extra), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-1500:]
        return r.stdout.strip().splitlines()[-1]

    assert run(dict(RANK="1", LOCAL_RANK="1", WORLD_SIZE="4")) == "None 0 (1, 1, 4) 0"          # the driver's launch line
    assert run(dict(RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="1")) == "1 1 (0, 0, 2) 1"
    assert run({}) == "None None (0, 0, 1) None"                                                # one process: untouched
    from lane_tracker_amd import distributed
    env = distributed.ensure_ipc_env({})
    assert env == {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}
