"""Sharding of independent frames over the GPUs of one node.

Frames with a fresh tracker each (BASELINE configs 2-4) do not interact: rank r of G processes the
contiguous block [r*N/G, (r+1)*N/G) with no halo and no data-path exchange.  The only collective is
one all-gather of the fixed 64-byte lane records (6 f64 coefficients + counts/flags) at the end --
32 KiB per rank for 4096 frames on 8 GPUs, latency-bound, so RCCL's ring/link bandwidth is
irrelevant here.  Results are bitwise identical for any G because the fit is computed from exact
integer moments (k_search.hip).

Transport: the product path is `lt_gather_*` of the C ABI (RCCL over xGMI, `_native.Gather`) -- one
process per GPU, no PyTorch anywhere.  `GlooTransport` (torch.distributed "gloo") exists for the
world-size-2 CPU tests only: it moves the same padded byte blocks, so the partitioning, padding and
re-assembly below are exercised without a GPU.

Launch: `python -m torch.distributed.run --nproc-per-node N script.py` (RANK / LOCAL_RANK / WORLD_SIZE in
the environment), or `spawn_ranks(N, argv)` from a parent that has not touched the GPU.  The ranks find
each other's RCCL id through a file named after their common parent process (`rendezvous_path`)."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np

from . import _native

RECORD_BYTES = _native.RECORD_DTYPE.itemsize


def shard_range(n_frames, rank, world):
    """Contiguous block of frame indices owned by `rank`: [lo, hi)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    return (rank * n_frames) // world, ((rank + 1) * n_frames) // world


def shard_sizes(n_frames, world):
    return [shard_range(n_frames, r, world)[1] - shard_range(n_frames, r, world)[0] for r in range(world)]


def pad_records(local_records, cap):
    """A rank's records padded with zero records to the largest shard (`cap`): every rank contributes the same
    number of bytes to the all-gather."""
    local = np.ascontiguousarray(local_records, dtype=_native.RECORD_DTYPE)
    out = np.zeros(cap, _native.RECORD_DTYPE)
    out[: local.shape[0]] = local
    return out


def assemble(gathered, sizes):
    """(world, cap) rank-major padded records -> the n_frames records in frame order."""
    gathered = np.asarray(gathered).reshape(len(sizes), -1)
    parts = [gathered[r, : sizes[r]] for r in range(len(sizes))]
    return np.concatenate(parts) if parts else np.zeros(0, _native.RECORD_DTYPE)


class GlooTransport:
    """CPU stand-in for the RCCL gather (tests only): torch.distributed all-gather of the padded byte blocks."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)

    def all_gather(self, padded):
        import torch
        send = torch.from_numpy(padded.view(np.uint8).reshape(-1).copy())
        recv = torch.empty(self.world * send.numel(), dtype=torch.uint8)
        self.dist.all_gather_into_tensor(recv, send, group=self.group)
        return np.frombuffer(recv.numpy().tobytes(), dtype=_native.RECORD_DTYPE).reshape(self.world, -1)


class RcclTransport:
    """The product transport for records that are already on the host: `lt_gather_host`, an RCCL all-gather of
    host bytes through page-locked staging buffers.  (Records still in a context's slots go device to device:
    `Gather.stage` + `gather_staged`.)"""

    def __init__(self, gather):
        self.g, self.world, self.rank = gather, gather.world, gather.rank

    def all_gather(self, padded):
        flat = self.g.host(padded.view(np.uint8).reshape(-1))
        return np.frombuffer(flat.tobytes(), dtype=_native.RECORD_DTYPE).reshape(self.world, -1)


def gather_records(local_records, n_frames, transport):
    """All-gather the per-rank record arrays (np structured, RECORD_DTYPE) into frame order.

    Every rank passes the records of its `shard_range`; every rank gets all `n_frames` records.
    Uneven shards are padded to the largest shard for the collective and trimmed afterwards."""
    sizes = shard_sizes(n_frames, transport.world)
    local = np.ascontiguousarray(local_records, dtype=_native.RECORD_DTYPE)
    if local.shape[0] != sizes[transport.rank]:
        raise ValueError(f"rank {transport.rank} holds {local.shape[0]} records, its shard has {sizes[transport.rank]}")
    cap = max(sizes) if sizes else 0
    if cap == 0:
        return np.zeros(0, _native.RECORD_DTYPE)
    return assemble(transport.all_gather(pad_records(local, cap)), sizes)


def gather_staged(gather, n_frames):
    """The device-side gather: every rank has staged the records of its shard (`Gather.stage`, positions
    0 .. shard size) straight from its context's HBM slots; one RCCL all-gather, frame order out."""
    sizes = shard_sizes(n_frames, gather.world)
    cap = max(sizes) if sizes else 0
    if cap == 0:
        return np.zeros(0, _native.RECORD_DTYPE)
    return assemble(gather.records(cap), sizes)


def process_shard(ctx, frames, first_frame, fp=None, sp=None, batch=256, gather=None):
    """Mask + sliding-window search + fit for a block of independent frames on one context.
    `frames` is (n, H, W, 3) u8 on the host; returns the n lane records tagged first_frame..
    With `gather` the records of every batch are also staged, device to device, at their shard position."""
    n = frames.shape[0]
    out = np.zeros(n, _native.RECORD_DTYPE)
    fp = fp or _native.filter_params()
    sp = sp or _native.search_params()
    ctx.reserve(min(batch, max(n, 1)))
    for lo in range(0, n, batch):
        m = min(batch, n - lo)
        ctx.upload_frame_rows(frames[lo:lo + m])      # only the camera rows the path reads cross the bus
        ctx.set_frame_base(m, first_frame + lo)
        ctx.mask_run(m, fp)
        ctx.sws_fit_run(m, sp)
        if gather is not None:
            gather.stage(m, at=lo)
        out[lo:lo + m] = ctx.download_records(m)
    return out


# ---- launch plumbing ---------------------------------------------------------------------------------------
IPC_ENV = ("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def ensure_ipc_env(env=None):
    """RCCL's intra-node transport hands device buffers from one rank process to another through HIP IPC handles
    (hipIpcGetMemHandle / hipIpcOpenMemHandle).  The hosts this runs on export such handles as dmabuf file descriptors only;
    the ROCm runtime's older ("legacy") IPC mode fails there with `hipIpcGetMemHandle: invalid argument` at
    `ncclCommInitRank`.  HSA_ENABLE_IPC_MODE_LEGACY=0 selects the dmabuf mode.  The runtime reads the variable when it
    initialises, so it has to be in the process environment BEFORE the first HIP call of a rank: every way a rank is
    started -- `spawn_ranks`, `python -m torch.distributed.run ... bench.py`, a user's own launcher calling `env_rank()`
    / `init_gather()` -- goes through here.  A value the caller has set is left alone."""
    env = os.environ if env is None else env
    env.setdefault(*IPC_ENV)
    return env


def env_rank():
    """(rank, local_rank, world) from the launcher's environment; (0, 0, 1) when not launched as a rank.  Call it before
    anything touches the GPU: with more than one rank it also puts the IPC mode RCCL needs into the environment."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        ensure_ipc_env()
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), world


if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1:      # a rank process: at import, i.e. before _native creates a context
    ensure_ipc_env()


def rendezvous_path():
    """File through which rank 0 publishes the RCCL id.  All ranks of one launch are children of the same launcher
    process (torch.distributed.run's agent, or `spawn_ranks`), so its pid names the job; the elastic agent's run id
    and restart count make the name new for every attempt (a restart keeps the agent's pid and port, and a stale id
    from an attempt that died before its barrier must never be read).  LT_GATHER_ID overrides -- the caller then owns
    the file's freshness (`spawn_ranks` removes it before it starts the ranks)."""
    p = os.environ.get("LT_GATHER_ID")
    if p:
        return p
    attempt = "%s_%s" % (os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"))
    attempt = "".join(ch if ch.isalnum() else "-" for ch in attempt)[:64]
    return os.path.join(tempfile.gettempdir(), "lt_gather_%d_%s_%s.id" % (os.getppid(), os.environ.get("MASTER_PORT", "0"), attempt))


def shares_devices():
    """LT_DEVICE_MODULO=1: several ranks may share a GPU (local rank r uses GPU r % visible GPUs).  For the tests that
    run the world-2 code on a one-GPU box (with tests/fake_rccl.c; the real RCCL refuses two ranks on one device);
    a run made this way is labelled as such by bench.py and is not an N-GPU measurement."""
    return os.environ.get("LT_DEVICE_MODULO", "0") not in ("", "0")


def local_device(local_rank=None):
    """The GPU of this rank: LOCAL_RANK (one process per GPU).  More ranks than GPUs is an error unless
    `shares_devices()`."""
    if local_rank is None:
        local_rank = env_rank()[1]
    ndev = _native.device_count()
    if shares_devices() and ndev > 0:
        return local_rank % ndev
    if local_rank >= ndev:
        raise RuntimeError("local rank %d but only %d GPU(s) visible" % (local_rank, ndev))
    return local_rank


def init_gather(ctx, timeout_s=120):
    """The rank's `_native.Gather`, from the launcher's environment.  Collective over all ranks."""
    rank, _, world = env_rank()        # (also: the IPC mode, if this is the rank's first call into this module)
    path = rendezvous_path()
    g = _native.Gather(ctx, rank, world, path, timeout_s)
    g.barrier()                       # every rank has read the id
    if rank == 0:
        try:
            os.remove(path)
        except OSError:
            pass
    return g


def visible_gpu_count():
    """Number of GPUs a fresh process of this interpreter sees -- asked in a child process, so that the caller
    itself never initialises the GPU (a launcher must stay GPU-free to start ranks)."""
    code = ("import sys; sys.path.insert(0, %r); from lane_tracker_amd import _native; print(_native.device_count())"
            % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    if r.returncode != 0:
        return 0
    try:
        return int(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        return 0


def spawn_ranks(world, argv, timeout=None):
    """Start `world` rank processes of `argv` (one per GPU: LOCAL_RANK = RANK) with the torch.distributed.run
    environment (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT) and wait for them.  Rank 0 inherits
    stdout.  Returns the largest exit code.  The caller must not have initialised the GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    id_path = os.path.join(tempfile.gettempdir(), "lt_gather_%d_%d.id" % (os.getpid(), port))
    try:
        os.remove(id_path)
    except OSError:
        pass
    procs = []
    for r in range(world):
        env = ensure_ipc_env(dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                                  MASTER_PORT=str(port), LT_GATHER_ID=id_path))
        procs.append(subprocess.Popen(list(argv), env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    import time
    worst, t0 = 0, time.monotonic()
    try:
        live = list(procs)
        while live:
            for p in list(live):
                rc = p.poll()
                if rc is not None:
                    live.remove(p)
                    worst = max(worst, abs(rc))
            if worst:                     # one rank failed: the others would wait for it in a collective
                break
            if timeout is not None and time.monotonic() - t0 > timeout:
                worst = 124
                break
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        try:
            os.remove(id_path)
        except OSError:
            pass
    return worst
