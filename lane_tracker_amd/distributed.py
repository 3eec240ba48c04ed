"""Sharding of independent frames over the GPUs of one node.

Frames with a fresh tracker each (BASELINE configs 2-4) do not interact: rank r of G processes the
contiguous block [r*N/G, (r+1)*N/G) with no halo and no data-path exchange.  The only collective is
one all-gather of the fixed 64-byte lane records (6 f64 coefficients + counts/flags) at the end --
32 KiB per rank for 4096 frames on 8 GPUs, latency-bound, so RCCL's ring/link bandwidth is
irrelevant here.  `torch.distributed` is plumbing: backend "nccl" is RCCL over xGMI on ROCm, "gloo"
is used by the CPU tests.  Results are bitwise identical for any G because the fit is computed
from exact integer moments (k_search.hip)."""
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


def gather_records(local_records, n_frames, group=None, device=None):
    """All-gather the per-rank record arrays (np structured, RECORD_DTYPE) into frame order.

    Every rank passes the records of its `shard_range`; every rank gets all `n_frames` records.
    Uneven shards are padded to the largest shard for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = shard_sizes(n_frames, world)
    local = np.ascontiguousarray(local_records, dtype=_native.RECORD_DTYPE)
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} records, its shard has {sizes[rank]}")
    cap = max(sizes) if sizes else 0
    send = torch.zeros(cap * RECORD_BYTES, dtype=torch.uint8)
    if local.shape[0]:
        send[: local.shape[0] * RECORD_BYTES] = torch.from_numpy(local.view(np.uint8).reshape(-1).copy())
    if device is not None:
        send = send.to(device)
    recv = torch.empty(world * cap * RECORD_BYTES, dtype=torch.uint8, device=send.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    flat = recv.cpu().numpy().reshape(world, cap * RECORD_BYTES)
    parts = [np.frombuffer(flat[r, : sizes[r] * RECORD_BYTES].tobytes(), dtype=_native.RECORD_DTYPE) for r in range(world)]
    return np.concatenate(parts) if parts else np.zeros(0, _native.RECORD_DTYPE)


def gather_device_records(ctx, n_local, n_frames, group=None):
    """Same gather with the records taken straight from a context's HBM slots [0, n_local) into a
    torch CUDA tensor (device-to-device copy, then RCCL all-gather): no host round trip on the send side."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = shard_sizes(n_frames, world)
    if n_local != sizes[rank]:
        raise ValueError(f"rank {rank} holds {n_local} records, its shard has {sizes[rank]}")
    cap = max(sizes)
    send = torch.zeros(cap * RECORD_BYTES, dtype=torch.uint8, device="cuda")
    if n_local:
        ctx.copy_records_to_device(n_local, send.data_ptr())
    recv = torch.empty(world * cap * RECORD_BYTES, dtype=torch.uint8, device="cuda")
    dist.all_gather_into_tensor(recv, send, group=group)
    flat = recv.cpu().numpy().reshape(world, cap * RECORD_BYTES)
    parts = [np.frombuffer(flat[r, : sizes[r] * RECORD_BYTES].tobytes(), dtype=_native.RECORD_DTYPE) for r in range(world)]
    return np.concatenate(parts)


def process_shard(ctx, frames, first_frame, fp=None, sp=None, batch=256):
    """Mask + sliding-window search + fit for a block of independent frames on one context.
    `frames` is (n, H, W, 3) u8 on the host; returns the n lane records tagged first_frame.."""
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
        out[lo:lo + m] = ctx.download_records(m)
    return out
