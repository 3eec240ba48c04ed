"""Worker of tests/test_gpu_distributed.py: one rank of a sharded run.  The gather is `lt_gather_*` of the C ABI
(RCCL); there is no PyTorch in this process.  Every rank processes its block of a 40-frame synthetic set on its
GPU, stages the records device to device and all-gathers them; rank 0 checks them against a straight
single-context run of all frames."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import _native, calib, distributed, synth  # noqa: E402

assert "torch" not in sys.modules
rank, local, world = distributed.env_rank()
try:
    local = distributed.local_device(local)          # LOCAL_RANK, or LOCAL_RANK % GPUs under LT_DEVICE_MODULO (fake-RCCL tests)
except RuntimeError as e:
    raise SystemExit("rank %d: %s" % (rank, e))
cal = calib.reference_calibration()
N = int(os.environ.get("LT_TEST_FRAMES", "40"))
r = synth.SceneRenderer(cal)
frames = np.stack([r.render(500 + i)[0] if i % 5 else np.full((720, 1280, 3), 128, np.uint8) for i in range(N)], 0)
lo, hi = distributed.shard_range(N, rank, world)
ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0],
                      device=local, capacity=16)
g = distributed.init_gather(ctx)
g.reserve(max(distributed.shard_sizes(N, world)))
mine = distributed.process_shard(ctx, frames[lo:hi], first_frame=lo, batch=16, gather=g)   # uneven last batch on purpose
dev = distributed.gather_staged(g, N)                                            # HBM slots -> RCCL -> host
host = distributed.gather_records(mine, N, distributed.RcclTransport(g))         # host records -> RCCL -> host
assert dev.tobytes() == host.tobytes(), "device-staged and host-staged gathers differ"
if os.environ.get("LT_TEST_MISMATCH"):
    # ranks that disagree on the record count must get an error from the agreement check, not a hang inside RCCL
    try:
        g.records(3 + rank)
        raise SystemExit("rank %d: mismatching counts went unnoticed" % rank)
    except _native.NativeError as e:
        if rank == 0:
            print("mismatch reported on rank 0: %s" % e)
    assert g.records(2).shape == (world, 2)          # and the communicator is still usable afterwards
times = g.host(np.array([float(rank)], np.float64))
assert times.reshape(-1).tolist() == [float(i) for i in range(world)]
if rank == 0:
    ref = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0],
                          device=local, capacity=N)
    want = distributed.process_shard(ref, frames, first_frame=0, batch=N)
    assert dev.tobytes() == want.tobytes(), "gathered records differ from the single-context run"
    assert list(dev["frame"]) == list(range(N)) and int(dev["detected"].sum()) == sum(1 for i in range(N) if i % 5)
    ref.close()
    print("distributed gpu ok: %d frames on %d rank(s), shards %s, torch loaded: %s"
          % (N, world, distributed.shard_sizes(N, world), "torch" in sys.modules))
g.barrier()
g.close()
ctx.close()
