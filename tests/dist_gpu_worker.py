"""Worker of tests/test_gpu_distributed.py: one rank of a sharded run over RCCL (torch.distributed "nccl").
Every rank processes its block of a 40-frame synthetic set on its GPU and all-gathers the lane records from
device memory; rank 0 checks them against a straight single-context run of all frames."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import _native, calib, distributed, synth  # noqa: E402

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
cal = calib.reference_calibration()
N = 40
r = synth.SceneRenderer(cal)
frames = np.stack([r.render(500 + i)[0] if i % 5 else np.full((720, 1280, 3), 128, np.uint8) for i in range(N)], 0)
lo, hi = distributed.shard_range(N, rank, world)
ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0],
                      device=local, capacity=16)
mine = distributed.process_shard(ctx, frames[lo:hi], first_frame=lo, batch=16)       # uneven last batch on purpose
host = distributed.gather_records(mine, N, device=torch.device("cuda", local))
# the device-side gather takes the records of the last batch straight from the context's slots
last = (hi - lo) - ((hi - lo - 1) // 16) * 16
send = torch.zeros(last * 64, dtype=torch.uint8, device="cuda")
ctx.copy_records_to_device(last, send.data_ptr())
assert np.frombuffer(send.cpu().numpy().tobytes(), _native.RECORD_DTYPE).tobytes() == mine[-last:].tobytes()
if rank == 0:
    ref = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0],
                          device=local, capacity=N)
    want = distributed.process_shard(ref, frames, first_frame=0, batch=N)
    assert host.tobytes() == want.tobytes(), "gathered records differ from the single-context run"
    assert list(host["frame"]) == list(range(N)) and int(host["detected"].sum()) == N - N // 5
    ref.close()
    print("distributed gpu ok: %d frames on %d rank(s)" % (N, world))
ctx.close()
dist.barrier()
dist.destroy_process_group()
