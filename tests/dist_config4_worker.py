"""Worker of tests/test_gpu_config4.py: one rank of BASELINE config 4 -- a 4096-frame synthetic stream sharded over the
ranks, 512 frames at a time through the rank's context, every chunk's records staged device to device and ONE
lt_gather_records at the end.  Rank 0 then checks the gathered records (a) bitwise against a plain run of all 4096
frames through a second context that never sees the gather (different batch shape: 256), and (b) for 64 evenly spaced
frames against the CPU oracle.  No PyTorch in this process.

Frames are rendered by a pool of worker processes forked BEFORE the GPU is initialised (a process that has touched
the GPU must not fork) and consumed chunk by chunk, so the host never holds more than two chunks."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lane_tracker_amd import _native, calib, distributed, synth  # noqa: E402

N = int(os.environ.get("LT_TEST_FRAMES", "4096"))
CHUNK = int(os.environ.get("LT_TEST_CHUNK", "512"))
SAMPLES = int(os.environ.get("LT_TEST_ORACLE_SAMPLES", "64"))
rank, local, world = distributed.env_rank()


def _render(i):
    global _R
    try:
        r = _R
    except NameError:
        r = _R = synth.SceneRenderer()
    return r.render(i)[0]


def main():
    from concurrent.futures import ProcessPoolExecutor
    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count() or 1
    try:                                             # the box shows 256 CPUs and grants a cgroup quota of 16: more workers only add set-up cost
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            cpus = max(1, min(cpus, int(float(q) / float(per) + 0.999)))
    except Exception:
        pass
    workers = max(1, min(48, cpus // world))
    pool = ProcessPoolExecutor(workers)
    list(pool.map(_render, range(workers)))          # forks every worker now, before any HIP call below

    def render(lo, hi):
        return np.stack(list(pool.map(_render, range(lo, hi), chunksize=max(1, (hi - lo) // (workers * 4)))), 0)

    t0 = time.time()
    device = distributed.local_device(local)
    cal = calib.reference_calibration()
    mk = lambda cap: _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"],
                                     cal["warp_matrices"][0], device=device, capacity=cap)
    ctx = mk(CHUNK)
    g = distributed.init_gather(ctx)
    lo, hi = distributed.shard_range(N, rank, world)
    g.reserve(max(distributed.shard_sizes(N, world)))
    sample_ids = sorted(set(np.linspace(0, N - 1, SAMPLES).astype(int).tolist())) if rank == 0 else []
    kept = {}                                        # sampled frames for the oracle check
    want = np.zeros(N, _native.RECORD_DTYPE) if rank == 0 else None
    ref = mk(256) if rank == 0 else None

    def plain(frames, first):                        # the straight single-context run of the same frames (rank 0)
        want[first:first + len(frames)] = distributed.process_shard(ref, frames, first_frame=first, batch=256)
        for i in sample_ids:
            if first <= i < first + len(frames):
                kept[i] = frames[i - first].copy()

    for c0 in range(lo, hi, CHUNK):
        c1 = min(c0 + CHUNK, hi)
        frames = render(c0, c1)
        distributed.process_shard(ctx, frames, first_frame=c0, batch=CHUNK, gather=_Shifted(g, c0 - lo))
        if rank == 0:
            plain(frames, c0)
    got = distributed.gather_staged(g, N)            # ONE all-gather of every rank's shard
    t_sharded = time.time() - t0
    if rank == 0:
        for r in range(1, world):                    # the other ranks' blocks, for the plain run only
            rlo, rhi = distributed.shard_range(N, r, world)
            for c0 in range(rlo, rhi, CHUNK):
                plain(render(c0, min(c0 + CHUNK, rhi)), c0)
        assert list(got["frame"]) == list(range(N)), "gathered records are not in frame order"
        assert got.tobytes() == want.tobytes(), "gathered records differ from the plain single-context run"
        from concurrent.futures import ThreadPoolExecutor
        from oracle import oracle as O
        from tests.helpers import coeff_close
        oc = O.make_calib(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0])
        with ThreadPoolExecutor(min(16, cpus)) as ex:
            res = list(ex.map(lambda i: O.frame_sws_fit(oc, kept[i]), sample_ids))
        for i, w in zip(sample_ids, res):
            rec = got[i]
            assert bool(rec["detected"]) == w["detected"], i
            assert (int(rec["n_left"]), int(rec["n_right"])) == (w["n_left"], w["n_right"]), i
            if w["detected"]:
                assert coeff_close(rec["left_coeffs"], w["coeffs"][0]) and coeff_close(rec["right_coeffs"], w["coeffs"][1]), i
        print("config4 ok: %d frames on %d rank(s) in chunks of %d, shards %s, %d/%d detected, %d oracle samples, "
              "sharded pass %.1f s (rendering included)" % (N, world, CHUNK, distributed.shard_sizes(N, world),
                                                           int(got["detected"].sum()), N, len(sample_ids), t_sharded))
        ref.close()
    g.barrier()
    g.close()
    ctx.close()
    pool.shutdown()


class _Shifted:
    """process_shard stages chunk-relative positions; this shifts them to the rank's shard position."""

    def __init__(self, gather, base):
        self.g, self.base = gather, base

    def stage(self, n, at=0, first=0):
        self.g.stage(n, at=self.base + at, first=first)


if __name__ == "__main__":
    main()
