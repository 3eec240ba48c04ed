#!/usr/bin/env python3
"""Where the time of the chained stream pipeline goes (process_batch, annotate=False): each phase alone, per frame."""
import json, sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from lane_tracker_amd import _native, calib, synth
from lane_tracker_amd.lane_tracker import LaneTracker

def t(fn, reps=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps

out = {}
for name, cal in (("1280x720", calib.reference_calibration()), ("1920x1080", calib.scaled_calibration(1.5))):
    base = synth.stream_lanes(32, seed=5, cal=cal)
    for n in (32, 128, 256):
        frames = np.concatenate([base, base[::-1]] * (n // 64 + 1), 0)[:n].copy()
        pin = _native.pinned_empty(frames.shape); pin[...] = frames
        lt = LaneTracker(**cal)
        ctx = lt._ctx
        ctx.reserve(n)
        sp = _native.search_params()
        r = {}
        r["upload_pageable_us"] = t(lambda: (ctx.upload_frame_rows_async(frames), ctx.sync())) / n * 1e6
        r["upload_pinned_us"] = t(lambda: (ctx.upload_frame_rows_async(pin), ctx.sync())) / n * 1e6
        r["mask_us"] = t(lambda: (ctx.mask_run(n), ctx.sync())) / n * 1e6
        for ch in (32, 64):
            def chunks():
                for lo in range(0, n, ch):
                    ctx.mask_run(min(ch, n - lo), first=lo)
                ctx.sync()
            r["mask_chunks%d_us" % ch] = t(chunks) / n * 1e6
        ctx.sws_fit_run(1, sp, first=0)
        r["chain_us"] = t(lambda: (ctx.band_fit_chain_run(n - 1, None, sp, first=1), ctx.sync())) / (n - 1) * 1e6
        r["download_records_us"] = t(lambda: ctx.download_records(n)) / n * 1e6
        rec = ctx.download_records(n)
        r["host_valid_many_us"] = t(lambda: lt._valid_many(rec["left_coeffs"], rec["right_coeffs"])) / n * 1e6
        lt.process_batch(frames, annotate=False)
        r["process_batch_pageable_fps"] = n / t(lambda: lt.process_batch(frames, annotate=False), 3)
        r["process_batch_pinned_fps"] = n / t(lambda: lt.process_batch(pin, annotate=False), 3)
        r["process_stream_pageable_fps"] = 6 * n / t(lambda: list(lt.process_stream([frames] * 6, annotate=False)), 2)
        r["success_ratio"] = lt.get_success_ratio()[0]
        bad = frames.copy()
        bad[9::10] = 0                                      # every tenth frame fails both tries
        lt.process_batch(bad, annotate=False)
        r["process_batch_fail_every_10_fps"] = n / t(lambda: lt.process_batch(bad, annotate=False), 3)
        out["%s_n%d" % (name, n)] = {k: round(v, 2) for k, v in r.items()}
        lt.close()
print(json.dumps(out, indent=1))
