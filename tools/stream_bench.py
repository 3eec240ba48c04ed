#!/usr/bin/env python3
"""Single stateful stream through LaneTracker.process() (host state machine + per-frame kernel chain):
frames/s of the drop-in API, which is latency-bound (one frame in flight), for the reference
calibration and for BASELINE config 5 (1920x1080 camera)."""
import json, sys, time
import numpy as np
sys.path.insert(0, ".")
from lane_tracker_amd import calib, synth
from lane_tracker_amd.lane_tracker import LaneTracker

out = {}
for name, cal in (("1280x720", calib.reference_calibration()), ("1920x1080_config5", calib.scaled_calibration(1.5))):
    frames = synth.stream_lanes(40, seed=5, cal=cal)
    frames = np.concatenate([frames, frames[::-1], frames, frames[::-1], frames], 0)      # 200 frames, smooth at the turning points
    lt = LaneTracker(**cal)
    for f in frames[:5]:
        lt.process(f)
    t0 = time.perf_counter()
    for f in frames[5:]:
        lt.process(f)
    dt = time.perf_counter() - t0
    ratio = lt.get_success_ratio()
    # the same without the presentation step (GPU overlay + 2.8 MB download + Pillow text)
    lt2 = LaneTracker(**cal)
    lt2.draw_lane = lambda img: img
    lt2.print_failure = lambda img: img
    for f in frames[:5]:
        lt2.process(f)
    t0 = time.perf_counter()
    for f in frames[5:]:
        lt2.process(f)
    dt2 = time.perf_counter() - t0
    out[name] = {"process_fps": round((len(frames) - 5) / dt, 1), "process_fps_without_overlay": round((len(frames) - 5) / dt2, 1),
                 "success_ratio": ratio[0]}
    lt.close(); lt2.close()
print(json.dumps(out))
# stream pipeline (process_batch): masks batched ahead, state machine trailing
out2 = {}
for name, cal in (("1280x720", calib.reference_calibration()), ("1920x1080_config5", calib.scaled_calibration(1.5))):
    frames = synth.stream_lanes(64, seed=5, cal=cal)
    lt = LaneTracker(**cal)
    lt.process_batch(frames[:32], annotate=False)
    t0 = time.perf_counter()
    for _ in range(3):
        lt.process_batch(frames[32:], annotate=False)
    dt = time.perf_counter() - t0
    out2[name] = {"process_batch_fps_no_overlay": round(96 / dt, 1), "success_ratio": lt.get_success_ratio()[0]}
    lt.process_batch(frames[:32])
    t0 = time.perf_counter()
    for _ in range(3):
        lt.process_batch(frames[32:])
    out2[name]["process_batch_fps_annotated"] = round(96 / (time.perf_counter() - t0), 1)
    # the same with the input frames in page-locked memory (what lane_tracker_amd.video.FrameSource hands over)
    from lane_tracker_amd._native import pinned_empty
    pf = pinned_empty((len(frames),) + frames[0].shape)
    pf[...] = np.stack(frames, 0)
    lt.process_batch(pf[:32])
    t0 = time.perf_counter()
    for _ in range(3):
        lt.process_batch(pf[32:])
    out2[name]["process_batch_fps_annotated_pinned_input"] = round(96 / (time.perf_counter() - t0), 1)
    lt.close()
print(json.dumps(out2))
