#!/usr/bin/env python3
"""Host wall time per context call inside LaneTracker.process() (one frame per call, annotated frame back)."""
import collections, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import calib, synth
from lane_tracker_amd.lane_tracker import LaneTracker
cal = calib.reference_calibration() if len(sys.argv) < 2 or sys.argv[1] in ("", "1.0") else calib.scaled_calibration(1.5)
if len(sys.argv) > 2 and sys.argv[2] == "bench":       # bench.py's stream (96 rendered frames played forwards and backwards into a window of 256)
    import bench
    frames = bench.stream_windows(bench.render_streams(96)["1280x720" if len(sys.argv[1]) == 0 or sys.argv[1] == "1.0" else "1920x1080"], 256, 1)[0]
else:
    frames = synth.stream_lanes(24, seed=5, cal=cal)
    frames = np.concatenate([frames, frames[::-1]] * 4, 0)
lt = LaneTracker(**cal)
if "engine" in sys.argv[3:]:                               # the frame's rows by the copy engine, not through the PCIe aperture
    lt._ctx.set_direct_upload(False)
for f in frames[:8]:
    lt.process(f)
acc, cnt = collections.defaultdict(float), collections.defaultdict(int)
def wrap(obj, name):
    fn = getattr(obj, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); acc[name] += time.perf_counter() - t0; cnt[name] += 1; return r
    setattr(obj, name, w)
for name in ("upload_frame_rows", "upload_frame_rest", "mask_run", "sws_fit_run", "band_fit_run", "download_record", "present_frame",
             "download_pixels"):
    wrap(lt._ctx, name)
class Lib:                                   # the library calls themselves, without their Python wrappers
    def __init__(self, lib):
        self._lib = lib
    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if name not in ("lt_present_frame", "lt_download_records", "lt_host_copy2d_async_group", "lt_host_copy_wait_group", "lt_present_lane_async",
                        "lt_present_finish", "lt_text_blend_host", "lt_upload_frame_rows", "lt_mask_run", "lt_band_fit_run", "lt_upload_frame_rest_rows"):
            return fn
        def w(*a):
            t0 = time.perf_counter(); r = fn(*a); acc[" " + name] += time.perf_counter() - t0; cnt[" " + name] += 1; return r
        return w
lt._ctx.lib = Lib(lt._ctx.lib)
from lane_tracker_amd import _native as _nat
_pe = _nat.pinned_empty
def _pe_timed(*a, **k):
    t0 = time.perf_counter(); r = _pe(*a, **k); acc[" pinned_empty"] += time.perf_counter() - t0; return r
_nat.pinned_empty = _pe_timed
_tb = _nat.text_blend
def _tb_timed(*a, **k):
    t0 = time.perf_counter(); r = _tb(*a, **k); acc[" text_blend"] += time.perf_counter() - t0; return r
_nat.text_blend = _tb_timed
for name in ("_record_success", "check_validity", "_points_packed", "get_curve_radius", "_lane_text", "_present", "_prepare_out", "_copies_done", "_lane_ahead"):
    wrap(lt, name)
n = len(frames) - 8
t0 = time.perf_counter()
for f in frames[8:]:
    out = lt.process(f)
total = time.perf_counter() - t0
print(json.dumps({"us_per_frame": round(total / n * 1e6, 1), "fps": round(n / total, 1),
                  "us_per_frame_by_call": {k: round(v / n * 1e6, 1) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])},
                  "calls_per_frame": {k: round(cnt[k] / n, 3) for k in ("mask_run", "sws_fit_run", "band_fit_run", "download_record") if k in cnt},
                  "success_ratio": round(lt.get_success_ratio()[0], 4)}))
