#!/usr/bin/env python3
"""Times of the reference's own NumPy stages a5-a8 (sliding_window_search, band_search, fit_poly, check_validity,
get_curve_radius), run UNMODIFIED by importing /root/reference in the build container (cv2 replaced by an empty module,
`np.int = int`, integral `partial`: SURVEY.md F5).  The reference cannot travel to the GPU box, so bench.py prints the
committed result (profiles/reference_numpy_timings.json) inside `cpu_baseline` as "reference NumPy, container".

usage (build container only): python tools/time_reference_numpy.py [--ref /root/reference]"""
import argparse
import json
import os
import platform
import sys
import time
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from lane_tracker_amd import calib, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--ref", default="/root/reference")
a = ap.parse_args()
sys.modules["cv2"] = types.ModuleType("cv2")
np.int = int
sys.path.insert(0, a.ref)
import lane_tracker as ref  # noqa: E402  (the reference, unmodified)

cal = calib.reference_calibration()
lt = ref.LaneTracker(img_size=cal["img_size"], warped_size=cal["warped_size"], cam_matrix=cal["cam_matrix"],
                     dist_coeffs=cal["dist_coeffs"], warp_matrices=cal["warp_matrices"], mpp_conversion=cal["mpp_conversion"])
mask = synth.synth_mask(1234, noise=1e-3)[0]


def best_ms(fn, reps=7):
    out = []
    for _ in range(reps):
        t = time.perf_counter()
        fn()
        out.append((time.perf_counter() - t) * 1e3)
    return round(min(out), 3), round(sorted(out)[len(out) // 2], 3)


res = {}
res["sliding_window_search (26 levels)"] = best_ms(lambda: lt.sliding_window_search(mask, 30, 40, 20, 0.1, 8, partial=1))
lt.sliding_window_search(mask, 30, 40, 20, 0.1, 8, partial=1)
n_left, n_right = len(lt.left_x), len(lt.right_x)
res["fit_poly (2 x np.polyfit)"] = best_ms(lambda: lt.fit_poly())
lf, rf = lt.fit_poly()
lt.last_left_coeffs, lt.last_right_coeffs = lf, rf
res["check_validity"] = best_ms(lambda: lt.check_validity(lf, rf))
res["get_curve_radius (2 more np.polyfit)"] = best_ms(lambda: lt.get_curve_radius())
res["band_search (whole-image nonzero + predicate)"] = best_ms(lambda: lt.band_search(mask, 25, partial=1))
out = {"what": "reference lane_tracker.py stages a5-a8 imported unmodified (cv2 stubbed), one thread, best / median of 7 runs in ms",
       "mask": "synthetic 1100x1080 lane mask, %d + %d lane pixels" % (n_left, n_right),
       "host": "%s, %d logical CPUs (build container, not the GPU box)" % (platform.processor() or platform.machine(), os.cpu_count()),
       "numpy": np.__version__, "stage_ms_best_median": res}
json.dump(out, open(os.path.join(ROOT, "profiles", "reference_numpy_timings.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
