"""Phase timing of k_sws_fit2 for one frame.  Needs a probe build:
   make -C lane_tracker_amd/csrc -B k_search.o CXXFLAGS='-O3 -std=c++17 -fPIC -ffp-contract=off -DLT_SWS2_PROBE' && make -C lane_tracker_amd/csrc"""
import sys, numpy as np
sys.path.insert(0, ".")
from lane_tracker_amd import _native, calib, synth
cal = calib.reference_calibration()
ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], capacity=1)
ctx.upload_frames(synth.SceneRenderer(cal).render(3)[0])
ctx.mask_run(1, _native.filter_params())
for _ in range(3):
    ctx.sws_fit_run(1, _native.search_params()); ctx.sync()
