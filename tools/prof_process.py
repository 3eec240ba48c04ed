import cProfile, pstats, sys, time
sys.path.insert(0, ".")
import numpy as np
from lane_tracker_amd import calib, synth
from lane_tracker_amd.lane_tracker import LaneTracker
cal = calib.reference_calibration()
frames = synth.stream_lanes(40, seed=5, cal=cal)
frames = np.concatenate([frames, frames[::-1], frames, frames[::-1], frames], 0)
lt = LaneTracker(**cal)
for f in frames[:5]:
    lt.process(f)
pr = cProfile.Profile()
pr.enable()
for f in frames[5:]:
    lt.process(f)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
