import ctypes as C, numpy as np, sys
sys.path.insert(0,'.')
from lane_tracker_amd import _native, calib, synth
cal=calib.reference_calibration()
B=64
ctx=_native.Context(cal["img_size"],cal["warped_size"],cal["cam_matrix"],cal["dist_coeffs"],cal["warp_matrices"][0],capacity=B)
r=synth.SceneRenderer(cal)
fr=np.stack([r.render(i)[0] for i in range(8)],0)[np.arange(B)%8]
ctx.upload_frames(fr)
fp=_native.filter_params(); fp.noise_thresh=-12345   # probe switch; mask_noise stays off so the value is otherwise unused
lib=ctx.lib; lib.lt_debug_cycles.argtypes=[C.c_void_p,C.c_void_p,C.c_int]
out=(C.c_longlong*16)()
ctx.mask_run(B,fp); lib.lt_debug_cycles(ctx._h,out,1)
ctx.mask_run(B,fp); lib.lt_debug_cycles(ctx._h,out,1)
o=list(out); n=o[8:12]
print('blocks per wave-slot',n)
print('staging cycles/wave', [o[i]/max(n[i],1) for i in range(4)])
print('phase cycles/wave  ', [o[4+i]/max(n[i],1) for i in range(4)])
