import numpy as np, sys
sys.path.insert(0,'.')
from lane_tracker_amd import _native, calib, synth
from oracle import oracle as O
cal=calib.reference_calibration()
ctx=_native.Context(cal["img_size"],cal["warped_size"],cal["cam_matrix"],cal["dist_coeffs"],cal["warp_matrices"][0],capacity=1)
oc=O.make_calib(cal["img_size"],cal["warped_size"],cal["cam_matrix"],cal["dist_coeffs"],cal["warp_matrices"][0])
f=synth.frame_uniform(1)
ctx.upload_frames(f[None]); ctx.mask_run(1)
R=ctx.download_plane(0,1)[0]; B=ctx.download_plane(1,1)[0]
bev=O.front_end(oc,f); wb=O.lab_b(bev)
bad=np.argwhere(B!=wb)
print('bad',len(bad),'x%4 hist',np.bincount(bad[:,1]%4,minlength=4))
g,c,k=O.lab_tables()
def labb(r,gg,b):
    Rr,G,Bb=int(g[r]),int(g[gg]),int(g[b])
    iy=(Rr*k[3]+G*k[4]+Bb*k[5]+2048)>>12; iz=(Rr*k[6]+G*k[7]+Bb*k[8]+2048)>>12
    v=(200*(int(c[iy])-int(c[iz]))+128*32768+16384)>>15
    return max(0,min(255,v))
for (y,x) in bad[:6]:
    r,gg,b=[int(v) for v in bev[y,x]]
    got=int(B[y,x])
    print((y,x),'rgb',(r,gg,b),'want',int(wb[y,x]),labb(r,gg,b),'got',got)
    # which single-channel substitution explains it?
    for name,fn in (('g',lambda v:labb(r,v,b)),('b',lambda v:labb(r,gg,v))):
        vals=[v for v in range(256) if fn(v)==got]
        print('   ',name,'candidates',vals[:8],'...' if len(vals)>8 else '')
    print('    neighbours rgb', [tuple(int(v) for v in bev[y,x+d]) for d in (-2,-1,1)])
