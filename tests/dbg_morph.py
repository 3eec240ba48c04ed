import numpy as np, sys
sys.path.insert(0,'.')
from lane_tracker_amd import _native
from oracle import oracle as O
c=_native.Context((2,2),(2,2),np.eye(3),np.zeros(5),np.eye(3))
rng=np.random.default_rng(0)
for k in (29,55):
    for shape in ((200,300),(1100,1080)):
        img=rng.integers(0,256,shape,dtype=np.uint8)
        for op,fn in (('erode',lambda a:O.erode(a,k)),('dilate',lambda a:O.dilate(a,k))):
            got=c.morph_ellipse(img,k,op); want=fn(img)
            bad=np.argwhere(got!=want)
            print(k,shape,op,'bad',len(bad), bad[:8].tolist())
            if len(bad):
                cols=np.bincount(bad[:,1]%128,minlength=128); print('  col%128 hist nonzero:',{i:int(v) for i,v in enumerate(cols) if v})
                rows=np.bincount(bad[:,0],minlength=shape[0]); print('  rows with errors:',int((rows>0).sum()),'first',np.flatnonzero(rows)[:10].tolist())
    # delta image: footprint probe at a seam
    img=np.full((120,300),255,np.uint8); img[60,64]=0; img[60,100]=7
    got=c.morph_ellipse(img,k,'erode'); want=O.erode(img,k)
    bad=np.argwhere(got!=want); print('delta',k,len(bad),bad[:12].tolist())
