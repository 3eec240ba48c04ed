#!/usr/bin/env python3
"""Diagnose the walking threshold kernels on one frame: merged plane of the GPU vs NumPy per-direction verdicts."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import _native, calib, synth

def sums(p, k, axis):
    P = np.cumsum(np.pad(p.astype(np.int64), [(k + 1, k + 1) if a == axis else (0, 0) for a in range(2)]), axis=axis)
    n = p.shape[axis]
    def take(lo, hi):
        sl = [slice(None)] * 2
        sl[axis] = slice(lo, hi)
        return P[tuple(sl)]
    # padded index j <-> original j-(k+1); prefix P[j] = sum of padded[0..j]
    before = take(k, k + n) - take(0, n)            # sum of the k pixels before
    after = take(2 * k + 1, 2 * k + 1 + n) - take(k + 1, k + 1 + n)   # sum of the k pixels after
    return before, after

def verdict(p, k, C, axis):
    b, a = sums(p, k, axis)
    t = k * p.astype(np.int64) - C * k
    return (b < t) & (a < t)

cal = calib.reference_calibration()
frame = synth.SceneRenderer(cal).render(7)[0]
ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0], device=0, capacity=1)
ctx.upload_frames(frame[None]); ctx.mask_run(1)
tr = ctx.download_plane(_native.PLANE_TOPHAT_R, 1)[0]; tb = ctx.download_plane(_native.PLANE_TOPHAT_B, 1)[0]
got = ctx.download_plane(_native.PLANE_MERGED, 1)[0] > 0
hr, vr, hb, vb = verdict(tr, 15, 8, 1), verdict(tr, 15, 8, 0), verdict(tb, 35, 5, 1), verdict(tb, 35, 5, 0)
passes = int(os.environ.get("LT_WALK_PASSES", "15"))
want = np.zeros_like(hr)
for bit, m in ((1, hr), (2, vr), (4, hb), (8, vb)):
    if passes & bit:
        want |= m
print("pixels set: got %d want %d, differing %d" % (got.sum(), want.sum(), (got != want).sum()))
for name, m in (("H_R", hr), ("V_R", vr), ("H_b", hb), ("V_b", vb)):
    print("%s: set %d, missing in got %d" % (name, m.sum(), (m & ~got).sum()))
extra = got & ~want
print("extra pixels %d; by row block:" % extra.sum(), [int(extra[r:r + 128].sum()) for r in range(0, 1100, 128)])
print("extra by column block:", [int(extra[:, c:c + 64].sum()) for c in range(0, 1080, 64)])
ys, xs = np.nonzero(extra)
print("first extras:", list(zip(ys[:12].tolist(), xs[:12].tolist())))
miss = want & ~got
print("missing %d; by row block:" % miss.sum(), [int(miss[r:r + 128].sum()) for r in range(0, 1100, 128)])
print("missing by column block:", [int(miss[:, c:c + 64].sum()) for c in range(0, 1080, 64)])
d = got != want
print("diff rows mod 128 histogram (first 16 lanes):", [int(d[r::128].sum()) for r in range(16)], "rows 64..79:", [int(d[r::128].sum()) for r in range(64, 80)])
print("diff by column, cols 0..95:", [int(d[:, c].sum()) for c in range(96)])
print("diff by column, cols 376..520 step 8:", [int(d[:, c:c + 8].sum()) for c in range(376, 520, 8)])
print("diff by column, cols 570..680 step 8:", [int(d[:, c:c + 8].sum()) for c in range(568, 680, 8)])
print("rows 0..127 diff by 64-col block:", [int(d[:128, c:c + 64].sum()) for c in range(0, 1080, 64)])
print("rows 0..127, cols 64..1023: diff by row:", [int(d[r, 64:1024].sum()) for r in range(0, 128)])
ys, xs = np.nonzero(d[:128, 64:1024])
print("examples (row, col, got, want):", [(int(y), int(x) + 64, int(got[y, x + 64]), int(want[y, x + 64])) for y, x in list(zip(ys, xs))[:24]])
for (y, x) in [(0, 67), (0, 84), (3, 100)]:
    row = tb[y].astype(int); k, C = 35, 5
    L = row[max(x - k, 0):x].sum(); R = row[x + 1:x + 1 + k].sum(); p = row[x]
    print("brute (%d,%d): p=%d k*p=%d L+Ck=%d R+Ck=%d pass=%s got=%d want=%d" % (y, x, p, k * p, L + C * k, R + C * k, (k * p > L + C * k) and (k * p > R + C * k), got[y, x], want[y, x]),
          "pixels:", row[x - 3:x + 4].tolist())
ctx.close()
