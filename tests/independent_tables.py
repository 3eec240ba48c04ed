"""Independent generators of the calibration tables (test infrastructure).

Written from the algorithm descriptions in SURVEY.md Appendix A -- not from oracle/lt_oracle.c or
csrc/lt_tables.cpp, whose coordinate generators are twin restatements by one hand.  Two generators per map:

* `*_f64`: vectorised NumPy in double precision following the operation order App. A gives for OpenCV
  (3x3 inverse by cofactors times 1/det, per-row start value plus repeated addition along the row, the
  64-pixel block association of warpPerspective).  NumPy ufuncs never contract a*b+c into an FMA, and
  `np.add.accumulate` adds sequentially, so this is bit-comparable with the C generators.
* `*_exact`: the mathematical definition evaluated directly (no running sums, no blocks, no stripes) in
  80-bit long double.  It can differ from an f64 evaluation only where a coordinate * 32 lies within
  rounding noise of a half-integer; `tie_distance` measures that.
"""
import numpy as np

INTER_BITS = 5
TAB = 1 << INTER_BITS


def _inv3_cofactor(a):
    """3x3 inverse, cofactors times the reciprocal determinant (double precision, NumPy scalars)."""
    a = np.asarray(a, np.float64).reshape(3, 3)
    det = (a[0, 0] * (a[1, 1] * a[2, 2] - a[1, 2] * a[2, 1]) - a[0, 1] * (a[1, 0] * a[2, 2] - a[1, 2] * a[2, 0])
           + a[0, 2] * (a[1, 0] * a[2, 1] - a[1, 1] * a[2, 0]))
    d = np.float64(1.0) / det
    t = np.empty((3, 3), np.float64)
    t[0, 0] = (a[1, 1] * a[2, 2] - a[1, 2] * a[2, 1]) * d
    t[0, 1] = (a[0, 2] * a[2, 1] - a[0, 1] * a[2, 2]) * d
    t[0, 2] = (a[0, 1] * a[1, 2] - a[0, 2] * a[1, 1]) * d
    t[1, 0] = (a[1, 2] * a[2, 0] - a[1, 0] * a[2, 2]) * d
    t[1, 1] = (a[0, 0] * a[2, 2] - a[0, 2] * a[2, 0]) * d
    t[1, 2] = (a[0, 2] * a[1, 0] - a[0, 0] * a[1, 2]) * d
    t[2, 0] = (a[1, 0] * a[2, 1] - a[1, 1] * a[2, 0]) * d
    t[2, 1] = (a[0, 1] * a[2, 0] - a[0, 0] * a[2, 1]) * d
    t[2, 2] = (a[0, 0] * a[1, 1] - a[0, 1] * a[1, 0]) * d
    return t


def _fixed(iu, iv, saturate_xy):
    """(integer coordinate * 32) -> (sx, sy) int16 pairs and the 5+5-bit fraction word, App. A.0."""
    sx, sy = iu >> INTER_BITS, iv >> INTER_BITS                       # arithmetic shift on int64
    if saturate_xy:
        sx, sy = np.clip(sx, -32768, 32767), np.clip(sy, -32768, 32767)
    xy = np.stack([sx, sy], -1).astype(np.int16)
    frac = ((iv & (TAB - 1)) * TAB + (iu & (TAB - 1))).astype(np.uint16)
    return xy, frac


def _rint_sat32(v):
    """round-half-even, saturating to the int32 range (cvRound semantics on clamped input)."""
    return np.clip(np.rint(v), -2147483648.0, 2147483647.0).astype(np.int64)


# ---- cv2.warpPerspective(src, M, (W, H)) : App. A.2 ---------------------------------------------------------
def warp_map_f64(M, W, H):
    m = _inv3_cofactor(M).reshape(9)
    y = np.arange(H, dtype=np.float64)[:, None]
    x = np.arange(W)
    xb = ((x // 64) * 64).astype(np.float64)[None, :]                 # block origin, then the offset inside the block
    x1 = (x % 64).astype(np.float64)[None, :]
    X0 = (m[0] * xb + m[1] * y) + m[2]
    Y0 = (m[3] * xb + m[4] * y) + m[5]
    W0 = (m[6] * xb + m[7] * y) + m[8]
    Wd = W0 + m[6] * x1
    with np.errstate(divide="ignore"):
        s = np.where(Wd != 0.0, np.float64(TAB) / Wd, 0.0)
    fX = np.clip((X0 + m[0] * x1) * s, -2147483648.0, 2147483647.0)
    fY = np.clip((Y0 + m[3] * x1) * s, -2147483648.0, 2147483647.0)
    return _fixed(_rint_sat32(fX), _rint_sat32(fY), True)


def warp_coords_exact(M, W, H):
    """Source coordinates * 32 of every bird's-eye pixel in long double, straight from the definition."""
    L = np.longdouble
    a = np.asarray(M, np.float64).reshape(3, 3).astype(L)
    cof = np.empty((3, 3), L)
    for i in range(3):
        for j in range(3):
            r = [k for k in range(3) if k != j]
            c = [k for k in range(3) if k != i]
            cof[i, j] = (a[r[0], c[0]] * a[r[1], c[1]] - a[r[0], c[1]] * a[r[1], c[0]]) * (-1) ** (i + j)
    det = a[0, 0] * cof[0, 0] + a[0, 1] * cof[1, 0] + a[0, 2] * cof[2, 0]
    m = cof / det
    y = np.arange(H).astype(L)[:, None]
    x = np.arange(W).astype(L)[None, :]
    w = m[2, 0] * x + m[2, 1] * y + m[2, 2]
    return (m[0, 0] * x + m[0, 1] * y + m[0, 2]) / w * TAB, (m[1, 0] * x + m[1, 1] * y + m[1, 2]) / w * TAB


# ---- cv2.undistort(img, K, D, None, K) : App. A.1 -----------------------------------------------------------
def undistort_map_f64(K, D, img_w, img_h, r0, r1):
    K = np.asarray(K, np.float64).reshape(3, 3)
    k1, k2, p1, p2, k3 = [np.float64(v) for v in np.asarray(D, np.float64).reshape(-1)[:5]]
    fx, fy, u0, v0 = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    stripe = min(max(1, 4096 // max(img_w, 1)), img_h)
    xy = np.empty((r1 - r0, img_w, 2), np.int16)
    frac = np.empty((r1 - r0, img_w), np.uint16)
    for row in range(r0, r1):
        y0 = (row // stripe) * stripe
        i = np.float64(row - y0)
        A = K.copy()
        A[1, 2] = K[1, 2] - y0
        ir = _inv3_cofactor(A).reshape(9)

        def walk(start, step):                   # start, start+step, (start+step)+step, ... sequentially
            seq = np.full(img_w, step, np.float64)
            seq[0] = start
            return np.add.accumulate(seq)
        xs, ys, ws = walk(i * ir[1] + ir[2], ir[0]), walk(i * ir[4] + ir[5], ir[3]), walk(i * ir[7] + ir[8], ir[6])
        iw = np.float64(1.0) / ws
        x, y = xs * iw, ys * iw
        x2, y2 = x * x, y * y
        r2, _2xy = x2 + y2, (2 * x) * y
        kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
        xd = (x * kr + p1 * _2xy) + p2 * (r2 + 2 * x2)
        yd = (y * kr + p1 * (r2 + 2 * y2)) + p2 * _2xy
        u, v = fx * xd + u0, fy * yd + v0
        a, b = _fixed(_rint_sat32(u * TAB), _rint_sat32(v * TAB), False)
        xy[row - r0], frac[row - r0] = a, b
    return xy, frac


def undistort_coords_exact(K, D, img_w, r0, r1):
    L = np.longdouble
    K = np.asarray(K, np.float64).reshape(3, 3).astype(L)
    k1, k2, p1, p2, k3 = [L(v) for v in np.asarray(D, np.float64).reshape(-1)[:5]]
    x = ((np.arange(img_w).astype(L) - K[0, 2]) / K[0, 0])[None, :]
    y = ((np.arange(r0, r1).astype(L) - K[1, 2]) / K[1, 1])[:, None]
    r2 = x * x + y * y
    kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
    xd = x * kr + p1 * (2 * x * y) + p2 * (r2 + 2 * x * x)
    yd = y * kr + p1 * (r2 + 2 * y * y) + p2 * (2 * x * y)
    return (K[0, 0] * xd + K[0, 2]) * TAB, (K[1, 1] * yd + K[1, 2]) * TAB


def table_coords(xy, frac):
    """(sx, sy, frac) -> the integer coordinate * 32 the table stands for (undoes App. A.0's split)."""
    iu = xy[..., 0].astype(np.int64) * TAB + (frac.astype(np.int64) & (TAB - 1))
    iv = xy[..., 1].astype(np.int64) * TAB + (frac.astype(np.int64) >> INTER_BITS)
    return iu, iv


def tie_distance(v):
    """distance of a long-double value from the nearest rounding tie (k + 0.5)"""
    f = v - np.floor(v)
    return np.abs(f - np.longdouble(0.5)).astype(np.float64)


# ---- RGB2LAB tables : App. A.4 ------------------------------------------------------------------------------
def lab_tables_exact():
    L = np.longdouble
    t = np.arange(256).astype(L) / 255
    lin = np.where(t <= L("0.04045"), t / L("12.92"), ((t + L("0.055")) / L("1.055")) ** L("2.4"))
    gamma = np.clip(np.rint(255 * 8 * lin), 0, 65535).astype(np.int64)
    u = np.arange(3072).astype(L) / (255 * 8)
    f = np.where(u < L("0.008856"), u * L("7.787") + L(16) / L(116), np.cbrt(u))
    cbrt = np.clip(np.rint(32768 * f), 0, 65535).astype(np.int64)
    rgb2xyz = np.array([["0.412453", "0.357580", "0.180423"], ["0.212671", "0.715160", "0.072169"],
                        ["0.019334", "0.119193", "0.950227"]])
    white = [L("0.950456"), L(1), L("1.088754")]
    coef = np.array([[int(np.rint(L(rgb2xyz[r, c]) / white[r] * 4096)) for c in range(3)] for r in range(3)]).reshape(9)
    return gamma, cbrt, coef


# ---- getStructuringElement(MORPH_ELLIPSE, (k, k)) : App. A.5 ------------------------------------------------
def ellipse_halfwidths_exact(k):
    """dx(dy) = round(sqrt(r^2 - dy^2)) in exact integer arithmetic (r^2 - dy^2 is an integer and (d + 1/2)^2 never
    is, so there are no ties)."""
    r = k // 2
    out = []
    for i in range(k):
        s = r * r - (i - r) ** 2
        d = 0
        while (2 * d + 1) ** 2 < 4 * s:       # (d + 1/2)^2 < s  <=>  round(sqrt(s)) > d
            d += 1
        out.append(d)
    return out, sum(min(r + d + 1, k) - max(r - d, 0) for d in out)


SURVEY_HALFWIDTHS = {   # SURVEY.md App. A.5, top row to centre row
    29: [0, 5, 7, 9, 10, 11, 11, 12, 13, 13, 13, 14, 14, 14, 14],
    55: [0, 7, 10, 12, 14, 16, 17, 18, 19, 20, 21, 22, 22, 23, 24, 24, 25, 25, 25, 26, 26, 26, 27, 27, 27, 27, 27, 27],
}
SURVEY_TAPS = {5: 17, 29: 641, 55: 2337}
