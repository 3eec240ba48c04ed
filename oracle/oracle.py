"""ctypes front end of the CPU oracle (oracle/lt_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/ (the parity tests, the fixture generator tests/gen_golden.py, the fuzz driver tests/fuzz_gpu.py),
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (lane_tracker_amd/) never does.  See lt_oracle.h for what is restated and the parity
status of each stage (cv2-backed stages: parity unpinned; NumPy-only stages: pinned by
tests/golden/, generated from the reference by tests/gen_golden.py).

The Python-level pieces of the reference that are plain NumPy on a handful of scalars
(get_poly_points lane_tracker.py:511-528, check_validity :561-627) are restated here in NumPy.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class Calib(C.Structure):
    _fields_ = [("img_w", C.c_int32), ("img_h", C.c_int32), ("warp_w", C.c_int32),
                ("warp_h", C.c_int32), ("K", C.c_double * 9), ("D", C.c_double * 5),
                ("M", C.c_double * 9)]


class FilterParams(C.Structure):
    _fields_ = [("filter_type", C.c_int32), ("ksize_r", C.c_int32), ("C_r", C.c_int32),
                ("ksize_b", C.c_int32), ("C_b", C.c_int32), ("mask_noise", C.c_int32),
                ("noise_thresh", C.c_int32), ("ksize_noise", C.c_int32), ("C_noise", C.c_int32)]


class SearchParams(C.Structure):
    _fields_ = [("window_width", C.c_int32), ("window_height", C.c_int32),
                ("search_range", C.c_int32), ("no_success_limit", C.c_int32),
                ("ignore_sides", C.c_int32), ("ignore_bottom", C.c_int32),
                ("bandwidth", C.c_int32), ("_pad", C.c_int32),
                ("mu", C.c_double), ("start_slice", C.c_double), ("partial", C.c_double)]


def build(force=False):
    if os.environ.get("LT_ORACLE_SANITIZED") == "1":   # tests/test_oracle_sanitized.py: the -fsanitize=address,undefined build
        so = os.path.join(_HERE, "liblt_oracle_asan.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "-s", "asan"])
        return so
    so = os.path.join(_HERE, "liblt_oracle.so")
    src = [os.path.join(_HERE, f) for f in ("lt_oracle.c", "lt_oracle.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liblt_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
    return _LIB


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a, a.ctypes.data_as(C.c_void_p)


def make_calib(img_size, warped_size, cam_matrix, dist_coeffs, M):
    c = Calib()
    c.img_w, c.img_h = int(img_size[0]), int(img_size[1])
    c.warp_w, c.warp_h = int(warped_size[0]), int(warped_size[1])
    c.K[:] = [float(v) for v in np.asarray(cam_matrix, dtype=np.float64).reshape(9)]
    c.D[:] = [float(v) for v in np.asarray(dist_coeffs, dtype=np.float64).reshape(-1)[:5]]
    c.M[:] = [float(v) for v in np.asarray(M, dtype=np.float64).reshape(9)]
    return c


def filter_params(filter_type="bilateral", ksize_r=15, C_r=8, ksize_b=35, C_b=5, mask_noise=False,
                  noise_thresh=140, ksize_noise=65, C_noise=10):
    ft = {"bilateral": 0, "neighborhood": 1}.get(filter_type, 2)
    return FilterParams(ft, ksize_r, C_r, ksize_b, C_b, int(bool(mask_noise)), noise_thresh,
                        ksize_noise, C_noise)


def search_params(window_width=30, window_height=40, search_range=20, mu=0.1, no_success_limit=8,
                  start_slice=0.25, ignore_sides=360, ignore_bottom=30, bandwidth=25, partial=1.0):
    return SearchParams(window_width, window_height, search_range, no_success_limit, ignore_sides,
                        ignore_bottom, bandwidth, 0, float(mu), float(start_slice), float(partial))


# ---- geometric front end ---------------------------------------------------------------------
def undistort_map(calib, r0=0, r1=None):
    r1 = calib.img_h if r1 is None else r1
    xy = np.empty((r1 - r0, calib.img_w, 2), np.int16)
    al = np.empty((r1 - r0, calib.img_w), np.uint16)
    lib().lto_undistort_map(C.byref(calib), r0, r1, xy.ctypes.data_as(C.c_void_p), al.ctypes.data_as(C.c_void_p))
    return xy, al


def warp_map(calib):
    xy = np.empty((calib.warp_h, calib.warp_w, 2), np.int16)
    al = np.empty((calib.warp_h, calib.warp_w), np.uint16)
    lib().lto_warp_map(C.byref(calib), xy.ctypes.data_as(C.c_void_p), al.ctypes.data_as(C.c_void_p))
    return xy, al


def warp_source_rows(calib):
    r0, r1 = C.c_int(), C.c_int()
    lib().lto_warp_source_rows(C.byref(calib), C.byref(r0), C.byref(r1))
    return r0.value, r1.value


def undistort(calib, frame):
    f, fp = _u8(frame)
    out = np.empty_like(f)
    lib().lto_undistort(C.byref(calib), fp, out.ctypes.data_as(C.c_void_p))
    return out


def warp(calib, und):
    f, fp = _u8(und)
    out = np.empty((calib.warp_h, calib.warp_w, 3), np.uint8)
    lib().lto_warp(C.byref(calib), fp, out.ctypes.data_as(C.c_void_p))
    return out


def front_end(calib, frame):
    f, fp = _u8(frame)
    out = np.empty((calib.warp_h, calib.warp_w, 3), np.uint8)
    lib().lto_front_end(C.byref(calib), fp, out.ctypes.data_as(C.c_void_p))
    return out


# ---- colour / morphology / thresholds -----------------------------------------------------------
def lab_tables():
    g = np.empty(256, np.uint16)
    c = np.empty(3072, np.uint16)
    k = np.empty(9, np.int32)
    lib().lto_lab_tables(g.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p), k.ctypes.data_as(C.c_void_p))
    return g, c, k


def lab_b(rgb):
    f, fp = _u8(rgb)
    out = np.empty(f.shape[:-1], np.uint8)
    lib().lto_lab_b(fp, int(out.size), out.ctypes.data_as(C.c_void_p))
    return out


def ellipse_kernel(k):
    el = np.empty((k, k), np.uint8)
    lib().lto_ellipse_kernel(k, el.ctypes.data_as(C.c_void_p))
    return el


def ellipse_halfwidths(k):
    dx = (C.c_int * k)()
    taps = lib().lto_ellipse_halfwidths(k, dx)
    return list(dx), taps


def _plane_op(name, img, *args):
    f, fp = _u8(img)
    out = np.empty_like(f)
    getattr(lib(), name)(fp, f.shape[0], f.shape[1], *args, out.ctypes.data_as(C.c_void_p))
    return out


def erode(img, k, brute=False):
    return _plane_op("lto_morph_ellipse_brute" if brute else "lto_morph_ellipse", img, k, 0)


def dilate(img, k, brute=False):
    return _plane_op("lto_morph_ellipse_brute" if brute else "lto_morph_ellipse", img, k, 1)


def tophat(img, k):
    return _plane_op("lto_tophat", img, k)


def morph_open(img, k=5):
    return _plane_op("lto_open", img, k)


def bilateral_adaptive_threshold(img, ksize=30, C_=0, mode="floor", true_value=255, false_value=0, fast=False):
    if mode not in ("floor", "ceil"):
        raise ValueError("Unexpected mode value. Expected value is 'floor' or 'ceil'.")
    return _plane_op("lto_bilateral_adaptive_threshold" + ("_fast" if fast else ""), img, int(ksize), int(C_),
                     0 if mode == "floor" else 1, int(true_value), int(false_value))


def adaptive_mean_threshold(img, block_size, C_, fast=False):
    return _plane_op("lto_adaptive_mean_threshold" + ("_fast" if fast else ""), img, int(block_size), int(C_))


def filter_lane_points(bev_rgb, fp=None, want_planes=False, fast=False):
    """fast=True: the running-sum thresholds (the CPU baseline's form; equal results, tests/test_oracle_units.py)."""
    fp = fp or filter_params()
    f, fptr = _u8(bev_rgb)
    h, w = f.shape[:2]
    mask = np.empty((h, w), np.uint8)
    planes = np.zeros((4, h, w), np.uint8) if want_planes else None
    rc = (lib().lto_filter_lane_points_fast if fast else lib().lto_filter_lane_points)(fptr, h, w, C.byref(fp), mask.ctypes.data_as(C.c_void_p),
                                      planes.ctypes.data_as(C.c_void_p) if want_planes else None)
    if rc:
        raise ValueError("Unexpected filter mode. Expected modes are 'bilateral' or 'neighborhood'.")
    return (mask, planes) if want_planes else mask


def mask_from_frame(calib, frame, fp=None):
    fp = fp or filter_params()
    f, fptr = _u8(frame)
    mask = np.empty((calib.warp_h, calib.warp_w), np.uint8)
    rc = lib().lto_mask_from_frame(C.byref(calib), fptr, C.byref(fp), mask.ctypes.data_as(C.c_void_p))
    if rc:
        raise ValueError("Unexpected filter mode. Expected modes are 'bilateral' or 'neighborhood'.")
    return mask


# ---- search + fit ------------------------------------------------------------------------------
def sliding_window_search(mask, sp=None):
    """-> dict(detected, left_y, left_x, right_y, right_x, left_centroids, right_centroids)"""
    sp = sp or search_params()
    m, mp = _u8(mask)
    h, w = m.shape
    nlev = max(int((sp.partial * (h - sp.ignore_bottom)) / sp.window_height), 1)
    cap = nlev * sp.window_height * sp.window_width + 16
    buf = np.zeros((4, cap), np.int32)
    cent = np.zeros((2, nlev + 2), np.int32)
    n = (C.c_int32 * 4)()
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    det = lib().lto_sliding_window_search(
        mp, h, w, C.byref(sp), p(buf[0]), p(buf[1]), C.byref(n, 0), p(buf[2]), p(buf[3]),
        C.byref(n, 4), p(cent[0]), C.byref(n, 8), p(cent[1]), C.byref(n, 12))
    return dict(detected=bool(det),
                left_y=buf[0, :n[0]].astype(np.int64), left_x=buf[1, :n[0]].astype(np.int64),
                right_y=buf[2, :n[1]].astype(np.int64), right_x=buf[3, :n[1]].astype(np.int64),
                left_centroids=cent[0, :n[2]].tolist(), right_centroids=cent[1, :n[3]].tolist())


def band_search(mask, left_coeffs, right_coeffs, sp=None):
    sp = sp or search_params()
    m, mp = _u8(mask)
    h, w = m.shape
    cap = h * w
    buf = np.zeros((4, cap), np.int32)
    n = (C.c_int32 * 2)()
    lc = (C.c_double * 3)(*[float(v) for v in left_coeffs])
    rc = (C.c_double * 3)(*[float(v) for v in right_coeffs])
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    det = lib().lto_band_search(mp, h, w, C.byref(sp), lc, rc, p(buf[0]), p(buf[1]), C.byref(n, 0),
                                p(buf[2]), p(buf[3]), C.byref(n, 4))
    return dict(detected=bool(det),
                left_y=buf[0, :n[0]].astype(np.int64), left_x=buf[1, :n[0]].astype(np.int64),
                right_y=buf[2, :n[1]].astype(np.int64), right_x=buf[3, :n[1]].astype(np.int64))


def polyfit2(y, x):
    y = np.ascontiguousarray(y, np.int32)
    x = np.ascontiguousarray(x, np.int32)
    out = (C.c_double * 3)()
    lib().lto_polyfit2(y.ctypes.data_as(C.c_void_p), x.ctypes.data_as(C.c_void_p), int(y.size), out)
    return np.array(out[:], np.float64)


def frame_sws_fit(calib, frame, fp=None, sp=None, want_mask=False, fast=False):
    fp = fp or filter_params()
    sp = sp or search_params()
    f, fptr = _u8(frame)
    mask = np.empty((calib.warp_h, calib.warp_w), np.uint8) if want_mask else None
    coef = (C.c_double * 6)()
    counts = (C.c_int32 * 3)()
    rc = (lib().lto_frame_sws_fit_fast if fast else lib().lto_frame_sws_fit)(C.byref(calib), fptr, C.byref(fp), C.byref(sp),
                                 mask.ctypes.data_as(C.c_void_p) if want_mask else None, coef, counts)
    if rc:
        raise ValueError("bad filter_type")
    out = dict(coeffs=np.array(coef[:], np.float64).reshape(2, 3), n_left=counts[0],
               n_right=counts[1], detected=bool(counts[2]))
    if want_mask:
        out["mask"] = mask
    return out


# ---- NumPy-level pieces of the reference (host logic) -------------------------------------------
def get_poly_points(warped_size, left_fit_coeffs, right_fit_coeffs, partial=1):
    """lane_tracker.py:511-528 with the 2017 NumPy behaviour made explicit (num -> int(), np.int)."""
    img_height, img_width = warped_size[1], warped_size[0]
    ploty = np.linspace(img_height * (1 - partial), img_height - 1, int(img_height * partial))
    lf, rf = left_fit_coeffs, right_fit_coeffs
    left_fitx = lf[0] * ploty ** 2 + lf[1] * ploty + lf[2]
    right_fitx = rf[0] * ploty ** 2 + rf[1] * ploty + rf[2]
    left_fit_x = left_fitx[(left_fitx <= img_width - 1) & (left_fitx >= 0)]
    right_fit_x = right_fitx[(right_fitx <= img_width - 1) & (right_fitx >= 0)]
    left_fit_y = np.linspace(img_height - len(left_fit_x), img_height - 1, len(left_fit_x))
    right_fit_y = np.linspace(img_height - len(right_fit_x), img_height - 1, len(right_fit_x))
    return (left_fit_y.astype(np.int64), left_fit_x.astype(np.int64),
            right_fit_y.astype(np.int64), right_fit_x.astype(np.int64))


def check_validity(warped_size, lf, rf, limits=None):
    """lane_tracker.py:561-627 -> bool (the reference stores it in self.valid_lane_lines).  `limits`:
    the seven numbers upstream hard-codes at :588-593, :617 (tracker_settings.md lists them per demo)."""
    L = dict(min_dist_y1=150, max_dist_y1=230, min_dist_y2=110, max_dist_y2=230, min_dist_y3=80, max_dist_y3=200,
             thresh=0.25)
    L.update(limits or {})
    ly, _, ry, _ = get_poly_points(warped_size, lf, rf)
    n = min(len(ly), len(ry))
    y1 = warped_size[0] - 1
    y2 = warped_size[0] - int(n * 0.35)
    y3 = warped_size[0] - int(n * 0.75)
    ev = lambda c, y: c[0] * (y ** 2) + c[1] * y + c[2]
    d1, d2, d3 = abs(ev(lf, y1) - ev(rf, y1)), abs(ev(lf, y2) - ev(rf, y2)), abs(ev(lf, y3) - ev(rf, y3))
    if ((d1 < L["min_dist_y1"]) | (d1 > L["max_dist_y1"]) | (d2 < L["min_dist_y2"]) | (d2 > L["max_dist_y2"])
            | (d3 < L["min_dist_y3"]) | (d3 > L["max_dist_y3"])):
        return False
    s = lambda c, y: 2 * c[0] * y + c[1]
    norm1, norm2 = abs(s(lf, y1) - s(rf, y1)), abs(s(lf, y3) - s(rf, y3))
    return not ((norm1 >= L["thresh"]) | (norm2 >= L["thresh"]))


# ---- presentation stage (draw_lane :629-662, create_split_view utils.py:57-103; cv2, unpinned) ------
def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.c_void_p)


def fill_poly(img, pts, color):
    """cv2.fillPoly(img, [pts], color) in place for one polygon; pts = (n, 2) (x, y)."""
    assert img.dtype == np.uint8 and img.flags.c_contiguous
    ch = 1 if img.ndim == 2 else img.shape[2]
    pts, pp = _i32(np.asarray(pts).reshape(-1, 2))
    col = (C.c_uint8 * 4)(*([int(c) for c in np.atleast_1d(color)] + [0] * 4)[:4])
    lib().lto_fill_poly(img.ctypes.data_as(C.c_void_p), img.shape[0], img.shape[1], ch, pp, len(pts), col)
    return img


def add_weighted(a, alpha, b, beta, gamma):
    a, pa = _u8(a)
    b, pb = _u8(b)
    assert a.shape == b.shape
    out = np.empty_like(a)
    f = lib().lto_add_weighted_u8
    f.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_double, C.c_double, C.c_size_t, C.c_void_p]
    f(pa, alpha, pb, beta, gamma, a.size, out.ctypes.data_as(C.c_void_p))
    return out


def draw_lane(calib, Minv, img, left_y, left_x, right_y, right_x):
    """The lane overlay of draw_lane() without the text lines."""
    img, pi = _u8(img)
    ly, ply = _i32(left_y)
    lx, plx = _i32(left_x)
    ry, pry = _i32(right_y)
    rx, prx = _i32(right_x)
    out = np.empty_like(img)
    mv = (C.c_double * 9)(*np.asarray(Minv, np.float64).ravel())
    lib().lto_draw_lane(C.byref(calib), mv, pi, ply, plx, len(ly), pry, prx, len(ry), out.ctypes.data_as(C.c_void_p))
    return out


def resize_linear(img, dsize):
    """cv2.resize(img, dsize=(w, h)) with the default INTER_LINEAR, u8."""
    img, pi = _u8(img)
    ch = 1 if img.ndim == 2 else img.shape[2]
    dw, dh = int(dsize[0]), int(dsize[1])
    out = np.empty((dh, dw) if img.ndim == 2 else (dh, dw, ch), np.uint8)
    lib().lto_resize_linear_u8(pi, img.shape[0], img.shape[1], ch, dh, dw, out.ctypes.data_as(C.c_void_p))
    return out
