"""Host side of the MI355X lane tracker: the reference's `LaneTracker` API and per-stream state
machine (reference lane_tracker.py:85-181, 795-1209) on top of the HIP kernel chain.

What runs where
  GPU (liblane_tracker_amd.so): undistort, perspective warp, filter_lane_points, the sliding-window
      and band searches and the polynomial fit -- the hot path.
  Host (this file): the ~20 scalars of cross-frame state, the two-try policy, check_validity and
      get_poly_points (a few f64 operations per frame; BASELINE north_star keeps them on the host),
      curve radius / eccentricity, and the lane overlay.

There is no CPU implementation of the hot path in this package: without the shared library and a
GPU, constructing a LaneTracker raises.
"""
import contextlib
import math
import os

import numpy as np

from . import hostcpu as _hostcpu
from . import _native
from . import overlay as _overlay
from . import utils as _utils
from .stream import StreamPipeline, _PackedPoly, _pack_deferred      # noqa: F401  (the window / stream pipeline: a mix-in)

__all__ = ["LaneTracker", "bilateral_adaptive_threshold"]

_default_ctx = None
_live_trackers = 0          # LaneTracker instances not yet closed: the last one to close trims the process-wide frame pool


def _context_for_module_functions():
    """Context used by the module-level `bilateral_adaptive_threshold` (it needs only a stream)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = _native.Context((2, 2), (2, 2), np.eye(3), np.zeros(5), np.eye(3), device=0, capacity=1)
    return _default_ctx


def bilateral_adaptive_threshold(img, ksize=30, C=0, mode='floor', true_value=255, false_value=0):
    """Cross-shaped adaptive threshold of a single-channel u8 image (reference lane_tracker.py:14-83).

    A pixel passes in 'floor' mode iff it is brighter by more than C than the mean of its `ksize`
    left AND right neighbours, or of its `ksize` upper AND lower neighbours (zeros beyond the
    border); 'ceil' mode tests darker instead.  Returns a u8 mask of `true_value` / `false_value`.
    """
    if mode not in ('floor', 'ceil'):
        raise ValueError("Unexpected mode value. Expected value is 'floor' or 'ceil'.")
    return _context_for_module_functions().bilateral_adaptive_threshold(
        img, ksize, C, 0 if mode == 'floor' else 1, true_value, false_value)


def _as_index(v):
    """The 2017 NumPy the reference ran on truncated float slice bounds / counts with int()."""
    return int(v)


class LaneTracker(StreamPipeline):
    """Tracks the left and right lines of the ego lane across the frames of one video stream.

    Drop-in for the reference class: same constructor, same `process()` signature and defaults,
    same public attributes.  One instance per stream; not thread-safe (reference :85-99).
    Extra keyword `device` selects the GPU.
    """

    def __init__(self, img_size, warped_size, cam_matrix, dist_coeffs, warp_matrices, mpp_conversion,
                 n_fail=8, n_reset=4, n_average=2, print_frame_count=False, device=0):
        self.img_size = img_size
        self.warped_size = warped_size
        self.cam_matrix = cam_matrix
        self.dist_coeffs = dist_coeffs
        self.M = warp_matrices[0]
        self.Minv = warp_matrices[1]
        self.mppv = mpp_conversion[0]
        self.mpph = mpp_conversion[1]
        self.n_reset = n_reset
        self.n_fail = n_fail
        self.n_average = n_average
        self.print_frame_count = print_frame_count

        # cross-frame state (reference :139-176)
        self.last_detection = n_reset + 1
        self.detected_pixels = False
        self.valid_lane_lines = False
        self.left_fit_coeffs = []
        self.right_fit_coeffs = []
        self.last_left_coeffs = None
        self.last_right_coeffs = None
        self.left_avg_coeffs = None
        self.right_avg_coeffs = None
        self.left_avg_y = np.array([])
        self.left_avg_x = np.array([])
        self.right_avg_y = np.array([])
        self.right_avg_x = np.array([])
        # lane pixels and window centroids of the last search: public attributes like upstream's, fetched from
        # the device on first read (properties below)
        self._lp = dict(left_y=None, left_x=None, right_y=None, right_x=None, left_window_centroids=None,
                        right_window_centroids=None)
        self.left_curve_radius = None
        self.right_curve_radius = None
        self.average_curve_radius = None
        self.average_curve_radii = []
        self.eccentricity = None
        self.counter = 0
        self.success = 0

        # device side
        self.device = device
        self._ctx = _native.Context(img_size, warped_size, cam_matrix, dist_coeffs, self.M, device=device, capacity=2)
        global _live_trackers
        _live_trackers += 1
        self._closed = False
        _hostcpu.find_blas_pools()       # (the one-time search for NumPy's BLAS library -- 0.1 s -- at set-up, not inside a frame's refit)
        self._slot = 0              # process() alternates between two slots (see process())
        self._aux_ctx = {}          # contexts for images that are not the calibration's BEV size
        self._fit = None            # (left_y array, right_y array, left coeffs, right coeffs) of the last search
        self._pending = None        # (ctx, slot): lane-pixel lists of the last search that found pixels, still on the device
        self._pending_cent = None   # (ctx, slot): window centroids of the last SLIDING-WINDOW search that found pixels (a band
                                    # search leaves them alone, reference :434-440 vs :492-495), still on the device
        self._overlay_ready = False
        self._have_font = False
        self._resident = None       # (frame array, slot) of the camera frame last uploaded to the main context
        self._in_stream = False     # a process_stream() generator is active: its windows own the context's slots
        # the presentation stage's tables now, not at the first annotated frame: lt_overlay_configure may widen the run of camera
        # rows the uploads bring by a row or two (so that the lane's rows need no upload of their own), and a frame uploaded before
        # that would lack them
        self._configure_overlay()

    # ------------------------------------------------------------------------------------------
    def get_success_ratio(self):
        return self.success / self.counter, self.success, self.counter


    def _lane_pixel_property(name, centroids=False):
        def get(self):
            (self._materialise_centroids if centroids else self._materialise_pixels)()
            return self._lp[name]

        def put(self, value):
            (self._materialise_centroids if centroids else self._materialise_pixels)()
            self._lp[name] = value
        return property(get, put, doc="lane pixels / window centroids of the last search (reference :434-440, :492-495)")

    left_y = _lane_pixel_property("left_y")
    left_x = _lane_pixel_property("left_x")
    right_y = _lane_pixel_property("right_y")
    right_x = _lane_pixel_property("right_x")
    left_window_centroids = _lane_pixel_property("left_window_centroids", True)
    right_window_centroids = _lane_pixel_property("right_window_centroids", True)
    del _lane_pixel_property

    # ---- the cross-frame state as data (reference :139-176; SURVEY.md section 5: "~20 scalars + two lists") ---------------
    STATE_VERSION = 1
    _STATE_SCALARS = ("last_detection", "detected_pixels", "valid_lane_lines", "left_curve_radius", "right_curve_radius",
                      "average_curve_radius", "eccentricity", "counter", "success")
    _STATE_VECTORS = ("last_left_coeffs", "last_right_coeffs", "left_avg_coeffs", "right_avg_coeffs")          # f64 or None
    _STATE_INT_ARRAYS = ("left_avg_y", "left_avg_x", "right_avg_y", "right_avg_x")                             # int64 (or the empty f64 array of a new tracker)
    _STATE_PIXELS = ("left_y", "left_x", "right_y", "right_x")

    def get_state(self):
        """Everything one frame hands to the next (the attributes of reference :139-176), as a plain dict of Python numbers and
        lists -- `json.dumps` takes it as it is, f64 values survive the round trip exactly.  A tracker built with the same
        constructor arguments, in this or another process, continues the stream after `set_state()` with the results this one
        would have produced (tests/test_gpu_state.py).  Call it between frames / windows, not inside a `process_stream()`."""
        if self._in_stream:
            raise RuntimeError("get_state() inside an active process_stream(): exhaust or close the generator first")
        self._materialise_pending()          # the lane pixels / centroids of the last search leave the device

        def num(v):
            if v is None:
                return None
            if isinstance(v, (bool, np.bool_)):
                return bool(v)
            if isinstance(v, (int, np.integer)):
                return int(v)
            return float(v)

        def arr(v):
            return None if v is None else np.asarray(v).tolist()
        st = {"version": self.STATE_VERSION, "n_fail": num(self.n_fail), "n_reset": num(self.n_reset), "n_average": num(self.n_average),
              "img_size": [int(v) for v in self.img_size], "warped_size": [int(v) for v in self.warped_size]}
        for k in self._STATE_SCALARS:
            st[k] = num(getattr(self, k))
        for k in self._STATE_VECTORS + self._STATE_INT_ARRAYS:
            st[k] = arr(getattr(self, k))
        st["left_fit_coeffs"] = [arr(c) for c in self.left_fit_coeffs]       # history: 3 values, or [] for a failed frame (:1142-1147)
        st["right_fit_coeffs"] = [arr(c) for c in self.right_fit_coeffs]
        st["average_curve_radii"] = [int(v) for v in self.average_curve_radii]
        for k in self._STATE_PIXELS:
            st[k] = arr(self._lp[k])
        for k in ("left_window_centroids", "right_window_centroids"):
            v = self._lp[k]
            st[k] = None if v is None else [num(c) for c in v]
        f = self._fit                        # the fit of the last search that found pixels (fit_poly() hands it out)
        st["last_search_fit"] = None if f is None else [arr(f[2]), arr(f[3])]
        st["outage_group"] = int(self._outage_group)
        st["avg_partial"] = None             # the `partial` the plot points of the averages (left_avg_x, ...) were formed with
        if self._avg_packed is not None:
            for (partial, _), b in self.__dict__.get("_packed", {}).items():
                if b is self._avg_packed[0]:
                    st["avg_partial"] = num(partial)
        return st

    def set_state(self, state):
        """Continue the stream `state` (a `get_state()` dict) was taken from.  The tracker must have been built for the same
        geometry and history lengths; ValueError otherwise."""
        if self._in_stream:
            raise RuntimeError("set_state() inside an active process_stream()")
        if state.get("version") != self.STATE_VERSION:
            raise ValueError("unknown tracker state version %r" % (state.get("version"),))
        for k in ("n_fail", "n_reset", "n_average"):
            if state[k] != getattr(self, k):
                raise ValueError("state was taken with %s=%r, this tracker has %r" % (k, state[k], getattr(self, k)))
        if list(state["img_size"]) != [int(v) for v in self.img_size] or list(state["warped_size"]) != [int(v) for v in self.warped_size]:
            raise ValueError("state was taken from a tracker of another geometry")
        self._all_copies_done()
        self._pending = self._pending_cent = None
        for k in self._STATE_SCALARS:
            setattr(self, k, state[k])
        for k in self._STATE_VECTORS:
            setattr(self, k, None if state[k] is None else np.array(state[k], np.float64))
        for k in self._STATE_INT_ARRAYS:
            v = state[k]
            setattr(self, k, np.array([]) if not v else np.array(v, np.int64))
        self.left_fit_coeffs = [np.array(c, np.float64) for c in state["left_fit_coeffs"]]
        self.right_fit_coeffs = [np.array(c, np.float64) for c in state["right_fit_coeffs"]]
        self.average_curve_radii = [int(v) for v in state["average_curve_radii"]]
        for k in self._STATE_PIXELS:
            self._lp[k] = None if state[k] is None else np.array(state[k], np.int64)
        for k in ("left_window_centroids", "right_window_centroids"):
            self._lp[k] = None if state[k] is None else list(state[k])
        f = state["last_search_fit"]
        self._fit = None if f is None else (self._lp["left_y"], self._lp["right_y"], np.array(f[0], np.float64), np.array(f[1], np.float64))
        self._outage_group = int(state.get("outage_group", 4))
        # the packed polygon of the averages, which a failed frame redraws (draw_lane): formed again from the coefficients --
        # the same lt_poly_points call that left left_avg_x & co. behind
        self._avg_packed = None
        if state.get("avg_partial") is not None and self.left_avg_coeffs is not None and self.right_avg_coeffs is not None:
            b = self._points_packed(self.left_avg_coeffs, self.right_avg_coeffs, state["avg_partial"], 'avg0')
            nl, nr = int(b[1][0]), int(b[1][1])
            if np.array_equal(b[2][:nl, 1], self.left_avg_x) and np.array_equal(b[3][:nr, 1], self.right_avg_x):
                self._avg_packed = (b, self.left_avg_x, self.right_avg_x)
        self._lane_in_flight = self._device_lane = None
        self._resident = None

    def close(self):
        try:
            self._all_copies_done()
            if self._group is not None:
                _native.host_copy_group_release(self._group)
                self._group = None
        except Exception:
            pass
        self._ctx.close()
        for c in self._aux_ctx.values():
            c.close()
        self._aux_ctx = {}
        if not getattr(self, "_closed", True):
            global _live_trackers
            self._closed = True
            _live_trackers -= 1
            if _live_trackers <= 0:      # nobody left to hand windows to: the pool of output blocks goes back to the system
                _native.frames_trim()

    # ---- device plumbing ----------------------------------------------------------------------
    def _ctx_for_plane(self, h, w):
        if (w, h) == (self._ctx.warp_w, self._ctx.warp_h):
            return self._ctx
        key = (h, w)
        if key not in self._aux_ctx:
            self._aux_ctx[key] = _native.Context((2, 2), (w, h), np.eye(3), np.zeros(5), np.eye(3),
                                                 device=self.device, capacity=1)
        return self._aux_ctx[key]

    def _collect_search(self, ctx, want_centroids, slot=0, lazy=False):
        """Pull the lane record of `slot`.  The pixel lists (self.left_y/left_x/right_y/right_x) and
        window centroids are fetched right away, or -- in the stream pipeline, `lazy=True` -- only if
        somebody reads them before the slot is reused (the state machine itself needs the record only)."""
        lf, rf, self.detected_pixels, flags = ctx.download_record(slot)
        self._fit_flags = flags
        if not self.detected_pixels:
            self._fit = None        # like the reference, a failed search leaves the previous pixel lists in place
            return
        self._pending = (ctx, slot)
        if want_centroids:
            self._pending_cent = (ctx, slot)
        if lazy and not flags:
            self._fit = ("pending", None, lf, rf)
            return
        self._materialise_pending()
        if flags & 1:
            lf = _minimum_norm_parabola(self._lp['left_y'], self._lp['left_x'])
        if flags & 2:
            rf = _minimum_norm_parabola(self._lp['right_y'], self._lp['right_x'])
        self._fit = (self._lp['left_y'], self._lp['right_y'], lf, rf)

    def _materialise_pixels(self):
        """Download the lane pixels of the most recent lazily collected search."""
        if self._pending is not None:
            ctx, slot = self._pending
            self._pending = None
            # The search that wrote these lists is complete (the host has seen its record), so the copy need not wait for
            # anything: in urgent mode a download waits for its own stream only -- outside it, for every stream of the context,
            # i.e. for all the uploads and masks queued ahead in a stream (7 ms per call there; get_curve_radius asks for the
            # lists when a radius lies within 1e-8 of an integer, which near-straight lanes do for frames on end).
            scope = ctx.urgent() if hasattr(ctx, "urgent") else contextlib.nullcontext()
            with scope:
                try:
                    if not hasattr(ctx, "download_lane_lists"):
                        raise _native.NativeError("no combined download")
                    self._lp['left_y'], self._lp['left_x'], self._lp['right_y'], self._lp['right_x'] = ctx.download_lane_lists(slot, False)[:4]
                except _native.NativeError:
                    self._lp['left_y'], self._lp['left_x'] = ctx.download_pixels(slot, 0)
                    self._lp['right_y'], self._lp['right_x'] = ctx.download_pixels(slot, 1)
            if self._fit is not None and isinstance(self._fit[0], str):
                self._fit = (self._lp['left_y'], self._lp['right_y'], self._fit[2], self._fit[3])

    def _materialise_centroids(self):
        """Download the window centroids of the most recent lazily collected sliding-window search."""
        if self._pending_cent is not None:
            ctx, slot = self._pending_cent
            self._pending_cent = None
            self._lp['left_window_centroids'] = ctx.download_centroids(slot, 0)
            self._lp['right_window_centroids'] = ctx.download_centroids(slot, 1)

    def _materialise_pending(self):
        # the usual case -- the lists of ONE sliding-window search, pixels and centroids of the same slot -- is one round trip to
        # the device instead of fourteen (lt_download_lane_lists)
        p, q = self._pending, self._pending_cent
        if p is not None and q is not None and p[0] is q[0] and p[1] == q[1] and hasattr(p[0], "download_lane_lists"):
            ctx, slot = p
            self._pending = self._pending_cent = None
            scope = ctx.urgent() if hasattr(ctx, "urgent") else contextlib.nullcontext()
            try:
                with scope:
                    ly, lx, ry, rx, cl, cr = ctx.download_lane_lists(slot, True)
            except _native.NativeError:          # (a list region beyond the staging buffer: the two calls)
                self._pending, self._pending_cent = p, q
            else:
                self._lp['left_y'], self._lp['left_x'], self._lp['right_y'], self._lp['right_x'] = ly, lx, ry, rx
                self._lp['left_window_centroids'], self._lp['right_window_centroids'] = cl, cr
                if self._fit is not None and isinstance(self._fit[0], str):
                    self._fit = (self._lp['left_y'], self._lp['right_y'], self._fit[2], self._fit[3])
                return
        self._materialise_pixels()
        self._materialise_centroids()

    # ---- filter_lane_points (reference :183-240) ------------------------------------------------
    def filter_lane_points(self, img, filter_type='bilateral', ksize_r=25, C_r=8, ksize_b=35, C_b=5,
                           mask_noise=False, ksize_noise=65, C_noise=10, noise_thresh=135):
        """RGB bird's-eye image -> binary lane mask {0,255}: R and Lab-b planes, 29x29 / 55x55
        elliptical top-hats, bilateral (or box-mean 'neighborhood') thresholds, OR-merge with the
        optional greenery mask, 5x5 elliptical open."""
        fp = _native.filter_params(filter_type, ksize_r, C_r, ksize_b, C_b, mask_noise, noise_thresh,
                                   ksize_noise, C_noise)
        return self._ctx.filter_lane_points(img, fp)

    # ---- searches (reference :242-500) --------------------------------------------------------------
    def sliding_window_search(self, img, window_width, window_height, search_range, mu, no_success_limit,
                              start_slice=0.25, ignore_sides=360, ignore_bottom=30, partial=1, diagnostics=False):
        """Bottom-up two-lane window search on a binary bird's-eye image; stores left/right pixel
        index arrays, window centroids and `detected_pixels`."""
        if diagnostics:
            print("Using sliding window search.")
        img = np.ascontiguousarray(img, np.uint8)
        ctx = self._ctx_for_plane(*img.shape[:2])
        ctx.upload_masks(img)
        self._search_uploaded(ctx, 'sws', dict(window_width=window_width, window_height=window_height,
                                               search_range=search_range, mu=mu, no_success_limit=no_success_limit,
                                               start_slice=start_slice, ignore_sides=ignore_sides,
                                               ignore_bottom=ignore_bottom, partial=partial), diagnostics)

    def band_search(self, img, bandwidth, ignore_bottom=30, partial=1, diagnostics=False):
        """Collects the non-zero pixels within `bandwidth` px of the last valid left/right fits."""
        if diagnostics:
            print("Using band search.")
        img = np.ascontiguousarray(img, np.uint8)
        ctx = self._ctx_for_plane(*img.shape[:2])
        ctx.upload_masks(img)
        self._search_uploaded(ctx, 'bs', dict(bandwidth=bandwidth, ignore_bottom=ignore_bottom, partial=partial),
                              diagnostics)

    def _search_uploaded(self, ctx, mode, kw, diagnostics, slot=0, lazy=False, between=None):
        """Launch the search over the mask in `slot` and collect its record; `between` (a callable) runs after the launch and
        before the wait -- host work that hides under the device's."""
        pix_here = self._pending is not None and self._pending[0] is ctx and self._pending[1] == slot
        cent_here = mode == 'sws' and self._pending_cent is not None and self._pending_cent[0] is ctx and self._pending_cent[1] == slot
        if pix_here and cent_here:
            self._materialise_pending()      # this search reuses the slot whose lists were not fetched yet: both in one round trip
        elif pix_here:
            self._materialise_pixels()
        elif cent_here:
            self._materialise_centroids()    # ... and a sliding-window search rewrites the slot's centroid lists
        if mode == 'sws':
            ctx.sws_fit_run(1, _native.search_params(**kw), first=slot)
        else:
            prev = np.concatenate([np.asarray(self.last_left_coeffs, np.float64).reshape(3),
                                   np.asarray(self.last_right_coeffs, np.float64).reshape(3)])
            ctx.band_fit_run(1, prev, _native.search_params(**kw), first=slot)
        if between is not None:
            between()
        self._collect_search(ctx, want_centroids=(mode == 'sws'), slot=slot, lazy=lazy)
        if diagnostics:
            print("Lane pixels found." if self.detected_pixels else "No lane pixels found.")

    # ---- fit (reference :502-509) ---------------------------------------------------------------------
    def fit_poly(self):
        """Second-degree least-squares fits x(y) of the current left/right lane pixels.  The fit is
        produced by the search kernel (exact int64 moments, 3x3 normal equations in f64)."""
        if self._fit is not None and (self._fit[0] == "pending" if isinstance(self._fit[0], str)
                                      else (self._fit[0] is self._lp['left_y'] and self._fit[1] is self._lp['right_y'])):
            return self._fit[2].copy(), self._fit[3].copy()
        # pixel arrays were replaced by the caller: fit them on the device from the lists
        lf = self._ctx.fit_poly2(self._lp['left_y'], self._lp['left_x'])
        rf = self._ctx.fit_poly2(self._lp['right_y'], self._lp['right_x'])
        return lf, rf

    # ---- host geometry (reference :511-528, 561-627) ---------------------------------------------------
    def _plot_rows(self, partial):
        """ploty and ploty ** 2 of get_poly_points for this `partial` (they only depend on the image height)."""
        cache = self.__dict__.setdefault("_ploty_cache", {})
        if partial not in cache:
            img_height = self.warped_size[1]
            ploty = np.linspace(img_height * (1 - partial), img_height - 1, _as_index(img_height * partial))
            cache[partial] = (ploty, ploty ** 2)
        return cache[partial]

    def _poly_inside(self, coeffs, partial):
        """x values of one parabola over the plot rows and the mask of those inside [0, W-1] (reference :519-524)."""
        ploty, ploty2 = self._plot_rows(partial)
        fitx = coeffs[0] * ploty2 + coeffs[1] * ploty + coeffs[2]
        return fitx, (fitx <= self.warped_size[0] - 1) & (fitx >= 0)

    _want_out = False      # process() is under way and will hand back an annotated frame ...
    _out = None            # ... which lands in this page-locked array, fetched while the device works
    _out_ahead = False     # ... and those runs are text rows, then lane rows: the lane can be drawn ahead of the text
    _out_rows = None       # address of the row runs still to come from the device when the other rows of _out are filled already
    _avg_packed = None     # (buffers, left_avg_x, right_avg_x) while the 'avg' buffers hold the polygon of left_avg_* / right_avg_*

    def _points_packed(self, left_fit_coeffs, right_fit_coeffs, partial, purpose):
        """get_poly_points through lt_poly_points (host C, the same f64 operations in the same order; pinned to the reference's
        fixtures by tests/test_host_geometry.py) into buffers this tracker keeps per (partial, purpose) -> the buffers:
        [0] the six coefficients, [1] int32 counts (left, right), [2] / [3] the left / right (y, x) pairs, [4] addresses."""
        b = self._packed_buffers(partial, purpose)
        co = b[0]
        co[:3] = left_fit_coeffs
        co[3:] = right_fit_coeffs
        rc = b[5](*b[4])
        if rc:
            _native._check(rc)
        return b

    def _packed_buffers(self, partial, purpose):
        """The buffers of _points_packed for (partial, purpose), created on first use (nothing is computed)."""
        packed = self.__dict__.setdefault("_packed", {})
        b = packed.get((partial, purpose))
        if b is None:
            ploty, ploty2 = self._plot_rows(partial)
            rows = len(ploty)
            co, cnt = np.empty(6, np.float64), np.zeros(2, np.int32)
            lyx, ryx = np.empty((max(rows, 1), 2), np.int32), np.empty((max(rows, 1), 2), np.int32)
            args = (int(self.warped_size[0]), int(self.warped_size[1]), co.ctypes.data, 1, ploty.ctypes.data, ploty2.ctypes.data,
                    rows, cnt.ctypes.data, cnt.ctypes.data + 4, lyx.ctypes.data, ryx.ctypes.data)
            b = packed[(partial, purpose)] = (co, cnt, lyx, ryx, args, _native.load().lt_poly_points, (ploty, ploty2))
        return b

    def get_poly_points(self, left_fit_coeffs, right_fit_coeffs, partial=1):
        img_height = self.warped_size[1]
        left_fitx, lin = self._poly_inside(left_fit_coeffs, partial)
        right_fitx, rin = self._poly_inside(right_fit_coeffs, partial)
        left_fit_x, right_fit_x = left_fitx[lin], right_fitx[rin]
        # np.linspace(H - n, H - 1, n) of the reference is exactly the integers H - n ... H - 1
        return (np.arange(img_height - len(left_fit_x), img_height, dtype=np.int64), left_fit_x.astype(np.int64),
                np.arange(img_height - len(right_fit_x), img_height, dtype=np.int64), right_fit_x.astype(np.int64))

    # separation limits at y1, y2, y3 and the tangent threshold: the values hard-coded upstream
    # (:588-593, :617; the "Demo 2" set of tracker_settings.md).  Overridable per instance.
    validity_limits = dict(min_dist_y1=150, max_dist_y1=230, min_dist_y2=110, max_dist_y2=230,
                           min_dist_y3=80, max_dist_y3=200, thresh=0.25)

    def check_validity(self, left_fit_coeffs, right_fit_coeffs, diagnostics=False):
        lim = self.validity_limits
        # only the number of plot points inside the image matters here (:565-569)
        cnt = self._points_packed(left_fit_coeffs, right_fit_coeffs, 1, 'validity')[1]
        n = min(int(cnt[0]), int(cnt[1]))
        # NB: the reference takes the image WIDTH as the bottom y (:571-573); reproduced as is
        y1 = self.warped_size[0] - 1
        y2 = self.warped_size[0] - int(n * 0.35)
        y3 = self.warped_size[0] - int(n * 0.75)
        # Python floats from here on: the same IEEE doubles and the same operations in the same order as NumPy's f64 scalars,
        # at a fifth of their cost per operation
        lf, rf = np.asarray(left_fit_coeffs, np.float64).tolist(), np.asarray(right_fit_coeffs, np.float64).tolist()
        at = lambda c, y: c[0] * (y ** 2) + c[1] * y + c[2]
        x1_diff, x2_diff, x3_diff = abs(at(lf, y1) - at(rf, y1)), abs(at(lf, y2) - at(rf, y2)), abs(at(lf, y3) - at(rf, y3))
        if ((x1_diff < lim['min_dist_y1']) or (x1_diff > lim['max_dist_y1']) or (x2_diff < lim['min_dist_y2'])
                or (x2_diff > lim['max_dist_y2']) or (x3_diff < lim['min_dist_y3']) or (x3_diff > lim['max_dist_y3'])):
            self.valid_lane_lines = False
            if diagnostics:
                print("No valid lane lines found, violated distance criterion: "
                      "x1_diff == {:.2f}, x2_diff == {:.2f}, x3_diff == {:.2f}".format(x1_diff, x2_diff, x3_diff))
            return
        slope = lambda c, y: 2 * c[0] * y + c[1]
        norm1 = abs(slope(lf, y1) - slope(rf, y1))
        norm2 = abs(slope(lf, y3) - slope(rf, y3))
        if (norm1 >= lim['thresh']) or (norm2 >= lim['thresh']):
            self.valid_lane_lines = False
            if diagnostics:
                print("No valid lane lines found, violated tangent criterion: norm1 == {:.3f}, norm2 == {:.3f}".format(norm1, norm2))
        else:
            self.valid_lane_lines = True
            if diagnostics:
                print("Valid lane lines found. Tangents: norm1 == {:.3f}, norm2 == {:.3f}. Distance: x1_diff == {:.2f}, "
                      "x2_diff == {:.2f}, x3_diff == {:.2f}".format(norm1, norm2, x1_diff, x2_diff, x3_diff))

    # ---- metrics (reference :530-559) -------------------------------------------------------------------
    def get_curve_radius(self):
        """Curve radius in metres, the reference's integers (:530-549).  The reference refits the pixels in metric
        units with two more np.polyfit calls.  A least-squares parabola is equivariant under axis scaling, so the
        metric coefficients follow from the pixel fit (a_m = a*mpph/mppv^2, b_m = b*mpph/mppv) up to floating-point
        rounding (<= 1.6e-10 relative over 3000 random lanes, nearly straight ones included); only when that value lies within
        1e-8 relative of an integer, so that `int()` could truncate
        differently are the lane pixels fetched and refitted exactly as upstream does."""
        lf, rf = self.fit_poly()
        y_eval = self.warped_size[1]

        def radius_of(fit_m):
            return ((1 + (2 * fit_m[0] * y_eval * self.mppv + fit_m[1]) ** 2) ** 1.5) / np.absolute(2 * fit_m[0])

        def quick(c):
            # radius_of on Python floats: the same doubles, the same operations (NumPy's f64 scalars call the same pow);
            # anything NumPy answers with inf / nan and a warning where Python raises goes the NumPy way
            mppv, mpph = self.mppv, self.mpph
            if type(mppv) is not float or type(mpph) is not float:
                return None
            a, b = float(c[0]) * mpph / (mppv ** 2), float(c[1]) * mpph / mppv
            try:
                val = ((1 + (2 * a * y_eval * mppv + b) ** 2) ** 1.5) / abs(2 * a)
            except (ZeroDivisionError, OverflowError):
                return None
            return val if math.isfinite(val) else None
        radii = []
        for side, c in enumerate((lf, rf)):
            val = quick(c)
            if val is None:
                val = radius_of((c[0] * self.mpph / (self.mppv ** 2), c[1] * self.mpph / self.mppv))
            if math.isfinite(val) and abs(val - round(val)) <= 1e-8 * max(1.0, abs(val)):
                ys, xs = (self.left_y, self.left_x) if side == 0 else (self.right_y, self.right_x)
                with _hostcpu.blas_limited():     # (LAPACK on no more threads than the CPUs granted, for this call only: hostcpu.py)
                    val = radius_of(np.polyfit(np.asarray(ys) * self.mppv, np.asarray(xs) * self.mpph, 2))
            radii.append(int(val))
        self._note_radii(radii)

    def _note_radii(self, radii):
        """left / right radius of this frame -> the running average over the window (reference :544-549)."""
        self.left_curve_radius, self.right_curve_radius = radii
        average_curve_radius = int(0.5 * (self.left_curve_radius + self.right_curve_radius))
        self.average_curve_radii.append(average_curve_radius)
        if len(self.average_curve_radii) > self.n_average:
            self.average_curve_radii.pop(0)
        real_curve_radii = [radius for radius in self.average_curve_radii if radius > 0]
        total = sum(real_curve_radii)
        # np.average of a few Python ints: exact sum as f64, one division -- the same value without the array round trip
        self.average_curve_radius = int(total / len(real_curve_radii)) if real_curve_radii and total < 2 ** 53 \
            else int(np.average(real_curve_radii))

    def get_eccentricity(self):
        left, right = self.left_avg_x[-1], self.right_avg_x[-1]
        mid = int(self.warped_size[0] / 2)
        self.eccentricity = (((mid - left) - (right - mid)) / 2) * self.mpph

    # ---- presentation (reference :629-793) ------------------------------------------
    def _lane_text(self):
        lines = ["Curve Radius: {} m".format(self.average_curve_radius),
                 "Eccentricity: {:.2f} m".format(self.eccentricity)]
        if self.print_frame_count:
            lines.append("Frame: {}".format(self.counter - 1))
        return lines

    def _failure_text(self):
        lines = ["Lane Line Detection Failed"]
        if self.print_frame_count:
            lines.append("Frame: {}".format(self.counter - 1))
        return lines

    def _configure_overlay(self):
        if not self._overlay_ready:
            self._ctx.overlay_configure(self.Minv)
            font = _overlay.font_atlas()
            self._have_font = font is not None
            if font is not None:
                self._ctx.overlay_set_font(font[0], font[1], font[2])
            self._overlay_ready = True

    def _annotate(self, polygons, texts, first=0):
        """Lane polygons (an empty one = plain copy) and text lines for the frames resident in slots
        first .. first + n - 1 -> the annotated frames (n, H, W, 3)."""
        ctx = self._ctx
        ctx.overlay_run(polygons, first=first)
        if self._have_font:
            ctx.overlay_text(texts, first=first)
        return ctx.download_overlay(len(polygons), first=first)

    def _overlay_slot(self, img):
        """Slot of the main context that holds `img` (uploading it if it is not the resident frame)."""
        self._configure_overlay()
        if self._resident is not None and self._resident[0] is img:
            return self._resident[1]
        self._ctx.upload_frames(img, first=0)
        self._resident, self._resident_partial = (img, 0), False
        return 0

    def draw_lane(self, img):
        """Highlight the lane between the averaged curves and print radius / eccentricity.  The polygon
        fill, its inverse warp with `Minv`, the 0.3 blend and the text run on the GPU (lt_overlay_run / lt_overlay_text); unlike
        upstream the caller's array is not written to."""
        slot = self._overlay_slot(img)
        ap = self._avg_packed
        if ap is not None and ap[1] is self.left_avg_x and ap[2] is self.right_avg_x:
            # the polygon of the averages is still packed where _record_success left it
            b = ap[0]
            a = b[4]
            return self._present(slot, a[7], a[8], a[9] if b[1][0] else None, a[10] if b[1][1] else None, self._lane_text(),
                                 drawn=self._lane_in_flight is b)
        return self._annotate([(self.left_avg_y, self.left_avg_x, self.right_avg_y, self.right_avg_x)], [self._lane_text()],
                              first=slot)[0]

    def _present_rows(self):
        """The two runs of camera rows an annotated frame can differ from its camera frame in -- the text lines and the rows the
        lane can reach (lt_overlay_rows) -- as four int32 {a0, a1, b0, b1}, or None when they cover (nearly) the whole frame.
        -> (array, its address, [a0, a1, b0, b1], lane rows behind text rows?, host-text: None or (lane-only runs {0, 0, l0, l1}:
        array, address; (l0, l1); (t0, t1); do rows of the lane run lie outside the rows the mask chain uploads?))."""
        if "_rows4" not in self.__dict__:
            self._configure_overlay()
            H = self.img_size[1]
            lane = self._ctx.overlay_rows()
            text = (0, 0)
            if self._have_font:
                gh = _overlay.font_atlas()[0].shape[1]
                text = (max(self._TEXT_ORIGIN[1], 0), min(self._TEXT_ORIGIN[1] + 2 * self._TEXT_STEP + gh, H))   # up to three lines
            if text[1] <= lane[0]:                          # the text above the lane rows (or no text): two runs, in this order --
                runs, ahead = [text, lane], True            # what drawing the lane ahead of the text needs (lt_present_lane_async)
            elif lane[1] <= text[0]:
                runs, ahead = [lane, text], False
            else:
                runs, ahead = [(min(text[0], lane[0]), max(text[1], lane[1])), (H, H)], False
            rows = np.array([runs[0][0], runs[0][1], runs[1][0], runs[1][1]], np.int32)
            covered = (rows[1] - rows[0]) + (rows[3] - rows[2])
            host = None
            disjoint = text[1] <= lane[0] or lane[1] <= text[0]
            if self.host_text and disjoint and lane[1] > lane[0] and self.img_size[0] % 4 == 0:
                # the text lines can be drawn on the host (lt_text_blend_host): of an annotated frame only the lane's run of rows is
                # the device's business
                lr = np.array([0, 0, lane[0], lane[1]], np.int32)
                src = self._ctx.source_rows()
                host = (lr, lr.ctypes.data, (int(lane[0]), int(lane[1])), (int(text[0]), int(text[1])),
                        not (src[0] <= lane[0] and lane[1] <= src[1]))
            self._rows4 = (rows, rows.ctypes.data, [int(v) for v in rows], ahead, host) if covered <= 0.8 * H else None
        return self._rows4

    # Alternatives kept for A/B runs and tests are plain class attributes (set them on the class or an instance; no environment
    # switches): host_text, host_copies_rows, enqueues_upload, draws_lane_on_device, speculates_lane; stream_lane_on_device,
    # strip_piece (stream.py).  None changes results.
    host_text = True            # False: the text lines are drawn by the device (lt_overlay_text), as in round 4

    _TEXT_ORIGIN, _TEXT_STEP = (20, 8), 35          # Context.present_frame / overlay_text defaults

    def _prepare_out(self, then=None):
        """While the device is busy with the frame (another 0.1 ms until its record is there): fetch the page-locked array the
        annotated frame will land in and copy the rows no overlay can touch from the camera frame the caller handed in -- they
        are the same bytes, and what the host copies now does not cross the bus later.  With the text drawn on the host
        (`_present_rows()[4]`) that is every row but the lane's run: the text is blended over the copied rows once it is known."""
        H, W = self.img_size[1], self.img_size[0]
        out = self._out = _native.pinned_empty((1, H, W, 3))
        self._out_rows = None
        self._out_host_text = False
        img = self._resident[0] if self._resident is not None else None
        rows = self._rows_for(img)
        if rows is None:
            return
        lib, dst, src, rb, grp = self._ctx.lib, out.ctypes.data, img.ctypes.data, W * 3, self._copy_group()
        if rows[4] is not None:
            l0, l1 = rows[4][2]
            gaps = ((0, l0), (l1, H))
            self._out_rows, self._out_ahead, self._out_host_text = rows[4][1], True, True
        else:
            a0, a1, b0, b1 = rows[2]
            gaps = ((0, a0), (a1, b0), (b1, H))
            self._out_rows, self._out_ahead = rows[1], rows[3]
        if then is not None:
            then()                       # (what needs the output frame but not its rows: the device's work goes first)
        for lo, hi in gaps:          # as rows of a 2-D copy: the copy threads share them (1080 rows of 5.8 KB: 4 MB, 0.3 ms on one thread)
            if hi > lo and lib.lt_host_copy2d_async_group(grp, dst + lo * rb, rb, src + lo * rb, rb, rb, hi - lo):
                raise _native.NativeError("lt_host_copy2d_async_group failed")
        self._copying = True             # until the group's wait: `out` and `img` must stay as they are

    _out_host_text = False      # the text lines of _out are drawn on the host (its row runs are then the lane's run alone)

    def _rows_for(self, img):
        """_present_rows() when `img` is a frame whose rows the host can copy as they are, else None."""
        if not self.host_copies_rows or not (isinstance(img, np.ndarray) and img.dtype == np.uint8 and img.flags["C_CONTIGUOUS"]
                                             and img.shape == (self.img_size[1], self.img_size[0], 3)):
            return None
        return self._present_rows()

    _copying = False            # the library's copy thread may still be filling rows of _out (or of the frame just handed out)

    _copy_keepalive = None      # the arrays the copy threads read and write, until they are done

    _group = None               # completion group of this tracker's host copies (lt_host_copy_group_create): process() and stand-alone windows
    _window_groups = ()         # groups of the windows of a stream whose host copies may still run, oldest first: [(group, keepalive)]

    def _copy_group(self):
        """This tracker's own completion group of the library's copy threads: its waits do not depend on what other trackers
        (or other threads) have queued."""
        if self._group is None:
            self._group = _native.host_copy_group()
        return self._group

    def _copies_done(self, window_group=None):
        """Wait for the host copies of this tracker -- process()'s and a stand-alone window's (its own group), and with
        `window_group` that window's of a stream (its group is given back)."""
        if self._copying:
            self._copying = False
            _native._check(self._ctx.lib.lt_host_copy_wait_group(self._copy_group()))
        self._copy_keepalive = None
        if window_group is not None:
            _native.host_copy_group_release(window_group)
            self._window_groups = [g for g in self._window_groups if g[0] != window_group]

    def _all_copies_done(self):
        for g, _ in list(self._window_groups):
            self._copies_done(g)
        self._copies_done()

    def _rows_for_window(self, frames):
        """_present_rows() when the annotated frames of this window can travel as row runs, else None."""
        if not self.host_copies_rows or frames.dtype != np.uint8 or not frames.flags["C_CONTIGUOUS"] or \
                frames.shape[1:] != (self.img_size[1], self.img_size[0], 3):
            return None
        return self._present_rows()

    _resident_partial = False   # of the resident frame only the rows process() reads and presents are on the device

    host_copies_rows = True     # False: the whole annotated frame comes back from the device

    def _present(self, slot, left_n, right_n, left_yx, right_yx, lines, line_len=40, drawn=False):
        """Polygon (packed, by address) and text lines onto the frame in `slot` -> the annotated frame.  drawn: _lane_ahead has
        drawn this polygon and sent its rows already; the text lines are what is left."""
        out, rows, self._out, self._out_rows, self._lane_in_flight = self._out, self._out_rows, None, None, None
        self._device_lane = None
        host_text, self._out_host_text = self._out_host_text, False
        if out is None:
            out, rows, host_text = _native.pinned_empty((1, self.img_size[1], self.img_size[0], 3)), None, False
        if len(lines) > 3:
            raise ValueError("at most three text lines")
        if rows is None and self._resident_partial and self._resident is not None:
            # the whole frame is wanted after all (draw_lane() by hand on the frame process() has just seen): bring the rest
            self._upload_keepalive = self._ctx.upload_frame_rest(self._resident[0], first=slot)
            self._resident_partial = False
        def packed():        # the lines as the library takes them (only where somebody still has to draw them)
            return b"".join(t.encode("ascii", "replace")[:line_len].ljust(line_len, b"\0") for t in lines) \
                if self._have_font and lines else None
        if host_text and rows is not None:
            # the device's part is the lane's run of rows (drawn and on its way already, or now); the text goes over the rows
            # the copy thread has brought from the caller's frame, on this thread, while the lane's rows land
            try:
                if not drawn:
                    self._ctx.present_lane_async(slot, left_n, right_n, left_yx, right_yx, out, rows)
            finally:
                early, self._text_in_flight = self._text_in_flight, None
                self._copies_done()      # (with the text of _text_early among them)
            if not (early is not None and early[0] is out and early[1] == lines):
                text = packed()
                if text is not None:
                    _native.text_blend(out, _overlay.font_atlas(), text, len(lines), line_len, self._TEXT_ORIGIN, self._TEXT_STEP)
            return self._ctx.present_finish(slot, None, 0, line_len, out, rows, origin=self._TEXT_ORIGIN, step=self._TEXT_STEP)[0]
        text = packed()
        try:
            if drawn and out is not None and rows is not None:
                return self._ctx.present_finish(slot, text, len(lines), line_len, out, rows, origin=self._TEXT_ORIGIN,
                                                step=self._TEXT_STEP)[0]
            return self._ctx.present_frame(slot, left_n, right_n, left_yx, right_yx, text, len(lines), line_len, out, rows,
                                           origin=self._TEXT_ORIGIN, step=self._TEXT_STEP)[0]
        finally:
            self._copies_done()          # the rows the host fills itself

    text_now = True             # False: the text lines as a job for a copy thread (lt_host_text_async_group), waited for in _present
    _text_in_flight = None      # (output frame, lines) whose text a copy thread is blending (_text_early), until _present has waited for it

    def _text_early(self):
        """The text lines of the frame being finished, blended into the output frame by one of the library's copy threads
        (lt_host_text_async_group: lt_text_blend_host's arithmetic) while this thread goes on with the frame's bookkeeping --
        10-13 us of the 40-odd between the record and the return.  The rows under the text were copied from the caller's frame by
        the same threads long ago (`_prepare_out`); they are waited for first all the same."""
        out, lines = self._out, self._lane_text()
        if out is None or len(lines) > 3:
            return
        if self.text_now:
            # ... or, since the overlap is only a few microseconds wide by now, drawn right here: ONE call waits for the rows and
            # draws (the lines spread over the copy threads that happen to be polling) -- no job, no second wait in _present
            text, nl = _native.text_bytes([lines])
            _native.host_text_now(self._copy_group(), out, _overlay.font_atlas(), text, nl, 40, self._TEXT_ORIGIN, self._TEXT_STEP)
            self._copying, self._copy_keepalive = False, None        # (the call has waited for the group)
            self._text_in_flight = (out, lines)
            return
        self._copies_done()
        text, nl = _native.text_bytes([lines])
        _native.host_text_async(self._copy_group(), out, out, (0, 0, 0, 0), _overlay.font_atlas(), text, nl, 40, self._TEXT_ORIGIN, self._TEXT_STEP)
        self._copying = True
        self._text_in_flight = (out, lines)

    def print_failure(self, img):
        """The failure message on a copy of the frame (upstream writes into the caller's array, :664-673)."""
        slot = self._overlay_slot(img)
        zero = self.__dict__.setdefault("_zero_counts", np.zeros(2, np.int32))
        return self._present(slot, zero.ctypes.data, zero.ctypes.data + 4, None, None, self._failure_text())

    def window_mask(self, img, window_width, window_height, center, level, ignore_bottom):
        return _overlay.window_mask(img, window_width, window_height, center, level, ignore_bottom)

    def visualize_sliding_window_search(self, binary_img, left_fit_coeffs, right_fit_coeffs, window_width,
                                        window_height, ignore_bottom):
        """Search windows, detected pixels and the fitted curves drawn over the mask (debugging aid)."""
        self._materialise_pending()
        return _overlay.visualize_sliding_window_search(
            binary_img, self.left_window_centroids, self.right_window_centroids, (self.left_y, self.left_x),
            (self.right_y, self.right_x), self.get_poly_points(left_fit_coeffs, right_fit_coeffs), window_width,
            window_height, ignore_bottom)

    def visualize_band_search(self, binary_img, left_fit_coeffs, right_fit_coeffs, bandwidth, partial):
        """Search bands around the previous curves, detected pixels and the new fitted curves (debugging aid)."""
        self._materialise_pending()
        return _overlay.visualize_band_search(
            binary_img, (self.left_y, self.left_x), (self.right_y, self.right_x),
            self.get_poly_points(self.last_left_coeffs, self.last_right_coeffs, partial),
            self.get_poly_points(left_fit_coeffs, right_fit_coeffs), bandwidth)

    def triple_split_view(self, images):
        """Annotated frame on top, bird's-eye image and search visualisation side by side below it."""
        img1_size = (images[0].shape[1], images[0].shape[0])
        img2_size = (images[1].shape[1], images[1].shape[0])
        positions = [(0, 0), (0, img1_size[1]), (round(0.5 * img1_size[0]), img1_size[1])]
        scale_factor = img2_size[0] / (0.5 * img1_size[0])
        scaled_size = (round(img2_size[0] / scale_factor), round(img2_size[1] / scale_factor))
        target_size = (img1_size[0], img1_size[1] + scaled_size[1])
        return _utils.create_split_view(target_size, images, positions, [img1_size, scaled_size, scaled_size])

    # ---- find_lane_points (reference :795-874) ---------------------------------------------------------------
    def find_lane_points(self, img, ksize_r=15, C_r=8, ksize_b=35, C_b=5, filter_type='bilateral', mask_noise=True,
                         noise_thresh=140, ksize_noise=65, C_noise=10, window_width=30, window_height=40,
                         search_range=20, mu=0.1, no_success_limit=8, start_slice=0.25, ignore_sides=360,
                         ignore_bottom=30, bandwidth=30, partial=0.5, diagnostics=False):
        """frame -> (binary bird's-eye mask, 'sws' | 'bs'); lane pixels are stored on the tracker."""
        mode = self._find_lane_points_device(img, ksize_r, C_r, ksize_b, C_b, filter_type, mask_noise, noise_thresh,
                                             ksize_noise, C_noise, window_width, window_height, search_range, mu,
                                             no_success_limit, start_slice, ignore_sides, ignore_bottom, bandwidth,
                                             partial, diagnostics)
        return self._ctx.download_masks(1)[0], mode

    def _find_lane_points_device(self, img, ksize_r, C_r, ksize_b, C_b, filter_type, mask_noise, noise_thresh,
                                 ksize_noise, C_noise, window_width, window_height, search_range, mu,
                                 no_success_limit, start_slice, ignore_sides, ignore_bottom, bandwidth, partial,
                                 diagnostics, reuse_frame=False, slot=0, have_mask=False, lazy=False):
        ctx = self._ctx
        self._device_lane = None
        if not reuse_frame:
            # only the camera rows the path reads are on the critical path; the rest of the frame (the overlay
            # shows it) follows on a copy stream while the mask chain runs
            # (not waited for: the engine's copy runs under the mask chain's launches; `img` -- or the contiguous copy the binding
            # made of it -- stays alive until the frame's record is on the host, and process() waits for the device on its way out
            # of an exception)
            self._rows_keepalive = ctx.upload_frame_rows(img, first=slot, enqueue=self.enqueues_upload)
        try:
            self._resident = (img, slot)
            if not have_mask:
                # (reuse_frame: the second try of this frame -- the bird's-eye planes are there, only the filter differs)
                ctx.mask_run(1, _native.filter_params(filter_type, ksize_r, C_r, ksize_b, C_b, mask_noise, noise_thresh,
                                                      ksize_noise, C_noise), first=slot, reuse_front=reuse_frame)

            if self._want_out and self._out is None and not reuse_frame:
                # The output array, and its untouched rows on their way on the library's copy threads, BEFORE the search is launched:
                # the device needs another 70 us of mask chain before it gets to the search, the host is ahead of it, and the rows
                # -- 1.8 MB at 1280x720, 4 MB at 1920x1080 -- should be in place when the record arrives and the text is due.
                self._prepare_out()

            def while_the_device_searches():
                # behind the search's launch (the device has the whole chain queued; the host's share of the frame must not delay it)
                if not reuse_frame:
                    # of the rest of the frame only the rows the overlay will read, when the annotated frame goes back in row runs
                    # (first: the overlay enqueued below waits for the copies known at that moment)
                    rows = self._rows_for(img) if self._want_out else None
                    if rows is not None and rows[4] is not None:
                        if rows[4][4]:           # (rows of the lane's run the mask chain does not read: none with the reference calibration)
                            self._upload_keepalive = ctx.upload_frame_rest(img, first=slot, rows=rows[4][1])
                    else:
                        self._upload_keepalive = ctx.upload_frame_rest(img, first=slot, rows=None if rows is None else rows[1])
                    self._resident_partial = rows is not None
                if self._want_out and self._out is None:
                    self._prepare_out()          # (a frame whose output array could not be prepared ahead of the search)
                if self._want_out and self._out is not None and self._device_lane is None and not reuse_frame and self.draws_lane_on_device \
                        and self.speculates_lane and self._out_host_text and self._out_rows is not None and self._out_ahead:
                    self._device_lane = self._lane_from_fit(partial, slot)   # the lane drawn by the device itself, right behind the search
            if self.last_detection > self.n_reset:                       # :851
                if diagnostics:
                    print("Using sliding window search.")
                self._search_uploaded(ctx, 'sws', dict(window_width=window_width, window_height=window_height,
                                                       search_range=search_range, mu=mu,
                                                       no_success_limit=no_success_limit, start_slice=start_slice,
                                                       ignore_sides=ignore_sides, ignore_bottom=ignore_bottom,
                                                       partial=partial), diagnostics, slot=slot, lazy=lazy,
                                      between=while_the_device_searches)
                return 'sws'
            if diagnostics:
                print("Using band search.")
            self._search_uploaded(ctx, 'bs', dict(bandwidth=bandwidth, ignore_bottom=ignore_bottom, partial=partial),
                                  diagnostics, slot=slot, lazy=lazy, between=while_the_device_searches)
            return 'bs'
        except BaseException:
            if not reuse_frame and self.enqueues_upload:     # the frame's rows may still be on their way: the caller's array must outlive the copy
                try:
                    ctx.sync()
                except Exception:
                    pass
            raise

    # ---- the two outcomes of a frame (reference :1142-1159, :1178-1202) ------------------------------------------
    def _record_failure(self):
        self.left_fit_coeffs.append(np.array([]))
        self.right_fit_coeffs.append(np.array([]))
        self.average_curve_radii.append(-1)
        if len(self.left_fit_coeffs) > self.n_average:
            self.left_fit_coeffs.pop(0)
            self.right_fit_coeffs.pop(0)
        if len(self.average_curve_radii) > self.n_average:
            self.average_curve_radii.pop(0)
        self.last_detection += 1

    def _averages_with(self, left_fit_coeffs, right_fit_coeffs):
        """left_avg_coeffs / right_avg_coeffs as _record_success would leave them for this fit (nothing is changed)."""
        L, R = self.left_fit_coeffs + [left_fit_coeffs], self.right_fit_coeffs + [right_fit_coeffs]
        if len(L) > self.n_average:
            L, R = L[1:], R[1:]
        return _mean_of_rows([c for c in L if c.size != 0]), _mean_of_rows([c for c in R if c.size != 0])

    enqueues_upload = True      # False: lt_upload_frame_rows waits for its copy
    _rows_keepalive = None
    draws_lane_on_device = True   # False: the host launches the overlay once it has the record
    _device_lane = None         # the device draws this frame's lane behind its search (lt_present_lane_from_fit_async was enqueued)
    _fit_flags = 0

    def _lane_from_fit(self, partial, slot):
        """Enqueue, behind the search that has just been launched over `slot`, the drawing of the lane this frame will have if it is
        valid: the device forms the running average with the fit of the slot's record itself (what _averages_with will form on the
        host: the same sum in the same order, this frame's fit last), its plot points and the polygon, and stores the lane's rows
        into the output frame -- 40 us before the host, which must see the record first, could launch that overlay.  -> truthy when
        enqueued."""
        L, R = self.left_fit_coeffs, self.right_fit_coeffs
        if len(L) + 1 > self.n_average:
            L, R = L[1:], R[1:]
        Lv, Rv = [c for c in L if c.size != 0], [c for c in R if c.size != 0]
        if len(Lv) != len(Rv):
            return None
        buf = self.__dict__.get("_prev_sum")
        if buf is None:
            buf = self._prev_sum = np.zeros(6, np.float64)
        if Lv:
            accl, accr = Lv[0], Rv[0]
            for k in range(1, len(Lv)):
                accl, accr = accl + Lv[k], accr + Rv[k]
            buf[:3] = accl
            buf[3:] = accr
        ploty, ploty2 = self._plot_rows(partial)
        self._configure_overlay()
        ok = self._ctx.present_lane_from_fit_async(slot, buf.ctypes.data, len(Lv) + 1, ploty.ctypes.data, ploty2.ctypes.data, len(ploty),
                                                   self._out, self._out_rows)
        return (len(Lv) + 1,) if ok else None

    speculates_lane = True      # False: the lane is drawn once the frame is known to be valid
    _lane_in_flight = None      # the packed-point buffers whose polygon lt_present_lane_async is drawing / has drawn for this frame

    def _lane_ahead(self, left_fit_coeffs, right_fit_coeffs, partial, slot):
        """The averaged curves this frame will have IF it is valid follow from its fit alone; whether it is valid, its radius and
        eccentricity take the host another 25 us.  So the polygon is drawn and the rows it can reach sent on their way now
        (lt_present_lane_async); a valid frame -- nearly all of a video -- only adds the text lines (lt_present_finish), an invalid
        one has everything drawn again by whatever is presented in the end.  -> (left average, right average, point buffers)."""
        la, ra = self._averages_with(left_fit_coeffs, right_fit_coeffs)
        # two buffer sets: the polygon of the last valid frame (redrawn on failures) must survive a speculation that fails
        keep = self._avg_packed[0] if self._avg_packed is not None else None
        purpose = 'avg1' if keep is not None and keep is self.__dict__.get("_packed", {}).get((partial, 'avg0')) else 'avg0'
        b = self._points_packed(la, ra, partial, purpose)
        if b is keep:
            return None
        self._configure_overlay()
        a = b[4]
        if self._device_lane is None or self._fit_flags:
            # (a fit the host refitted -- a rank-deficient side -- is not the record's: the device has drawn nothing for it)
            self._ctx.present_lane_async(slot, a[7], a[8], a[9] if b[1][0] else None, a[10] if b[1][1] else None, self._out, self._out_rows)
        self._lane_in_flight = b
        return la, ra, b

    def _record_success(self, left_fit_coeffs, right_fit_coeffs, partial, ahead=None, on_text=None):
        self.left_fit_coeffs.append(left_fit_coeffs)
        self.right_fit_coeffs.append(right_fit_coeffs)
        self.last_left_coeffs = left_fit_coeffs
        self.last_right_coeffs = right_fit_coeffs
        if len(self.left_fit_coeffs) > self.n_average:
            self.left_fit_coeffs.pop(0)
            self.right_fit_coeffs.pop(0)
        self.last_detection = 0
        self.success += 1
        if ahead is not None:       # _lane_ahead has formed the averages and their plot points already
            self.left_avg_coeffs, self.right_avg_coeffs, b = ahead
        else:
            self.left_avg_coeffs = _mean_of_rows([c for c in self.left_fit_coeffs if c.size != 0])
            self.right_avg_coeffs = _mean_of_rows([c for c in self.right_fit_coeffs if c.size != 0])
            keep = self._avg_packed[0] if self._avg_packed is not None else None
            purpose = 'avg1' if keep is not None and keep is self.__dict__.get("_packed", {}).get((partial, 'avg0')) else 'avg0'
            b = self._points_packed(self.left_avg_coeffs, self.right_avg_coeffs, partial, purpose)     # = get_poly_points
        nl, nr, H = int(b[1][0]), int(b[1][1]), self.warped_size[1]
        early = on_text is not None and nl > 0 and nr > 0
        if early:
            # what the frame's text lines need, first: the radius, and the eccentricity from the last plot point of each averaged
            # curve (get_eccentricity's operands and operations: left_avg_x[-1], right_avg_x[-1] as int64) -- then the text goes to
            # a copy thread while this one builds the rest of the state
            self.get_curve_radius()
            mid = int(self.warped_size[0] / 2)
            self.eccentricity = (((mid - np.int64(b[2][nl - 1, 1])) - (np.int64(b[3][nr - 1, 1]) - mid)) / 2) * self.mpph
            on_text()
        self.left_avg_y, self.left_avg_x = np.arange(H - nl, H, dtype=np.int64), b[2][:nl, 1].astype(np.int64)
        self.right_avg_y, self.right_avg_x = np.arange(H - nr, H, dtype=np.int64), b[3][:nr, 1].astype(np.int64)
        self._avg_packed = (b, self.left_avg_x, self.right_avg_x)
        if not early:
            self.get_curve_radius()
            self.get_eccentricity()

    fast_tail = True            # False: a valid first try's bookkeeping by the Python functions one after the other (A/B, tests)

    def _tail_fast(self, partial):
        """The frame process() sees nearly always -- a first try the device has searched, fitted and drawn, about to turn out valid
        -- between its record and its text lines in ONE host call (lt_frame_tail: check_validity, the running average, its plot
        points, radius, eccentricity: the operations of fit_poly / _lane_ahead / check_validity / _record_success / get_curve_radius
        in their order) instead of 29 us of Python.  -> True: valid, the state is what _record_success leaves (text lines on their
        way); False: check_validity says no (nothing else changed); None: not this frame -- the caller goes the long way, from an
        unchanged state (a radius whose int() upstream's refit decides, a fit the host refitted, no device-drawn lane, ...)."""
        fit = self._fit
        if not (fit is not None and isinstance(fit[0], str) and not self._fit_flags and self._device_lane is not None
                and self._out is not None and self._out_rows is not None and self._out_ahead and self._out_host_text
                and self._have_font and self.speculates_lane and type(self.mppv) is float and type(self.mpph) is float):
            return None
        keep = self._avg_packed[0] if self._avg_packed is not None else None
        purpose = 'avg1' if keep is not None and keep is self.__dict__.get("_packed", {}).get((partial, 'avg0')) else 'avg0'
        b = self._packed_buffers(partial, purpose)
        if b is keep:
            return None
        t = self.__dict__.get("_tail")
        if t is None:
            t = self._tail = (np.zeros(24, np.float64), np.zeros(8, np.float64), {}, _native.load().lt_frame_tail)
        inp, out, argsets, fn = t
        args = argsets.get((partial, purpose))
        if args is None:
            pv, pv2 = self._plot_rows(1)
            pp, pp2 = b[6]
            args = argsets[(partial, purpose)] = (int(self.warped_size[0]), int(self.warped_size[1]), inp.ctypes.data, pv.ctypes.data,
                                                  pv2.ctypes.data, len(pv), pp.ctypes.data, pp2.ctypes.data, len(pp), b[0].ctypes.data,
                                                  b[1].ctypes.data, b[1].ctypes.data + 4, b[2].ctypes.data, b[3].ctypes.data, out.ctypes.data)
        lim = self.validity_limits
        try:
            inp[0:3] = fit[2]
            inp[3:6] = fit[3]
            inp[6:12] = self._prev_sum
            inp[12] = self._device_lane[0]
            inp[13:22] = (lim['min_dist_y1'], lim['max_dist_y1'], lim['min_dist_y2'], lim['max_dist_y2'], lim['min_dist_y3'],
                          lim['max_dist_y3'], lim['thresh'], self.mppv, self.mpph)
        except (TypeError, ValueError, KeyError):
            return None
        if fn(*args) or out[1]:
            return None
        if not out[0]:
            self.valid_lane_lines = False
            return False
        # valid: what _lane_ahead, check_validity and _record_success (with the lane ahead and the text early) leave
        lf, rf = fit[2].copy(), fit[3].copy()
        self.valid_lane_lines = True
        self._lane_in_flight = b
        self.left_fit_coeffs.append(lf)
        self.right_fit_coeffs.append(rf)
        self.last_left_coeffs = lf
        self.last_right_coeffs = rf
        if len(self.left_fit_coeffs) > self.n_average:
            self.left_fit_coeffs.pop(0)
            self.right_fit_coeffs.pop(0)
        self.last_detection = 0
        self.success += 1
        co = b[0]
        self.left_avg_coeffs, self.right_avg_coeffs = co[:3].copy(), co[3:].copy()
        nl, nr, H = int(b[1][0]), int(b[1][1]), self.warped_size[1]
        self._note_radii([int(out[2]), int(out[3])])
        self.eccentricity = out[4] * 1.0          # (an np.float64 of its own, as the NumPy scalars of get_eccentricity give)
        self._text_early()
        self.left_avg_y, self.left_avg_x = np.arange(H - nl, H, dtype=np.int64), b[2][:nl, 1].astype(np.int64)
        self.right_avg_y, self.right_avg_x = np.arange(H - nr, H, dtype=np.int64), b[3][:nr, 1].astype(np.int64)
        self._avg_packed = (b, self.left_avg_x, self.right_avg_x)
        return True

    # ---- process (reference :876-1209) -----------------------------------------------------------------------
    def process(self, img, ksize_r=15, C_r=8, ksize_b=35, C_b=5, filter_type='bilateral', mask_noise=False,
                noise_thresh=140, ksize_noise=65, C_noise=10, window_width=30, window_height=40, search_range=20,
                mu=0.1, no_success_limit=8, start_slice=0.25, ignore_sides=360, ignore_bottom=30, bandwidth=25,
                partial=1.0, n_tries=2, visualize_search=False, split_view=False, diagnostics=False):
        """One video frame (RGB u8, the constructor's img_size) in, the annotated frame out.

        Up to two detection attempts per frame: the caller's parameters first, then the
        'neighborhood' parameter set (reference :1081-1099).  A valid detection updates the
        running average over `n_average` fits and resets `last_detection`; a failure re-draws the
        last averaged lane for up to `n_fail` frames and falls back from band search to
        sliding-window search after `n_reset` misses.
        """
        first_try = (ksize_r, C_r, ksize_b, C_b, filter_type, mask_noise, noise_thresh, ksize_noise, C_noise,
                     window_width, window_height, search_range, mu, no_success_limit, start_slice, ignore_sides,
                     ignore_bottom, bandwidth, partial)
        # The lane-pixel lists stay on the device until somebody reads lt.left_x & co.  Frames alternate between
        # two slots, so that the lists of the previous search are still there when this frame's searches find
        # nothing (upstream keeps the old lists in that case).
        if self._in_stream:
            raise RuntimeError("process() inside an active process_stream() would overwrite its frames")
        self._slot ^= 1
        try:
            return self._step(img, first_try, n_tries, diagnostics, slot=self._slot, have_mask=False, lazy=True, annotate=True,
                              visualize_search=visualize_search, split_view=split_view)
        finally:
            if self._lane_in_flight is not None or self._device_lane is not None:     # an exception between the two halves: the copy into _out may still be running
                self._lane_in_flight = self._device_lane = None
                self._ctx.sync()
            self._copies_done()
            self._want_out, self._out, self._out_rows, self._text_in_flight = False, None, None, None

    def _step(self, img, first_try, n_tries, diagnostics, slot, have_mask, lazy, annotate, visualize_search=False,
              split_view=False, defer=None):
        """One frame of the per-stream state machine (reference :1026-1209).  With `defer` (a list) the
        annotated frame is not produced here: ('lane', polygon, text) or ('fail', None, text) is appended
        and the caller renders all frames of the window in one overlay launch."""
        partial = first_try[-1]
        self.counter += 1
        self.detected_pixels = False
        self.valid_lane_lines = False
        left_fit_coeffs = right_fit_coeffs = None
        used = first_try                      # the parameter set of the most recent attempt
        self._want_out = annotate and defer is None

        search_mode = self._find_lane_points_device(img, *first_try, diagnostics, reuse_frame=have_mask, slot=slot,
                                                    have_mask=have_mask, lazy=lazy)
        spec = None
        fast = None
        if self.detected_pixels:
            if self.fast_tail and self._want_out and not (diagnostics or visualize_search or split_view):
                fast = self._tail_fast(partial)
            if fast is None:
                left_fit_coeffs, right_fit_coeffs = self.fit_poly()
                if self._out_rows is not None and self._out_ahead and self.speculates_lane:
                    spec = self._lane_ahead(left_fit_coeffs, right_fit_coeffs, partial, slot)
                self.check_validity(left_fit_coeffs, right_fit_coeffs, diagnostics)
                if diagnostics and self.valid_lane_lines:
                    print("Success at first attempt!")
            if not self.valid_lane_lines:
                spec = self._lane_in_flight = None      # drawn for nothing: whatever is presented later draws everything again

        if ((not self.detected_pixels) or (not self.valid_lane_lines)) and ((n_tries >= 2) or (n_tries == -1)):
            if diagnostics:
                print("No success at first attempt, now trying second.")
            partial = 1.0                                               # the second parameter set (:1081-1099)
            second_try = self._SECOND_TRY
            used = second_try
            search_mode = self._find_lane_points_device(img, *second_try, diagnostics, reuse_frame=True, slot=slot,
                                                        lazy=lazy)
            if self.detected_pixels:
                left_fit_coeffs, right_fit_coeffs = self.fit_poly()
                self.check_validity(left_fit_coeffs, right_fit_coeffs, diagnostics)
                if diagnostics and self.valid_lane_lines:
                    print("Success at second attempt!")

        search_visualization = warped_img = None
        if visualize_search or split_view:                              # :1130-1137
            binary_img = self._ctx.download_masks(1, first=slot)[0]
            if self.detected_pixels:
                if search_mode == 'sws':
                    search_visualization = self.visualize_sliding_window_search(binary_img, left_fit_coeffs,
                                                                                right_fit_coeffs, used[9], used[10],
                                                                                used[16])
                else:
                    search_visualization = self.visualize_band_search(binary_img, left_fit_coeffs, right_fit_coeffs,
                                                                      used[17], used[18])
            else:
                search_visualization = binary_img
            if split_view:
                # the bird's-eye image the detector saw (upstream warps the raw, still distorted frame here, :1035)
                warped_img = self._ctx.download_bev(1, first=slot)[0]

        def present(annotated):
            if visualize_search:
                return annotated, search_visualization
            if split_view:
                return self.triple_split_view([annotated, warped_img, search_visualization])
            return annotated

        if not self.valid_lane_lines:                                   # :1142-1173
            if diagnostics:
                print("No success after all attempts.")
            self._record_failure()
            if not annotate:
                return None
            redraw = (self.left_avg_y.size != 0) and (self.last_detection <= self.n_fail)
            if defer is not None:
                defer.append(('lane', (self.left_avg_y, self.left_avg_x, self.right_avg_y, self.right_avg_x),
                              self._lane_text()) if redraw else ('fail', None, self._failure_text()))
                return None
            return present(self.draw_lane(img) if redraw else self.print_failure(img))

        # success (:1178-1209)
        if fast:                            # (recorded by _tail_fast already, text lines included)
            return present(self.draw_lane(img))
        on_text = None
        if spec is not None and self._want_out and self._out is not None and self._out_host_text and self._out_rows is not None \
                and self._have_font and not (visualize_search or split_view):
            on_text = self._text_early      # the lane is being drawn by the device: the text lines go to a copy thread as soon as they are known
        self._record_success(left_fit_coeffs, right_fit_coeffs, partial, spec, on_text)
        if not annotate:
            return None
        if defer is not None:
            defer.append(('lane', (self.left_avg_y, self.left_avg_x, self.right_avg_y, self.right_avg_x),
                          self._lane_text()))
            return None
        return present(self.draw_lane(img))


def _mean_of_rows(rows):
    """np.average(rows, axis=0) for a short list of equally long f64 vectors, bit for bit (the sequential sum NumPy's
    reduction over the first axis forms, then one division by the count) at a fifth of its call overhead
    (tests/test_host_geometry.py::test_mean_of_rows_is_numpy_average)."""
    acc = rows[0]
    for r in rows[1:]:
        acc = acc + r
    return acc / len(rows)


def _minimum_norm_parabola(y, x):
    """np.polyfit's answer for rank-deficient input (fewer than 3 distinct y): column-scaled
    Vandermonde, minimum-norm least squares with rcond = len(y)*eps.  Only reached for degenerate
    pixel sets; the regular fit is done on the GPU."""
    y = np.asarray(y, np.float64)
    lhs = np.vander(y, 3)
    scale = np.sqrt((lhs * lhs).sum(axis=0))
    scale[scale == 0] = 1.0
    with _hostcpu.blas_limited():
        c = np.linalg.lstsq(lhs / scale, np.asarray(x, np.float64), rcond=len(y) * np.finfo(np.float64).eps)[0]
    return c / scale
