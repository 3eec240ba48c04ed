"""ctypes binding of liblane_tracker_amd.so (include/lane_tracker_amd.h).

The library is HIP-only.  There is deliberately no fallback: if the shared object is missing or no
GPU is visible, constructing a `Context` raises -- nothing in this package computes on the CPU.
"""
import ctypes as C
import math
import os
import subprocess
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# LANE_TRACKER_AMD_LIB: load another build of the same library (tools/toolchain_cases.sh compares variant builds)
LIB_PATH = os.environ.get("LANE_TRACKER_AMD_LIB") or os.path.join(_HERE, "liblane_tracker_amd.so")
NUM_STAGES = 12
ABI_VERSION = 5          # LT_ABI_VERSION of include/lane_tracker_amd.h this table was written against

PLANE_R, PLANE_LAB_B, PLANE_TOPHAT_R, PLANE_TOPHAT_B, PLANE_MERGED, PLANE_MASK = range(6)


class NativeError(RuntimeError):
    pass


class Calib(C.Structure):
    _fields_ = [("img_w", C.c_int32), ("img_h", C.c_int32), ("warp_w", C.c_int32), ("warp_h", C.c_int32),
                ("cam_matrix", C.c_double * 9), ("dist_coeffs", C.c_double * 5), ("M", C.c_double * 9)]


class FilterParams(C.Structure):
    _fields_ = [("filter_type", C.c_int32), ("ksize_r", C.c_int32), ("C_r", C.c_int32), ("ksize_b", C.c_int32),
                ("C_b", C.c_int32), ("mask_noise", C.c_int32), ("noise_thresh", C.c_int32),
                ("ksize_noise", C.c_int32), ("C_noise", C.c_int32)]


class SearchParams(C.Structure):
    _fields_ = [("window_width", C.c_int32), ("window_height", C.c_int32), ("search_range", C.c_int32),
                ("no_success_limit", C.c_int32), ("ignore_sides", C.c_int32), ("ignore_bottom", C.c_int32),
                ("bandwidth", C.c_int32), ("_pad", C.c_int32), ("mu", C.c_double), ("start_slice", C.c_double),
                ("partial", C.c_double)]


class LaneRecord(C.Structure):
    _fields_ = [("left_coeffs", C.c_double * 3), ("right_coeffs", C.c_double * 3), ("n_left", C.c_int32),
                ("n_right", C.c_int32), ("detected", C.c_uint8), ("fit_flags", C.c_uint8), ("mode", C.c_uint8),
                ("_pad", C.c_uint8), ("frame", C.c_int32)]


RECORD_DTYPE = np.dtype([("left_coeffs", "<f8", 3), ("right_coeffs", "<f8", 3), ("n_left", "<i4"),
                         ("n_right", "<i4"), ("detected", "u1"), ("fit_flags", "u1"), ("mode", "u1"),
                         ("_pad", "u1"), ("frame", "<i4")])
assert RECORD_DTYPE.itemsize == 64 and C.sizeof(LaneRecord) == 64


class Info(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("capacity", C.c_int32),
                ("cu_count", C.c_int32), ("src_row0", C.c_int32), ("src_row1", C.c_int32),
                ("max_pixels_per_side", C.c_int32), ("max_levels", C.c_int32), ("alg_bytes_mask", C.c_int64),
                ("alg_bytes_search", C.c_int64), ("device_name", C.c_char * 64)]


# name -> (restype, argtypes); every symbol include/lane_tracker_amd.h declares
_P = C.c_void_p
_SIGNATURES = {
    "lt_last_error": (C.c_char_p, []),
    "lt_abi_version": (C.c_int, []),
    "lt_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "lt_create": (C.c_int, [C.POINTER(Calib), C.c_int, C.POINTER(_P)]),
    "lt_destroy": (None, [_P]),
    "lt_reserve": (C.c_int, [_P, C.c_int]),
    "lt_get_info": (C.c_int, [_P, C.POINTER(Info)]),
    "lt_sync": (C.c_int, [_P]),
    "lt_set_streams": (C.c_int, [_P, C.c_int]),
    "lt_upload_frames": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "lt_get_source_rows": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "lt_upload_frame_rows": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "lt_upload_frame_rows_async": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "lt_upload_frame_rows_enqueue": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "lt_upload_frame_rest": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "lt_upload_frame_rest_rows": (C.c_int, [_P, _P, C.c_int, C.c_int, _P]),
    "lt_upload_masks": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "lt_download_masks": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "lt_download_plane": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    "lt_download_undistorted": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "lt_download_records": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "lt_download_pixels": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, C.c_int, C.POINTER(C.c_int)]),
    "lt_download_centroids": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.POINTER(C.c_int)]),
    "lt_download_lane_lists": (C.c_int, [_P, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int,
                               C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "lt_copy_records_to_device": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "lt_enqueue_records_to_device": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "lt_overlay_configure": (C.c_int, [_P, _P]),
    "lt_overlay_run": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, C.c_double]),
    "lt_overlay_set_font": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int]),
    "lt_overlay_text": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "lt_overlay_rows": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "lt_present_frame": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, C.c_double, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "lt_present_lane_async": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, C.c_double, _P, _P]),
    "lt_present_finish": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "lt_overlay_run_strip_coeffs": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, C.c_int, C.c_double]),
    "lt_present_lane_from_fit_async": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P, C.c_int, C.c_double, _P, _P]),
    "lt_lane_spans_from_fit": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, C.c_int, _P, _P, C.c_int, _P]),
    "lt_lane_polygon_spans": (C.c_int, [C.c_int, _P, C.c_int, _P, C.c_int, _P]),
    "lt_download_overlay": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "lt_download_overlay_async": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "lt_download_bev": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "lt_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "lt_host_free": (C.c_int, [_P]),
    "lt_host_copy_async": (C.c_int, [_P, _P, C.c_size_t]),
    "lt_host_copy2d_async": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.c_size_t, C.c_size_t]),
    "lt_overlay_run_rows": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, C.c_double, _P]),
    "lt_download_overlay_rows_async": (C.c_int, [_P, C.c_int, C.c_int, _P, _P]),
    "lt_host_copy_wait": (C.c_int, []),
    "lt_host_copy_group_create": (C.c_int, [C.POINTER(C.c_int)]),
    "lt_host_copy_group_destroy": (C.c_int, [C.c_int]),
    "lt_host_copy_async_group": (C.c_int, [C.c_int, _P, _P, C.c_size_t]),
    "lt_host_copy2d_async_group": (C.c_int, [C.c_int, _P, C.c_size_t, _P, C.c_size_t, C.c_size_t, C.c_size_t]),
    "lt_host_copy_wait_group": (C.c_int, [C.c_int]),
    "lt_host_touch_async_group": (C.c_int, [C.c_int, _P, C.c_size_t]),
    "lt_shutdown": (C.c_int, []),
    "lt_host_copy_stats": (C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_longlong), C.POINTER(C.c_int)]),
    "lt_warm": (C.c_int, [_P, C.POINTER(SearchParams), C.POINTER(SearchParams), C.c_int]),
    "lt_overlay_run_strip": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, _P, C.c_double]),
    "lt_strip_download_async": (C.c_int, [_P, C.c_int, C.c_int, _P, C.c_size_t, C.c_int]),
    "lt_text_blend_host": (C.c_int, [_P, C.c_size_t, C.c_int, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_int]),
    "lt_host_text_async_group": (C.c_int, [C.c_int, _P, C.c_size_t, _P, C.c_size_t, C.c_int, _P, C.c_int, C.c_int, _P, _P,
                                           C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "lt_host_text_now_group": (C.c_int, [C.c_int, _P, C.c_int, C.c_int, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int,
                                         C.c_int, C.c_int, C.c_int]),
    "lt_mask_run": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(FilterParams)]),
    "lt_mask_rerun": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(FilterParams)]),
    "lt_upload_bev": (C.c_int, [_P, _P, C.c_int, C.c_int]),
    "lt_filter_run": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(FilterParams)]),
    "lt_sws_fit_run": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(SearchParams)]),
    "lt_band_fit_run": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(SearchParams), _P]),
    "lt_band_fit_chain_run": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(SearchParams), _P]),
    "lt_band_fit_chain_collect": (C.c_int, [_P, C.c_int, C.c_int, _P]),
    "lt_band_fit_chain_cancel": (C.c_int, [_P]),
    "lt_set_search_cus": (C.c_int, [_P, C.c_int]),
    "lt_set_walk_min_frames": (C.c_int, [_P, C.c_int]),
    "lt_set_direct_upload": (C.c_int, [_P, C.c_int]),
    "lt_direct_upload_count": (C.c_ulonglong, [_P]),
    "lt_set_urgent": (C.c_int, [_P, C.c_int]),
    "lt_poly_points": (C.c_int, [C.c_int, C.c_int, _P, C.c_int, _P, _P, C.c_int, _P, _P, _P, _P]),
    "lt_frame_tail": (C.c_int, [C.c_int, C.c_int, _P, _P, _P, C.c_int, _P, _P, C.c_int, _P, _P, _P, _P, _P, _P]),
    "lt_download_overlay_wait": (C.c_int, [_P]),
    "lt_set_frame_base": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "lt_mask_batch": (C.c_int, [_P, _P, C.c_int, C.POINTER(FilterParams), _P]),
    "lt_sws_fit_batch": (C.c_int, [_P, _P, C.c_int, C.POINTER(SearchParams), _P]),
    "lt_band_fit_batch": (C.c_int, [_P, _P, C.c_int, C.POINTER(SearchParams), _P, _P]),
    "lt_bilateral_adaptive_threshold": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.c_int, _P]),
    "lt_filter_lane_points": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(FilterParams), _P]),
    "lt_morph_ellipse": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "lt_fit_poly2": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "lt_calib_source_rows": (C.c_int, [C.POINTER(Calib), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "lt_calib_warp_table": (C.c_int, [C.POINTER(Calib), _P, _P]),
    "lt_calib_undistort_table": (C.c_int, [C.POINTER(Calib), C.c_int, C.c_int, _P, _P]),
    "lt_calib_lab_tables": (C.c_int, [_P, _P, _P]),
    "lt_calib_ellipse": (C.c_int, [C.c_int, _P, C.POINTER(C.c_int)]),
    "lt_gather_init": (C.c_int, [_P, C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(_P)]),
    "lt_gather_world": (C.c_int, [_P, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "lt_gather_reserve": (C.c_int, [_P, C.c_int]),
    "lt_gather_stage": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "lt_gather_records": (C.c_int, [_P, C.c_int, _P]),
    "lt_gather_host": (C.c_int, [_P, _P, C.c_size_t, _P]),
    "lt_gather_barrier": (C.c_int, [_P]),
    "lt_gather_destroy": (None, [_P]),
    "lt_timer_start": (C.c_int, [_P]),
    "lt_timer_stop": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "lt_set_stage_timing": (C.c_int, [_P, C.c_int]),
    "lt_stage_reset": (C.c_int, [_P]),
    "lt_stage_ms": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_int32), C.c_int]),
    "lt_stage_name": (C.c_char_p, [C.c_int]),
    "lt_device_cache_trim": (C.c_int, [C.c_size_t]),
    "lt_device_cache_stats": (C.c_int, [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "lt_device_cache_counters": (C.c_int, [C.POINTER(C.c_ulonglong)] * 4),
    "lt_host_memory_stats": (C.c_int, [C.POINTER(C.c_size_t)] * 3),
    "lt_set_download_method": (C.c_int, [_P, C.c_int]),
    "lt_download_stats": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "lt_last_threshold_path": (C.c_int, [_P]),
    "lt_last_adaptive_path": (C.c_int, [_P]),
}

_lib = None


def build(force=False):
    """Compile the shared library in-tree with hipcc for gfx950 (csrc/Makefile)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc, "-s", "-j8"] + (["-B"] if force else [])
    subprocess.check_call(cmd)
    return LIB_PATH


def load():
    """dlopen the library and bind every declared symbol.  Raises NativeError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  lane_tracker_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and this table ever diverge
        fn.restype = res
        fn.argtypes = args
    if lib.lt_abi_version() != ABI_VERSION:
        raise NativeError("ABI version mismatch between _native.py and liblane_tracker_amd.so")
    _lib = lib
    return lib


def exported_symbols():
    return list(_SIGNATURES)


def _check(rc):
    if rc != 0:
        msg = load().lt_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(msg)
        raise NativeError(f"lane_tracker_amd error {rc}: {msg}")


class _PinnedPool:
    """Page-locked host blocks behind NumPy arrays (lt_host_alloc).  An array handed out keeps its block until
    the array and every view of it are gone; the block then goes back to the pool (or to the driver once the pool
    holds enough of that size).  Beyond `limit` bytes outstanding, or when the allocation fails, callers get a
    plain NumPy array -- slower copies, same results."""

    def __init__(self, limit=8 << 30, keep_per_size=4):
        self.limit, self.keep = limit, keep_per_size
        self.free, self.outstanding, self.kinds = {}, 0, {}

    def empty(self, shape, dtype=np.uint8):
        nbytes = math.prod(shape) * np.dtype(dtype).itemsize
        if nbytes == 0 or self.outstanding + nbytes > self.limit:
            return np.empty(shape, dtype)
        blocks = self.free.get(nbytes)
        if blocks:
            ptr = blocks.pop()
        else:
            out = C.c_void_p()
            if load().lt_host_alloc(nbytes, C.byref(out)) != 0 or not out.value:
                return np.empty(shape, dtype)
            ptr = out.value
        kind = self.kinds.get(nbytes)
        if kind is None:
            kind = self.kinds[nbytes] = C.c_uint8 * nbytes
        buf = kind.from_address(ptr)
        self.outstanding += nbytes
        fin = weakref.finalize(buf, self._release, nbytes, ptr)
        fin.atexit = False                       # at interpreter exit the driver reclaims everything
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def _release(self, nbytes, ptr):
        self.outstanding -= nbytes
        blocks = self.free.setdefault(nbytes, [])
        if len(blocks) < self.keep:
            blocks.append(ptr)
        else:
            load().lt_host_free(ptr)


_pinned = _PinnedPool()


class _FramePool:
    """Ordinary (pageable) host blocks behind the NumPy arrays a window of annotated frames is handed out in.  A fresh
    anonymous mapping costs a page fault -- and a cleared page -- per 4 KB (2 MB with transparent huge pages) at first touch:
    0.7 GB per window of 256 1280x720 frames, 40-50 ms spread over the copy threads, as much as the rest of the window
    takes.  So the blocks are kept: an array handed out keeps its block until the array and every view of it are gone, then the
    block waits here for the next window of that size.  At most `keep` blocks per size and `limit` bytes idle are kept
    (`frames_pool_limit` changes them; `frames_trim()` gives the idle blocks back -- `LaneTracker.close()` calls it when the last
    tracker of the process closes).  Returned frames are VIEWS of one window-sized block: keeping one frame keeps its window's
    block alive (copy the frame to let the window go)."""

    def __init__(self, limit=8 << 30, keep_per_size=5):
        self.limit, self.keep = limit, keep_per_size
        self.free, self.idle_bytes = {}, 0

    def empty(self, shape, dtype=np.uint8):
        import mmap
        nbytes = math.prod(shape) * np.dtype(dtype).itemsize
        if nbytes < (1 << 20):
            return np.empty(shape, dtype)
        blocks = self.free.get(nbytes)
        if blocks:
            mm = blocks.pop()
            self.idle_bytes -= nbytes
        else:
            try:
                mm = mmap.mmap(-1, nbytes)
                if hasattr(mmap, "MADV_HUGEPAGE"):
                    try:
                        mm.madvise(mmap.MADV_HUGEPAGE)
                    except OSError:
                        pass
            except (OSError, ValueError):
                return np.empty(shape, dtype)
        buf = (C.c_uint8 * nbytes).from_buffer(mm)
        fin = weakref.finalize(buf, self._release, nbytes, mm)
        fin.atexit = False
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def _release(self, nbytes, mm):
        blocks = self.free.setdefault(nbytes, [])
        if len(blocks) < self.keep and self.idle_bytes + nbytes <= self.limit:
            blocks.append(mm)
            self.idle_bytes += nbytes
        # else: the mapping goes with its last reference

    def trim(self):
        self.free, self.idle_bytes = {}, 0

    def prefault(self, shape, count, dtype=np.uint8):
        """Make sure `count` blocks for arrays of this shape wait in the pool with every page touched (by the library's copy
        threads): a stream's first windows then find their output memory in place.  First touch of fresh memory runs at about
        10 GB/s on the GPU boxes whatever the thread count (0.7 GB per window of 256 1280x720 frames: 70 ms)."""
        nbytes = math.prod(shape) * np.dtype(dtype).itemsize
        fresh = [self.empty(shape, dtype) for _ in range(min(count, self.keep))]      # (blocks that wait already come first: touching them again costs nothing)
        if fresh and nbytes >= (1 << 20):
            g = host_copy_group()
            lib = load()
            for a in fresh:
                rc = lib.lt_host_touch_async_group(g, a.ctypes.data, nbytes)
                if rc:
                    _check(rc)
            host_copy_group_release(g)
        del fresh                        # (back into the pool, touched)


_frames = _FramePool()


def frames_trim():
    """Give the idle blocks of the frame pool back to the system (blocks behind arrays still alive stay with their arrays)."""
    _frames.trim()


def frames_pool_limit(idle_bytes=None, keep_per_size=None):
    """Limits of the frame pool: at most `idle_bytes` of idle blocks and `keep_per_size` idle blocks per window size (defaults:
    8 GB, 5 -- a stream keeps four windows in flight).  -> (idle_bytes, keep_per_size) in force."""
    if idle_bytes is not None:
        _frames.limit = int(idle_bytes)
    if keep_per_size is not None:
        _frames.keep = int(keep_per_size)
    return _frames.limit, _frames.keep


def frames_prefault(shape, count):
    """`count` touched blocks for arrays of `shape` in the frame pool (see _FramePool.prefault)."""
    _frames.prefault(shape, count)


def frames_empty(shape, dtype=np.uint8):
    """An uninitialised array in ordinary host memory out of a pool of kept blocks (no first-touch page faults after the first
    windows of a stream): what process_batch / process_stream return their annotated frames in."""
    return _frames.empty(shape, dtype)


def device_cache_trim(keep_bytes=0):
    """Hand the device memory closed contexts left in the library's cache back to the driver (all of it beyond keep_bytes)."""
    _check(load().lt_device_cache_trim(int(keep_bytes)))


def host_copy_group():
    """A fresh completion group of the library's host copy threads (lt_host_copy_group_create)."""
    g = C.c_int(0)
    _check(load().lt_host_copy_group_create(C.byref(g)))
    return g.value


def host_copy_group_release(group):
    """Wait for the group's copies and forget it (lt_host_copy_group_destroy)."""
    _check(load().lt_host_copy_group_destroy(int(group)))


def text_bytes(lines_per_frame, line_len=40):
    """[[str, ...], ...] -> (bytes, n_lines): every frame's lines NUL-padded to line_len, frames padded to the longest list."""
    nl = max((len(l) for l in lines_per_frame), default=0)
    blank = b"\0" * line_len
    return b"".join(b"".join(t.encode("ascii", "replace")[:line_len].ljust(line_len, b"\0") for t in lines) + blank * (nl - len(lines))
                    for lines in lines_per_frame), nl


def text_blend(frames, font, text, n_lines, line_len=40, origin=(20, 8), step=35):
    """The text lines of `frames` (n, H, W, 3) u8 drawn in place on the calling thread (lt_text_blend_host: lt_overlay_text's
    arithmetic on the host).  font = (atlas (g, gh, gw) u8, advance (g,) u8, first_char); text = n * n_lines * line_len bytes."""
    f = _font_args.get(id(font))
    if f is None or f[0] is not font:                 # (addresses and sizes of a font, once: .ctypes.data costs a microsecond a time)
        atlas, advance, first_char = font
        f = _font_args[id(font)] = (font, atlas.ctypes.data, advance.ctypes.data, int(first_char), atlas.shape[0], atlas.shape[2], atlas.shape[1])
    n, H, W = frames.shape[0], frames.shape[1], frames.shape[2]
    rc = (_lib or load()).lt_text_blend_host(frames.ctypes.data, H * W * 3, n, H, W, f[1], f[2], f[3], f[4], f[5], f[6], text, n_lines, line_len,
                                 int(origin[0]), int(origin[1]), int(step))
    if rc:
        _check(rc)


_font_args = {}


def host_text_async(group, dst, src, rows, font, text, n_lines, line_len=40, origin=(20, 8), step=35):
    """On the library's copy threads, in `group`: the two runs of rows `rows` = (a0, a1, b0, b1) (or one run (a0, a1)) of every frame
    of `src` into `dst` (both (n, H, W, 3) u8, C-contiguous), then that frame's text lines drawn over them
    (lt_host_text_async_group).  font None / n_lines 0: only the rows."""
    n, H, W = dst.shape[0], dst.shape[1], dst.shape[2]
    if font is None or not n_lines:
        atlas = advance = None
        first_char = g = gw = gh = 0
        text, n_lines = None, 0
    else:
        atlas, advance, first_char = font
        g, gh, gw = atlas.shape
    r4 = np.array((list(rows) + [H, H])[:4] if len(rows) == 2 else list(rows), np.int32)
    _check(load().lt_host_text_async_group(int(group), dst.ctypes.data, H * W * 3, src.ctypes.data, H * W * 3, n, r4.ctypes.data, H, W,
                                           None if atlas is None else atlas.ctypes.data, None if advance is None else advance.ctypes.data,
                                           int(first_char), int(g), int(gw), int(gh), text, int(n_lines), int(line_len), int(origin[0]),
                                           int(origin[1]), int(step)))


def host_text_now(group, frame, font, text, n_lines, line_len=40, origin=(20, 8), step=35):
    """The text lines of ONE frame `frame` (1, H, W, 3) u8, drawn before the call returns and behind the copies of `group`
    (lt_host_text_now_group): the wait for the rows under the text and the drawing in one call."""
    f = _font_args.get(id(font))
    if f is None or f[0] is not font:
        atlas, advance, first_char = font
        f = _font_args[id(font)] = (font, atlas.ctypes.data, advance.ctypes.data, int(first_char), atlas.shape[0], atlas.shape[2], atlas.shape[1])
    rc = (_lib or load()).lt_host_text_now_group(group, frame.ctypes.data, frame.shape[1], frame.shape[2], f[1], f[2], f[3], f[4], f[5], f[6], text,
                                                 n_lines, line_len, origin[0], origin[1], step)
    if rc:
        _check(rc)


def host_copy_stats():
    """{'busy_s', 'bytes', 'pieces', 'threads'} of the library's copy threads since the process started (lt_host_copy_stats)."""
    b, y, p, t = C.c_double(), C.c_double(), C.c_longlong(), C.c_int()
    _check(load().lt_host_copy_stats(C.byref(b), C.byref(y), C.byref(p), C.byref(t)))
    return {"busy_s": b.value, "bytes": y.value, "pieces": p.value, "threads": t.value}


def device_cache_stats():
    """{'kept_bytes', 'live_bytes', 'limit_bytes', 'kept_blocks'} of the library's device-memory cache (lt_device_cache_stats)."""
    k, l, m, n = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_int()
    _check(load().lt_device_cache_stats(C.byref(k), C.byref(l), C.byref(m), C.byref(n)))
    return {"kept_bytes": k.value, "live_bytes": l.value, "limit_bytes": m.value, "kept_blocks": n.value}


def host_memory_stats():
    """{'staging_bytes', 'queued_pieces', 'pending_pieces'}: page-locked staging the library holds, and the copy threads' backlog."""
    v = [C.c_size_t() for _ in range(3)]
    _check(load().lt_host_memory_stats(*[C.byref(x) for x in v]))
    return dict(zip(("staging_bytes", "queued_pieces", "pending_pieces"), (int(x.value) for x in v)))


def device_cache_counters():
    """{'hits', 'misses', 'evicted_blocks', 'evicted_bytes'} of the device-memory cache since the process started."""
    v = [C.c_ulonglong() for _ in range(4)]
    _check(load().lt_device_cache_counters(*[C.byref(x) for x in v]))
    return dict(zip(("hits", "misses", "evicted_blocks", "evicted_bytes"), (int(x.value) for x in v)))


def pinned_empty(shape, dtype=np.uint8):
    """An uninitialised array in page-locked host memory (falls back to np.empty)."""
    return _pinned.empty(shape, dtype)


def _u8(a, shape_tail=None):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return a


def filter_params(filter_type="bilateral", ksize_r=15, C_r=8, ksize_b=35, C_b=5, mask_noise=False,
                  noise_thresh=140, ksize_noise=65, C_noise=10):
    ft = {"bilateral": 0, "neighborhood": 1}.get(filter_type, 2)   # 2 -> ValueError from the library (:220)
    return FilterParams(ft, int(ksize_r), int(C_r), int(ksize_b), int(C_b), int(bool(mask_noise)),
                        int(noise_thresh), int(ksize_noise), int(C_noise))


def search_params(window_width=30, window_height=40, search_range=20, mu=0.1, no_success_limit=8,
                  start_slice=0.25, ignore_sides=360, ignore_bottom=30, bandwidth=25, partial=1.0):
    return SearchParams(int(window_width), int(window_height), int(search_range), int(no_success_limit),
                        int(ignore_sides), int(ignore_bottom), int(bandwidth), 0, float(mu), float(start_slice),
                        float(partial))


def make_calib(img_size, warped_size, cam_matrix, dist_coeffs, M):
    cal = Calib()
    cal.img_w, cal.img_h = int(img_size[0]), int(img_size[1])
    cal.warp_w, cal.warp_h = int(warped_size[0]), int(warped_size[1])
    cal.cam_matrix[:] = [float(v) for v in np.asarray(cam_matrix, np.float64).reshape(9)]
    d = np.asarray(dist_coeffs, np.float64).reshape(-1)
    cal.dist_coeffs[:] = [float(v) for v in (list(d[:5]) + [0.0] * 5)[:5]]
    cal.M[:] = [float(v) for v in np.asarray(M, np.float64).reshape(9)]
    return cal


def calib_tables(cal):
    """The host-built calibration tables of a `Calib` (no GPU needed): dict with the warp map, the source-row
    window, the undistortion map of those rows, the Lab tables and the ellipse half-widths."""
    lib = load()
    r0, r1 = C.c_int(0), C.c_int(0)
    _check(lib.lt_calib_source_rows(C.byref(cal), C.byref(r0), C.byref(r1)))
    wxy = np.empty((cal.warp_h, cal.warp_w, 2), np.int16)
    wfr = np.empty((cal.warp_h, cal.warp_w), np.uint16)
    _check(lib.lt_calib_warp_table(C.byref(cal), wxy.ctypes.data, wfr.ctypes.data))
    uxy = np.empty((r1.value - r0.value, cal.img_w, 2), np.int16)
    ufr = np.empty((r1.value - r0.value, cal.img_w), np.uint16)
    _check(lib.lt_calib_undistort_table(C.byref(cal), r0.value, r1.value, uxy.ctypes.data, ufr.ctypes.data))
    gamma, cbrt, coef = np.empty(256, np.uint16), np.empty(3072, np.uint16), np.empty(9, np.int32)
    _check(lib.lt_calib_lab_tables(gamma.ctypes.data, cbrt.ctypes.data, coef.ctypes.data))
    ell = {}
    for k in (5, 29, 55):
        dx, taps = np.empty(k, np.int32), C.c_int(0)
        _check(lib.lt_calib_ellipse(k, dx.ctypes.data, C.byref(taps)))
        ell[k] = (dx, taps.value)
    return dict(source_rows=(r0.value, r1.value), warp_xy=wxy, warp_frac=wfr, und_xy=uxy, und_frac=ufr, gamma=gamma,
                cbrt=cbrt, lab_coeffs=coef, ellipse=ell)


def pack_polygons(polygons):
    """[(left_y, left_x, right_y, right_x), ...] -> (left counts, right counts, left (y, x) pairs, right (y, x) pairs), the
    form lt_overlay_run takes."""
    ln = np.array([len(p[0]) for p in polygons], np.int32)
    rn = np.array([len(p[2]) for p in polygons], np.int32)

    def pairs(ys, xs, counts):
        out = np.empty((int(counts.sum()), 2), np.int32)
        at = 0
        for y, x, m in zip(ys, xs, counts):
            out[at:at + m, 0] = y
            out[at:at + m, 1] = x
            at += m
        return out
    return ln, rn, pairs([p[0] for p in polygons], [p[1] for p in polygons], ln), pairs([p[2] for p in polygons], [p[3] for p in polygons], rn)


def poly_points(warped_size, coeffs, ploty, ploty2):
    """get_poly_points (reference :511-528) for many pairs of parabolas at once, packed for overlay_run_packed: coeffs (n, 6)
    = left a, b, c, right a, b, c; ploty / ploty2 as LaneTracker._plot_rows gives them.  Host-only (lt_poly_points)."""
    lib = load()
    coeffs = np.ascontiguousarray(coeffs, np.float64).reshape(-1, 6)
    ploty, ploty2 = np.ascontiguousarray(ploty, np.float64), np.ascontiguousarray(ploty2, np.float64)
    n, rows = coeffs.shape[0], ploty.shape[0]
    ln, rn = np.empty(n, np.int32), np.empty(n, np.int32)
    lyx, ryx = np.empty((n * rows, 2), np.int32), np.empty((n * rows, 2), np.int32)
    _check(lib.lt_poly_points(int(warped_size[0]), int(warped_size[1]), coeffs.ctypes.data, n, ploty.ctypes.data, ploty2.ctypes.data,
                              rows, ln.ctypes.data, rn.ctypes.data, lyx.ctypes.data, ryx.ctypes.data))
    return ln, rn, lyx[:int(ln.sum())], ryx[:int(rn.sum())]


class Context:
    """One device context = one HIP stream + calibration tables + `capacity` frame slots in HBM."""

    def __init__(self, img_size, warped_size, cam_matrix, dist_coeffs, M, device=0, capacity=1):
        self._h = None
        lib = load()
        cal = make_calib(img_size, warped_size, cam_matrix, dist_coeffs, M)
        h = _P()
        _check(lib.lt_create(C.byref(cal), int(device), C.byref(h)))
        self._h = h
        self.lib = lib
        self.img_w, self.img_h, self.warp_w, self.warp_h = cal.img_w, cal.img_h, cal.warp_w, cal.warp_h
        self.reserve(capacity)

    def close(self):
        if self._h is not None:
            self.lib.lt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- bookkeeping
    def reserve(self, capacity):
        _check(self.lib.lt_reserve(self._h, int(capacity)))
        self.capacity = max(getattr(self, "capacity", 0), int(capacity))

    def info(self):
        i = Info()
        _check(self.lib.lt_get_info(self._h, C.byref(i)))
        return i

    def sync(self):
        _check(self.lib.lt_sync(self._h))

    def set_streams(self, n):
        """Spread the slots over n HIP streams (slices of the capacity overlap each other's stages)."""
        _check(self.lib.lt_set_streams(self._h, int(n)))

    # -- data movement
    def source_rows(self):
        """Camera rows [row0, row1) the path reads."""
        a, b = C.c_int(0), C.c_int(0)
        _check(self.lib.lt_get_source_rows(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def upload_frame_rows(self, frames, first=0, enqueue=False):
        """Like upload_frames, but only the camera rows the path reads cross the bus (not enough for the overlay).
        enqueue: no wait for the copy (lt_upload_frame_rows_enqueue) -- the array handed to the library is returned and must stay
        alive and unchanged until a call that waits for work launched over these slots afterwards (download_record, sync)."""
        f = _u8(frames).reshape(-1, self.img_h, self.img_w, 3)
        if enqueue:
            _check(self.lib.lt_upload_frame_rows_enqueue(self._h, f.ctypes.data, first, f.shape[0]))
            return f
        _check(self.lib.lt_upload_frame_rows(self._h, f.ctypes.data, first, f.shape[0]))

    def upload_frame_rows_async(self, frames, first=0):
        """Stream-ordered upload_frame_rows (no host wait).  `frames` must be a C-contiguous u8 array that stays alive
        and unchanged until the next sync() -- use pinned_empty() for it; returns the array handed to the library."""
        f = np.asarray(frames)
        if f.dtype != np.uint8 or not f.flags["C_CONTIGUOUS"]:
            raise ValueError("upload_frame_rows_async needs a C-contiguous uint8 array (no hidden copy may be made)")
        f = f.reshape(-1, self.img_h, self.img_w, 3)
        _check(self.lib.lt_upload_frame_rows_async(self._h, f.ctypes.data, first, f.shape[0]))
        return f

    def upload_frame_rest(self, frames, first=0, rows=None):
        """The rows upload_frame_rows left out, on a copy stream beside the compute streams (for the overlay); with `rows` (the
        ADDRESS of four int32 {a0, a1, b0, b1}) only those of them inside the two runs -- what present_frame with the same runs
        reads.  Returns the array actually handed to the library: keep it alive until the next sync() / download."""
        f = _u8(frames).reshape(-1, self.img_h, self.img_w, 3)
        if rows is None:
            _check(self.lib.lt_upload_frame_rest(self._h, f.ctypes.data, first, f.shape[0]))
        else:
            _check(self.lib.lt_upload_frame_rest_rows(self._h, f.ctypes.data, first, f.shape[0], rows))
        return f

    def upload_frames(self, frames, first=0):
        f = _u8(frames).reshape(-1, self.img_h, self.img_w, 3)
        _check(self.lib.lt_upload_frames(self._h, f.ctypes.data, first, f.shape[0]))
        return f.shape[0]

    def upload_masks(self, masks, first=0):
        m = _u8(masks).reshape(-1, self.warp_h, self.warp_w)
        _check(self.lib.lt_upload_masks(self._h, m.ctypes.data, first, m.shape[0]))
        return m.shape[0]

    def upload_bev(self, bev, first=0):
        b = _u8(bev).reshape(-1, self.warp_h, self.warp_w, 3)
        _check(self.lib.lt_upload_bev(self._h, b.ctypes.data, first, b.shape[0]))
        return b.shape[0]

    def download_masks(self, n, first=0):
        out = np.empty((n, self.warp_h, self.warp_w), np.uint8)
        _check(self.lib.lt_download_masks(self._h, first, n, out.ctypes.data))
        return out

    def download_plane(self, plane, n, first=0):
        out = np.empty((n, self.warp_h, self.warp_w), np.uint8)
        _check(self.lib.lt_download_plane(self._h, plane, first, n, out.ctypes.data))
        return out

    def download_undistorted(self, n, first=0):
        i = self.info()
        out = np.empty((n, i.src_row1 - i.src_row0, self.img_w, 3), np.uint8)
        _check(self.lib.lt_download_undistorted(self._h, first, n, out.ctypes.data))
        return out

    # ---- presentation stage (draw_lane overlay, bird's-eye image) ----
    def overlay_configure(self, Minv):
        m = np.ascontiguousarray(Minv, np.float64).reshape(9)
        _check(self.lib.lt_overlay_configure(self._h, m.ctypes.data))

    def overlay_run(self, polygons, first=0, alpha=0.3, rows=None):
        """polygons: one (left_y, left_x, right_y, right_x) tuple per slot (empty arrays: plain copy); rows: None or the
        ADDRESS of four int32 {a0, a1, b0, b1} -- only these two runs of rows of every frame are drawn."""
        self.overlay_run_packed(*pack_polygons(polygons), first=first, alpha=alpha, rows=rows)

    def overlay_run_packed(self, ln, rn, lyx, ryx, first=0, alpha=0.3, rows=None):
        """The same with the polygons already packed (pack_polygons / poly_points): int32 counts per slot and the (y, x) pairs
        of all slots back to back."""
        ln, rn = np.ascontiguousarray(ln, np.int32), np.ascontiguousarray(rn, np.int32)
        lyx, ryx = np.ascontiguousarray(lyx, np.int32), np.ascontiguousarray(ryx, np.int32)
        if len(rn) != len(ln) or lyx.size != 2 * int(ln.sum()) or ryx.size != 2 * int(rn.sum()):
            raise ValueError("point lists do not match their counts")
        _check(self.lib.lt_overlay_run_rows(self._h, first, len(ln), ln.ctypes.data, rn.ctypes.data,
                                            lyx.ctypes.data if lyx.size else None, ryx.ctypes.data if ryx.size else None,
                                            float(alpha), rows))

    def overlay_run_strip_packed(self, ln, rn, lyx, ryx, first=0, alpha=0.3):
        """overlay_run_packed in strip mode (lt_overlay_run_strip): only the rows the lane can reach, packed per slot."""
        ln, rn = np.ascontiguousarray(ln, np.int32), np.ascontiguousarray(rn, np.int32)
        lyx, ryx = np.ascontiguousarray(lyx, np.int32), np.ascontiguousarray(ryx, np.int32)
        if len(rn) != len(ln) or lyx.size != 2 * int(ln.sum()) or ryx.size != 2 * int(rn.sum()):
            raise ValueError("point lists do not match their counts")
        _check(self.lib.lt_overlay_run_strip(self._h, first, len(ln), ln.ctypes.data, rn.ctypes.data,
                                             lyx.ctypes.data if lyx.size else None, ryx.ctypes.data if ryx.size else None, float(alpha)))

    def overlay_run_strip_coeffs(self, coeffs, draw, ploty, ploty2, first=0, alpha=0.3):
        """Strips from the lanes' averaged coefficients (lt_overlay_run_strip_coeffs): coeffs (n, 6) f64, draw (n,) u8 (0: no lane in
        that frame); plot points and polygon intervals are formed on the device.  -> False where that form does not exist."""
        coeffs = np.ascontiguousarray(coeffs, np.float64).reshape(-1, 6)
        draw = np.ascontiguousarray(draw, np.uint8)
        if len(draw) != len(coeffs):
            raise ValueError("one draw byte per frame")
        rc = self.lib.lt_overlay_run_strip_coeffs(self._h, first, len(coeffs), coeffs.ctypes.data, draw.ctypes.data, ploty.ctypes.data,
                                                  ploty2.ctypes.data, len(ploty), float(alpha))
        if rc == -5:
            return False
        if rc:
            _check(rc)
        return True

    def strip_download_async(self, out, first, group):
        """The strips of slots first .. first+len(out)-1 into rows overlay_rows() of the frames `out` (n, H, W, 3) u8, C-contiguous,
        ordinary memory; complete when the host-copy group `group` has been waited for (lt_strip_download_async)."""
        if out.dtype != np.uint8 or not out.flags["C_CONTIGUOUS"] or out.shape[1:] != (self.img_h, self.img_w, 3):
            raise ValueError("strip_download_async needs a C-contiguous uint8 array (n, H, W, 3)")
        _check(self.lib.lt_strip_download_async(self._h, first, out.shape[0], out.ctypes.data, self.img_h * self.img_w * 3, int(group)))

    def warm(self, sws=None, band=None, annotate=0):
        """Set up now what the first searches / chains / overlays would set up on the way (lt_warm); annotate: 0 none, 1 whole
        annotated frames, 2 strips."""
        _check(self.lib.lt_warm(self._h, None if sws is None else C.byref(sws), None if band is None else C.byref(band), int(annotate)))

    def overlay_set_font(self, atlas, advance, first_char=32):
        """atlas: (n_glyphs, glyph_h, glyph_w) u8 alpha cells; advance: (n_glyphs,) u8."""
        atlas = np.ascontiguousarray(atlas, np.uint8)
        advance = np.ascontiguousarray(advance, np.uint8)
        _check(self.lib.lt_overlay_set_font(self._h, atlas.ctypes.data, advance.ctypes.data, int(first_char),
                                            atlas.shape[0], atlas.shape[2], atlas.shape[1]))

    def overlay_text(self, lines_per_slot, first=0, origin=(20, 8), step=35, line_len=40):
        """lines_per_slot: one list of strings per slot (ASCII)."""
        n = len(lines_per_slot)
        nl = max((len(l) for l in lines_per_slot), default=0)
        if n == 0 or nl == 0:
            return
        blank = b"\0" * line_len
        buf = b"".join(b"".join(t.encode("ascii", "replace")[:line_len].ljust(line_len, b"\0") for t in lines) + blank * (nl - len(lines))
                       for lines in lines_per_slot)
        _check(self.lib.lt_overlay_text(self._h, first, n, buf, nl, line_len, int(origin[0]), int(origin[1]), int(step)))

    def download_overlay(self, n, first=0):
        out = pinned_empty((n, self.img_h, self.img_w, 3))
        _check(self.lib.lt_download_overlay(self._h, first, n, out.ctypes.data))
        return out

    def download_overlay_async(self, out, first=0, rows=None):
        """Enqueue the copy of the annotated frames of slots first .. first+len(out)-1 into `out` (a C-contiguous u8 array
        (n, H, W, 3), from pinned_empty()); valid after the next sync().  rows: None or the ADDRESS of four int32 -- only these two
        runs of rows of every frame (the caller fills the others)."""
        if out.dtype != np.uint8 or not out.flags["C_CONTIGUOUS"] or out.shape[1:] != (self.img_h, self.img_w, 3):
            raise ValueError("download_overlay_async needs a C-contiguous uint8 array (n, H, W, 3)")
        _check(self.lib.lt_download_overlay_rows_async(self._h, first, out.shape[0], out.ctypes.data, rows))

    def set_download_method(self, method):
        """-1: engine or kernel by measurement (default); 0: copy engine; 1: kernel."""
        _check(self.lib.lt_set_download_method(self._h, int(method)))

    def download_stats(self):
        eg, kg = C.c_double(), C.c_double()
        ec, kc, m = C.c_int(), C.c_int(), C.c_int()
        _check(self.lib.lt_download_stats(self._h, C.byref(eg), C.byref(ec), C.byref(kg), C.byref(kc), C.byref(m)))
        return {"engine_GBs": round(eg.value, 1), "engine_copies": ec.value, "kernel_GBs": round(kg.value, 1), "kernel_copies": kc.value,
                "method": "kernel" if m.value == 1 else "engine"}

    def download_overlay_wait(self):
        """Block until the frames of every download_overlay_async have landed (later uploads / masks keep running)."""
        _check(self.lib.lt_download_overlay_wait(self._h))

    def download_bev(self, n, first=0):
        out = pinned_empty((n, self.warp_h, self.warp_w, 3))
        _check(self.lib.lt_download_bev(self._h, first, n, out.ctypes.data))
        return out

    def download_records(self, n, first=0):
        out = np.zeros(n, RECORD_DTYPE)
        _check(self.lib.lt_download_records(self._h, first, n, out.ctypes.data))
        return out

    def download_record(self, slot):
        """The record of one slot -> (left coeffs, right coeffs, detected, fit_flags): download_records(1, slot) without
        the per-call array and field lookups (LaneTracker.process() asks once per frame, with the device idle behind it)."""
        v = self.__dict__.get("_rec1")
        if v is None:
            r = np.zeros(1, RECORD_DTYPE)
            v = self._rec1 = (r, r.ctypes.data, r["left_coeffs"][0], r["right_coeffs"][0], r["detected"], r["fit_flags"])
        rc = self.lib.lt_download_records(self._h, slot, 1, v[1])
        if rc:
            _check(rc)
        return v[2].copy(), v[3].copy(), bool(v[4][0]), int(v[5][0])

    def overlay_rows(self):
        """(row0, row1): the camera rows in which the lane overlay can change a pixel (lt_overlay_rows)."""
        a, b = C.c_int(0), C.c_int(0)
        _check(self.lib.lt_overlay_rows(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def present_frame(self, slot, left_n, right_n, left_yx, right_yx, text, n_lines, line_len, out, rows=None, origin=(20, 8),
                      step=35, alpha=0.3):
        """draw_lane() / print_failure() of ONE slot in one library call (lt_present_frame), for a caller that keeps the packed
        polygon in buffers of its own: left_n / right_n / left_yx / right_yx are ADDRESSES (int32 count; int32 (y, x) pairs, or
        None when the count is 0), text is None or n_lines * line_len bytes (lines padded with NUL), out the (1, H, W, 3) array
        the frame lands in, rows None or the ADDRESS of four int32 {a0, a1, b0, b1}: only these two runs of rows are written."""
        rc = self.lib.lt_present_frame(self._h, slot, left_n, right_n, left_yx, right_yx, alpha, text, n_lines, line_len, origin[0],
                                       origin[1], step, out.ctypes.data, rows)
        if rc:
            _check(rc)
        return out

    def present_lane_async(self, slot, left_n, right_n, left_yx, right_yx, out, rows, alpha=0.3):
        """First half of present_frame (lt_present_lane_async): the polygon drawn, the rows the lane can reach on their way into
        `out`; no wait.  `rows` (address of four int32) is required."""
        rc = self.lib.lt_present_lane_async(self._h, slot, left_n, right_n, left_yx, right_yx, alpha, out.ctypes.data, rows)
        if rc:
            _check(rc)

    def present_lane_from_fit_async(self, slot, prev_sum_addr, count, ploty_addr, ploty2_addr, n_rows, out, rows, alpha=0.3):
        """The lane drawn by the device itself behind the slot's search (lt_present_lane_from_fit_async) -> True, or False where
        that form does not exist (LT_ERR_STATE: the caller draws once it has the record).  Addresses of f64 buffers the caller keeps."""
        rc = self.lib.lt_present_lane_from_fit_async(self._h, slot, prev_sum_addr, count, ploty_addr, ploty2_addr, n_rows, alpha,
                                                     out.ctypes.data, rows)
        if rc == -5:
            return False
        if rc:
            _check(rc)
        return True

    def lane_spans_from_fit(self, fit6, prev_sum, count, ploty, ploty2, detected=True, fit_flags=0):
        """Test hook (lt_lane_spans_from_fit): the row intervals (warp_h, 2) int16 the device forms for a fit given by value."""
        fit6 = np.ascontiguousarray(fit6, np.float64).reshape(6)
        ps = np.ascontiguousarray(prev_sum if prev_sum is not None else np.zeros(6), np.float64).reshape(6)
        ploty, ploty2 = np.ascontiguousarray(ploty, np.float64), np.ascontiguousarray(ploty2, np.float64)
        out = np.empty((self.warp_h, 2), np.int16)
        _check(self.lib.lt_lane_spans_from_fit(self._h, fit6.ctypes.data, int(bool(detected)), int(fit_flags), ps.ctypes.data, int(count),
                                               ploty.ctypes.data, ploty2.ctypes.data, len(ploty), out.ctypes.data))
        return out

    def present_finish(self, slot, text, n_lines, line_len, out, rows, origin=(20, 8), step=35):
        """Second half (lt_present_finish): the text lines, their rows into `out`, and the wait for both halves."""
        rc = self.lib.lt_present_finish(self._h, slot, text, n_lines, line_len, origin[0], origin[1], step, out.ctypes.data, rows)
        if rc:
            _check(rc)
        return out

    def download_pixels(self, slot, side):
        cnt = C.c_int(0)
        _check(self.lib.lt_download_pixels(self._h, slot, side, None, None, 0, C.byref(cnt)))
        n = cnt.value
        ys, xs = np.empty(n, np.int32), np.empty(n, np.int32)
        if n:
            _check(self.lib.lt_download_pixels(self._h, slot, side, ys.ctypes.data, xs.ctypes.data, n, C.byref(cnt)))
        return ys.astype(np.int64), xs.astype(np.int64)

    def download_centroids(self, slot, side):
        cap = self.info().max_levels + 2
        out = np.zeros(cap, np.int32)
        cnt = C.c_int(0)
        _check(self.lib.lt_download_centroids(self._h, slot, side, out.ctypes.data, cap, C.byref(cnt)))
        return [int(v) for v in out[:cnt.value]]

    def download_lane_lists(self, slot, want_centroids=True):
        """(left_y, left_x, right_y, right_x, left centroids or None, right centroids or None) of the slot's last search in ONE round
        trip to the device (lt_download_lane_lists); the lists are what download_pixels / download_centroids return."""
        cap = 1 << 16
        for _ in range(2):
            arrs = self.__dict__.get("_list_buf")
            if arrs is None or arrs[0].size < cap:
                arrs = self._list_buf = [np.empty(cap, np.int32) for _ in range(4)]       # (kept: 1 MB, filled by every call)
            cnt, ccnt = (C.c_int * 2)(), (C.c_int * 2)()
            ccap = self.info().max_levels + 2 if want_centroids else 0
            cl, cr = np.zeros(max(ccap, 1), np.int32), np.zeros(max(ccap, 1), np.int32)
            _check(self.lib.lt_download_lane_lists(self._h, int(slot), arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, arrs[3].ctypes.data,
                                                   arrs[0].size, cnt, int(bool(want_centroids)), cl.ctypes.data, cr.ctypes.data, ccap, ccnt))
            if max(cnt[0], cnt[1]) <= arrs[0].size:
                break
            cap = max(cnt[0], cnt[1])
        nl, nr = cnt[0], cnt[1]
        out = (arrs[0][:nl].astype(np.int64), arrs[1][:nl].astype(np.int64), arrs[2][:nr].astype(np.int64), arrs[3][:nr].astype(np.int64))
        if not want_centroids:
            return out + (None, None)
        return out + ([int(v) for v in cl[:ccnt[0]]], [int(v) for v in cr[:ccnt[1]]])

    def copy_records_to_device(self, n, dst_ptr, first=0):
        _check(self.lib.lt_copy_records_to_device(self._h, first, n, C.c_void_p(int(dst_ptr))))

    def enqueue_records_to_device(self, n, dst_ptr, first=0):
        """Stream-ordered copy (no host wait): valid in dst after the next sync()."""
        _check(self.lib.lt_enqueue_records_to_device(self._h, first, n, C.c_void_p(int(dst_ptr))))

    def set_frame_base(self, n, first_frame, first=0):
        _check(self.lib.lt_set_frame_base(self._h, first, n, int(first_frame)))

    # -- compute (asynchronous on the context's stream)
    def mask_run(self, n, fp=None, first=0, reuse_front=False):
        """undistort + warp + filter_lane_points over slots first .. first+n-1.  reuse_front: the slots' frames have been through
        mask_run already and only the filter parameters differ (the second try of a frame): lt_mask_rerun skips the front end where
        the slots' planes are still current."""
        fp = fp or filter_params()
        _check((self.lib.lt_mask_rerun if reuse_front else self.lib.lt_mask_run)(self._h, first, n, C.byref(fp)))

    def filter_run(self, n, fp=None, first=0):
        fp = fp or filter_params()
        _check(self.lib.lt_filter_run(self._h, first, n, C.byref(fp)))

    def sws_fit_run(self, n, sp=None, first=0):
        sp = sp or search_params()
        _check(self.lib.lt_sws_fit_run(self._h, first, n, C.byref(sp)))

    def band_fit_run(self, n, prev_coeffs, sp=None, first=0):
        sp = sp or search_params()
        prev = np.ascontiguousarray(prev_coeffs, np.float64).reshape(n, 6)
        _check(self.lib.lt_band_fit_run(self._h, first, n, C.byref(sp), prev.ctypes.data))

    def band_fit_chain_run(self, n, seed_coeffs=None, sp=None, first=0):
        """Band search + fit of slots first .. first+n-1 as consecutive frames of one stream, chained on the device:
        frame k+1 searches around frame k's fit; the first around `seed_coeffs` (6 doubles) or, if None, around the
        record of slot first-1.  Records behind a frame the chain could not build on come back with mode 255."""
        sp = sp or search_params()
        seed = None if seed_coeffs is None else np.ascontiguousarray(seed_coeffs, np.float64).reshape(6)
        _check(self.lib.lt_band_fit_chain_run(self._h, first, n, C.byref(sp), None if seed is None else seed.ctypes.data))

    def set_search_cus(self, n):
        """Reserve n CUs for the chained search (the compute streams are recreated without them); 0 undoes it."""
        _check(self.lib.lt_set_search_cus(self._h, int(n)))

    def set_walk_min_frames(self, frames):
        """Calls of at least `frames` frames take the walking threshold kernels (0: always; negative: the default, 80)."""
        _check(self.lib.lt_set_walk_min_frames(self._h, int(frames)))

    def set_direct_upload(self, on=-1):
        """One frame's rows (upload_frame_rows(enqueue=True) of a frame or two) stored by the calling thread through the PCIe
        aperture instead of copied by the engine: True / False allows / forbids, -1 asks.  -> does this context take the aperture?"""
        r = self.lib.lt_set_direct_upload(self._h, -1 if on == -1 else int(bool(on)))
        if r < 0:
            _check(r)
        return bool(r)

    def direct_upload_count(self):
        return int(self.lib.lt_direct_upload_count(self._h))

    def urgent(self):
        """Context manager: the stage calls inside run on the context's urgent stream (lt_set_urgent) -- behind the work of their
        own slots only, not behind masks of later frames already queued -- and downloads wait for that stream only."""
        import contextlib

        @contextlib.contextmanager
        def scope():                     # re-entrant: the mode ends with the outermost scope
            depth = getattr(self, "_urgent_depth", 0)
            if depth == 0:
                _check(self.lib.lt_set_urgent(self._h, 1))
            self._urgent_depth = depth + 1
            try:
                yield self
            finally:
                self._urgent_depth -= 1
                if self._urgent_depth == 0:
                    _check(self.lib.lt_set_urgent(self._h, 0))
        return scope()

    def band_fit_chain_cancel(self):
        """Chains enqueued so far stop at their next frame (their speculation has been rejected)."""
        _check(self.lib.lt_band_fit_chain_cancel(self._h))

    def band_fit_chain_collect(self, n, first=0):
        """Records of slots first .. first+n-1 as the most recent chain covering them left them; waits for that chain only."""
        out = np.zeros(n, RECORD_DTYPE)
        _check(self.lib.lt_band_fit_chain_collect(self._h, first, n, out.ctypes.data))
        return out

    # -- single-image operators
    def bilateral_adaptive_threshold(self, img, ksize, C_, mode, true_value, false_value):
        a = _u8(img)
        if a.ndim != 2:
            raise ValueError("bilateral_adaptive_threshold expects a single-channel image")
        out = np.empty_like(a)
        _check(self.lib.lt_bilateral_adaptive_threshold(self._h, a.ctypes.data, a.shape[0], a.shape[1], int(ksize),
                                                        int(C_), int(mode), int(true_value), int(false_value),
                                                        out.ctypes.data))
        return out

    def filter_lane_points(self, bev, fp=None):
        fp = fp or filter_params()
        a = _u8(bev)
        if a.ndim != 3 or a.shape[2] != 3:
            raise ValueError("filter_lane_points expects an RGB image (H, W, 3)")
        out = np.empty(a.shape[:2], np.uint8)
        _check(self.lib.lt_filter_lane_points(self._h, a.ctypes.data, a.shape[0], a.shape[1], C.byref(fp),
                                              out.ctypes.data))
        return out

    def morph_ellipse(self, img, k, op, direct=False):
        """op: 'erode' | 'dilate' | 'tophat' | 'open' with the k x k ellipse (k in 5, 29, 55)."""
        a = _u8(img)
        if a.ndim != 2:
            raise ValueError("morph_ellipse expects a single-channel image")
        out = np.empty_like(a)
        code = {"erode": 0, "dilate": 1, "tophat": 2, "open": 3}[op]
        _check(self.lib.lt_morph_ellipse(self._h, a.ctypes.data, a.shape[0], a.shape[1], int(k), code,
                                         int(bool(direct)), out.ctypes.data))
        return out

    def fit_poly2(self, ys, xs):
        """np.polyfit(ys, xs, 2) on the device; rank-deficient inputs get NumPy's minimum-norm answer."""
        ys = np.ascontiguousarray(ys, np.int32)
        xs = np.ascontiguousarray(xs, np.int32)
        coef = (C.c_double * 3)()
        bad = C.c_int(0)
        _check(self.lib.lt_fit_poly2(self._h, ys.ctypes.data, xs.ctypes.data, int(ys.size), self.warp_h, self.warp_w,
                                     coef, C.byref(bad)))
        if bad.value:
            from .lane_tracker import _minimum_norm_parabola
            return _minimum_norm_parabola(ys, xs)
        return np.array(coef[:], np.float64)

    # -- measurement
    def timer_start(self):
        _check(self.lib.lt_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float(0)
        _check(self.lib.lt_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def set_stage_timing(self, enabled):
        _check(self.lib.lt_set_stage_timing(self._h, int(bool(enabled))))

    def last_threshold_path(self):
        """1 = long-walk threshold kernels, 0 = tile kernel, -1 = no bilateral chain has run yet."""
        return int(self.lib.lt_last_threshold_path(self._h))

    def last_adaptive_path(self):
        """'neighborhood' calls: 1 = running box sums, 0 = per-pixel windows, -1 = none yet."""
        return int(self.lib.lt_last_adaptive_path(self._h))

    def stage_reset(self):
        _check(self.lib.lt_stage_reset(self._h))

    def stage_ms(self):
        ms = (C.c_float * NUM_STAGES)()
        ln = (C.c_int32 * NUM_STAGES)()
        _check(self.lib.lt_stage_ms(self._h, ms, ln, NUM_STAGES))
        return {self.lib.lt_stage_name(i).decode(): (ms[i], ln[i]) for i in range(NUM_STAGES)}


class Gather:
    """The RCCL all-gather of lane records between the ranks of one node (lt_gather_*; one per rank, bound to
    the rank's Context).  Collective calls: every rank must make them in the same order."""

    def __init__(self, ctx, rank, world, id_path, timeout_s=120):
        self._g = None
        self.ctx, self.lib = ctx, ctx.lib
        g = _P()
        _check(self.lib.lt_gather_init(ctx._h, int(rank), int(world), os.fsencode(id_path), int(timeout_s), C.byref(g)))
        self._g = g
        self.rank, self.world = int(rank), int(world)

    def reserve(self, records_per_rank):
        _check(self.lib.lt_gather_reserve(self._g, int(records_per_rank)))

    def stage(self, n, at=0, first=0):
        """Records of slots [first, first+n) -> send buffer position `at`; stream-ordered, no host wait."""
        _check(self.lib.lt_gather_stage(self._g, int(first), int(n), int(at)))

    def records(self, n_records):
        """(world, n_records) records: the first n_records staged records of every rank."""
        out = np.zeros((self.world, int(n_records)), RECORD_DTYPE)
        _check(self.lib.lt_gather_records(self._g, int(n_records), out.ctypes.data))
        return out

    def host(self, array):
        """All-gather of a small host array: (world,) + array.shape."""
        a = np.ascontiguousarray(array)
        out = np.empty((self.world,) + a.shape, a.dtype)
        _check(self.lib.lt_gather_host(self._g, a.ctypes.data, a.nbytes, out.ctypes.data))
        return out

    def barrier(self):
        _check(self.lib.lt_gather_barrier(self._g))

    def close(self):
        if self._g is not None:
            self.lib.lt_gather_destroy(self._g)
            self._g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_count():
    n = C.c_int(0)
    _check(load().lt_device_count(C.byref(n)))
    return n.value


def lane_polygon_spans(warp_h, left_y, left_x, right_y, right_x):
    """(warp_h, 2) int16 (lo, hi) column interval per bird's-eye row of draw_lane's filled polygon
    (host-only helper of the overlay stage; rows the polygon does not touch are (32767, -32768))."""
    lyx = np.ascontiguousarray(np.stack([np.asarray(left_y), np.asarray(left_x)], 1).reshape(-1, 2), np.int32)
    ryx = np.ascontiguousarray(np.stack([np.asarray(right_y), np.asarray(right_x)], 1).reshape(-1, 2), np.int32)
    out = np.empty((int(warp_h), 2), np.int16)
    _check(load().lt_lane_polygon_spans(int(warp_h), lyx.ctypes.data if len(lyx) else None, len(lyx),
                                        ryx.ctypes.data if len(ryx) else None, len(ryx), out.ctypes.data))
    return out
