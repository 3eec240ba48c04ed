"""A CPU stand-in for `lane_tracker_amd._native.Context`, backed by the oracle, for tests of the HOST logic of the tracker
(the state machine of `_step`, the chained stream driver `_run_window_chained`, `process_stream`'s window hand-over) where
no GPU is present.  It implements the calls those paths make -- uploads, mask_run, the searches, the chained band search
with its collect / cancel, record / pixel / centroid downloads -- with the reference semantics the oracle restates; the
overlay entry points are absent (annotate=False only).  Test infrastructure: never imported by the product."""
import hashlib

import numpy as np

from lane_tracker_amd import _native
from oracle import oracle as O


def _osp(sp):
    return O.search_params(sp.window_width, sp.window_height, sp.search_range, sp.mu, sp.no_success_limit, sp.start_slice,
                           sp.ignore_sides, sp.ignore_bottom, sp.bandwidth, sp.partial)


def _ofp(fp):
    ft = {0: "bilateral", 1: "neighborhood"}[fp.filter_type]
    return O.filter_params(ft, fp.ksize_r, fp.C_r, fp.ksize_b, fp.C_b, bool(fp.mask_noise), fp.noise_thresh, fp.ksize_noise, fp.C_noise)


class FakeContext:
    mask_cache = {}                  # shared by all instances: (frame digest, filter tuple) -> mask
    calls = None                     # optional list: every chain launch is appended (first, n, seeded by value)

    def __init__(self, img_size, warped_size, cam_matrix, dist_coeffs, M, device=0, capacity=1):
        self.img_w, self.img_h = int(img_size[0]), int(img_size[1])
        self.warp_w, self.warp_h = int(warped_size[0]), int(warped_size[1])
        self.oc = O.make_calib(img_size, warped_size, cam_matrix, dist_coeffs, M)
        self.capacity = capacity
        self.slots = {}
        self.tickets = []
        self.cancels = 0

    def _slot(self, i):
        return self.slots.setdefault(i, dict(frame=None, mask=None, rec=np.zeros(1, _native.RECORD_DTYPE)[0], pix=None, cent=None))

    # -- bookkeeping / data movement
    def reserve(self, capacity):
        if capacity > self.capacity:
            self.capacity = capacity
            self.slots = {}          # growing drops what is on the device, like the real context

    def sync(self):
        pass

    def set_search_cus(self, n):
        pass

    def warm(self, sws=None, band=None, annotate=0):
        pass

    def overlay_configure(self, Minv):       # (the tracker configures the presentation stage at construction; nothing here draws)
        pass

    def overlay_set_font(self, atlas, advance, first_char=32):
        pass

    def urgent(self):
        import contextlib
        return contextlib.nullcontext(self)

    def close(self):
        pass

    def upload_frame_rows(self, frames, first=0, enqueue=False):
        f = np.asarray(frames).reshape(-1, self.img_h, self.img_w, 3)
        for k in range(f.shape[0]):
            assert first + k < self.capacity
            self._slot(first + k)["frame"] = f[k]
    upload_frames = upload_frame_rows

    def upload_frame_rows_async(self, frames, first=0):
        self.upload_frame_rows(frames, first)
        return frames

    def upload_frame_rest(self, frames, first=0, rows=None):
        return frames

    def download_masks(self, n, first=0):
        return np.stack([self._slot(first + k)["mask"] for k in range(n)], 0)

    def download_records(self, n, first=0):
        out = np.zeros(n, _native.RECORD_DTYPE)
        for k in range(n):
            out[k] = self._slot(first + k)["rec"]
        return out

    def download_record(self, slot):
        r = self.download_records(1, first=slot)[0]
        return (np.array(r["left_coeffs"], np.float64), np.array(r["right_coeffs"], np.float64), bool(r["detected"]),
                int(r["fit_flags"]))

    def download_pixels(self, slot, side):
        p = self._slot(slot)["pix"]
        return (p[0], p[1]) if side == 0 else (p[2], p[3])

    def download_centroids(self, slot, side):
        return list(self._slot(slot)["cent"][side])

    # -- compute
    def mask_run(self, n, fp=None, first=0, reuse_front=False):
        fp = fp or _native.filter_params()
        key_fp = tuple(getattr(fp, f[0]) for f in fp._fields_)
        for k in range(n):
            s = self._slot(first + k)
            key = (hashlib.sha1(s["frame"].tobytes()).hexdigest(), key_fp)
            if key not in FakeContext.mask_cache:
                FakeContext.mask_cache[key] = O.mask_from_frame(self.oc, s["frame"], _ofp(fp))
            s["mask"] = FakeContext.mask_cache[key]

    def _store(self, slot, o, mode, cent=None):
        s = self._slot(slot)
        rec = np.zeros(1, _native.RECORD_DTYPE)[0]
        rec["frame"] = s["rec"]["frame"]
        rec["mode"] = mode
        rec["detected"] = 1 if o["detected"] else 0
        if o["detected"]:
            rec["n_left"], rec["n_right"] = len(o["left_y"]), len(o["right_y"])
            flags = (1 if len(np.unique(o["left_y"])) < 3 else 0) | (2 if len(np.unique(o["right_y"])) < 3 else 0)
            rec["fit_flags"] = flags
            if not flags & 1:
                rec["left_coeffs"] = O.polyfit2(o["left_y"], o["left_x"])
            if not flags & 2:
                rec["right_coeffs"] = O.polyfit2(o["right_y"], o["right_x"])
            s["pix"] = (o["left_y"], o["left_x"], o["right_y"], o["right_x"])
            if cent is not None:
                s["cent"] = cent
        s["rec"] = rec
        return rec

    def sws_fit_run(self, n, sp=None, first=0):
        sp = sp or _native.search_params()
        for k in range(n):
            o = O.sliding_window_search(self._slot(first + k)["mask"], _osp(sp))
            self._store(first + k, o, 0, (o["left_centroids"], o["right_centroids"]))

    def band_fit_run(self, n, prev_coeffs, sp=None, first=0):
        sp = sp or _native.search_params()
        prev = np.asarray(prev_coeffs, np.float64).reshape(n, 6)
        for k in range(n):
            self._store(first + k, O.band_search(self._slot(first + k)["mask"], prev[k, :3], prev[k, 3:], _osp(sp)), 1)

    def band_fit_chain_run(self, n, seed_coeffs=None, sp=None, first=0):
        sp = sp or _native.search_params()
        if 2 * sp.bandwidth + 2 > 64:
            raise _native.NativeError("chained band search needs a band of at most 64 columns")
        if FakeContext.calls is not None:
            FakeContext.calls.append((first, n, seed_coeffs is not None))
        if seed_coeffs is None:
            if first < 1:
                raise ValueError("a chain without seed coefficients continues from the record of slot first - 1")
            r = self._slot(first - 1)["rec"]
            carry = np.concatenate([r["left_coeffs"], r["right_coeffs"]]) if (r["detected"] and not r["fit_flags"]) else None
            lo = first - 1
        else:
            carry, lo = np.asarray(seed_coeffs, np.float64).reshape(6), first
        for k in range(n):
            if carry is None:        # the walk has stopped: not searched
                s = self._slot(first + k)
                rec = np.zeros(1, _native.RECORD_DTYPE)[0]
                rec["frame"], rec["mode"] = s["rec"]["frame"], 255
                s["rec"] = rec
                continue
            rec = self._store(first + k, O.band_search(self._slot(first + k)["mask"], carry[:3], carry[3:], _osp(sp)), 1)
            carry = np.concatenate([rec["left_coeffs"], rec["right_coeffs"]]) if (rec["detected"] and not rec["fit_flags"]) else None
        self.tickets.append((lo, first + n - lo, first))

    def band_fit_chain_collect(self, n, first=0):
        hit = [i for i, (lo, cnt, own) in enumerate(self.tickets) if own <= first and first + n <= lo + cnt]
        if not hit:              # (a chain that searched the first slot itself goes before one that only holds it as its seed)
            hit = [i for i, (lo, cnt, own) in enumerate(self.tickets) if lo <= first and first + n <= lo + cnt]
        if not hit:
            raise _native.NativeError("no chained search covers slots [%d, %d)" % (first, first + n))
        del self.tickets[:hit[-1] + 1]
        return self.download_records(n, first)

    def band_fit_chain_cancel(self):
        self.cancels += 1
