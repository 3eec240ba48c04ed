"""The stream pipeline of the tracker (reference lane_tracker.py:851-872, 1064-1128, 1178-1199):
windows of consecutive frames of ONE video through the device with the searches chained there (`lt_band_fit_chain_run`), the
host replaying check_validity / the histories exactly as `process()` does; outages in speculative groups; annotated frames of a
window as strips.  `StreamPipeline` is the part of `LaneTracker` (lane_tracker.py) that does this: a mix-in, it has no state of
its own and is never instantiated alone."""
import os

import numpy as np

from . import _native
from . import overlay as _overlay


class StreamPipeline:
    """`process_batch`, `process_stream`, `warm` and what they are made of; `LaneTracker` supplies the per-frame state machine
    (`_step`, `_record_success`, `_record_failure`, ...) and the device context (`self._ctx`)."""

    def _commit_valid_run(self, LF, RF, g, skip, annotate, partial, deferred, commit):
        """Frames skip .. g-1 of a run of valid first tries (raw fits LF, RF; `commit(j)` = what `_step` does for frame j on
        success).  With annotation every frame leaves a picture: the first n_average - 1 frames (their averages reach back before
        the run) and the last one (it leaves the state) go the ordinary way, the ones between all at once (`_record_successes`)
        -- except around a frame whose radius needs the scalar route's care (a near-straight lane, `_delicate_radii`): that
        frame and the n_average - 1 behind it (their averaging windows contain it) go the ordinary way, the stretches between
        such frames all at once."""
        j, k_avg = skip, int(self.n_average)
        if annotate and k_avg >= 1 and g >= 2 * k_avg + 4:
            for j in range(k_avg - 1):
                commit(j)
            plain = ~self._delicate_radii(LF[:g], RF[:g])
            clean = plain.copy()                       # clean[f]: no delicate frame in f's averaging window [f - k + 1, f]
            for t in range(1, k_avg):
                clean[t:] &= plain[:-t]
            j = k_avg - 1
            while j < g - 1:
                if clean[j]:
                    e = j
                    while e < g - 1 and clean[e]:
                        e += 1
                    if e - j >= 4 and self._record_successes(LF[:g], RF[:g], j, e, partial, deferred):
                        j = e
                        continue
                commit(j)
                j += 1
        for j in range(j, g):
            commit(j)

    def _delicate_radii(self, LF, RF):
        """Per frame: does `get_curve_radius` need the scalar route for this pair of raw fits -- a radius that is not finite,
        huge, or within 2e-8 (relative) of an integer, where only the exact refit on the lane pixels decides what `int()`
        gives upstream?  (Every radius above 2.5e7 m qualifies: near-straight lanes.)"""
        LF, RF = np.asarray(LF, np.float64).reshape(-1, 3), np.asarray(RF, np.float64).reshape(-1, 3)
        y_eval = self.warped_size[1]
        bad = np.zeros(len(LF), bool)
        for Cf in (LF, RF):
            a_m, b_m = Cf[:, 0] * self.mpph / (self.mppv ** 2), Cf[:, 1] * self.mpph / self.mppv
            with np.errstate(all="ignore"):
                v = ((1 + (2 * a_m * y_eval * self.mppv + b_m) ** 2) ** 1.5) / np.absolute(2 * a_m)
                bad |= ~np.isfinite(v) | (v >= 2.0 ** 50) | (np.abs(v - np.rint(v)) <= 2e-8 * np.maximum(1.0, np.abs(v)))
        return bad

    def _record_successes(self, LF, RF, lo, hi, partial, deferred):
        """`_record_success` + the deferred picture for the frames lo .. hi-1 of a run of valid first tries (raw fits LF, RF,
        frames 0 .. lo-1 of the run already recorded; lo >= n_average - 1, so every average stays inside the run), all at
        once: the same IEEE operations element-wise (averages: the sequential sum np.average forms, then the division;
        radii: `get_curve_radius`; eccentricity) and `lt_poly_points` for `get_poly_points`.  Leaves the histories as they
        are before frame `hi`.  Returns False -- nothing touched -- whenever a frame needs the scalar route's care: a radius
        within 2e-8 of an integer (the exact refit decides those), not finite or huge, no positive radius in an averaging
        window, or a parabola without a point inside the image."""
        k = int(self.n_average)
        LF, RF = np.asarray(LF, np.float64), np.asarray(RF, np.float64)
        y_eval = self.warped_size[1]

        def radii(Cf):                   # get_curve_radius for every frame of the run (their own raw fits)
            a_m, b_m = Cf[:, 0] * self.mpph / (self.mppv ** 2), Cf[:, 1] * self.mpph / self.mppv
            with np.errstate(all="ignore"):
                return ((1 + (2 * a_m * y_eval * self.mppv + b_m) ** 2) ** 1.5) / np.absolute(2 * a_m)
        vl, vr = radii(LF[:hi]), radii(RF[:hi])
        w0 = max(0, lo - k + 1)              # the frames whose radii enter an averaging window of lo .. hi-1
        if self._delicate_radii(LF[w0:hi], RF[w0:hi]).any():
            return False
        with np.errstate(all="ignore"):      # (frames in front of w0 may be delicate: their entries of r are never read)
            r = np.trunc(0.5 * (np.trunc(vl) + np.trunc(vr)))
        r = np.where(np.isfinite(r) & (np.abs(r) < 2.0 ** 62), r, 0).astype(np.int64)                          # per frame, :545
        m = hi - lo
        idx = np.arange(lo, hi)
        total, count = np.zeros(m), np.zeros(m, np.int64)
        suml, sumr = LF[idx - k + 1].copy(), RF[idx - k + 1].copy()
        for t in range(k):               # window entry t of every frame: frame j - k + 1 + t
            w = r[idx - k + 1 + t]
            total += np.where(w > 0, w, 0)
            count += w > 0
            if t:
                suml += LF[idx - k + 1 + t]
                sumr += RF[idx - k + 1 + t]
        if not np.all(count > 0):
            return False
        avg_radius = np.trunc(total / count).astype(np.int64)
        avg = np.concatenate([suml / k, sumr / k], axis=1)
        ploty, ploty2 = self._plot_rows(partial)
        mid = int(self.warped_size[0] / 2)
        polys = None
        if self.stream_lane_on_device and len(ploty):
            # The pictures are drawn from the averaged coefficients by the device (lt_overlay_run_strip_coeffs forms get_poly_points
            # and the polygons there); what the host needs of the plot points is the LAST kept point of each curve, for the
            # eccentricity (:551-559) -- the bottom plot row's, when it lies inside the image, which is the rule: the same two
            # products and two sums, element-wise.  A frame whose bottom point is outside takes the points' way below.
            xmax = self.warped_size[0] - 1
            xl = (avg[:, 0] * ploty2[-1] + avg[:, 1] * ploty[-1]) + avg[:, 2]
            xr = (avg[:, 3] * ploty2[-1] + avg[:, 4] * ploty[-1]) + avg[:, 5]
            if np.all((xl <= xmax) & (xl >= 0) & (xr <= xmax) & (xr >= 0)):
                ecc = (((mid - xl.astype(np.int64)) - (xr.astype(np.int64) - mid)) / 2) * self.mpph
                plot = (ploty, ploty2)
                polys = [_CoeffPoly(avg[q], self.warped_size, plot) for q in range(m)]
        if polys is None:
            ln, rn, lyx, ryx = _native.poly_points(self.warped_size, avg, ploty, ploty2)
            if not (np.all(ln > 0) and np.all(rn > 0)):
                return False
            le, re = np.cumsum(ln), np.cumsum(rn)
            ecc = (((mid - lyx[le - 1, 1].astype(np.int64)) - (ryx[re - 1, 1].astype(np.int64) - mid)) / 2) * self.mpph
            polys = [_PackedPoly(lyx[le[q] - ln[q]:le[q]], ryx[re[q] - rn[q]:re[q]]) for q in range(m)]
        for q in range(m):
            self.counter += 1
            lines = ["Curve Radius: {} m".format(int(avg_radius[q])), "Eccentricity: {:.2f} m".format(float(ecc[q]))]
            if self.print_frame_count:
                lines.append("Frame: {}".format(self.counter - 1))
            deferred.append(('lane', polys[q], lines))
        self.success += m
        self.left_fit_coeffs = [np.array(c) for c in LF[hi - k:hi]]
        self.right_fit_coeffs = [np.array(c) for c in RF[hi - k:hi]]
        self.average_curve_radii = [int(v) for v in r[hi - k:hi]]
        return True

    _coeff_strips = True         # lt_overlay_run_strip_coeffs exists for this context (cleared on its first refusal)
    stream_lane_on_device = True     # False: plot points and polygon intervals of a window's frames on the host (A/B, tests)

    # ---- the chained stream pipeline (reference :851-872, :1064-1128, :1178-1199) ------------
    search_cus = 1                   # CUs kept free of the mask chain for the chained search (lt_set_search_cus); 0: shared; >= 2: the
                                     # others belong to the kernel that copies annotated frames back when lt_set_download_method asks for it
    chain_searches = True            # False: process_batch searches frame by frame (one record round trip per frame)
    chain_chunk = None               # frames per upload + mask launch, and per chain, inside a window; None: by window size --
                                     # 32 for a stand-alone window (its head and tail count), half a window up to 128 in a
                                     # stream of windows (the walking threshold kernels take launches of >= 80 frames)
    chain_depth = 3                  # chains kept in flight behind the one the host is checking
    outage_groups = True             # False: a frame whose first try failed is handled alone (`_step`), not in speculative groups
    _outage_group = 4                # frames in the next such group: 4, doubling up to 32 while every frame of a group fails
    stream_lookahead = 2             # process_stream: windows fed (uploads + masks) ahead of the one being searched; with 1 the
                                     # uploads pause between windows (the next-but-one window would reuse the slots still searched)

    def _valid_many(self, LF, RF):
        """check_validity (:561-627) for m fits at once: the same f64 operations in the same order, element by element,
        so every entry equals what `check_validity` would have stored in valid_lane_lines."""
        lim = self.validity_limits
        ploty, ploty2 = self._plot_rows(1)
        W = self.warped_size[0]

        # plot points inside the image, per side (:565-569): lt_poly_points counts them (the same f64 operations in the same
        # order as fitx = a * ploty**2 + b * ploty + c; tests/test_host_geometry.py)
        ln, rn, _, _ = _native.poly_points(self.warped_size, np.concatenate([np.asarray(LF, np.float64).reshape(-1, 3),
                                                                             np.asarray(RF, np.float64).reshape(-1, 3)], axis=1), ploty, ploty2)
        n = np.minimum(ln, rn).astype(np.int64)
        y1 = np.full(len(LF), W - 1, np.int64)
        y2 = W - (n * 0.35).astype(np.int64)
        y3 = W - (n * 0.75).astype(np.int64)

        def at(Cf, y):
            return Cf[:, 0] * (y ** 2) + Cf[:, 1] * y + Cf[:, 2]

        def slope(Cf, y):
            return 2 * Cf[:, 0] * y + Cf[:, 1]
        x1, x2, x3 = (np.abs(at(LF, y) - at(RF, y)) for y in (y1, y2, y3))
        dist_bad = ((x1 < lim['min_dist_y1']) | (x1 > lim['max_dist_y1']) | (x2 < lim['min_dist_y2'])
                    | (x2 > lim['max_dist_y2']) | (x3 < lim['min_dist_y3']) | (x3 > lim['max_dist_y3']))
        norm1 = np.abs(slope(LF, y1) - slope(RF, y1))
        norm2 = np.abs(slope(LF, y3) - slope(RF, y3))
        return ~dist_bad & ~((norm1 >= lim['thresh']) | (norm2 >= lim['thresh']))

    _SECOND_TRY = (15, 5, 35, 5, 'neighborhood', False, 140, 65, 10, 30, 40, 20, 0.1, 50, 0.25, 360, 30, 30, 1.0)   # :1081-1099

    def _fail_group(self, frames, base, i, k, first_try, fp, n_tries, annotate, deferred, speculate=None):
        """Frames i .. i+k-1 of a window (slots base+i ..; first-try masks computed), the first of which is known or
        expected to fail its first try: all of them at once, speculating that every one fails both tries.  While frames
        fail, nothing a frame needs depends on the frame before it except the count of misses: its search mode (sliding
        windows once `last_detection > n_reset`, :851) and the band seed (the last valid fits, unchanged) are known in
        advance.  So: the first-try searches of the whole group in one launch, one record download; the second-try masks
        and searches of the frames in front of the first first-try success likewise; then the frames are committed in order up
        to and including the first success (whatever was computed behind it under the wrong hypothesis is dropped, and the
        first-try masks the second try overwrote are computed again).  When a first try succeeds behind failing frames, the
        frames behind it are chained from its record at once (`speculate(position)`, the caller's launcher), beside the second
        tries still to run in front of it: those usually fail too, and the chain has then done its work under them.
        Returns (frames committed >= 1, ended with a success, that chain or None).
        State after every frame = `_step` frame by frame (tests/test_stream_driver_cpu.py, tests/fuzz_chain.py)."""
        ctx = self._ctx
        tries = [first_try] + ([self._SECOND_TRY] if (n_tries >= 2 or n_tries == -1) else [])
        d0 = int(self.last_detection)
        n_bs = max(0, min(k, int(self.n_reset) - d0 + 1))          # frames j with d0 + j <= n_reset search a band, the rest windows
        if n_bs and (self.last_left_coeffs is None or self.last_right_coeffs is None or np.size(self.last_left_coeffs) != 3):
            k = 0
        seed = None if not n_bs or not k else np.concatenate([np.asarray(self.last_left_coeffs, np.float64).reshape(3),
                                                              np.asarray(self.last_right_coeffs, np.float64).reshape(3)])
        for handle, fetch in ((self._pending, self._materialise_pixels), (self._pending_cent, self._materialise_centroids)):
            if handle is not None and handle[0] is ctx and base + i <= handle[1] < base + i + k:
                fetch()                  # (lists not fetched yet that these searches would overwrite)

        def search(t, lo, hi):           # try t of frames lo .. hi-1 -> their records
            q = tries[t]
            try:
                if lo < min(n_bs, hi):
                    m = min(n_bs, hi) - lo
                    ctx.band_fit_run(m, np.tile(seed, (m, 1)), _native.search_params(bandwidth=q[17], ignore_bottom=q[16], partial=q[18]),
                                     first=base + i + lo)
                if max(n_bs, lo) < hi:
                    a = max(n_bs, lo)
                    ctx.sws_fit_run(hi - a, _native.search_params(window_width=q[9], window_height=q[10], search_range=q[11], mu=q[12],
                                                                  no_success_limit=q[13], start_slice=q[14], ignore_sides=q[15],
                                                                  ignore_bottom=q[16], partial=q[18]), first=base + i + a)
            except _native.NativeError:  # geometry outside the kernels' limits
                return None
            return ctx.download_records(hi - lo, first=base + i + lo)

        def verdicts(rec):               # per record: 1 valid, 0 failed, -1 needs the frame-by-frame route (rank-deficient fit)
            det = rec["detected"] != 0
            v = np.zeros(len(rec), np.int64)
            v[det & (rec["fit_flags"] != 0)] = -1
            idx = np.flatnonzero(det & (rec["fit_flags"] == 0))
            if len(idx):
                v[idx] = self._valid_many(rec["left_coeffs"][idx], rec["right_coeffs"][idx])
            return v

        def first_where(cond, default):
            hits = np.flatnonzero(cond)
            return int(hits[0]) if len(hits) else default

        rec1 = search(0, 0, k) if k else None
        if rec1 is None:
            self._step(frames[i], first_try, n_tries, False, slot=base + i, have_mask=True, lazy=True, annotate=annotate, defer=deferred)
            return 1, bool(self.valid_lane_lines), None
        v1 = verdicts(rec1)
        end = first_where(v1 != 0, k)                              # frames [0, end) failed their first try for certain
        rec2, v2, e2 = None, None, 0
        spec = None
        if len(tries) == 2 and end:
            e2 = end
            if speculate is not None and end < k and v1[end] == 1:
                spec = speculate(i + end + 1)                      # seeded on the device by the record the first try of frame `end` left
            ctx.mask_run(e2, _native.filter_params(*[self._SECOND_TRY[x] for x in (4, 0, 1, 2, 3, 5, 6, 7, 8)]), first=base + i, reuse_front=True)
            rec2 = search(1, 0, e2)
            if rec2 is None:
                if spec is not None:
                    ctx.band_fit_chain_cancel()
                ctx.mask_run(e2, fp, first=base + i, reuse_front=True)
                self._step(frames[i], first_try, n_tries, False, slot=base + i, have_mask=True, lazy=True, annotate=annotate, defer=deferred)
                return 1, bool(self.valid_lane_lines), None
            v2 = verdicts(rec2)
            end = min(end, first_where(v2 != 0, e2))
        # frames [0, end) failed every try; frame `end` (if inside the group) is a success or needs care
        win = None                       # (try, record) of the success that ends the group
        if end < k:
            if rec2 is not None and end < e2 and v2[end] == 1:
                win = (1, rec2[end])
            elif (rec2 is None or end >= e2) and v1[end] == 1:
                win = (0, rec1[end])
        committed = end + (1 if win else 0)
        if spec is not None and not (win is not None and win[0] == 0 and i + committed == spec[0]):
            ctx.band_fit_chain_cancel()  # a second try succeeded in front of it (or a frame needs care): its seed is not the stream's state
            spec = None
        last = None                      # the most recent search of the committed frames that found pixels: (frame, try, record)
        for j in range(end):
            final = rec2[j] if rec2 is not None else rec1[j]
            self.counter += 1
            self.detected_pixels = bool(final["detected"])
            self.valid_lane_lines = False
            self._record_failure()
            if annotate:
                redraw = (self.left_avg_y.size != 0) and (self.last_detection <= self.n_fail)
                deferred.append(('lane', (self.left_avg_y, self.left_avg_x, self.right_avg_y, self.right_avg_x), self._lane_text())
                                if redraw else ('fail', None, self._failure_text()))
            if rec1[j]["detected"]:
                last = (j, 0, rec1[j])
            if rec2 is not None and rec2[j]["detected"]:
                last = (j, 1, rec2[j])
        if win:
            t, r = win
            if t == 1 and rec1[end]["detected"]:
                last = (end, 0, rec1[end])
            last = (end, t, r)
        if committed:
            jl = committed - 1
            self._resident, self._resident_partial = (frames[i + jl], base + i + jl), self._window_rows is not None and annotate
        if last is not None:
            j, t, r = last
            if t == 0 and rec2 is not None and j < e2:
                # the second try of that frame ran on its slot afterwards and found nothing: the lists this search left are
                # the tracker's, so it is run again (same mask, same search: same lists)
                ctx.mask_run(1, fp, first=base + i + j, reuse_front=True)
                search(0, j, j + 1)
            self._pending = (ctx, base + i + j)
            if j >= n_bs:
                self._pending_cent = (ctx, base + i + j)
            final_search_found = (j == committed - 1) and (win is not None or t == len(tries) - 1 or rec2 is None)
            self._fit = ("pending", None, np.array(r["left_coeffs"], np.float64), np.array(r["right_coeffs"], np.float64)) \
                if final_search_found else None
        elif committed:
            self._fit = None
        if win:
            t, r = win
            self.counter += 1
            self.detected_pixels = True
            self.valid_lane_lines = True
            self._record_success(np.array(r["left_coeffs"], np.float64), np.array(r["right_coeffs"], np.float64), tries[t][18])
            if annotate:
                deferred.append(('lane', (self.left_avg_y, self.left_avg_x, self.right_avg_y, self.right_avg_x), self._lane_text()))
        if e2 > committed:               # frames behind the last committed one still carry second-try masks
            ctx.mask_run(e2 - committed, fp, first=base + i + committed, reuse_front=True)
        if not committed:                # frame i itself needs the frame-by-frame route (a rank-deficient fit)
            self._step(frames[i], first_try, n_tries, False, slot=base + i, have_mask=True, lazy=True, annotate=annotate, defer=deferred)
            return 1, bool(self.valid_lane_lines), None
        return committed, win is not None, spec

    def _run_window_chained(self, frames, first_try, fp, n_tries, annotate, deferred, base=0, prefed=0, ahead=None, flush=None):
        """The frame loop of a window with the searches chained on the device.  State after every frame, and every
        attribute at the end, equal those of `_step` frame by frame (tests/test_gpu_chain.py, tests/fuzz_chain.py).
        The window's frames live in slots base .. base+n-1; the first `prefed` of them already have their upload and
        first-try mask enqueued (by an earlier window); `ahead` = the windows that follow, in order, as mutable lists
        [frames, first slot, frames fed so far]: they are fed, in order, as this window drains (the third entry is updated).
        A generator: it yields exactly once, when the window's first searches are in flight and the first of them is checked, but
        before anything of the window is committed to the tracker's state (`process_stream` uses that moment to wait for the
        previous window's annotated frames); run it to exhaustion."""
        ctx, n = self._ctx, frames.shape[0]
        ahead = ahead or []
        total = n + sum(len(a[0]) for a in ahead)
        partial = first_try[-1]
        sws_kw = dict(window_width=first_try[9], window_height=first_try[10], search_range=first_try[11], mu=first_try[12],
                      no_success_limit=first_try[13], start_slice=first_try[14], ignore_sides=first_try[15],
                      ignore_bottom=first_try[16], partial=partial)
        sp_sws = _native.search_params(**sws_kw)
        sp_band = _native.search_params(bandwidth=first_try[17], ignore_bottom=first_try[16], partial=partial)
        if self.chain_chunk is not None:
            chunk = max(2, int(self.chain_chunk)) & ~1
        elif self._in_stream:
            chunk = max(32, min(128, (n // 2) & ~1))
        else:
            chunk = 32 if n < 512 else 64
        masked = prefed + (sum(a[2] for a in ahead) if prefed >= n else 0)
        # stream positions [0, masked) have their upload + first-try mask enqueued; positions >= n are frames of the windows
        # ahead (fed strictly in order, so a later window has frames fed only if the ones before it are fed completely)
        head = not self._in_stream       # a stand-alone window: nothing is in flight when it starts

        def span(at):                    # frames per launch at position `at`: short at the head of a stand-alone window (the
            return min(chunk, max(16, at & ~1)) if head else chunk   # first records come back early), then `chunk`

        def chain_span(at):              # frames per chain: in an annotated stream a window starts with short chains (32, 32, 64,
            if annotate and self._in_stream and self.chain_chunk is None:    # ...), so that its first frames are on their way
                return min(chunk, max(32, at & ~1))    # back soon after the frames of the window before have landed (the host waits
            return span(at)              # for those before it commits anything of this window) and each piece's bookkeeping hides
                                         # under the copy of the piece before

        rest_rows = self._window_rows[1] if self._window_rows is not None else None   # annotated frames travel as row runs
        rest_needed = annotate
        if self._window_rows is not None and self._window_rows[4] is not None:         # ... as strips: the lane's run of rows alone,
            rest_rows, rest_needed = self._window_rows[4][1], annotate and self._window_rows[4][4]   # which the mask chain has uploaded

        def feed(upto):                  # keep the device supplied with masks ahead of the searches
            nonlocal masked
            while masked < min(total, upto):
                if masked < n:
                    m = min(span(masked), n - masked)
                    ctx.upload_frame_rows_async(frames[masked:masked + m], first=base + masked)
                    ctx.mask_run(m, fp, first=base + masked)
                    if rest_needed:      # the rest of these frames, for the overlay: behind their rows on the copy stream
                        ctx.upload_frame_rest(frames[masked:masked + m], first=base + masked, rows=rest_rows)
                else:
                    q = masked - n
                    for a in ahead:                       # the window position `masked` falls into
                        if q < len(a[0]):
                            break
                        q -= len(a[0])
                    m = min(chunk, len(a[0]) - q)
                    ctx.upload_frame_rows_async(a[0][q:q + m], first=a[1] + q)
                    ctx.mask_run(m, fp, first=a[1] + q)
                    if rest_needed:
                        ctx.upload_frame_rest(a[0][q:q + m], first=a[1] + q, rows=rest_rows)
                    a[2] = q + m
                masked += m
        feed(2 * chunk)
        depth = max(1, int(self.chain_depth))

        def launch(at):
            """Enqueue a chain at frame `at` from the tracker's state (host seed, or a sliding-window search of `at` and a
            chain behind it).  Returns (first, length, search mode of the first frame) or None (frame by frame)."""
            feed(at + (depth + 1) * chunk)
            L = min(chain_span(at), min(masked, n) - at)
            mode = 'sws' if self.last_detection > self.n_reset else 'bs'          # :851
            if self._pending is not None and self._pending[0] is ctx and base + at <= self._pending[1] < base + n:
                self._materialise_pixels()        # (cannot happen inside a window: committed frames lie before `at`)
            if mode == 'sws' and self._pending_cent is not None and self._pending_cent[0] is ctx and self._pending_cent[1] == base + at:
                self._materialise_centroids()
            try:
                if mode == 'sws':
                    if L < 2:
                        return None
                    ctx.sws_fit_run(1, sp_sws, first=base + at)
                    ctx.band_fit_chain_run(L - 1, None, sp_band, first=base + at + 1)
                else:
                    seed = np.concatenate([np.asarray(self.last_left_coeffs, np.float64).reshape(3),
                                           np.asarray(self.last_right_coeffs, np.float64).reshape(3)])
                    ctx.band_fit_chain_run(L, seed, sp_band, first=base + at)
            except _native.NativeError:  # geometry outside the chain kernel's limits
                return None
            return at, L, mode

        def launch_behind(prev):
            """Speculate further: the chain continues on the device from the last record of `prev` (not yet checked)."""
            at = prev[0] + prev[1]
            if at >= n:
                feed(at + (depth + 1) * chunk)   # nothing left to chain in this window: keep feeding the next one
                return None
            feed(at + (depth + 1) * chunk)
            L = min(chain_span(at), min(masked, n) - at)
            try:
                ctx.band_fit_chain_run(L, None, sp_band, first=base + at)
            except _native.NativeError:
                return None
            return at, L, 'bs'

        # chains in flight, oldest first: each but the first is seeded on the device by the last record of the one before it,
        # so each waits for the masks of its own frames only and the host checks one while the next ones run
        i, flight, started = 0, [], False
        while i < n:
            if not flight:
                first_chain = launch(i)
                if first_chain is None:
                    if not started:
                        started = True
                        yield
                    self._step(frames[i], first_try, n_tries, False, slot=base + i, have_mask=True, lazy=True, annotate=annotate,
                               defer=deferred)
                    i += 1
                    continue
                flight.append(first_chain)
            while len(flight) < depth + 1:
                more = launch_behind(flight[-1])
                if more is None:
                    break
                flight.append(more)
            first, L, mode = flight.pop(0)
            rec = ctx.band_fit_chain_collect(L, first=base + first)
            good = (rec["mode"] != 255) & (rec["detected"] != 0) & (rec["fit_flags"] == 0)
            LF, RF = rec["left_coeffs"], rec["right_coeffs"]
            g = L if good.all() else int(np.argmin(good))              # frames [0, g) were found, with regular fits
            if g:
                ok = self._valid_many(LF[:g], RF[:g])
                if not ok.all():
                    g = int(np.argmin(ok))
            if not started:              # nothing of this window has touched the tracker's state yet
                started = True
                yield
            # frames first .. first+g-1: first try valid.  Without annotation only the last n_average of them leave a
            # trace in the state (histories are that long; every other attribute is overwritten by each success).
            skip = 0 if annotate else max(0, g - max(int(self.n_average), 1))
            if g and mode == 'sws':
                self._pending_cent = (ctx, base + first)    # the sliding-window search of the chain's first frame found pixels (:439-440)
            if skip:
                self.counter += skip
                self.success += skip
            def commit(j):
                self.counter += 1
                self.detected_pixels = True
                self.valid_lane_lines = True
                lf, rf = np.array(LF[j], np.float64), np.array(RF[j], np.float64)
                self._pending = (ctx, base + first + j)
                self._fit = ("pending", None, lf, rf)
                self._resident, self._resident_partial = (frames[first + j], base + first + j), self._window_rows is not None and annotate
                self._record_success(lf, rf, partial)
                if annotate:
                    deferred.append(('lane', (self.left_avg_y, self.left_avg_x, self.right_avg_y, self.right_avg_x),
                                     self._lane_text()))
            self._commit_valid_run(LF, RF, g, skip, annotate, partial, deferred, commit)
            i = first + g
            if flush is not None:
                flush(False)             # render and download what has been committed so far, under the searches still running
            if g < L:
                # frame i: first try failed (or needs the host's exact fit): the ordinary route, second try included;
                # whatever was chained behind it is dropped (and told to stop)
                if flight:
                    ctx.band_fit_chain_cancel()
                flight = []
                while i < n:             # groups of frames, speculating that the outage lasts (`_fail_group`), until one succeeds
                    k = min(max(1, int(self._outage_group)), n - i) if self.outage_groups else 1
                    feed(i + k)
                    k = min(k, min(masked, n) - i)
                    def speculate(at):  # a chain behind a first try that succeeded inside the group, from the masks already there
                        L = min(chain_span(at), min(masked, n) - at) if at < n else 0
                        if L < 1:
                            return None
                        try:
                            ctx.band_fit_chain_run(L, None, sp_band, first=base + at)
                        except _native.NativeError:
                            return None
                        return at, L, 'bs'
                    with ctx.urgent():   # not behind the masks of later frames queued on the slots' streams
                        done, recovered, spec = self._fail_group(frames, base, i, k, first_try, fp, n_tries, annotate, deferred,
                                                                 speculate if self.outage_groups else None)
                    i += done
                    if flush is not None:
                        flush(False)
                    if recovered or not self.outage_groups:
                        self._outage_group = 4
                        if spec is not None:
                            flight = [spec]   # already running: the frames behind the recovered one
                        break
                    if done == k:        # every frame of the group failed: a longer group next
                        self._outage_group = min(32, 2 * max(1, int(self._outage_group)))
        if flush is not None:
            flush(True)
        if not started:                  # (an empty window)
            yield


    _search_cus_set = False

    def _reserve_search_cus(self):
        """CUs of their own for the stream pipeline's long-running kernels (lt_set_search_cus), at the first call that uses that
        pipeline -- not in the constructor: a tracker that only ever serves process() has no chained search to protect, runs its
        kernels on every CU, and creates no CU-masked streams (which, never destroyed, make a process die in the runtime's exit
        handlers under rocprofv3: NOTES_r06 E.7).  The call synchronises the (idle) context and replaces its streams."""
        if not self._search_cus_set:
            self._search_cus_set = True
            if self.search_cus:
                self._ctx.set_search_cus(self.search_cus)

    def _batch_arguments(self, kwargs):
        """process()'s keywords with its defaults -> (keyword dict, first-try parameter tuple, filter parameters).  (Every entry of
        the stream pipeline -- warm, process_batch, process_stream -- comes through here first.)"""
        self._reserve_search_cus()
        import inspect
        sig = inspect.signature(type(self).process)
        k = {name: v.default for name, v in sig.parameters.items() if name not in ("self", "img")}
        unknown = set(kwargs) - set(k)
        if unknown:
            raise TypeError("unexpected keyword(s): " + ", ".join(sorted(unknown)))
        k.update(kwargs)
        if k["visualize_search"] or k["split_view"]:
            raise NotImplementedError("search visualisation / split view are not available in the stream pipeline")
        first_try = (k["ksize_r"], k["C_r"], k["ksize_b"], k["C_b"], k["filter_type"], k["mask_noise"], k["noise_thresh"],
                     k["ksize_noise"], k["C_noise"], k["window_width"], k["window_height"], k["search_range"], k["mu"],
                     k["no_success_limit"], k["start_slice"], k["ignore_sides"], k["ignore_bottom"], k["bandwidth"],
                     k["partial"])
        fp = _native.filter_params(k["filter_type"], k["ksize_r"], k["C_r"], k["ksize_b"], k["C_b"],
                                   k["mask_noise"], k["noise_thresh"], k["ksize_noise"], k["C_noise"])
        return k, first_try, fp

    def warm(self, window=256, annotate=True, output_pool=True, **kwargs):
        """Set up, ahead of the first window, what `process_stream` / `process_batch` over windows of up to `window` frames would
        otherwise set up on the way (`kwargs`: process()'s keywords, as for those calls): the slot regions of a stream
        (`stream_lookahead` + 1 windows, one more with annotation) sized once, the search and chain buffers for both parameter
        sets, the presentation stage's buffers, the glyph atlas and -- `output_pool` -- the memory the annotated frames of the first
        windows will be returned in, every page touched (first touch of fresh memory runs at ~10 GB/s: 70 ms per window of 256
        1280x720 frames, which a stream that was not warmed pays inside its first three or four windows).  Optional -- a stream
        that was not warmed does the same work on the way -- and repeatable (a no-op the second time).  Returns the seconds it took."""
        import time
        t0 = time.perf_counter()
        k, first_try, fp = self._batch_arguments(kwargs)
        ctx = self._ctx
        if annotate:
            self._configure_overlay()
        look = max(1, int(self.stream_lookahead))
        regions = look + (2 if annotate else 1)
        size = (max(int(window), 1) + 1) & ~1
        if regions * size > ctx.capacity:
            self._materialise_pending()      # growing the context drops what is still on the device
            ctx.reserve(regions * size)
        mode = 0
        if annotate:
            rows = self._present_rows() if self.host_copies_rows else None
            mode = 2 if (rows is not None and rows[4] is not None) else 1
        if mode == 2 and output_pool and annotate != "inplace":   # the pool of output frames: a window being filled, one landing, one with the caller, one to spare
            _native.frames_prefault((int(window), ctx.img_h, ctx.img_w, 3), regions)
        for q in (first_try, self._SECOND_TRY):
            ctx.warm(_native.search_params(window_width=q[9], window_height=q[10], search_range=q[11], mu=q[12], no_success_limit=q[13],
                                           start_slice=q[14], ignore_sides=q[15], ignore_bottom=q[16], partial=q[18]),
                     _native.search_params(bandwidth=q[17], ignore_bottom=q[16], partial=q[18]), mode)
        return time.perf_counter() - t0

    @staticmethod
    def _as_window(frames):
        frames = np.ascontiguousarray(frames, np.uint8)
        if frames.ndim != 4:
            raise ValueError("expected frames of shape (n, H, W, 3)")
        return frames

    _window_rows = None         # _present_rows() while an annotated window / stream sends its frames back as row runs
    strip_piece = 32            # frames per overlay launch + strip download of a committed run
    _annotate_inplace = False   # annotate="inplace": annotated frames are the caller's own arrays, drawn over (strips only)

    def _window_renderer(self, deferred, base, n, piece=32, frames=None):
        """(flush, out) for a window of n frames in slots base..: `flush(force)` renders the frames committed to `deferred`
        since the last call -- overlay kernels, then the copy towards `out`, all only enqueued -- once at least `piece` of them
        have gathered (or `force`); `out` is complete after `_copies_done(flush.group)` (strips) / the next sync and
        `_copies_done()` (row runs, whole frames).
        Strips (`_window_rows[4]`, the default): `out` is ordinary memory.  Of every frame only the lane's run of rows is drawn on the
        device (lt_overlay_run_strip) and comes back, packed, through the library's page-locked staging blocks
        (lt_strip_download_async); the rows above and below it are copied from `frames`, the window as the caller handed it in,
        by the library's copy threads, which also draw the text lines (lt_host_text_async_group).  One completion group per window.
        Row runs (`_window_rows` without strips: `host_text = False`) and whole frames (`host_copies_rows = False`): round 4's ways, `out` page-locked.
        In place (`annotate="inplace"`, strips only): `out` IS `frames` -- the strips land in the caller's own window and the text is
        drawn over it; no row is copied on the host (0.9 instead of 2.8 MB per 1280x720 frame through the copy threads)."""
        self._configure_overlay()
        ctx = self._ctx
        empty = np.zeros(0, np.int64)
        done = [0]
        wr = self._window_rows if (frames is not None and n) else None
        H, rb, fb = ctx.img_h, ctx.img_w * 3, ctx.img_h * ctx.img_w * 3
        if wr is not None and wr[4] is not None:
            inplace = self._annotate_inplace and frames.flags.writeable
            out = frames if inplace else _native.frames_empty((n, ctx.img_h, ctx.img_w, 3))
            (l0, l1), (t0, t1) = wr[4][2], wr[4][3]
            font = _overlay.font_atlas() if self._have_font else None
            if font is None:
                t0 = t1 = 0
            group = _native.host_copy_group()
            self._window_groups = list(self._window_groups) + [(group, (out, frames))]
            # every row the device does not deliver: above and below the lane's run (in place: they are where they belong)
            host_rows = (0, 0, 0, 0) if inplace else (0, l0, l1, H)

            def flush(force):
                lo, hi = done[0], len(deferred)
                if hi <= lo or (hi - lo < piece and not force):
                    return
                # A committed run is up to 128 frames; one overlay launch and one download for all of them would let the first strip
                # leave the device only when the last frame is drawn, and the window would be handed out 2.4 ms (1280x720; 5.6 ms at
                # 1920x1080) after its last commit -- time the driving thread spends waiting (tools/annot_cprofile.py).  In pieces
                # of `strip_piece` frames the strips of a piece cross the bus while the next piece is drawn and the copy threads
                # place the piece before.
                for a in range(lo, hi, self.strip_piece):
                    b = min(a + self.strip_piece, hi)
                    part = deferred[a:b]
                    # Lanes that came as averaged coefficients (`_record_successes`: nearly all of a video) are drawn from them -- plot
                    # points and polygons on the device --, the others (the frames at the seams of a run, redrawn lanes, second tries)
                    # from their points: runs of one kind, each one overlay launch; frames without a lane go with either.
                    g0 = 0
                    while g0 < len(part):
                        kind, g1 = None, g0
                        while g1 < len(part):
                            d = part[g1]
                            kd = None if d[0] != 'lane' else ('c' if isinstance(d[1], _CoeffPoly) and self._coeff_strips else 'p')
                            if kd is not None and kind is not None and kd != kind:
                                break
                            kind = kind or kd
                            g1 += 1
                        run = part[g0:g1]
                        if kind == 'c':
                            first_poly = next(d[1] for d in run if d[0] == 'lane')
                            co = np.zeros((len(run), 6), np.float64)
                            dr = np.zeros(len(run), np.uint8)
                            for i, d in enumerate(run):
                                if d[0] == 'lane':
                                    co[i] = d[1].coeffs
                                    dr[i] = 1
                            if not ctx.overlay_run_strip_coeffs(co, dr, first_poly.plot[0], first_poly.plot[1], first=base + a + g0):
                                self._coeff_strips = False           # (this context has no such form: points from now on)
                                ctx.overlay_run_strip_packed(*_pack_deferred(run), first=base + a + g0)
                        else:
                            ctx.overlay_run_strip_packed(*_pack_deferred(run), first=base + a + g0)
                        g0 = g1
                    ctx.strip_download_async(out[a:b], base + a, group)
                    # the host's share of these frames, piece by piece as they are committed (all of a window's untouched rows at
                    # once, at its start, sat in the copy threads' queue in front of the last strips of the window before: 9 ms per
                    # window of 1920x1080 frames waiting for them), a frame at a time: its rows from the caller's window, then its text
                    text, nl = _native.text_bytes([d[2] for d in part]) if font is not None else (None, 0)
                    _native.host_text_async(group, out[a:b], frames[a:b], host_rows, font, text, nl, 40, self._TEXT_ORIGIN, self._TEXT_STEP)
                done[0] = hi
            flush.group = group
            return flush, out
        out = _native.pinned_empty((n, ctx.img_h, ctx.img_w, 3))
        rows = wr[1] if wr is not None else None
        group = None
        if rows is not None:
            # the rows no overlay can touch: from the caller's window into `out` on the library's copy threads, from now on.  In a
            # stream every window has a completion group of its own (the window before is handed out after waiting for ITS
            # copies only); a stand-alone window uses the tracker's.
            a0, a1, b0, b1 = wr[2]
            lib, dst, src = ctx.lib, out.ctypes.data, frames.ctypes.data
            if self._in_stream:
                group = _native.host_copy_group()
                self._window_groups = list(self._window_groups) + [(group, (out, frames))]
            else:
                self._copying = True
                self._copy_keepalive = (out, frames)
            for lo, hi in ((0, a0), (a1, b0), (b1, H)):
                if hi > lo and lib.lt_host_copy2d_async_group(group if group is not None else self._copy_group(), dst + lo * rb, fb,
                                                              src + lo * rb, fb, (hi - lo) * rb, n):
                    raise _native.NativeError("lt_host_copy2d_async_group failed")

        def flush(force):
            lo, hi = done[0], len(deferred)
            if hi <= lo or (hi - lo < piece and not force):
                return
            part = deferred[lo:hi]
            ctx.overlay_run_packed(*_pack_deferred(part), first=base + lo, rows=rows)
            if self._have_font:
                ctx.overlay_text([d[2] for d in part], first=base + lo)
            ctx.download_overlay_async(out[lo:hi], first=base + lo, rows=rows)
            done[0] = hi
        flush.group = group
        return flush, out

    def _render_window(self, deferred, base):
        """One overlay launch and one download for a whole window; a failed frame has no polygon (plain copy)."""
        self._configure_overlay()
        ctx = self._ctx
        ctx.overlay_run_packed(*_pack_deferred(deferred), first=base)
        if self._have_font:
            ctx.overlay_text([d[2] for d in deferred], first=base)
        return list(ctx.download_overlay(len(deferred), first=base))

    def process_batch(self, frames, annotate=True, **kwargs):
        """The same result as calling `process()` on each frame of `frames` in order (one stateful
        stream), arranged for throughput:

          * the frames are uploaded (only the camera rows the path reads) and their first-try masks (undistort + warp
            + filter) computed a few dozen at a time, ahead of the searches -- that stage is stateless;
          * the searches of consecutive frames are chained on the device (`lt_band_fit_chain_run`): frame
            k+1's band is drawn around frame k's fit without a host round trip, speculating that frame k
            will be found valid; the host collects the records of a whole run once, replays
            check_validity / the history exactly as `process()` does, and at the first frame that was
            not detected, not valid (or whose fit was rank deficient) drops the speculative tail, runs that
            frame the ordinary way (second try included) and starts the next chain behind it;
          * the second-try mask (different filter parameters) is computed lazily, only for a frame
            whose first try failed;
          * lane-pixel lists stay on the device unless somebody reads them.

        `kwargs` are `process()`'s keywords.  Returns the list of annotated frames, or None for every
        frame when `annotate=False` (state and attributes are updated identically).  `annotate="inplace"` (not in the
        reference: its draw_lane returns a new image) draws into `frames` itself where the frames travel as strips -- a
        C-contiguous, writeable uint8 window -- and returns its frames; otherwise it behaves like `annotate=True`.  For
        consecutive windows of one video prefer `process_stream`, which keeps the device busy across window boundaries."""
        if self._in_stream:
            raise RuntimeError("process_batch() inside an active process_stream() would overwrite its frames")
        k, first_try, fp = self._batch_arguments(kwargs)
        self._annotate_inplace = isinstance(annotate, str) and annotate == "inplace"   # (a window _as_window had to copy: into the copy)
        frames = self._as_window(frames)
        n = frames.shape[0]
        ctx = self._ctx
        self._materialise_pending()      # growing the context below drops what is still on the device
        ctx.reserve(max(n, 1))
        deferred = []
        if self.chain_searches and not k["diagnostics"]:
            self._window_rows = self._rows_for_window(frames) if annotate else None
            try:
                flush, out = self._window_renderer(deferred, 0, n, frames=frames) if (annotate and n) else (None, None)
                for _ in self._run_window_chained(frames, first_try, fp, k["n_tries"], annotate, deferred, flush=flush):
                    pass
                self._materialise_pending()  # the attributes describe the last frame, as after process() (also waits for `out`)
                if out is not None:
                    ctx.sync()
                    self._copies_done(flush.group)
                    return list(out)
                return [None] * n
            finally:
                self._all_copies_done()
                self._window_rows = None
        else:
            ctx.upload_frame_rows(frames)        # the camera rows the path reads; the rest only if frames are annotated
            ctx.mask_run(n, fp)
            if annotate:
                self._upload_keepalive = ctx.upload_frame_rest(frames)     # beside the mask chain, for the overlay
            for i in range(n):
                self._step(frames[i], first_try, k["n_tries"], k["diagnostics"], slot=i, have_mask=True, lazy=True,
                           annotate=annotate, defer=deferred)
        self._materialise_pending()      # the attributes describe the last frame, as after process()
        return self._render_window(deferred, 0) if annotate else [None] * n

    def process_stream(self, windows, annotate=True, **kwargs):
        """Generator over consecutive windows of ONE video: `windows` yields arrays (n, H, W, 3); for each, what
        `process_batch` would return is yielded, and the tracker's state after it is what `process()` frame by frame
        leaves.  The context holds `stream_lookahead + 1` windows side by side: while the searches of one window drain, the
        uploads and masks of the next ones are already running, so neither the bus nor the device idles at window boundaries
        (a window's head and tail cost about a quarter of a 256-frame `process_batch` call).  Do not call `process()` / `process_batch()` on this
        tracker until the generator is exhausted or closed.  `annotate="inplace"`: see `process_batch` (every window must be a
        C-contiguous, writeable uint8 array; a window that is not comes back as new frames)."""
        k, first_try, fp = self._batch_arguments(kwargs)
        self._annotate_inplace = isinstance(annotate, str) and annotate == "inplace"
        if not (self.chain_searches and not k["diagnostics"]):
            for w in windows:            # the frame-by-frame route has nothing to overlap
                yield self.process_batch(w, annotate=annotate, **kwargs)
            return
        it = iter(windows)
        cur = next(it, None)
        if cur is None:
            return
        if self._in_stream:
            raise RuntimeError("this tracker already runs a process_stream()")
        cur = self._as_window(cur)
        ctx = self._ctx
        look = max(1, int(self.stream_lookahead))
        # windows resident side by side: the one being searched and `look` being fed -- and, with annotation, the one before,
        # whose frames are still on their way back while the next one's first searches start
        regions = look + (2 if annotate else 1)
        size = 0                         # slots per region
        free = []                        # first slots of the regions nobody lives in
        queue = []                       # windows ahead of `cur`, in order: [frames, first slot, frames fed]; [.., None, 0]: not placed
        cur = [cur, None, 0]
        landing = None                   # (page-locked frames, first slot) of the window before `cur`, annotated frames being copied back

        def landed():
            nonlocal landing
            arrays, region, group = landing
            landing = None
            if not (self._window_rows is not None and self._window_rows[4] is not None):
                ctx.download_overlay_wait()  # these frames have landed; the uploads, masks and searches of the next windows run on
            self._copies_done(group)     # ... and so have the rows the host copies itself and -- strips -- the rows from the device (this window's group only)
            free.append(region)
            return list(arrays)
        self._in_stream = True
        self._window_rows = self._rows_for_window(cur[0]) if annotate else None
        try:
            while cur is not None:
                while len(queue) < look:             # know the next windows
                    w = next(it, None)
                    if w is None:
                        break
                    queue.append([self._as_window(w), None, 0])
                n = cur[0].shape[0]
                if cur[1] is None:                   # first window, or one that did not fit the regions: (re)size the context
                    if landing is not None:
                        yield landed()
                    if n > size:
                        self._materialise_pending()  # growing the context drops what is still on the device
                        size = (n + 1) & ~1
                        ctx.reserve(regions * size)
                        self.warm(n, annotate, output_pool=False, **kwargs)   # (a no-op when the caller has warmed the tracker for this window size)
                    free = [r * size for r in range(regions)]
                    for q in queue:                  # nothing can have been fed ahead of an unplaced window
                        q[1], q[2] = None, 0
                    cur[1] = free.pop(0)
                ahead = []
                for q in queue:                      # place the windows ahead while they fit and regions are free, in order
                    if q[1] is None:
                        if not (0 < q[0].shape[0] <= size) or not free:
                            break
                        q[1] = free.pop(0)
                    ahead.append(q)
                deferred = []
                flush, frames_out = self._window_renderer(deferred, cur[1], n, frames=cur[0]) if (annotate and n) else (None, None)
                if n:
                    for _ in self._run_window_chained(cur[0], first_try, fp, k["n_tries"], annotate, deferred, base=cur[1],
                                                      prefed=cur[2], ahead=ahead, flush=flush):
                        if landing is not None:      # this window's first searches are in flight: now wait for the frames of the one before
                            yield landed()
                if landing is not None:
                    yield landed()
                if frames_out is not None:
                    landing = (frames_out, cur[1], flush.group)   # handed out when the next window is under way (or the stream ends)
                    cur = queue.pop(0) if queue else None
                else:
                    free.append(cur[1])  # its frames, masks and records are not needed any more
                    cur = queue.pop(0) if queue else None
                    yield [None] * n
            if landing is not None:
                yield landed()
            self._materialise_pending()  # the attributes describe the last frame, as after process()
        finally:
            self._in_stream = False
            self._window_rows = None
            if annotate:                 # a generator closed early: no copy may still be writing into page-locked arrays
                try:                     # that go back to the pool with their last reference
                    ctx.band_fit_chain_cancel()
                    ctx.sync()
                    self._all_copies_done()
                except Exception:
                    pass



class _PackedPoly:
    """A lane polygon as lt_poly_points leaves it -- int32 (y, x) pairs of the left and of the right curve -- in the entries
    `_record_successes` defers for the overlay (the other entries carry upstream's four int64 arrays)."""
    __slots__ = ("lyx", "ryx")

    def __init__(self, lyx, ryx):
        self.lyx, self.ryx = lyx, ryx

    def as_tuple(self):
        a, b = self.lyx.astype(np.int64), self.ryx.astype(np.int64)
        return a[:, 0], a[:, 1], b[:, 0], b[:, 1]


class _CoeffPoly:
    """A lane polygon as its averaged coefficients (left a, b, c, right a, b, c): `_record_successes` defers these when the device
    forms plot points and polygon itself (lt_overlay_run_strip_coeffs); `points()` is the host's form, for the pieces that go the
    points' way (a piece that also holds redrawn or failed-try frames, row runs, whole frames)."""
    __slots__ = ("coeffs", "size", "plot")

    def __init__(self, coeffs, size, plot):
        self.coeffs, self.size, self.plot = coeffs, size, plot

    def points(self):
        _, _, lyx, ryx = _native.poly_points(self.size, self.coeffs[None], self.plot[0], self.plot[1])
        return lyx, ryx

    def as_tuple(self):
        lyx, ryx = self.points()
        a, b = lyx.astype(np.int64), ryx.astype(np.int64)
        return a[:, 0], a[:, 1], b[:, 0], b[:, 1]


_NO_POINTS = np.zeros((0, 2), np.int32)


def _pack_deferred(part):
    """Deferred pictures ('lane', polygon, text) / ('fail', None, text) -> (left counts, right counts, left (y, x) pairs,
    right (y, x) pairs) as Context.overlay_run_packed takes them; a failed frame has no polygon (plain copy)."""
    L, R = [], []
    for d in part:
        p = d[1] if d[0] == 'lane' else None
        if p is None:
            L.append(_NO_POINTS)
            R.append(_NO_POINTS)
        elif isinstance(p, _PackedPoly):
            L.append(p.lyx)
            R.append(p.ryx)
        elif isinstance(p, _CoeffPoly):
            lyx, ryx = p.points()
            L.append(lyx)
            R.append(ryx)
        else:
            L.append(np.stack([p[0], p[1]], 1).astype(np.int32))
            R.append(np.stack([p[2], p[3]], 1).astype(np.int32))
    ln, rn = np.array([len(a) for a in L], np.int32), np.array([len(a) for a in R], np.int32)
    return ln, rn, (np.concatenate(L) if L else _NO_POINTS), (np.concatenate(R) if R else _NO_POINTS)
