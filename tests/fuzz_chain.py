#!/usr/bin/env python3
"""Randomised differential run of the chained stream pipeline (LaneTracker.process_batch, lt_band_fit_chain_run) against
frame-by-frame process() on the GPU box:   python tests/fuzz_chain.py [iterations] [seed] [1080] [long]

Each iteration builds a stream from a pool of frames -- drifting lanes, lanes that jump sideways (valid masks, but the
band search of the next frame finds nothing or a fit check_validity rejects), noise, flat grey, black -- with random
failure bursts, draws random tracker parameters (n_average, n_reset, n_fail, bandwidth, n_tries, partial, window sizes,
chain chunk / depth), random window splits and annotation on / off, and requires the tracker state, the lane-pixel lists
and (when annotated) the returned frames to equal those of process() after every window.  Prints the first mismatch and
exits 1."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lane_tracker_amd import calib, synth  # noqa: E402
from lane_tracker_amd.lane_tracker import LaneTracker  # noqa: E402


def state(lt):
    b = lambda a: None if a is None else np.asarray(a).tobytes()
    return dict(detected=lt.detected_pixels, valid=lt.valid_lane_lines, last_detection=lt.last_detection, success=lt.success,
                counter=lt.counter, left_avg=b(lt.left_avg_coeffs), right_avg=b(lt.right_avg_coeffs), last_left=b(lt.last_left_coeffs),
                last_right=b(lt.last_right_coeffs), hist=[b(c) for c in lt.left_fit_coeffs] + [b(c) for c in lt.right_fit_coeffs],
                radii=list(lt.average_curve_radii), radius=lt.average_curve_radius, lr=(lt.left_curve_radius, lt.right_curve_radius),
                ecc=lt.eccentricity, pts=(b(lt.left_avg_y), b(lt.left_avg_x), b(lt.right_avg_y), b(lt.right_avg_x)),
                pix=(b(lt.left_y), b(lt.left_x), b(lt.right_y), b(lt.right_x)), cent=(lt.left_window_centroids, lt.right_window_centroids))


def _until_overflow(pairs, rest):
    """Iterate `pairs`; upstream's get_curve_radius raises OverflowError (`int(inf)`, lane_tracker.py:540-545) for a lane whose
    quadratic coefficient is exactly zero, and so does this implementation.  When the batch side raises it, the frame-by-frame
    tracker must raise it too on the frames it has not seen yet: yields (None, 0) then, (None, 1) if it does not."""
    it = iter(pairs)
    while True:
        try:
            item = next(it)
        except StopIteration:
            return
        except OverflowError:
            seq, frames, kw = rest()
            try:
                for f in frames:
                    seq.process(f, **kw)
            except OverflowError:
                yield None, 0
                return
            print("ONLY THE BATCH SIDE RAISED OverflowError")
            yield None, 1
            return
        yield item


def main(iters=20, seed=1, cal=None, verbose=True, long_runs=False):
    """long_runs: streams of up to 220 frames, outages of up to 45 frames (the speculative groups of `_fail_group` reach 32),
    windows of up to 130 frames (runs of valid frames long enough for `_record_successes`), annotation half of the time."""
    rng = np.random.default_rng(seed)
    cal = cal or calib.reference_calibration()
    h, w = cal["img_size"][1], cal["img_size"][0]
    # pools: two drifting streams (the second one offset sideways: a jump between them breaks the band search)
    a = synth.stream_lanes(24, seed=101, cal=cal)
    bpool = synth.stream_lanes(24, seed=202, cal=cal)
    noise = [synth.frame_uniform(900 + i, img_size=(w, h)) for i in range(3)]
    grey, black = np.full((h, w, 3), 128, np.uint8), np.zeros((h, w, 3), np.uint8)
    bad = 0
    for it in range(iters):
        n = int(rng.integers(60, 220)) if long_runs else int(rng.integers(8, 90))
        frames, src, pos = [], a, int(rng.integers(0, 24))
        burst = 0
        for i in range(n):
            if burst > 0:
                burst -= 1
                frames.append([grey, black, noise[i % 3]][int(rng.integers(0, 3))])
                continue
            r = rng.random()
            if r < 0.06:
                burst = int(rng.integers(0, 46 if long_runs else 9))   # outage: more bad frames (beyond n_reset: back to sliding windows)
                frames.append(black)
                continue
            if r < 0.12:
                src = bpool if src is a else a               # the lane jumps
            pos = (pos + 1) % 24
            frames.append(src[pos] if (pos // 24) % 2 == 0 else src[23 - pos])
        frames = np.stack(frames, 0)
        ctor = dict(n_fail=int(rng.integers(1, 10)), n_reset=int(rng.integers(0, 6)), n_average=int(rng.integers(1, 5)),
                    print_frame_count=bool(rng.random() < 0.3))
        kw = dict(bandwidth=int(rng.choice([25, 31, 12, 30])), n_tries=int(rng.choice([2, 2, 1, -1])), partial=float(rng.choice([1.0, 1.0, 0.5])),
                  window_height=int(rng.choice([40, 40, 118])), no_success_limit=int(rng.choice([8, 3, 50])))
        annotate = bool(rng.random() < (0.5 if long_runs else 0.3))
        seq, bat = LaneTracker(**cal, **ctor), LaneTracker(**cal, **ctor)
        bat.chain_chunk, bat.chain_depth = (int(rng.choice([2, 8, 16, 32, 64, 0])) or None), int(rng.choice([1, 2, 3]))
        try:
            wins, lo = [], 0
            while lo < n:
                wlen = int(rng.integers(1, 130 if long_runs else 40))
                wins.append(frames[lo:lo + wlen])
                lo += wlen
            stream = bool(rng.random() < 0.5)              # process_stream (windows overlap on the device) or one process_batch per window
            results = bat.process_stream(wins, annotate=annotate, **kw) if stream else (bat.process_batch(w, annotate=annotate, **kw) for w in wins)
            lo = 0
            for win, outs in _until_overflow(zip(wins, results), lambda: (seq, frames[lo:], kw)):
                if win is None:          # both trackers raised upstream's OverflowError (int(inf): a lane fitted exactly straight)
                    bad += outs
                    break
                wlen = len(win)
                outs_seq = [seq.process(f, **kw) for f in win]
                s1, s2 = state(seq), state(bat)
                if s1 != s2:
                    keys = [k for k in s1 if s1[k] != s2[k]]
                    print("STATE MISMATCH it", it, "seed", seed, "window", lo, wlen, "keys", keys, ctor, kw, bat.chain_chunk, bat.chain_depth, "stream" if stream else "batch")
                    bad += 1
                    break
                if annotate and any(not np.array_equal(g, s) for g, s in zip(outs, outs_seq)):
                    print("FRAME MISMATCH it", it, "seed", seed, "window", lo, wlen, ctor, kw, "stream" if stream else "batch")
                    bad += 1
                    break
                lo += wlen
            if stream:
                results.close()
            if verbose:
                print("it %d: %d frames, %d/%d valid, chunk %s depth %d%s%s" % (it, n, bat.success, bat.counter, bat.chain_chunk, bat.chain_depth,
                                                                               ", annotated" if annotate else "", ", stream" if stream else ""))
        finally:
            seq.close()
            bat.close()
        if bad:
            break
    return bad


if __name__ == "__main__":
    it = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    sd = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    big = "1080" in sys.argv[3:]
    n_bad = main(it, sd, cal=calib.scaled_calibration(1.5) if big else None, long_runs="long" in sys.argv[3:])
    print("chain fuzz: %d iterations, %d mismatches" % (it, n_bad))
    sys.exit(1 if n_bad else 0)
