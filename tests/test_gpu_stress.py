"""Repetition tests: the same inputs many times in one process must give the same bytes every time.  A data race, a
missed wait state around hand-issued instructions or a hazard that depends on what ran before shows up as a run that
differs; a deterministic code-generation problem does not (tools/toolchain_cases.sh covers those)."""
import os

import numpy as np
import pytest

from helpers import golden_files, params_of, unpack_mask

pytestmark = pytest.mark.gpu


def test_sliding_window_golden_set_200_times_in_one_process():
    """Every sws_* fixture 200 times (alternating with the other fixtures, so that each run follows a different
    predecessor): records, centroid lists and lane-pixel lists must equal the first run's, which the fixture test has
    compared with the reference."""
    from lane_tracker_amd import _native
    fixtures = []
    for path in golden_files("sws_"):
        d = np.load(path)
        fixtures.append((os.path.basename(path), unpack_mask(d), params_of(d)))
    shapes = sorted({m.shape for _, m, _ in fixtures})
    ctxs = {s: _native.Context((2, 2), (s[1], s[0]), np.eye(3), np.zeros(5), np.eye(3), device=0, capacity=1) for s in shapes}
    try:
        first = {}
        for rep in range(200):
            order = list(range(len(fixtures)))
            if rep % 2:
                order.reverse()
            if rep % 3 == 2:
                order = order[1::2] + order[0::2]
            for i in order:
                name, mask, p = fixtures[i]
                c = ctxs[mask.shape]
                c.upload_masks(mask)
                c.sws_fit_run(1, _native.search_params(**p))
                rec = c.download_records(1)[0]
                key = [rec.tobytes(), tuple(c.download_centroids(0, 0)), tuple(c.download_centroids(0, 1))]
                if rep % 20 == 0:            # the pixel lists cost two more downloads: every twentieth repetition
                    key += [a.tobytes() for a in c.download_pixels(0, 0) + c.download_pixels(0, 1)]
                else:
                    key += first[name][3:] if name in first else []
                if name not in first:
                    first[name] = key
                assert key[:3] == first[name][:3], (name, rep, "record or centroids changed between repetitions")
                if rep % 20 == 0:
                    assert key == first[name], (name, rep, "lane pixels changed between repetitions")
    finally:
        for c in ctxs.values():
            c.close()


def test_mask_chain_100_times_in_one_process(monkeypatch):
    """The default mask chain (top-hats, walking thresholds with their inline-assembly lane writes, open) on the same four
    frames 100 times, interleaved with a second parameter set that takes the tile kernel: identical masks every time."""
    import zlib
    from lane_tracker_amd import _native, calib, synth
    cal = calib.reference_calibration()
    r = synth.SceneRenderer(cal)
    frames = np.stack([r.render(300 + i)[0] for i in range(3)] + [synth.frame_uniform(9)], 0)
    ctx = _native.Context(cal["img_size"], cal["warped_size"], cal["cam_matrix"], cal["dist_coeffs"], cal["warp_matrices"][0],
                          device=0, capacity=4)
    ctx.set_walk_min_frames(0)                             # four frames: the walking kernels all the same
    try:
        ctx.upload_frames(frames)
        want = {}
        for rep in range(100):
            for tag, fp in (("walk", _native.filter_params()), ("tile", _native.filter_params(ksize_r=17, ksize_b=33))):
                ctx.mask_run(4, fp)
                crc = [zlib.crc32(m.tobytes()) for m in ctx.download_masks(4)]
                assert ctx.last_threshold_path() == (1 if tag == "walk" else 0)
                if tag not in want:
                    want[tag] = crc
                assert crc == want[tag], (tag, rep)
    finally:
        ctx.close()
