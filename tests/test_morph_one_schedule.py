"""The schedule of k_morph_one (csrc/k_tophat.hip: the walk of a band split over the Q waves of a workgroup by row pairs, partial
results through a ring of rows) restated in NumPy and held against the direct evaluation of the footprint -- no GPU: what is
checked is the index arithmetic of the kernel's header comment (which offsets a wave delivers in which step, when a row is
complete, that a ring of 3 x 2Q rows is enough), for toy structuring elements and for the 29x29 ellipse."""
import numpy as np
import pytest


def _run(K, dx, Q, h, w, band_rows, seed):
    R, S = K // 2, 2 * Q
    RING = 3 * S
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, (h, w)).astype(np.int64)

    def H(row, d):      # horizontal minimum of half-width d of a (clamped) row, columns clamped
        r = img[min(max(row, 0), h - 1)]
        idx = np.clip(np.arange(w)[:, None] + np.arange(-d, d + 1)[None, :], 0, w - 1)
        return r[idx].min(1)
    ref = np.full((h, w), 255, np.int64)        # out-of-image taps ignored
    for y in range(h):
        for i in range(K):
            yy = y + i - R
            if 0 <= yy < h:
                for x in range(w):
                    ref[y, x] = min(ref[y, x], img[yy, max(x - dx[i], 0):min(x + dx[i], w - 1) + 1].min())
    out = np.full((h, w), -1, np.int64)
    for band in range((h + band_rows - 1) // band_rows):
        yb0 = band * band_rows
        yb1 = min(yb0 + band_rows, h)
        y_first, y_base = yb0 - R, yb0 - 2 * R
        npairs = (yb1 - yb0 + 2 * R + 1) // 2
        nsteps = (npairs + Q - 1) // Q
        A = [[np.full(w, 255, np.int64) for _ in range(K + 1)] for _ in range(Q)]     # position j = -1 .. K - 1 at index j + 1
        ring = [[None] * RING for _ in range(Q)]
        written_in_step = [[-9] * RING for _ in range(Q)]
        for t in range(nsteps):
            for wv in range(Q):
                yy = y_first + 2 * (t * Q + wv)
                Ha = {d: H(yy, d) for d in set(dx)}
                Hb = {d: H(yy + 1, d) for d in set(dx)}
                An = [None] * (K + 1)
                for p in range(K + 1):
                    j = p - 1
                    terms = []
                    if j + S <= K - 1:
                        terms.append(A[wv][p + S])
                    if j <= K - 2:
                        terms.append(Ha[dx[j + 1]])
                    if j >= 0:
                        terms.append(Hb[dx[j]])
                    An[p] = np.minimum.reduce(terms)
                o0 = S * t + 2 * wv
                for q in range(S):
                    # a slot is never rewritten before the step after the one that combines it (the ring is long enough)
                    assert written_in_step[wv][(o0 + q) % RING] <= t - 2
                    ring[wv][(o0 + q) % RING] = An[q]
                    written_in_step[wv][(o0 + q) % RING] = t
                for p in range(S, K + 1):
                    A[wv][p] = An[p]
            for wv in range(Q):         # behind the step's barrier
                o0 = S * t + 2 * wv
                for rr in range(2):
                    y = y_base + o0 + rr
                    if yb0 <= y < yb1:
                        assert all(written_in_step[q][(o0 + rr) % RING] in (t - 1, t) for q in range(Q))
                        out[y] = np.minimum.reduce([ring[q][(o0 + rr) % RING] for q in range(Q)])
    return out, ref


ELL29 = [0, 5, 7, 9, 10, 11, 11, 12, 13, 13, 13, 14, 14, 14, 14, 14, 14, 14, 13, 13, 13, 12, 11, 11, 10, 9, 7, 5, 0]


@pytest.mark.parametrize("K,dx,Q,h,w,band_rows", [(7, [0, 2, 3, 3, 3, 2, 0], 2, 23, 17, 6), (9, [0, 2, 3, 4, 4, 4, 3, 2, 0], 4, 31, 13, 10),
                                                   (9, [0, 2, 3, 4, 4, 4, 3, 2, 0], 2, 30, 13, 7), (29, ELL29, 4, 50, 12, 20),
                                                   (29, ELL29, 8, 44, 10, 44)])
def test_wave_split_walk_equals_the_footprint(K, dx, Q, h, w, band_rows):
    out, ref = _run(K, dx, Q, h, w, band_rows, seed=K + Q)
    assert np.array_equal(out, ref)
