"""Synthetic inputs: seeded binary masks for the search/fit stages, and
camera frames -- (S1) iid-uniform bytes, (S2) lane-like road scenes -- for the whole path.
All generators are pure NumPy on `default_rng(seed)` (PCG64, platform-stable)."""
import numpy as np

from . import calib as _calib


# ---- bird's-eye binary masks (inputs of sliding_window_search / band_search) ----------------------
def lane_polys(rng, h=1100, w=1080, left_base=(419, 459), sep=(160, 198), slope=0.05, curv=1e-4):
    """Two parabolas x(y) = a*(y-(h-1))^2 + s*(y-(h-1)) + xb given as np.polyfit-order coeffs in y."""
    xb = rng.uniform(*left_base)
    s = rng.uniform(-slope, slope)
    a = rng.uniform(-curv, curv)
    d = rng.uniform(*sep)
    y0 = h - 1

    def expand(xb_):
        return np.array([a, s - 2 * a * y0, a * y0 * y0 - s * y0 + xb_])
    return expand(xb), expand(xb + d)


def synth_mask(seed, h=1100, w=1080, noise=1e-3, dashed_right=True, line_width=12, left_base=(419, 459),
               sep=(160, 198), slope=0.05, curv=1e-4, drop_left=False, drop_right=False, value=255):
    """Lane-like binary mask {0,value}: solid left line, dashed right line, salt noise."""
    rng = np.random.default_rng(seed)
    lc, rc = lane_polys(rng, h, w, left_base, sep, slope, curv)
    yy = np.arange(h)[:, None].astype(np.float64)
    xx = np.arange(w)[None, :].astype(np.float64)
    m = np.zeros((h, w), bool)
    if not drop_left:
        m |= np.abs(xx - (lc[0] * yy * yy + lc[1] * yy + lc[2])) <= line_width / 2
    if not drop_right:
        r = np.abs(xx - (rc[0] * yy * yy + rc[1] * yy + rc[2])) <= line_width / 2
        if dashed_right:
            phase = int(rng.integers(0, 150))
            r &= (((np.arange(h) + phase) % 150) < 60)[:, None]
        m |= r
    if noise > 0:
        m |= rng.random((h, w)) < noise
    return (m.astype(np.uint8) * value), lc, rc


def random_mask(seed, h=1100, w=1080, density=0.5, value=255):
    rng = np.random.default_rng(seed)
    return ((rng.random((h, w)) < density).astype(np.uint8) * value)


# ---- camera frames --------------------------------------------------------------------------------
def frame_uniform(seed, img_size=_calib.IMAGE_WIDTH_HEIGHT):
    """(S1) iid-uniform u8 frame: bit-exactness stress, worst-case mask density."""
    w, h = img_size
    return np.random.default_rng(seed).integers(0, 256, (h, w, 3), dtype=np.uint8)


class SceneRenderer:
    """(S2) lane-like frames: a bird's-eye road canvas (grey asphalt + noise, yellow solid left
    line, white dashed right line) seen through the calibration's camera model.  The camera->BEV
    coordinate map (distortion + homography, float) is built once per calibration; it only creates
    inputs, so its arithmetic is not part of any parity claim."""

    def __init__(self, cal=None):
        cal = cal or _calib.reference_calibration()
        self.cal = cal
        (self.w, self.h), (self.bw, self.bh) = cal["img_size"], cal["warped_size"]
        K, D, M = cal["cam_matrix"], np.asarray(cal["dist_coeffs"]).reshape(-1), cal["warp_matrices"][0]
        u, v = np.meshgrid(np.arange(self.w, dtype=np.float64), np.arange(self.h, dtype=np.float64))
        xd, yd = (u - K[0, 2]) / K[0, 0], (v - K[1, 2]) / K[1, 1]
        x, y = xd.copy(), yd.copy()
        k1, k2, p1, p2, k3 = D[:5]
        for _ in range(8):  # invert the distortion model by fixed-point iteration
            r2 = x * x + y * y
            kr = 1 + ((k3 * r2 + k2) * r2 + k1) * r2
            dx = 2 * p1 * x * y + p2 * (r2 + 2 * x * x)
            dy = p1 * (r2 + 2 * y * y) + 2 * p2 * x * y
            x, y = (xd - dx) / kr, (yd - dy) / kr
        uu, vu = K[0, 0] * x + K[0, 2], K[1, 1] * y + K[1, 2]
        den = M[2, 0] * uu + M[2, 1] * vu + M[2, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            bx = (M[0, 0] * uu + M[0, 1] * vu + M[0, 2]) / den
            by = (M[1, 0] * uu + M[1, 1] * vu + M[1, 2]) / den
        ok = (np.abs(den) > 1e-9) & (bx >= 0) & (bx <= self.bw - 1.001) & (by >= 0) & (by <= self.bh - 1.001)
        self.ok = ok
        bx, by = np.where(ok, bx, 0.0), np.where(ok, by, 0.0)
        x0, y0 = np.floor(bx).astype(np.int64), np.floor(by).astype(np.int64)
        self.idx = (y0 * self.bw + x0)[ok]
        fx, fy = (bx - x0)[ok], (by - y0)[ok]
        self.wts = np.stack([(1 - fx) * (1 - fy), fx * (1 - fy), (1 - fx) * fy, fx * fy], 0).astype(np.float32)

    def canvas(self, rng, lc, rc, dashed_phase=0, line_width=12):
        bh, bw = self.bh, self.bw
        can = np.empty((bh, bw, 3), np.float32)
        can[:] = 90.0 + rng.integers(-10, 11, (bh, bw, 1)).astype(np.float32)
        yy = np.arange(bh, dtype=np.float64)[:, None]
        xx = np.arange(bw, dtype=np.float64)[None, :]
        left = np.abs(xx - (lc[0] * yy * yy + lc[1] * yy + lc[2])) <= line_width / 2
        right = np.abs(xx - (rc[0] * yy * yy + rc[1] * yy + rc[2])) <= line_width / 2
        right &= (((np.arange(bh) + dashed_phase) % 150) < 60)[:, None]
        can[left] = (220.0, 190.0, 60.0)
        can[right] = (235.0, 235.0, 235.0)
        return can

    def render(self, seed, lc=None, rc=None, dashed_phase=None):
        """-> (frame u8 HxWx3, left coeffs, right coeffs) ; coeffs are BEV parabolas in y."""
        rng = np.random.default_rng(seed)
        if lc is None:
            lc, rc = lane_polys(rng, self.bh, self.bw)
        if dashed_phase is None:
            dashed_phase = int(rng.integers(0, 150))
        can = self.canvas(rng, lc, rc, dashed_phase).reshape(-1, 3)
        frame = rng.integers(96, 224, (self.h, self.w, 3), dtype=np.uint8)  # sky / surroundings
        i = self.idx
        val = (can[i] * self.wts[0][:, None] + can[i + 1] * self.wts[1][:, None]
               + can[i + self.bw] * self.wts[2][:, None] + can[i + self.bw + 1] * self.wts[3][:, None])
        frame[self.ok] = np.clip(np.rint(val), 0, 255).astype(np.uint8)
        return frame, lc, rc


def frames_lanes(seeds, cal=None):
    """Batch of S2 frames, one per seed (fresh scene each): (n, H, W, 3) u8."""
    r = SceneRenderer(cal)
    return np.stack([r.render(int(s))[0] for s in seeds], 0)


def stream_lane_params(n, seed=0, bh=1100):
    """The per-frame arguments of `SceneRenderer.render` for a config-5 style stream (one scene whose lane geometry
    drifts slowly): a list of (render seed, left coeffs, right coeffs, dashed phase)."""
    rng = np.random.default_rng(seed)
    xb, s, a, d = rng.uniform(429, 449), 0.0, 0.0, rng.uniform(170, 195)
    y0 = bh - 1
    out = []
    for i in range(n):
        xb += rng.uniform(-1.0, 1.0)
        s = float(np.clip(s + rng.uniform(-0.004, 0.004), -0.05, 0.05))
        a = float(np.clip(a + rng.uniform(-4e-6, 4e-6), -1e-4, 1e-4))
        ex = lambda b: np.array([a, s - 2 * a * y0, a * y0 * y0 - s * y0 + b])
        out.append((seed * 100003 + i, ex(xb), ex(xb + d), (i * 20) % 150))
    return out


def stream_lanes(n, seed=0, cal=None):
    """Config-5 style stream: one scene whose lane geometry drifts slowly from frame to frame."""
    r = SceneRenderer(cal)
    return np.stack([r.render(sd, lc, rc, dashed_phase=ph)[0] for sd, lc, rc, ph in stream_lane_params(n, seed, r.bh)], 0)
