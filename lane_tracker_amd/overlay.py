"""Presentation of the result on the camera frame (reference draw_lane / print_failure,
lane_tracker.py:629-673).  SURVEY.md section 8(f) row N1: outside the accelerated path and not
bit-matched -- OpenCV's fillPoly rasteriser, its inverse warp of the polygon and its anti-aliased
Hershey text cannot be reproduced without OpenCV.  This is plain NumPy (+ Pillow for text when it
is installed) so that `LaneTracker.process()` returns an annotated frame like the reference."""
import numpy as np


class LaneOverlay:
    def __init__(self, img_size, warped_size, M):
        self.w, self.h = int(img_size[0]), int(img_size[1])
        self.bw, self.bh = int(warped_size[0]), int(warped_size[1])
        M = np.asarray(M, np.float64)
        u, v = np.meshgrid(np.arange(self.w, dtype=np.float64), np.arange(self.h, dtype=np.float64))
        den = M[2, 0] * u + M[2, 1] * v + M[2, 2]
        with np.errstate(divide="ignore", invalid="ignore"):
            bx = np.rint((M[0, 0] * u + M[0, 1] * v + M[0, 2]) / den)
            by = np.rint((M[1, 0] * u + M[1, 1] * v + M[1, 2]) / den)
        ok = np.isfinite(bx) & np.isfinite(by) & (np.abs(den) > 1e-12)
        ok &= (bx >= 0) & (bx < self.bw) & (by >= 0) & (by < self.bh)
        self.cam_idx = np.flatnonzero(ok.ravel())
        self.bx = bx[ok].astype(np.int32)      # bird's-eye pixel under each camera pixel that sees the road plane
        self.by = by[ok].astype(np.int32)
        self.cam_g = self.cam_idx * 3 + 1      # flat offset of the green byte of those camera pixels

    def draw(self, img, left_y, left_x, right_y, right_x, alpha=0.3):
        """Green lane polygon between the two averaged curves, blended like addWeighted(img,1,lane,0.3,0)."""
        out = np.array(img, dtype=np.uint8, copy=True)
        ly, lx = np.asarray(left_y, np.int64), np.asarray(left_x, np.int64)
        ry, rx = np.asarray(right_y, np.int64), np.asarray(right_x, np.int64)
        if ly.size and ry.size:
            # per bird's-eye row: the span between the two averaged curves (rows where both exist)
            lo = np.full(self.bh, 1 << 30, np.int32)
            hi = np.full(self.bh, -1, np.int32)
            lrow = np.full(self.bh, -1, np.int32)
            rrow = np.full(self.bh, -1, np.int32)
            lrow[np.clip(ly, 0, self.bh - 1)] = lx
            rrow[np.clip(ry, 0, self.bh - 1)] = rx
            both = (lrow >= 0) & (rrow >= 0)
            lo[both] = np.minimum(lrow, rrow)[both]
            hi[both] = np.maximum(lrow, rrow)[both]
            # evaluated only at the camera pixels, not over the whole bird's-eye image
            hit = (self.bx >= lo[self.by]) & (self.bx <= hi[self.by])
            sel = self.cam_g[hit]
            flat = out.reshape(-1)
            flat[sel] = np.minimum(255, flat[sel].astype(np.int16) + int(np.rint(255 * alpha))).astype(np.uint8)
        return out


_FONT = None


def _font():
    global _FONT
    if _FONT is None:
        from PIL import ImageFont
        try:
            _FONT = ImageFont.load_default(size=28)
        except Exception:
            _FONT = ImageFont.load_default()
    return _FONT


def put_lines(img, lines, origin=(20, 8), step=35):
    """White text lines at the reference's positions ((20,35), (20,70), ... baselines).  Only the text
    strip goes through Pillow."""
    try:
        from PIL import Image, ImageDraw
    except Exception:
        return img
    img = np.ascontiguousarray(img)
    strip_h = min(img.shape[0], origin[1] + len(lines) * step + 8)
    pil = Image.fromarray(img[:strip_h])
    d = ImageDraw.Draw(pil)
    for i, text in enumerate(lines):
        d.text((origin[0], origin[1] + i * step), text, fill=(255, 255, 255), font=_font())
    if not img.flags.writeable:
        img = img.copy()
    img[:strip_h] = np.asarray(pil)
    return img
