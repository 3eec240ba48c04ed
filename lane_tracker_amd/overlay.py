"""Presentation helpers around the accelerated path.

* The lane overlay of `draw_lane()` (lane_tracker.py:629-662: fillPoly -> warpPerspective(Minv) ->
  addWeighted) runs on the GPU (`lt_overlay_run`, csrc/k_overlay.hip); this module only adds the text.
* The search visualisations (`visualize_sliding_window_search` :687-729, `visualize_band_search`
  :731-771) are debugging aids made of NumPy indexing plus three cv2 calls (merge, fillPoly,
  addWeighted); they are restated here in NumPy, the polygon through the library's host-side
  `lt_lane_polygon_spans`.
* Text: OpenCV's anti-aliased Hershey glyphs cannot be reproduced without OpenCV; the lines are drawn
  with Pillow's default font at the reference's positions when Pillow is installed."""
import numpy as np

from . import _native


def add_weighted(a, alpha, b, beta, gamma=0.0):
    """cv2.addWeighted on u8 arrays: f32 products and sums, round-half-even, saturate."""
    t = a.astype(np.float32) * np.float32(alpha) + b.astype(np.float32) * np.float32(beta)
    t = t + np.float32(gamma)
    return np.clip(np.rint(t), 0, 255).astype(np.uint8)


def fill_band(img, ys, x_lo, x_hi, color):
    """cv2.fillPoly of the polygon (x_lo, ys) followed by the reversed (x_hi, ys), in place."""
    h, w = img.shape[:2]
    spans = _native.lane_polygon_spans(h, ys, x_lo, ys, x_hi)
    rows = np.flatnonzero(spans[:, 0] <= spans[:, 1])
    for y in rows:
        lo, hi = max(int(spans[y, 0]), 0), min(int(spans[y, 1]), w - 1)
        if lo <= hi:
            img[y, lo:hi + 1] = color
    return img


def _window_box(shape, window_width, window_height, center, level, ignore_bottom):
    """Row and column slices of search window `level` (lane_tracker.py:684, same int() truncations)."""
    img_height = shape[0] - ignore_bottom
    return (slice(int(img_height - (level + 1) * window_height), int(img_height - level * window_height)),
            slice(max(int(center - window_width / 2), 0), min(int(center + window_width / 2), shape[1])))


def window_mask(img, window_width, window_height, center, level, ignore_bottom):
    """lane_tracker.py:675-685: 1 inside the search window, 0 elsewhere."""
    output = np.zeros_like(img)
    output[_window_box(img.shape, window_width, window_height, center, level, ignore_bottom)] = 1
    return output


def visualize_sliding_window_search(binary_img, left_centroids, right_centroids, left_yx, right_yx, fit_points,
                                    window_width, window_height, ignore_bottom):
    """lane_tracker.py:687-729: windows in half-transparent green over the mask, lane pixels red / blue,
    fitted curves yellow."""
    sides = []
    for cents in (left_centroids, right_centroids):
        pts = np.zeros_like(binary_img)
        for level, center in enumerate(cents):
            pts[_window_box(binary_img.shape, window_width, window_height, center, level, ignore_bottom)] = 255
        sides.append(pts)
    green = (sides[1] + sides[0]).astype(np.uint8)                # u8 wrap-around (254) where both overlap, as upstream
    out = np.repeat(binary_img[:, :, None], 3, axis=2)
    out[:, :, 1] = add_weighted(binary_img, 1, green, 0.5, 0.0)   # the template's red and blue channels are zero
    out[left_yx[0], left_yx[1]] = (255, 0, 0)
    out[right_yx[0], right_yx[1]] = (0, 0, 255)
    left_fit_y, left_fit_x, right_fit_y, right_fit_x = fit_points
    out[left_fit_y, left_fit_x] = (255, 235, 0)
    out[right_fit_y, right_fit_x] = (255, 235, 0)
    return out


def visualize_band_search(binary_img, left_yx, right_yx, band_points, fit_points, bandwidth):
    """lane_tracker.py:731-771: lane pixels red / blue, the two search bands (previous curves +- bandwidth)
    in transparent green, the new fitted curves yellow."""
    out = np.repeat(binary_img[:, :, None], 3, axis=2)
    out[left_yx[0], left_yx[1]] = (255, 0, 0)
    out[right_yx[0], right_yx[1]] = (0, 0, 255)
    band = np.zeros(binary_img.shape, np.uint8)
    left_band_y, left_band_x, right_band_y, right_band_x = band_points
    fill_band(band, left_band_y, left_band_x - bandwidth, left_band_x + bandwidth, 255)
    fill_band(band, right_band_y, right_band_x - bandwidth, right_band_x + bandwidth, 255)
    out[:, :, 1] = add_weighted(out[:, :, 1], 1, band, 0.3, 0)
    left_fit_y, left_fit_x, right_fit_y, right_fit_x = fit_points
    out[left_fit_y, left_fit_x] = (255, 235, 0)
    out[right_fit_y, right_fit_x] = (255, 235, 0)
    return out


_ATLAS = None


def font_atlas(size=28, first_char=32, last_char=126):
    """Glyph atlas for the text lines of draw_lane / print_failure (lt_overlay_set_font): alpha cells of the
    printable ASCII characters rendered once with Pillow's default font, and each character's advance.
    Returns (atlas (n, gh, gw) u8, advance (n,) u8, first_char), or None when Pillow is not installed.
    OpenCV's anti-aliased Hershey glyphs cannot be reproduced without OpenCV; this is the build's own font."""
    global _ATLAS
    if _ATLAS is None:
        try:
            from PIL import Image, ImageDraw, ImageFont
        except Exception:
            _ATLAS = False
            return None
        try:
            font = ImageFont.load_default(size=size)
        except Exception:
            font = ImageFont.load_default()
        chars = [chr(c) for c in range(first_char, last_char + 1)]
        ascent, descent = font.getmetrics() if hasattr(font, "getmetrics") else (size, size // 4)
        gh = int(ascent + descent)
        adv = [max(1, int(np.ceil(font.getlength(ch)))) if hasattr(font, "getlength") else size // 2 for ch in chars]
        gw = int(max(adv)) + 2
        atlas = np.zeros((len(chars), gh, gw), np.uint8)
        for i, ch in enumerate(chars):
            cell = Image.new("L", (gw, gh), 0)
            ImageDraw.Draw(cell).text((0, 0), ch, fill=255, font=font)
            atlas[i] = np.asarray(cell)
        _ATLAS = (atlas, np.minimum(np.asarray(adv), gw).astype(np.uint8), first_char)
    return _ATLAS or None
