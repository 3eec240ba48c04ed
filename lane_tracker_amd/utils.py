"""Calibration loaders with the reference's names and return values (reference utils.py:13-55).
Pickle files written by the reference (`cam_calib.p`, `warp_params.p`) load as before; `.npz` files
with the same keys load without unpickling anything."""
import pickle

import numpy as np


def _load_mapping(filepath):
    if str(filepath).endswith(".npz"):
        with np.load(filepath, allow_pickle=False) as z:
            return {k: z[k] for k in z.files}
    with open(filepath, "rb") as f:
        return pickle.load(f)


def load_camera_calib(filepath):
    """-> (cam_matrix, dist_coeffs)"""
    d = _load_mapping(filepath)
    cam_matrix, dist_coeffs = d['cam_matrix'], d['dist_coeffs']
    print("Camera matrix and distortion coefficients loaded.")
    return cam_matrix, dist_coeffs


def load_warp_params(filepath):
    """-> (M, Minv, image_width_height, warped_width_height, mppv, mpph)"""
    d = _load_mapping(filepath)
    as_size = lambda v: tuple(int(t) for t in np.asarray(v).reshape(-1)[:2])
    out = (d['M'], d['Minv'], as_size(d['image_width_height']), as_size(d['warped_width_height']),
           float(d['mppv']), float(d['mpph']))
    print("Warp parameters loaded.")
    return out


def save_calibration_npz(cam_path, warp_path, cam_matrix, dist_coeffs, M, Minv, image_width_height,
                         warped_width_height, mppv, mpph):
    np.savez(cam_path, cam_matrix=np.asarray(cam_matrix), dist_coeffs=np.asarray(dist_coeffs))
    np.savez(warp_path, M=np.asarray(M), Minv=np.asarray(Minv), image_width_height=np.asarray(image_width_height),
             warped_width_height=np.asarray(warped_width_height), mppv=mppv, mpph=mpph)


def _resize_taps(src_len, dst_len):
    """cv2.resize INTER_LINEAR tap positions and 11-bit coefficients along one axis."""
    f = ((np.arange(dst_len, dtype=np.float64) + 0.5) * (src_len / dst_len) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = f - s.astype(np.float32)
    low, high = s < 0, s >= src_len - 1
    s = np.where(low, 0, np.where(high, src_len - 1, s))
    f = np.where(low | high, np.float32(0), f).astype(np.float32)
    c1 = np.rint(f * np.float32(2048)).astype(np.int64)
    c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
    return s, np.minimum(s + 1, src_len - 1), c0, c1


def resize_linear(img, dsize):
    """cv2.resize(img, dsize=(w, h)) with the default bilinear interpolation on u8 images: half-pixel
    centres, 11-bit coefficients, OpenCV's two-stage fixed-point rounding."""
    img = np.asarray(img, np.uint8)
    dw, dh = int(dsize[0]), int(dsize[1])
    x0, x1, a0, a1 = _resize_taps(img.shape[1], dw)
    y0, y1, b0, b1 = _resize_taps(img.shape[0], dh)
    src = img.astype(np.int64)
    if src.ndim == 3:
        a0, a1 = a0[None, :, None], a1[None, :, None]
        b0, b1 = b0[:, None, None], b1[:, None, None]
    else:
        a0, a1 = a0[None, :], a1[None, :]
        b0, b1 = b0[:, None], b1[:, None]
    rows = src[:, x0] * a0 + src[:, x1] * a1                     # horizontal pass, all source rows
    out = (((b0 * (rows[y0] >> 4)) >> 16) + ((b1 * (rows[y1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def create_split_view(target_size, images, positions, sizes, captions=[]):
    """Place images on a canvas of `target_size` (w, h) (reference utils.py:57-103): each image is
    resized to its slot with the bilinear `resize_linear`; one-channel images fill all three
    channels; captions are not rendered (they need OpenCV's Hershey glyphs)."""
    assert len(images) == len(positions) == len(sizes)
    x_max, y_max = target_size
    canvas = np.zeros((y_max, x_max, 3), dtype=np.uint8)
    for img, (x, y), (w, h) in zip(images, positions, sizes):
        if img.shape[0] != h or img.shape[1] != w:
            img = resize_linear(img, (w, h))
        if img.ndim == 2:
            img = img[:, :, None]
        canvas[y:min(y + h, y_max), x:min(x + w, x_max), :] = img[:min(h, y_max - y), :min(w, x_max - x)]
    return canvas
