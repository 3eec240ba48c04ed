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


def create_split_view(target_size, images, positions, sizes, captions=[]):
    """Place images on a canvas of `target_size` (w, h) (reference utils.py:57-103).  Nearest-neighbour
    resize; captions are not rendered (presentation only)."""
    assert len(images) == len(positions) == len(sizes)
    x_max, y_max = target_size
    canvas = np.zeros((y_max, x_max, 3), dtype=np.uint8)
    for img, (x, y), (w, h) in zip(images, positions, sizes):
        if img.shape[0] != h or img.shape[1] != w:
            yy = (np.arange(h) * img.shape[0] / h).astype(np.int64)
            xx = (np.arange(w) * img.shape[1] / w).astype(np.int64)
            img = img[yy][:, xx]
        if img.ndim == 2:
            img = img[:, :, None]
        canvas[y:min(y + h, y_max), x:min(x + w, x_max), :] = img[:min(h, y_max - y), :min(w, x_max - x)]
    return canvas
