"""Runs under a SECOND interpreter that has scikit-image (tests/test_skimage_crosscheck.py starts it): independent,
third-party implementations of two of the cv2-backed stages, in floating point.

    python skimage_side.py in.npz out.npz

in.npz:  colors (N,3) u8;  image (H,W,3) u8;  M (3,3) f64;  out_hw (2,) int
out.npz: lab_b (N,) f64 = CIE b* of the sRGB colours (D65, skimage.color.rgb2lab);
         warped (h,w,3) f64 = the image under cv2.warpPerspective(image, M, (w, h)) semantics -- dst(x, y) = src(M^-1 (x, y)),
         bilinear, constant 0 outside (skimage.transform.warp with the inverse map)."""
import sys

import numpy as np
import skimage
from skimage import color, transform

d = np.load(sys.argv[1])
lab = color.rgb2lab(d["colors"][None].astype(np.float64) / 255.0)[0]
h, w = (int(v) for v in d["out_hw"])
tf = transform.ProjectiveTransform(matrix=np.linalg.inv(d["M"]))
warped = transform.warp(d["image"].astype(np.float64), tf, output_shape=(h, w), order=1, mode="constant", cval=0.0, preserve_range=True)
np.savez(sys.argv[2], lab_b=lab[:, 2], warped=warped, version=np.array(skimage.__version__))
