"""Shared helpers for the parity tests."""
import glob
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def unpack_mask(d):
    h, w = [int(v) for v in d["mask_shape"]]
    bits = np.unpackbits(d["mask_bits"])[: h * w].reshape(h, w)
    return (bits * int(d["mask_value"])).astype(np.uint8)


def params_of(d):
    return {k[len("param_"):]: d[k].item() for k in d.files if k.startswith("param_")}


def coeff_close(got, want, h=1100, tol=1e-4):
    """BASELINE tolerance: 1e-4 relative per coefficient with the absolute floor of SURVEY 8(a):
    |da| H^2, |db| H and |dc| each <= tol * max(1, |c|)."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    lim = tol * max(1.0, abs(float(want[2])))
    return (abs(got[0] - want[0]) * h * h <= lim and abs(got[1] - want[1]) * h <= lim
            and abs(got[2] - want[2]) <= lim)
