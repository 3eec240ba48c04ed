"""Calibration constants of the reference (cam_calib.p / warp_params.p) and the
config-5 rescaling to a 1920x1080 camera.  Kept as literals so that nothing has to unpickle foreign
files on the GPU box; `utils.load_camera_calib` / `load_warp_params` still read the pickles when a
user has them (reference utils.py:13-55)."""
import numpy as np

CAM_MATRIX = np.array([[1154.3293544733699, 0.0, 669.68287497444635],
                       [0.0, 1148.4715517793131, 385.86265402405462],
                       [0.0, 0.0, 1.0]], dtype=np.float64)
DIST_COEFFS = np.array([[-0.24180123999440323, -0.047799949862003206, -0.0011385469776010269,
                         -0.00011245666608284012, 0.018317194299296482]], dtype=np.float64)
M = np.array([[-1.6192154913816159e-01, -1.2786360662302192e+00, 6.4141214769155897e+02],
              [-1.6944778913341452e-14, -3.0195465710372815e+00, 1.3808914123085760e+03],
              [-1.4745149545802860e-17, -2.3776238777793780e-03, 1.0]], dtype=np.float64)
MINV = np.array([[5.3932875397536018e-01, -5.0395955194543518e-01, 3.4998140303318553e+02],
                 [8.8817841970012523e-16, -3.3117555118762820e-01, 4.5731747460155714e+02],
                 [-3.0357660829594124e-18, -7.8741089824045977e-04, 1.0]], dtype=np.float64)
IMAGE_WIDTH_HEIGHT = (1280, 720)
WARPED_WIDTH_HEIGHT = (1080, 1100)
MPPV = 0.03048
MPPH = 0.0146304


def reference_calibration():
    """dict with the constructor arguments process_video.py:28-37 passes to LaneTracker."""
    return dict(img_size=IMAGE_WIDTH_HEIGHT, warped_size=WARPED_WIDTH_HEIGHT,
                cam_matrix=CAM_MATRIX.copy(), dist_coeffs=DIST_COEFFS.copy(),
                warp_matrices=(M.copy(), MINV.copy()), mpp_conversion=(MPPV, MPPH))


def scaled_calibration(scale=1.5):
    """BASELINE config 5 (1920x1080 camera): scale the INPUT side only -- K' = S K, M' = M S^-1,
    Minv' = S Minv -- and keep the bird's-eye view at 1080x1100, because every pixel constant
    downstream of the warp is absolute."""
    S = np.diag([scale, scale, 1.0])
    Sinv = np.diag([1.0 / scale, 1.0 / scale, 1.0])
    w, h = IMAGE_WIDTH_HEIGHT
    return dict(img_size=(int(round(w * scale)), int(round(h * scale))),
                warped_size=WARPED_WIDTH_HEIGHT, cam_matrix=S @ CAM_MATRIX,
                dist_coeffs=DIST_COEFFS.copy(), warp_matrices=(M @ Sinv, S @ MINV),
                mpp_conversion=(MPPV, MPPH))
