"""Drop-in module for `from utils import load_camera_calib, load_warp_params` (process_video.py:16-17)."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)

from lane_tracker_amd.utils import create_split_view, load_camera_calib, load_warp_params  # noqa: E402,F401
