import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from lane_tracker_amd.video import VideoFileClip  # noqa: E402,F401
