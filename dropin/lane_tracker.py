"""Drop-in module: put this directory on sys.path (before the reference's) and
`from lane_tracker import bilateral_adaptive_threshold, LaneTracker` (process_video.py:14-15)
resolves to the MI355X implementation."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _root not in sys.path:
    sys.path.insert(0, _root)

from lane_tracker_amd.lane_tracker import LaneTracker, bilateral_adaptive_threshold  # noqa: E402,F401
