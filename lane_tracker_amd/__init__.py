"""MI355X-native lane-tracker hot path (undistort -> warp -> filter_lane_points ->
sliding_window_search / band_search -> fit_poly) behind the reference's LaneTracker API."""
__version__ = "0.1.0"
