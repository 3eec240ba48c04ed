"""The author's three demo configurations (reference tracker_settings.md:1-111; SURVEY.md section 8(f),
row N3): keyword sets for `LaneTracker.process()` and the lane-separation / tangent limits of
`check_validity()`, which upstream edits in the source for each video (`lane_tracker.py:588-593, 617`)
and which `LaneTracker.validity_limits` exposes per instance.

    lt = LaneTracker(**calibration)
    settings.apply(lt, settings.DEMO_1)                 # sets lt.validity_limits
    out = lt.process(frame, **settings.DEMO_1["process"])
"""

_COMMON = dict(ksize_r=15, C_r=8, ksize_b=35, C_b=5, filter_type='bilateral', mask_noise=True, noise_thresh=140,
               ksize_noise=65, C_noise=10, window_width=30, window_height=40, search_range=20, mu=0.1,
               no_success_limit=50, start_slice=0.25, ignore_sides=360, ignore_bottom=30, bandwidth=30, partial=1.0,
               n_tries=2)

# highway, good contrast, greenery next to the road (tracker_settings.md:1-33)
DEMO_1 = dict(process=dict(_COMMON),
              validity=dict(min_dist_y1=150, max_dist_y1=245, min_dist_y2=150, max_dist_y2=255, min_dist_y3=150,
                            max_dist_y3=255, thresh=0.25))

# low-contrast concrete highway; these validity limits are the ones hard-coded upstream (:35-72)
DEMO_2 = dict(process=dict(_COMMON, ksize_r=20, C_r=5, mask_noise=False, n_tries=1),
              validity=dict(min_dist_y1=150, max_dist_y1=230, min_dist_y2=110, max_dist_y2=230, min_dist_y3=80,
                            max_dist_y3=200, thresh=0.25))

# mountain road with sharp turns: half the look-ahead, wider tolerances (:74-111)
DEMO_3 = dict(process=dict(_COMMON, partial=0.5),
              validity=dict(min_dist_y1=150, max_dist_y1=245, min_dist_y2=140, max_dist_y2=265, min_dist_y3=125,
                            max_dist_y3=290, thresh=0.46))

DEMOS = {"demo1": DEMO_1, "demo2": DEMO_2, "demo3": DEMO_3}


def apply(tracker, demo):
    """Install a demo's validity limits on a tracker; returns its process() keywords."""
    tracker.validity_limits = dict(demo["validity"])
    return dict(demo["process"])
