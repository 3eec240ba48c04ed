"""Stand-in for moviepy on machines without moviepy/ffmpeg: `moviepy.editor.VideoFileClip` over frame
sequences (lane_tracker_amd/video.py).  Opt in by putting `dropin/frames_backend` on PYTHONPATH; leave it
off when the real moviepy is installed."""
