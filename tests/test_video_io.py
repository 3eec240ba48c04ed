"""Frame source / sink: round trips on the CPU; on the GPU the windowed pipeline and a caller
of the reference's API surface (imports, keyword constructor, fl_image/write_videofile, get_success_ratio)
run against dropin/ + the moviepy stand-in."""
import os
import sys

import numpy as np
import pytest

from lane_tracker_amd import video

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _frames(n=5, h=24, w=32, seed=0):
    return np.random.default_rng(seed).integers(0, 256, (n, h, w, 3), dtype=np.uint8)


def test_npy_and_raw_round_trip(tmp_path):
    fr = _frames()
    for name in ("clip.npy", "clip.rgb"):
        with video.FrameSink(tmp_path / name, (32, 24), n=len(fr)) as sink:
            sink.write(fr[:2])
            sink.write(fr[2])            # a single frame
            sink.write(fr[3:])
        src = video.FrameSource(tmp_path / name, size=(32, 24))
        assert len(src) == 5 and src.size == (32, 24)
        assert np.array_equal(src.read(0, 5), fr)
        assert np.array_equal(src.read(3, 99), fr[3:])
        assert np.array_equal(np.stack(list(src)), fr)


def test_png_directory_round_trip_and_order(tmp_path):
    fr = _frames(12)
    with video.FrameSink(tmp_path / "out", (32, 24)) as sink:
        sink.write(fr)
    names = sorted(os.listdir(tmp_path / "out"))
    assert names[0] == "frame_000000.png" and names[-1] == "frame_000011.png"
    src = video.FrameSource(tmp_path / "out")
    assert np.array_equal(src.read(0, 12), fr)        # PNG is lossless, order = name order


def test_source_errors(tmp_path):
    with pytest.raises(ValueError):
        video.FrameSource(tmp_path)                                # empty directory
    (tmp_path / "x.rgb").write_bytes(b"\0" * 100)
    with pytest.raises(ValueError):
        video.FrameSource(tmp_path / "x.rgb")                      # raw needs a size
    with pytest.raises(ValueError):
        video.FrameSource(tmp_path / "x.rgb", size=(32, 24))       # not a whole number of frames
    np.save(tmp_path / "bad.npy", np.zeros((3, 4, 5), np.uint8))
    with pytest.raises(ValueError):
        video.FrameSource(tmp_path / "bad.npy")
    np.save(tmp_path / "ok.npy", _frames(2))
    with pytest.raises(ValueError):
        video.FrameSource(tmp_path / "ok.npy", size=(64, 48))      # size mismatch
    with pytest.raises(ValueError):
        video.FrameSink(tmp_path / "o.npy", (32, 24))              # .npy needs n
    with video.FrameSink(tmp_path / "o.rgb", (32, 24)) as s:
        with pytest.raises(ValueError):
            s.write(np.zeros((2, 10, 10, 3), np.uint8))


def test_clip_path_resolution(tmp_path):
    np.save(tmp_path / "drive.npy", _frames(3))
    assert video.resolve_clip_path(str(tmp_path / "drive.mp4")) == str(tmp_path / "drive.npy")
    os.makedirs(tmp_path / "road")
    assert video.resolve_clip_path(str(tmp_path / "road.mp4")) == str(tmp_path / "road")
    assert video.resolve_clip_path(str(tmp_path / "out.mp4"), must_exist=False) == str(tmp_path / "out")
    assert video.resolve_clip_path(str(tmp_path / "x.npy")) == str(tmp_path / "x.npy")
    with pytest.raises(FileNotFoundError):
        video.resolve_clip_path(str(tmp_path / "missing.mp4"))


def test_clip_with_plain_function(tmp_path):
    fr = _frames(4)
    np.save(tmp_path / "in.npy", fr)
    clip = video.VideoFileClip(str(tmp_path / "in.mp4"))
    assert clip.size == (32, 24)
    clip.fl_image(lambda img: 255 - img).write_videofile(str(tmp_path / "neg.npy"), audio=False)
    assert np.array_equal(np.load(tmp_path / "neg.npy"), 255 - fr)


# ---- GPU ------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_process_frames_equals_process(tmp_path):
    from lane_tracker_amd import calib, synth
    from lane_tracker_amd.lane_tracker import LaneTracker
    cal = calib.reference_calibration()
    lanes = synth.stream_lanes(14, seed=2)
    frames = np.stack([lanes[i] if i not in (6, 7) else np.full_like(lanes[0], 128) for i in range(14)], 0)
    np.save(tmp_path / "in.npy", frames)
    a, b = LaneTracker(**cal), LaneTracker(**cal)
    try:
        want = [a.process(f) for f in frames]
        with video.FrameSink(tmp_path / "out.rgb", cal["img_size"]) as sink:
            n, _ = video.process_frames(b, video.FrameSource(tmp_path / "in.npy"), sink, window=5)   # 5 + 5 + 4
        assert n == 14 and a.get_success_ratio() == b.get_success_ratio()
        got = video.FrameSource(tmp_path / "out.rgb", size=cal["img_size"]).read(0, 14)
        for i in range(14):
            assert np.array_equal(got[i], want[i]), i
    finally:
        a.close()
        b.close()


def _drive_the_dropin(workdir):
    """What a caller of the reference's API surface does, written for this test (the contract is the four
    imports, the keyword constructor, `VideoFileClip(...).fl_image(lt.process).write_videofile(...)` and
    `get_success_ratio()` -- reference process_video.py:14-17, 28-37, 42-44, 47)."""
    import importlib
    tracker_mod = importlib.import_module("lane_tracker")
    loaders = importlib.import_module("utils")
    editor = importlib.import_module("moviepy.editor")
    assert callable(tracker_mod.bilateral_adaptive_threshold)
    K, D = loaders.load_camera_calib(os.path.join(workdir, "cam_calib.npz"))
    warp = loaders.load_warp_params(os.path.join(workdir, "warp_params.npz"))
    assert len(warp) == 6
    kwargs = dict(zip(("img_size", "warped_size"), warp[2:4]))
    kwargs.update(cam_matrix=K, dist_coeffs=D, warp_matrices=tuple(warp[:2]), mpp_conversion=tuple(warp[4:]),
                  n_fail=8, n_reset=4, n_average=2, print_frame_count=False)
    tracker = tracker_mod.LaneTracker(**kwargs)
    # moviepy hands fl_image's callable exactly one positional array per frame: the bound method itself
    clip = editor.VideoFileClip(os.path.join(workdir, "drive.npy")).fl_image(tracker.process)
    clip.write_videofile(os.path.join(workdir, "annotated.npy"), audio=False)
    ratio = tracker.get_success_ratio()
    # and a plain callable goes through the same stand-in frame by frame
    shapes = []
    second = tracker_mod.LaneTracker(**kwargs)

    def per_frame(frame):
        out = second.process(frame)
        shapes.append((frame.shape, out.shape, out.dtype))
        return out
    editor.VideoFileClip(os.path.join(workdir, "drive.npy")).fl_image(per_frame).write_videofile(
        os.path.join(workdir, "annotated_again.npy"), audio=False)
    for t in (tracker, second):
        if hasattr(t, "close"):
            t.close()
    return shapes, ratio, second.get_success_ratio()


@pytest.mark.gpu
def test_reference_style_caller_runs_against_the_dropin(tmp_path, monkeypatch):
    """dropin/ + the moviepy stand-in satisfy the call pattern of the reference's driver script."""
    from lane_tracker_amd import calib, synth, utils
    cal = calib.reference_calibration()
    utils.save_calibration_npz(tmp_path / "cam_calib.npz", tmp_path / "warp_params.npz", cal["cam_matrix"],
                               cal["dist_coeffs"], cal["warp_matrices"][0], cal["warp_matrices"][1], cal["img_size"],
                               cal["warped_size"], *cal["mpp_conversion"])
    frames = np.stack(synth.stream_lanes(6, seed=4), 0)
    np.save(tmp_path / "drive.npy", frames)
    monkeypatch.syspath_prepend(os.path.join(ROOT, "dropin", "frames_backend"))
    monkeypatch.syspath_prepend(os.path.join(ROOT, "dropin"))
    for name in ("lane_tracker", "utils", "moviepy", "moviepy.editor"):
        monkeypatch.delitem(sys.modules, name, raising=False)
    shapes, first_ratio, second_ratio = _drive_the_dropin(str(tmp_path))
    H, W = cal["img_size"][1], cal["img_size"][0]
    assert shapes == [((H, W, 3), (H, W, 3), np.dtype(np.uint8))] * 6
    assert tuple(first_ratio) == (1.0, 6, 6) and tuple(second_ratio) == (1.0, 6, 6)
    out = video.FrameSource(tmp_path / "annotated.npy")
    assert len(out) == 6 and out.size == tuple(cal["img_size"])
    got = out.read(0, 6)
    assert not np.array_equal(got, frames), "the overlay must have been drawn into the frames"
    # windowed (bound-method) and frame-by-frame (plain callable) runs give the same annotated frames
    assert np.array_equal(got, video.FrameSource(tmp_path / "annotated_again.npy").read(0, 6))
