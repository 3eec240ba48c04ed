"""Frame source / sink either side of the tracker.

The reference drives `LaneTracker.process` through moviepy (`process_video.py:41-44`: VideoFileClip ->
fl_image -> write_videofile).  moviepy and ffmpeg do not exist in this image, so video containers are out
of scope; what a lane-tracking run needs is an ordered sequence of RGB u8 frames in and the annotated
frames out.  Supported on both sides:

  * a directory of images (PNG/JPEG/BMP/PPM, sorted by name; Pillow),
  * a `.npy` array of shape (n, H, W, 3) (memory-mapped, so a long clip never has to fit in RAM),
  * a headerless raw RGB24 file (`.rgb` / `.raw`; what `ffmpeg -pix_fmt rgb24 -f rawvideo` writes and reads).

`process_frames()` feeds the tracker in windows through `LaneTracker.process_stream` (the stream pipeline:
masks of the whole window batched ahead on the GPU, state machine trailing), which gives exactly the
frames `process()` would return one by one.  `VideoFileClip` is the small part of moviepy's interface that
`process_video.py` uses, on top of the same sources and sinks.

CLI:  python -m lane_tracker_amd.video IN OUT [--cam cam_calib.p] [--warp warp_params.p] [--size WxH] ...
"""
import argparse
import os
import sys
import time

import numpy as np

_IMAGE_EXT = (".png", ".jpg", ".jpeg", ".bmp", ".ppm")


_VIDEO_EXT = (".mp4", ".avi", ".mov", ".mkv", ".webm")


def _is_raw(path):
    return str(path).lower().endswith((".rgb", ".raw"))


def resolve_clip_path(path, must_exist=True):
    """A video file name as process_video.py spells it ('clip.mp4') -> the frame sequence standing in for
    it: 'clip.mp4' itself if it is a directory, else 'clip/' , 'clip.npy' or 'clip.rgb' next to it.  For
    outputs (must_exist=False) a video extension is dropped and the result is a directory of PNGs."""
    path = str(path)
    stem, ext = os.path.splitext(path)
    if ext.lower() not in _VIDEO_EXT:
        return path
    if not must_exist:
        return stem
    for cand in (path, stem, stem + ".npy", stem + ".rgb"):
        if os.path.isdir(cand) or (cand != path and os.path.isfile(cand)):
            return cand
    raise FileNotFoundError(f"{path}: video containers cannot be decoded here (no ffmpeg); put the frames in "
                            f"{stem}/ (images), {stem}.npy or {stem}.rgb")


class FrameSource:
    """Ordered RGB u8 frames.  `size` = (width, height) is required for raw files only."""

    def __init__(self, path, size=None):
        self.path = str(path)
        self._files = None
        self._array = None
        if os.path.isdir(self.path):
            self._files = sorted(os.path.join(self.path, f) for f in os.listdir(self.path)
                                 if f.lower().endswith(_IMAGE_EXT))
            if not self._files:
                raise ValueError(f"no image files in {self.path}")
            first = self._read_image(self._files[0])
            self.height, self.width = first.shape[:2]
            self._n = len(self._files)
        elif self.path.lower().endswith(".npy"):
            self._array = np.load(self.path, mmap_mode="r")
            if self._array.ndim != 4 or self._array.shape[3] != 3 or self._array.dtype != np.uint8:
                raise ValueError("expected a uint8 array of shape (n, H, W, 3)")
            self._n, self.height, self.width = self._array.shape[:3]
        elif _is_raw(self.path):
            if size is None:
                raise ValueError("raw RGB24 input needs size=(width, height)")
            self.width, self.height = int(size[0]), int(size[1])
            fb = self.width * self.height * 3
            nbytes = os.path.getsize(self.path)
            if nbytes % fb:
                raise ValueError(f"{self.path}: {nbytes} bytes is not a whole number of {self.width}x{self.height} frames")
            self._n = nbytes // fb
            self._array = np.memmap(self.path, np.uint8, "r", shape=(self._n, self.height, self.width, 3)) if self._n \
                else np.zeros((0, self.height, self.width, 3), np.uint8)
        else:
            raise ValueError(f"unsupported frame source {self.path!r} (directory of images, .npy, .rgb/.raw)")
        if size is not None and (self.width, self.height) != (int(size[0]), int(size[1])):
            raise ValueError(f"frames are {self.width}x{self.height}, expected {size[0]}x{size[1]}")

    @staticmethod
    def _buffer(shape):
        """Frames are read straight into page-locked memory when the library is there: the tracker's uploads then run
        at the PCIe rate instead of the pageable-copy rate."""
        try:
            from ._native import pinned_empty
            return pinned_empty(shape)
        except Exception:
            return np.empty(shape, np.uint8)

    @staticmethod
    def _read_image(path):
        from PIL import Image
        with Image.open(path) as im:
            return np.asarray(im.convert("RGB"), np.uint8)

    def __len__(self):
        return self._n

    @property
    def size(self):
        return (self.width, self.height)

    def read(self, start, stop):
        """Frames [start, stop) as one contiguous (n, H, W, 3) array."""
        start, stop = max(0, start), min(self._n, stop)
        out = self._buffer((max(stop - start, 0), self.height, self.width, 3))
        if self._files is None:
            out[...] = self._array[start:stop]
            return out
        for i in range(start, stop):
            img = self._read_image(self._files[i])
            if img.shape != out.shape[1:]:
                raise ValueError(f"{self._files[i]}: {img.shape[1]}x{img.shape[0]}, expected {self.width}x{self.height}")
            out[i - start] = img
        return out

    def __iter__(self):
        for i in range(self._n):
            yield self.read(i, i + 1)[0]


class FrameSink:
    """Where processed frames go.  Directories get `frame_000000.png`, ...; `.npy` needs `n` up front."""

    def __init__(self, path, size, n=None):
        self.path = str(path)
        self.width, self.height = int(size[0]), int(size[1])
        self.count = 0
        self._raw = None
        self._array = None
        if self.path.lower().endswith(".npy"):
            if n is None:
                raise ValueError(".npy output needs the number of frames")
            self._array = np.lib.format.open_memmap(self.path, "w+", np.uint8, (int(n), self.height, self.width, 3))
        elif _is_raw(self.path):
            self._raw = open(self.path, "wb")
        else:
            os.makedirs(self.path, exist_ok=True)

    def write(self, frames):
        frames = np.asarray(frames, np.uint8)
        if frames.ndim == 3:
            frames = frames[None]
        if frames.shape[1:] != (self.height, self.width, 3):
            raise ValueError(f"sink takes {self.width}x{self.height} RGB frames, got {frames.shape[1:]}")
        if self._array is not None:
            self._array[self.count:self.count + len(frames)] = frames
        elif self._raw is not None:
            self._raw.write(np.ascontiguousarray(frames).tobytes())
        else:
            from PIL import Image
            for k, f in enumerate(frames):
                Image.fromarray(f).save(os.path.join(self.path, "frame_{:06d}.png".format(self.count + k)))
        self.count += len(frames)

    def close(self):
        if self._raw is not None:
            self._raw.close()
            self._raw = None
        if self._array is not None:
            self._array.flush()
            self._array = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def process_frames(tracker, source, sink=None, window=64, **process_kwargs):
    """Run every frame of `source` through the tracker, in order, `window` frames per GPU batch; write the
    annotated frames to `sink` if there is one.  Returns (frames, seconds)."""
    t0 = time.perf_counter()
    n = len(source)
    windows = (source.read(start, start + window) for start in range(0, n, window))
    # process_stream: the uploads and masks of window k+1 run while the searches of window k drain
    for out in tracker.process_stream(windows, annotate=sink is not None, **process_kwargs):
        if sink is not None:
            sink.write(np.stack(out, 0))
    return n, time.perf_counter() - t0


class VideoFileClip:
    """The slice of `moviepy.editor.VideoFileClip` that process_video.py touches (`:41-44`), over frame
    sequences: `clip.fl_image(fn)` returns a lazy clip, `write_videofile(path)` evaluates it.  When `fn`
    is the bound `process` of a `lane_tracker_amd` LaneTracker, evaluation goes through `process_stream`
    windows instead of one call per frame (same frames, higher throughput)."""

    def __init__(self, filename, size=None, fps=25.0, _fn=None, _source=None):
        self.filename = filename
        self.fps = fps
        self._source = _source if _source is not None else FrameSource(resolve_clip_path(filename), size)
        self._fn = _fn
        self.size = self._source.size

    def fl_image(self, image_func):
        return VideoFileClip(self.filename, fps=self.fps, _fn=image_func, _source=self._source)

    def iter_frames(self):
        for f in self._source:
            yield f if self._fn is None else self._fn(f)

    def write_videofile(self, filename, audio=False, window=64, **_ignored):
        from .lane_tracker import LaneTracker
        fn = self._fn
        tracker = getattr(fn, "__self__", None)
        n = len(self._source)
        with FrameSink(resolve_clip_path(filename, must_exist=False), self.size, n=n) as sink:
            if isinstance(tracker, LaneTracker) and getattr(fn, "__func__", None) is LaneTracker.process:
                process_frames(tracker, self._source, sink, window=window)
            else:
                for f in self.iter_frames():
                    sink.write(f)
        return filename


def _parse_size(text):
    w, h = text.lower().split("x")
    return int(w), int(h)


def main(argv=None):
    ap = argparse.ArgumentParser(description="Lane tracking over a frame sequence (process_video.py without moviepy)")
    ap.add_argument("input", help="directory of images, .npy (n,H,W,3) or raw .rgb")
    ap.add_argument("output", help="directory (PNG), .npy or raw .rgb; '-' to discard the frames")
    ap.add_argument("--cam", default="cam_calib.p", help="camera calibration (.p pickle or .npz)")
    ap.add_argument("--warp", default="warp_params.p", help="warp parameters (.p pickle or .npz)")
    ap.add_argument("--size", type=_parse_size, default=None, help="WxH of raw input frames")
    ap.add_argument("--window", type=int, default=64, help="frames per GPU batch")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--frame-count", action="store_true", help="print the frame number onto each image")
    ap.add_argument("--settings", choices=("default", "demo1", "demo2", "demo3"), default="default",
                    help="parameter set of tracker_settings.md (process() keywords + validity limits)")
    a = ap.parse_args(argv)
    from .lane_tracker import LaneTracker
    from .utils import load_camera_calib, load_warp_params
    cam_matrix, dist_coeffs = load_camera_calib(a.cam)
    M, Minv, image_wh, warped_wh, mppv, mpph = load_warp_params(a.warp)
    src = FrameSource(a.input, a.size or image_wh)
    lt = LaneTracker(img_size=image_wh, warped_size=warped_wh, cam_matrix=cam_matrix, dist_coeffs=dist_coeffs,
                     warp_matrices=(M, Minv), mpp_conversion=(mppv, mpph), n_fail=8, n_reset=4, n_average=2,
                     print_frame_count=a.frame_count, device=a.device)
    try:
        sink = None if a.output == "-" else FrameSink(a.output, src.size, n=len(src))
        kw = {}
        if a.settings != "default":
            from . import settings
            kw = settings.apply(lt, settings.DEMOS[a.settings])
        n, dt = process_frames(lt, src, sink, window=a.window, **kw)
        if sink is not None:
            sink.close()
        ratio, success, total = lt.get_success_ratio() if lt.counter else (0.0, 0, 0)
        print("Frames: {}  ({:.1f} frames/s including I/O)".format(n, n / dt if dt > 0 else 0.0))
        print("Success ratio: ", ratio)
        print("Success absolute: ", success)
        print("Total frames: ", total)
    finally:
        lt.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
