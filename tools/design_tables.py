#!/usr/bin/env python3
"""Regenerate the measured tables of DESIGN.md from a kept bench line, so that the document and the profile file it cites cannot
disagree:

    python tools/design_tables.py [profiles/r06_bench.json] [--check]

Everything between `<!-- bench:NAME -->` and `<!-- /bench:NAME -->` in DESIGN.md is replaced by the table NAME built from the
JSON (one bench.py line, or a list of lines: then every figure is shown as min - max over the lines).  --check: exit 1 if
DESIGN.md would change (tests/test_design_tables.py runs this)."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path):
    txt = open(path).read().strip()
    try:
        d = json.loads(txt)
    except json.JSONDecodeError:
        d = [json.loads(l) for l in txt.split("\n") if l.startswith("{")]
    return d if isinstance(d, list) else [d]


def k(v):
    """frames/s as '12.3 k'"""
    return "%.1f k" % (v / 1e3) if v >= 1000 else "%.0f" % v


def span(lines, get, fmt=k):
    vals = []
    for d in lines:
        try:
            v = get(d)
        except (KeyError, TypeError, IndexError):
            continue
        if v is not None:
            vals.append(v)
    if not vals:
        return "—"
    lo, hi = min(vals), max(vals)
    return fmt(lo) if fmt(lo) == fmt(hi) else "%s – %s" % (fmt(lo), fmt(hi))


def tables(lines):
    ms = lambda v: "%.2f ms" % v
    ms1 = lambda v: "%.0f ms" % v
    pct = lambda v: "%.1f %%" % (100 * v)
    t = {}
    t["status"] = "\n".join([
        "| | |",
        "|---|---|",
        "| batch of 256 HBM-resident 1280×720 frames, mask chain + sliding-window search + fit, every step on the same slots (`value`) | **%s frames/s** (%s with steps alternating between two resident copies) |"
        % (span(lines, lambda d: d["value"]), span(lines, lambda d: d["overlapped_batches_frames_per_s"])),
        "| warp + threshold stage | %s per 256 frames = %s of the 8 TB/s HBM roofline |"
        % (span(lines, lambda d: d["roofline"]["stage_ms_per_launch"], ms), span(lines, lambda d: d["roofline"]["frac"], pct)),
        "| CPU port beside it (threads = the cgroup's CPUs) | %s frames/s on %s threads, %s on one |"
        % (span(lines, lambda d: d["cpu_baseline"]["value"], lambda v: "%.0f" % v), span(lines, lambda d: d["cpu_baseline"]["cores"], lambda v: "%d" % v),
           span(lines, lambda d: d["cpu_baseline"]["single_thread_frames_per_s"], lambda v: "%.0f" % v)),
    ])
    rows = ["| | 1280×720 | 1920×1080 (config 5) |", "|---|---|---|"]

    def row(label, get, fmt=k):
        rows.append("| %s | %s | %s |" % (label, span(lines, lambda d: get(d["stream"]["1280x720"]), fmt), span(lines, lambda d: get(d["stream"]["1920x1080"]), fmt)))
    row("`process()`, one frame per call, annotated frame back", lambda s: s["process_fps"])
    row("`process_batch`, 256 frames per call", lambda s: s["process_batch_fps"])
    row("`process_batch`, annotated", lambda s: s["process_batch_annotated_fps"])
    row("`process_stream`, later passes of the long-lived tracker (4096 frames each): median of three", lambda s: s["process_stream_fps"])
    row("… min / max of the three", lambda s: s["process_stream_passes"]["min"])
    row("… ", lambda s: s["process_stream_passes"]["max"])
    row("`process_stream`, first pass of a fresh tracker after `warm()`: median of three (min – max)",
        lambda s: s["process_stream_first_pass"]["frames_per_s"]["median"])
    row("… min", lambda s: s["process_stream_first_pass"]["frames_per_s"]["min"])
    row("… max", lambda s: s["process_stream_first_pass"]["frames_per_s"]["max"])
    row("… time to the first window", lambda s: s["process_stream_first_pass"]["time_to_first_window_ms"]["median"], ms1)
    row("… `warm()` itself (median; the first tracker of a process: max)", lambda s: s["process_stream_first_pass"]["warm_ms"]["median"], ms1)
    row("… a fresh tracker that was NOT warmed", lambda s: s["process_stream_first_pass"]["not_warmed"]["frames_per_s"])
    row("`process_stream`, annotated, later passes of the long-lived tracker: median of three", lambda s: s["process_stream_annotated_fps"])
    row("… min", lambda s: s["process_stream_annotated_passes"]["min"])
    row("… max", lambda s: s["process_stream_annotated_passes"]["max"])
    row("… copy threads' busy share; bytes the device cache gave back to the driver meanwhile", lambda s: s["copy_threads_busy_share_annotated_stream"], lambda v: "%.2f" % v)
    row("… ", lambda s: s["device_cache_during_annotated_passes"]["evicted_bytes"], lambda v: "%d B" % v)
    row("annotated, first pass after `warm()`: median of three", lambda s: s["process_stream_annotated_first_pass"]["frames_per_s"]["median"])
    row("… min", lambda s: s["process_stream_annotated_first_pass"]["frames_per_s"]["min"])
    row("… max", lambda s: s["process_stream_annotated_first_pass"]["frames_per_s"]["max"])
    row("… time to the first window", lambda s: s["process_stream_annotated_first_pass"]["time_to_first_window_ms"]["median"], ms1)
    row("… `warm()` itself: median", lambda s: s["process_stream_annotated_first_pass"]["warm_ms"]["median"], ms1)
    row("… `warm()` of the first tracker of the process (touches the output pool)", lambda s: s["process_stream_annotated_first_pass"]["warm_ms"]["max"], ms1)
    row("… a fresh tracker that was NOT warmed", lambda s: s["process_stream_annotated_first_pass"]["not_warmed"]["frames_per_s"])
    row("`process_stream(annotate=\"inplace\")`: drawn into the caller's windows (not in the reference)", lambda s: s["process_stream_annotated_inplace_fps"])
    row("with outages (four of 16 frames per window)", lambda s: s["process_stream_outages_fps"])
    rows.append("| Demo 1 settings (`mask_noise`) | %s | — |" % span(lines, lambda d: d["stream"]["1280x720"]["process_stream_demo1_fps"]))
    t["stream"] = "\n".join(rows)
    prow = ["| | 1280×720 | 1920×1080 |", "|---|---|---|"]
    prow.append("| `process()`: median of three stretches of 0.15 s, frames/s | %s | %s |" % (
        span(lines, lambda d: d["stream"]["1280x720"]["process_fps"]), span(lines, lambda d: d["stream"]["1920x1080"]["process_fps"])))
    prow.append("| … slowest / fastest stretch | %s / %s | %s / %s |" % (
        span(lines, lambda d: min(d["stream"]["1280x720"]["process_fps_stretches"])), span(lines, lambda d: max(d["stream"]["1280x720"]["process_fps_stretches"])),
        span(lines, lambda d: min(d["stream"]["1920x1080"]["process_fps_stretches"])), span(lines, lambda d: max(d["stream"]["1920x1080"]["process_fps_stretches"]))))
    usf = lambda v: "%.0f µs" % v
    prow.append("| … the frame itself (process_fps is frames per wall time, the MEAN frame): median / p90 / p99 | %s / %s / %s | %s / %s / %s |" % tuple(
        span(lines, lambda d, z=z, q=q: d["stream"][z]["process_frame_us"][q], usf) for z in ("1280x720", "1920x1080") for q in ("median", "p90", "p99")))
    us = lambda v: "%.0f µs" % (v * 1e3)
    prow.append("| BASELINE config 1: `tests/golden/photo_test4.png` through `process()` with its defaults (two tries, both rejected, as in the reference's own run): median of 200 calls | %s (%s frames/s) | — |" % (
        span(lines, lambda d: d["config1"]["process_defaults"]["median_ms"], us), span(lines, lambda d: d["config1"]["process_defaults"]["frames_per_s"])))
    prow.append("| … the first frame of a video under the Demo 1 limits (first try valid, lane drawn) | %s | — |" % span(lines, lambda d: d["config1"]["process_demo1_first_frame"]["median_ms"], us))
    prow.append("| … the CPU port's two tries of the same frame on ONE thread (no drawing) | %s | — |" % span(lines, lambda d: d["config1"]["cpu_port_one_thread"]["median_ms"], lambda v: "%.1f ms" % v))
    t["process"] = "\n".join(prow)
    srows = ["| | frames/s | mask stage ms | threshold + merge ms |", "|---|---|---|---|"]
    for key, label in (("process_defaults", "`process()` defaults"), ("demo1", "Demo 1 (`mask_noise`)"), ("demo2", "Demo 2 (k_r = 20)"),
                       ("demo3", "Demo 3 (`mask_noise`)"), ("second_try", "second try (`'neighborhood'`, no top-hats)")):
        srows.append("| %s | %s | %s | %s |" % (label, span(lines, lambda d: d["settings"][key]["frames_per_s"]),
                                               span(lines, lambda d: d["settings"][key]["mask_stage_ms"], lambda v: "%.2f" % v),
                                               span(lines, lambda d: d["settings"][key]["threshold_ms"], lambda v: "%.2f" % v)))
    t["settings"] = "\n".join(srows)
    krows = ["| kernel (stage name) | ms per 256 frames |", "|---|---|"]
    names = []
    for d in lines:
        for n in d.get("kernels_ms_per_step", {}):
            if n not in names:
                names.append(n)
    for n in names:
        krows.append("| `%s` | %s |" % (n, span(lines, lambda d: d["kernels_ms_per_step"][n], lambda v: "%.3f" % v)))
    t["kernels"] = "\n".join(krows)
    return t


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    path = args[0] if args else os.path.join(ROOT, "profiles", "r06_bench.json")
    lines = load(path)
    t = tables(lines)
    design = os.path.join(ROOT, "DESIGN.md")
    txt = open(design).read()
    new = txt
    for name, body in t.items():
        pat = re.compile(r"(<!-- bench:%s -->)\n.*?(<!-- /bench:%s -->)" % (name, name), re.S)
        new = pat.sub(lambda m: m.group(1) + "\n" + body + "\n" + m.group(2), new)
    if "--check" in sys.argv:
        if new != txt:
            print("DESIGN.md is out of date with", os.path.relpath(path, ROOT))
            return 1
        return 0
    open(design, "w").write(new)
    print("DESIGN.md tables rebuilt from", os.path.relpath(path, ROOT), "(%d bench line(s))" % len(lines))
    return 0


if __name__ == "__main__":
    sys.exit(main())
