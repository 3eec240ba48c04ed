"""Soak test of the drop-in (reference process_video.py:41-44: one long-lived tracker per video).  A child
process -- started before anything here touches the GPU, KILLED (never re-exec'd) on its deadline -- runs tools/soak.py:
`process_stream(annotate=True)` over windows with outages for 60 s (LT_SOAK_SECONDS=600: the ten-minute form), two tracker close /
reopen cycles on the way.  After the first sixth of the run everything the process holds must be flat -- resident memory,
page-locked bytes, the device cache -- nothing may go back to the driver (an eviction is a wipe is half-rate downloads: the bug of
round 5's long-lived annotated stream, which this test would have failed), the copy threads' queue must be empty between passes,
and the last quarter of the run must not be markedly slower than the first."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_annotated_stream_soak_is_flat():
    seconds = float(os.environ.get("LT_SOAK_SECONDS", "60"))
    interval = 5.0 if seconds <= 120 else 30.0
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "soak.py"), "--seconds", str(seconds), "--interval", str(interval)],
                         cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    try:
        out, err = p.communicate(timeout=seconds * 2 + 240)
    except subprocess.TimeoutExpired:
        p.kill()
        out, err = p.communicate()
        pytest.fail("the soak run did not finish within its deadline:\n" + out[-1500:] + err[-1500:])
    assert p.returncode == 0, out[-1500:] + err[-3000:]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    open(os.path.join(ROOT, "gpurun_out", "soak_last.jsonl"), "w").write(out)             # (kept for the notes: profiles/r06_soak.jsonl)
    lines = [json.loads(l) for l in out.splitlines() if l.startswith("{")]
    verdict = [json.loads(l[len("VERDICT "):]) for l in out.splitlines() if l.startswith("VERDICT ")]
    assert verdict and verdict[0]["reopened"] == 2, out[-1500:]
    samples = [l for l in lines if "fps" in l]
    assert len(samples) >= 6, samples
    warm = max(2, len(samples) // 6)
    steady = samples[warm:]
    first = steady[0]
    held = lambda s: s["cache_kept"] + s["cache_live"]
    for s in steady:
        assert s["queued_pieces"] == 0 and s["pending_pieces"] == 0, s                     # nothing left in the copy threads' queue between passes
        assert s["staging_bytes"] <= first["staging_bytes"] + (96 << 20), (first, s)       # page-locked staging: a pool of a few blocks, at most three more after a reopen
        assert s["pinned_pool_outstanding"] + s["pinned_pool_idle"] <= first["pinned_pool_outstanding"] + first["pinned_pool_idle"], (first, s)
        assert s["evicted_bytes"] == first["evicted_bytes"], (first, s)                    # the device cache gives NOTHING back to the driver (no wipe, no half-rate downloads)
        assert held(s) <= held(first) * 1.02 + (64 << 20), (first, s)                      # ... and what the process holds on the device does not grow (a reopened
        assert s["cache_misses"] - first["cache_misses"] <= 64, (first, s)                 #     tracker finds its slot buffers in the cache; a few small buffers of other sizes may be new)
        assert s["rss"] <= first["rss"] * 1.03 + (64 << 20), (first["rss"], s["rss"])      # resident memory flat (3 % + 64 MB of allocator noise)
        assert s["threads"] <= first["threads"] + 1, (first, s)
        assert s["success_ratio"] > 0.5
    last_third = [s for s in steady if s["t"] >= samples[-1]["t"] * 0.72]                  # behind the second reopen: strictly flat
    assert len({s["staging_bytes"] for s in last_third}) == 1 and len({held(s) for s in last_third}) == 1, last_third
    # Speed: resources are what this test pins (an eviction, a leak, a queue that does not drain -- the causes); the rate itself swings
    # +-15 % from one five-second interval to the next on the shared hosts (NOTES_r05 D.10: it follows the copy threads' own speed),
    # so the check on it is coarse: the last quarter of the run must not be a quarter slower than the first.
    rates = [s["fps"] for s in steady]
    k = max(1, len(rates) // 4)
    early, late = sorted(rates[:k])[k // 2], sorted(rates[-k:])[k // 2]
    tol = 0.25 if seconds <= 120 else 0.15
    assert late >= early * (1.0 - tol), ("the stream slowed down", rates)
