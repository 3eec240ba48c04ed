#!/usr/bin/env python3
"""bench.py's stream leg by itself, N times in fresh processes: the frames/s figures of both sizes, one line per run
(box-to-box and run-to-run spread of the first-pass figures)."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 2):
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-settings", "--no-host-fed"],
                       cwd=root, capture_output=True, text=True, timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    s = json.loads(line)["stream"]
    for k in ("1280x720", "1920x1080"):
        print(i, k, {a: b for a, b in s[k].items() if "fps" in a}, flush=True)
