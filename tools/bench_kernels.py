#!/usr/bin/env python3
"""Print value and the per-stage times of one bench.py run (A/B helper): python tools/bench_kernels.py [bench args]"""
import json
import subprocess
import sys

out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--steps", "10", "--warmup", "3"] + sys.argv[1:],
                     capture_output=True, text=True)
line = [l for l in out.stdout.splitlines() if l.startswith("{")]
if not line:
    print(out.stdout[-2000:], out.stderr[-2000:])
    sys.exit(1)
d = json.loads(line[-1])
k = d.get("kernels_ms_per_step", {})
print(f"value {d['value']:.0f} fps  ms_per_step {d['ms_per_step']:.4f}  sum {sum(k.values()):.4f}  " +
      "  ".join(f"{a} {b:.4f}" for a, b in k.items()))
