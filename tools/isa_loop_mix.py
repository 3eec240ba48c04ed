#!/usr/bin/env python3
"""Instruction mix of the hottest loop of every kernel in a gfx950 assembly listing.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ilane_tracker_amd/csrc -S --cuda-device-only -o /tmp/k.s <file.hip>
    python tools/isa_loop_mix.py /tmp/k.s [name-substring]

For each kernel it prints the VALU / LDS / vector-memory / scalar instruction counts of the longest backward-branch
loop, and the most frequent opcodes: the counts DESIGN.md quotes next to SQ_INSTS_VALU.
"""
import collections
import re
import subprocess
import sys


def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        return n


def main():
    lines = open(sys.argv[1]).read().split("\n")
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    starts = [(i, m.group(1)) for i, l in enumerate(lines) if (m := re.match(r"^(_Z\w+):\s*; @", l))]
    for k, (s, name) in enumerate(starts):
        e = next((i for i in range(s, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end")), len(lines))
        pretty = demangle(name)
        if want not in pretty:
            continue
        body = lines[s:e]
        labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
        loops = []
        for i, l in enumerate(body):
            m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                loops.append((labels[m.group(1)], i))
        if not loops:
            continue
        a, b = max(loops, key=lambda t: t[1] - t[0])
        c = collections.Counter()
        for l in body[a:b + 1]:
            l = l.strip()
            if not l or l[0] in ";." or l.endswith(":"):
                continue
            c[l.split()[0]] += 1
        cls = lambda p: sum(v for k_, v in c.items() if k_.startswith(p))
        print(pretty[:150])
        print(f"  loop of {b - a + 1} lines: VALU {cls('v_')}  DS {cls('ds_')}  VMEM {cls('buffer_') + cls('global_') + cls('flat_')}  SALU {cls('s_')}")
        print("  " + ", ".join(f"{k_} {v}" for k_, v in sorted(c.items(), key=lambda t: -t[1])[:18]))


if __name__ == "__main__":
    main()
