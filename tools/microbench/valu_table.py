#!/usr/bin/env python3
"""Text table of tools/microbench/valu_issue's JSON: wave-instructions per cycle per SIMD by waves per SIMD."""
import json, sys
d = json.load(open(sys.argv[1]))
Ws = ["1", "2", "3", "4", "5", "6", "8"]
print("%s, %d CUs, clock attribute %d kHz, %d instructions per wave" % (d["device"], d["cus"], d["clock_rate_khz"], d["insts_per_wave"]))
print("rate = wave64 instructions per shader cycle per SIMD (0.25 = 16 lanes/clk, 0.5 = 32 lanes/clk); columns: waves per SIMD")
print("%-34s" % "instruction / chain" + "".join("%8s" % w for w in Ws) + "   GHz(wall, W=8)")
for name, r in d["ops"].items():
    ghz = r["8"]["wave_insts_per_ns_per_simd"] / r["8"]["rate"] if r["8"]["rate"] else 0.0
    print("%-34s" % name + "".join("%8.3f" % r[w]["rate"] for w in Ws) + "   %.2f" % ghz)
