#!/usr/bin/env python3
"""Text table + summary JSON of tools/microbench/valu_issue's output.
usage: valu_table.py gpurun_out/mb/valu_issue.json [summary.json]

Throughput is taken from the hipEvent wall time (wave-instructions per ns per SIMD with every SIMD of the chip
busy); the shader clock during the run is estimated from the one-wave-per-SIMD rows, where the s_memtime tick count
of a wave and the wall time describe the same interval (tick = shader cycle, MI355X_MICROARCH.md)."""
import json, re, sys
d = json.load(open(sys.argv[1]))
Ws = ["1", "2", "3", "4", "5", "6", "8"]
ops = d["ops"]
ghz = sorted(r["1"]["wave_insts_per_ns_per_simd"] / r["1"]["rate"] for r in ops.values() if r["1"]["rate"] > 0)
clock = ghz[len(ghz) // 2]
print("%s, %d CUs, %d instructions per wave; shader clock during the run ~%.2f GHz (median of s_memtime ticks / wall ns, W=1)"
      % (d["device"], d["cus"], d["insts_per_wave"], clock))
print("columns: wave64 instructions per ns per SIMD at W waves per SIMD (all %d SIMDs busy); last: cycles per instruction at W=8" % (d["cus"] * 4))
print("%-36s" % "instruction / chain" + "".join("%8s" % w for w in Ws) + "   cyc/inst")
summary = {}
for name, r in ops.items():
    m = re.search(r"_x(\d+)insts", name)
    mult = int(m.group(1)) if m else 1
    best = max(r[w]["wave_insts_per_ns_per_simd"] for w in Ws) * mult
    cyc = clock / (r["8"]["wave_insts_per_ns_per_simd"] * mult)
    summary[name] = {"insts_per_ns_per_simd_W8": round(r["8"]["wave_insts_per_ns_per_simd"] * mult, 4),
                     "best_insts_per_ns_per_simd": round(best, 4), "cycles_per_inst_W8": round(cyc, 2),
                     "insts_per_ns_per_simd_W1": round(r["1"]["wave_insts_per_ns_per_simd"] * mult, 4)}
    print("%-36s" % name + "".join("%8.3f" % (r[w]["wave_insts_per_ns_per_simd"] * mult) for w in Ws) + "   %6.2f" % cyc)
four = [v["cycles_per_inst_W8"] for k, v in summary.items() if k.split("/")[0] in
        ("pk_minimum3_f16", "pk_maximum3_f16", "pk_mad_u16", "pk_add_u16", "pk_sub_i16", "pk_max_u16", "perm_b32", "mov_dpp_row_shr1",
         "mov_dpp_wave_shr1", "and_or_b32", "sad_u8", "cmp_gt_i16_vcc", "writelane_b32")]
two = [v["cycles_per_inst_W8"] for k, v in summary.items() if k.split("/")[0] in ("add_u32", "xor_b32", "mov_b32", "fma_f32")]
res = {"device": d["device"], "clock_ghz_during_run": round(clock, 3),
       "packed16_dpp_perm_cmp_cycles_per_inst": round(sum(four) / len(four), 2),
       "plain_vop2_32bit_cycles_per_inst": round(sum(two) / len(two), 2),
       "peak_wave_insts_per_cycle_per_simd": round(1.0 / (sum(four) / len(four)), 4),
       "reading": "the packed 16-bit (VOP3P), VOP3, DPP, v_perm, v_cmp and v_writelane instructions the mask-stage kernels are made of "
                  "issue one wave64 instruction per ~4 cycles per SIMD (16 lanes/clk): the 39.3 T lane-ops/s ceiling, not 78.6 T; "
                  "plain 32-bit VOP2/VOP1 (v_add_u32, v_xor_b32, v_mov_b32) and v_fma_f32 issue in ~2 cycles (32 lanes/clk)",
       "per_instruction": summary}
print(json.dumps({k: v for k, v in res.items() if k != "per_instruction"}, indent=1))
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
