#!/usr/bin/env python3
"""profiles/<tag>_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/prof_pmc.sh.
usage: tools/make_traffic.py gpurun_out/<pmc dir> profiles/r01_traffic.json [frames_per_launch]"""
import collections, csv, glob, json, re, sys
root, out = sys.argv[1], sys.argv[2]
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 256
MASK = ("k_undistort_rows", "k_warp_split", "k_morph_runs", "k_bilateral_tile", "k_bilateral_walk", "k_or4_bits", "k_pack_merge",
        "k_erode5_bits", "k_dilate5_mask", "k_dilate5_bits", "k_adaptive_mean", "k_morph_ellipse", "k_merge")
# FETCH_SIZE reads 1/2 of the bytes of every coalesced load width (1, 2, 4, 8, 16 bytes per lane) and 1/1.714 of the
# unaligned overlapping 8-byte taps of the remap kernels; WRITE_SIZE is exact (profiles/r02_fetch_calib.json, measured with
# tools/microbench/fetch_calib.hip on 1 GiB buffers).
FETCH_FACTOR_TAPS, FETCH_FACTOR = 1.714, 2.0
def fetch_factor(kernel):
    return FETCH_FACTOR_TAPS if kernel.startswith(("k_undistort_rows", "k_warp_split")) else FETCH_FACTOR
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU"):
            continue
        m = re.search(r"(k_[a-z0-9_]+)(<[^>]*>)?", r["Kernel_Name"])
        if m:
            k = m.group(1) + (m.group(2) or "").replace("lt::(anonymous namespace)::", "")
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
per = {}
for k, v in acc.items():
    # counters are KB per dispatch; with --streams 1 one dispatch of a kernel covers the whole batch
    mean = lambda x: sum(x) / len(x) if x else 0.0
    per[k] = {"fetch_bytes_raw": mean(v["FETCH_SIZE"]) * 1024, "fetch_factor": fetch_factor(k),
              "fetch_bytes": mean(v["FETCH_SIZE"]) * 1024 * fetch_factor(k), "write_bytes": mean(v["WRITE_SIZE"]) * 1024,
              "dispatches": len(v["FETCH_SIZE"]), "valu_wave_insts": mean(v["SQ_INSTS_VALU"])}
fetch = sum(v["fetch_bytes"] for k, v in per.items() if k.startswith(MASK))
write = sum(v["write_bytes"] for k, v in per.items() if k.startswith(MASK))
valu = sum(v["valu_wave_insts"] for k, v in per.items() if k.startswith(MASK))
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/prof_pmc.sh); bench.py --steps 1 --warmup 1 "
                     "--streams 1 (mean per dispatch), %d frames per launch" % frames,
           "calibration": "FETCH_SIZE x 2.0 (x 1.714 for the unaligned 8-byte taps of the two remap kernels), WRITE_SIZE x 1.0: factors "
                          "measured per access pattern on 1 GiB buffers, profiles/r02_fetch_calib.json",
           "frames_per_launch": frames, "mask_stage_fetch_bytes_per_launch": fetch, "mask_stage_write_bytes_per_launch": write,
           "mask_stage_traffic_bytes_per_launch": fetch + write,
           "mask_stage_valu_wave_insts_per_launch": valu,   # SQ_INSTS_VALU: wave64 VALU instructions, 64 lane-operations each
           "per_kernel": per}, open(out, "w"), indent=1)
print("mask stage: fetch %.3f GB  write %.3f GB  per %d frames" % (fetch / 1e9, write / 1e9, frames))
