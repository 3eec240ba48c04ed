#!/usr/bin/env python3
"""profiles/<tag>_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/prof_pmc.sh.
usage: tools/make_traffic.py gpurun_out/<pmc dir> profiles/r01_traffic.json [frames_per_launch]"""
import collections, csv, glob, json, re, sys
root, out = sys.argv[1], sys.argv[2]
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 256
MASK = ("k_undistort_rows", "k_warp_split", "k_morph_runs", "k_bilateral_tile", "k_pack_merge", "k_erode5_bits", "k_dilate5_mask",
        "k_adaptive_mean", "k_morph_ellipse", "k_merge")
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
    per[k] = {"fetch_bytes": mean(v["FETCH_SIZE"]) * 1024, "write_bytes": mean(v["WRITE_SIZE"]) * 1024, "dispatches": len(v["FETCH_SIZE"]),
              "valu_wave_insts": mean(v["SQ_INSTS_VALU"])}
fetch = sum(v["fetch_bytes"] for k, v in per.items() if k.startswith(MASK))
write = sum(v["write_bytes"] for k, v in per.items() if k.startswith(MASK))
valu = sum(v["valu_wave_insts"] for k, v in per.items() if k.startswith(MASK))
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/prof_pmc.sh); bench.py --steps 1 --warmup 1 "
                     "--streams 1 (mean per dispatch), %d frames per launch" % frames,
           "calibration": "WRITE_SIZE of k_dilate5_mask equals the %d u8 masks it writes; no x2 correction applied (the x2 rule of "
                          "MI355X_MICROARCH.md is for 16 B/lane streaming loads, none in the mask stage)" % frames,
           "frames_per_launch": frames, "mask_stage_fetch_bytes_per_launch": fetch, "mask_stage_write_bytes_per_launch": write,
           "mask_stage_traffic_bytes_per_launch": fetch + write,
           "mask_stage_valu_wave_insts_per_launch": valu,   # SQ_INSTS_VALU: wave64 VALU instructions, 64 lane-operations each
           "per_kernel": per}, open(out, "w"), indent=1)
print("mask stage: fetch %.3f GB  write %.3f GB  per %d frames" % (fetch / 1e9, write / 1e9, frames))
