#!/usr/bin/env python3
"""profiles/<tag>_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/prof_round.sh.
usage: tools/make_traffic.py gpurun_out/<pmc dir> profiles/r03_traffic.json [frames_per_launch] [commit]"""
import collections, csv, glob, json, re, sys
root, out = sys.argv[1], sys.argv[2]
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 256
commit = sys.argv[4] if len(sys.argv) > 4 else "unknown"
MASK = ("k_undistort_rows", "k_warp_split", "k_morph_runs", "k_bilateral_tile", "k_bilateral_walk", "k_or4_bits", "k_pack_merge",
        "k_erode5_bits", "k_dilate5_mask", "k_dilate5_bits", "k_adaptive_mean", "k_adaptive_box_walk", "k_morph_ellipse", "k_merge")
# FETCH_SIZE reads 1/2 of the bytes of every coalesced, aligned load width (1, 2, 4, 8, 16 bytes per lane); WRITE_SIZE is exact
# (profiles/r02_fetch_calib.json, measured with tools/microbench/fetch_calib.hip on 1 GiB buffers).  Every kernel of the chain
# loads aligned pieces since round 2 (the remap kernels' unaligned 8-byte taps, factor 1.714, are gone), so the factor is 2.0
# throughout.
FETCH_FACTOR = 2.0
def fetch_factor(kernel):
    return FETCH_FACTOR

# Compulsory bytes per frame of each kernel AS DESIGNED (every input plane read once, every output written once; 1080 x 1100
# planes, 238 x 1280 undistorted rows of RGBX dwords, top-hat planes with the 1088-byte pitch, 149,600-byte bit planes) --
# the denominator of the per-kernel "measured / compulsory" column.  The algorithmic bytes of the whole stage (SURVEY 8(d):
# camera rows in + mask out = 2,101,920 B) are the denominator of traffic_over_algorithmic.
PLANE, PITCHED, BITS, UND, CAM = 1100 * 1080, 1100 * 1088, 149600, 238 * 1280 * 4, 238 * 1280 * 3
COMPULSORY = {"k_undistort_rows": (CAM, UND), "k_warp_split4": (UND, 2 * PLANE),
              "k_morph_runs2<SE29, false": (PLANE, PLANE), "k_morph_runs2<SE29, true": (2 * PLANE, PITCHED),
              "k_morph_runs2<SE55, false": (PLANE, PLANE), "k_morph_runs2<SE55, true": (2 * PLANE, PITCHED),
              "k_bilateral_walk_hv": (PITCHED, 2 * BITS), "k_adaptive_box_walk": (2 * PLANE, 2 * BITS),
              "k_merge_open5<6": (6 * BITS, 2 * BITS), "k_merge_open5<2": (2 * BITS, 2 * BITS), "k_merge_open5": (4 * BITS, 2 * BITS)}
def compulsory(kernel):
    for k, v in COMPULSORY.items():
        if kernel.startswith(k):
            return v
    return None
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] not in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_LDS_IDX_ACTIVE"):
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
              "dispatches": len(v["FETCH_SIZE"]), "valu_wave_insts": mean(v["SQ_INSTS_VALU"]),
              "lds_idx_active_cycles": mean(v["SQ_LDS_IDX_ACTIVE"])}
    c = compulsory(k)
    if c:
        per[k]["compulsory_read_bytes"], per[k]["compulsory_write_bytes"] = c[0] * frames, c[1] * frames
        per[k]["read_over_compulsory"] = round(per[k]["fetch_bytes"] / (c[0] * frames), 3)
        per[k]["write_over_compulsory"] = round(per[k]["write_bytes"] / (c[1] * frames), 3)
fetch = sum(v["fetch_bytes"] for k, v in per.items() if k.startswith(MASK))
write = sum(v["write_bytes"] for k, v in per.items() if k.startswith(MASK))
valu = sum(v["valu_wave_insts"] for k, v in per.items() if k.startswith(MASK))
lds = sum(v["lds_idx_active_cycles"] for k, v in per.items() if k.startswith(MASK))
json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/prof_round.sh); bench.py --steps 1 --warmup 1 "
                     "--streams 1 (mean per dispatch), %d frames per launch" % frames,
           "commit": commit,
           "calibration": "FETCH_SIZE x 2.0, WRITE_SIZE x 1.0: factors measured per access pattern on 1 GiB buffers, "
                          "profiles/r02_fetch_calib.json (every load of the chain is an aligned piece)",
           "mask_stage_compulsory_bytes_per_launch": sum(sum(compulsory(k)) * frames for k in per if k.startswith(MASK) and compulsory(k)),
           "alg_bytes_per_launch": 2101920 * frames,
           "traffic_over_algorithmic": round((fetch + write) / (2101920.0 * frames), 2),
           "frames_per_launch": frames, "mask_stage_fetch_bytes_per_launch": fetch, "mask_stage_write_bytes_per_launch": write,
           "mask_stage_traffic_bytes_per_launch": fetch + write,
           "mask_stage_valu_wave_insts_per_launch": valu,   # SQ_INSTS_VALU: wave64 VALU instructions, 64 lane-operations each
           "mask_stage_lds_idx_active_cycles_per_launch": lds,   # SQ_LDS_IDX_ACTIVE: cycles the CUs' LDS index units were busy, summed over the CUs
           "per_kernel": per}, open(out, "w"), indent=1)
print("mask stage: fetch %.3f GB  write %.3f GB  per %d frames" % (fetch / 1e9, write / 1e9, frames))
