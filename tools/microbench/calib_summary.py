#!/usr/bin/env python3
"""Correction factors for FETCH_SIZE / WRITE_SIZE from the counter passes of tools/microbench/fetch_calib.
usage: calib_summary.py gpurun_out/mb  -> JSON on stdout"""
import csv, glob, json, re, sys
root = sys.argv[1]
G = float(1 << 30)
TRUE = {  # kernel-name fragment -> (counter, true bytes, what it stands for in the mask stage)
    "k_read<unsigned char>": ("FETCH_SIZE", G, "coalesced byte loads (top-hat source rows)"),
    "k_read<unsigned short>": ("FETCH_SIZE", G, "coalesced 2-byte loads"),
    "k_read<unsigned int>": ("FETCH_SIZE", G, "coalesced dword loads (band staging, minuend)"),
    "k_read<unsigned long>": ("FETCH_SIZE", G, "coalesced 8-byte loads (bit-plane words)"),
    "k_read<u128>": ("FETCH_SIZE", G, "coalesced 16-byte loads (the guide's x2 case)"),
    "k_read_tap8": ("FETCH_SIZE", G, "unaligned overlapping 8-byte taps at 3-byte pitch (undistort / warp)"),
    "k_read_rows_dword": ("FETCH_SIZE", float((1 << 30) // 1080 * 1080), "row-wise dword loads, one row per wave (threshold staging)"),
    "k_write<unsigned char>": ("WRITE_SIZE", G, "coalesced byte stores"),
    "k_write<unsigned int>": ("WRITE_SIZE", G, "coalesced dword stores (planes)"),
    "k_write<unsigned long>": ("WRITE_SIZE", G, "coalesced 8-byte stores"),
    "k_write<u128>": ("WRITE_SIZE", G, "coalesced 16-byte stores"),
    "k_write_lane0_u64": ("WRITE_SIZE", G / 8, "one 8-byte store per wave (ballot words)"),
}
vals = {}
for p in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        vals.setdefault((r["Kernel_Name"], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
out = {"unit_note": "counter values are KiB per dispatch (x1024 = bytes); factor = true bytes / reported bytes", "patterns": {}}
for frag, (ctr, true_b, what) in TRUE.items():
    for (kn, cn), v in vals.items():
        if cn == ctr and frag in kn:
            rep = sum(v) / len(v) * 1024.0
            out["patterns"][frag] = {"counter": ctr, "true_bytes": true_b, "reported_bytes": rep,
                                     "factor": round(true_b / rep, 4) if rep else None, "stands_for": what}
# the other direction of every kernel (what a read kernel writes and vice versa): sanity, should be ~0
json.dump(out, sys.stdout, indent=1)
print()
