#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel (mean per dispatch)."""
import collections, csv, glob, sys
path = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(path)):
    k = r["Kernel_Name"].split("(")[0].replace("lt::(anonymous namespace)::", "").replace("void ", "")
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k][r["Counter_Name"]] += 1
for k in sorted(acc):
    if k.startswith("__amd"):
        continue
    print(k)
    print("   " + "  ".join(f"{c}={acc[k][c] / cnt[k][c]:.4g}" for c in sorted(acc[k])))
