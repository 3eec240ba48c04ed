#!/usr/bin/env python3
"""Per-kernel means of every counter found under a tools/prof_round.sh output directory.
usage: tools/pmc_kernels.py gpurun_out/<dir> [substring ...]   (kernels whose name contains any substring; default all)"""
import collections, csv, glob, re, sys
root, subs = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(p)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"].replace("lt::(anonymous namespace)::", "").replace("void ", ""))
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if subs and not any(s in k for s in subs):
        continue
    c = {n: sum(v) / len(v) for n, v in acc[k].items()}
    print(k)
    print("   " + "  ".join("%s=%.3g" % (n.replace("SQ_", ""), v) for n, v in sorted(c.items())))
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        print("   per wave-cycle: " + "  ".join("%s %.1f%%" % (n.replace("SQ_", ""), 100 * c[n] / wc) for n in
              ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY") if n in c))
    if "FETCH_SIZE" in c:
        print("   HBM side: fetch %.1f MB (x2 calibrated: %.1f MB), write %.1f MB" % (c["FETCH_SIZE"] / 1024, c["FETCH_SIZE"] / 512, c.get("WRITE_SIZE", 0) / 1024))
