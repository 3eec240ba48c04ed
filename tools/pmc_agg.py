#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc csv outputs of tools/prof_pmc.sh per kernel and print derived ratios."""
import collections, csv, glob, re, sys
root = sys.argv[1]
def agg(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(lambda: collections.defaultdict(int))
    for r in csv.DictReader(open(path)):
        m = re.search(r'(k_[a-z0-9_]+)(<[^>]*>)?', r["Kernel_Name"])
        if not m: continue
        k = m.group(1) + (m.group(2) or '').replace('lt::(anonymous namespace)::', '')
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
    return {k: {c: acc[k][c] / cnt[k][c] for c in acc[k]} for k in acc}
A = {}
for p in glob.glob(root + '/**/*counter_collection.csv', recursive=True):
    for k, v in agg(p).items():
        A.setdefault(k, {}).update(v)
for k, v in A.items():
    wc = v.get('SQ_WAVE_CYCLES', 0) or 1
    g = lambda n: v.get(n, 0)
    print(k)
    print(f"   waves={g('SQ_WAVES'):.0f} busy_cycles={g('SQ_BUSY_CYCLES'):.3g} gui_active={g('GRBM_GUI_ACTIVE'):.3g}")
    print(f"   per-wave time: VALU {g('SQ_ACTIVE_INST_VALU')/wc:.1%}  LDS {g('SQ_ACTIVE_INST_LDS')/wc:.1%}  any-active {g('SQ_ACTIVE_INST_ANY')/wc:.1%}  wait_any {g('SQ_WAIT_ANY')/wc:.1%}  wait_inst {g('SQ_WAIT_INST_ANY')/wc:.1%}  wait_lds {g('SQ_WAIT_INST_LDS')/wc:.1%}")
    print(f"   wave-insts: VALU={g('SQ_INSTS_VALU'):.3g} LDS={g('SQ_INSTS_LDS'):.3g} SALU={g('SQ_INSTS_SALU'):.3g} VMEM_RD={g('SQ_INSTS_VMEM_RD'):.3g}  lds_bank_conflict={g('SQ_LDS_BANK_CONFLICT'):.3g}/{g('SQ_LDS_IDX_ACTIVE'):.3g}")
    print(f"   FETCH_SIZE={g('FETCH_SIZE'):.4g} KB  WRITE_SIZE={g('WRITE_SIZE'):.4g} KB")
