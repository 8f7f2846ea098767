#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counter_collection.csv files per kernel: tools/pmc_summary.py DIR [DIR...] [--kernels a,b]"""
import collections, csv, glob, re, sys
dirs = [a for a in sys.argv[1:] if not a.startswith("--")]
want = None
for a in sys.argv[1:]:
    if a.startswith("--kernels="):
        want = a.split("=", 1)[1].split(",")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d in dirs:
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"zg::(k_\w+(?:<\d+>)?)", r["Kernel_Name"])
            k = m.group(1) if m else r["Kernel_Name"].split("(")[0]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in sorted(agg.items()):
    if want and k not in want:
        continue
    extra = {}
    if "SQ_THREAD_CYCLES_VALU" in v and v.get("SQ_ACTIVE_INST_VALU"):
        extra["valu_lane_util"] = v["SQ_THREAD_CYCLES_VALU"] / (v["SQ_ACTIVE_INST_VALU"] * 64)
    if "SQ_LDS_BANK_CONFLICT" in v and v.get("SQ_LDS_IDX_ACTIVE"):
        extra["lds_conflict_frac"] = v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"]
    print(k, {**{a: f"{b:.4g}" for a, b in v.items()}, **{a: f"{b:.3f}" for a, b in extra.items()}})
