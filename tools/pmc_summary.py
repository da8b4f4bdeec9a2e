#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc csv output per kernel: mean counter value per dispatch.  usage: pmc_summary.py <dir> [filter]"""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else "anonymous namespace"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        full = r["Kernel_Name"]
        if filt not in full:
            continue
        m = re.search(r"(\w+<[^>]*>|\w+)\(", full.replace("(anonymous namespace)::", ""))
        name = m.group(1) if m else full[:60]
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in sorted(agg.items()):
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
