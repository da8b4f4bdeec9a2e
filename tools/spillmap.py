#!/usr/bin/env python3
"""Where do the register spills of a kernel fall?  Per basic block: MFMA count, scratch ops, loop depth.
usage: spillmap.py <file.s> <mangled-kernel-name-substring>"""
import re, sys
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and l.rstrip().endswith(":") or (key in l and re.match(r"^_Z\S+:", l)))
end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i])
blk, stats, order = "entry", {"entry": dict(mfma=0, scr=0, depth="")}, ["entry"]
for l in lines[start:end]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blk = m.group(1); stats[blk] = dict(mfma=0, scr=0, depth=""); order.append(blk); continue
    if "v_mfma" in l: stats[blk]["mfma"] += 1
    if "scratch_" in l: stats[blk]["scr"] += 1
    m = re.search(r"Depth[= ](\d)", l)
    if m: stats[blk]["depth"] = m.group(1)
for b in order:
    st = stats[b]
    if st["mfma"] or st["scr"]: print(f"{b:12s} mfma={st['mfma']:4d} scratch={st['scr']:4d} depth={st['depth']}")
