#!/usr/bin/env python3
"""Aggregates a rocprofv3 kernel trace over the M x M section of one cold E-step (from the end of the previous step's
syrk_reduce to the start of this step's moments kernel).  usage: mxm_timeline.py <kernel_trace.csv> [step] [--list]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
step = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 4
idx = [i for i, r in enumerate(rows) if "panel1_kernel" in r["Kernel_Name"] or "panel_kernel<double, 1, 1" in r["Kernel_Name"]]
i = idx[step]
a = max(j for j in range(i) if "syrk_reduce" in rows[j]["Kernel_Name"])
def short(n):
    n = re.sub(r"void |at::native::\(anonymous namespace\)::|at::native::|\(anonymous namespace\)::", "", n)
    return n[:56]
t0 = int(rows[a]["End_Timestamp"]); t1 = int(rows[i]["Start_Timestamp"])
print(f"M x M section (epilogue of step {step - 1} + prelude + K_fu fill of step {step}): wall {(t1 - t0) / 1e3:.1f} us, {i - a - 1} dispatches")
agg = collections.OrderedDict(); busy = gaps = 0; last = t0
for r in rows[a + 1:i]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    d = agg.setdefault(short(r["Kernel_Name"]), [0, 0.0]); d[0] += 1; d[1] += (e - s) / 1e3
    busy += e - s; gaps += max(0, s - last); last = max(last, e)
    if "--list" in sys.argv:
        print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:8.1f}  {short(r['Kernel_Name'])}")
print(f"   busy {busy / 1e3:.1f} us, idle gaps {gaps / 1e3:.1f} us")
for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:30]:
    print(f"   {t:8.1f} us  x{n:<4d} {k}")
