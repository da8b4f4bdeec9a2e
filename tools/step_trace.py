#!/usr/bin/env python3
"""Per-step breakdown of a rocprofv3 kernel trace of bench.py: for the last `n` E-steps of the trace (the replayed ones) the
durations of the moments and site-sum kernels, the wall time of the M x M section between them (end of syrk_reduce -> start of the
next moments kernel), the step period, and the kernels of that section by total time.
usage: step_trace.py <kernel_trace.csv> [n]"""
import collections, csv, re, sys
import numpy as np
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
name = lambda r: r["Kernel_Name"]
S = lambda r: int(r["Start_Timestamp"])
E = lambda r: int(r["End_Timestamp"])
mom = [i for i, r in enumerate(rows) if "panel1_kernel" in name(r) and "(PanelArgs" in name(r) or "panel1_kernel" in name(r)]
mom = mom[-(n + 1):]
def short(x):
    x = re.sub(r"void |at::native::\(anonymous namespace\)::|at::native::|\(anonymous namespace\)::", "", x)
    return x[:64]
per = []; agg = collections.OrderedDict()
for a, b in zip(mom[:-1], mom[1:]):
    seg = rows[a:b]
    acc = [r for r in seg if "syrk1" in name(r) or "syrk_kernel" in name(r)]
    red = [r for r in seg if "syrk_reduce" in name(r)]
    keep = [r for r in seg if "clock_keeper" in name(r)]
    if not acc or not red:
        continue
    t_red = E(red[-1])
    per.append(dict(period=(S(rows[b]) - S(rows[a])) / 1e3, moments=(E(rows[a]) - S(rows[a])) / 1e3,
                    accum=sum(E(r) - S(r) for r in acc) / 1e3, mxm=(S(rows[b]) - t_red) / 1e3,
                    npass=(t_red - S(rows[a])) / 1e3, keeper=sum(E(r) - S(r) for r in keep) / 1e3, nk=len(seg)))
    for r in seg:
        if S(r) >= t_red and "clock_keeper" not in name(r):
            d = agg.setdefault(short(name(r)), [0, 0.0]); d[0] += 1; d[1] += (E(r) - S(r)) / 1e3
print(f"{len(per)} steps; mean (min .. max) in us")
for k in ("period", "npass", "moments", "accum", "mxm", "keeper", "nk"):
    v = np.array([p[k] for p in per])
    print(f"  {k:8s} {v.mean():9.1f}  ({v.min():.1f} .. {v.max():.1f})")
print("kernels of the M x M sections, per step:")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print(f"   {t / len(per):8.1f} us  x{c / len(per):<5.1f} {k}")
