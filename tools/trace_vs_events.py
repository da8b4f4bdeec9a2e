#!/usr/bin/env python3
"""Do the HIP-event kernel times that bench.py prints agree with the rocprofv3 kernel trace of the SAME run?
usage: trace_vs_events.py <kernel_trace.csv> <bench stdout (the JSON line is its last line starting with '{')>"""
import csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
line = [l for l in open(sys.argv[2]) if l.startswith("{")][-1]
b = json.loads(line)
W, K = b["warmup"], b["steps"]
print(f"rocprofv3 --kernel-trace of bench.py --steps {K} --warmup {W} (workload {b['config']['workload'][:40]}): per-launch durations from the")
print("trace against the HIP-event means bench.py printed in the SAME run (timed launches = launches W+1 .. W+K of each kernel).\n")
# round 3: the fp64 moments / site sums run panel1_kernel / syrk1_kernel; older builds panel_kernel<double, 1, 1, true> / syrk_kernel<double>
def first_present(*pats):
    for pat in pats:
        if any(pat in r["Kernel_Name"] for r in rows):
            return pat
    return pats[0]
for pat, key, extra in ((first_present("panel1_kernel", "panel_kernel<double, 1, 1, true>"), "tsvgp_moments", None),
                        (first_present("syrk1_kernel", "syrk_kernel<double>"), "tsvgp_site_accum", "syrk_reduce_kernel<double>")):
    ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if pat in r["Kernel_Name"]]
    timed = ms[W:W + K]
    ex = 0.0
    if extra:
        e = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if extra in r["Kernel_Name"]]
        ex = sum(e[W:W + K]) / max(len(e[W:W + K]), 1)
    print(f"{pat}: {len(ms)} launches, ms: {[round(x, 2) for x in ms]}")
    print(f"   mean of the {K} timed launches in the trace{' (+ ' + extra + f' {ex:.3f})' if extra else ''} : {sum(timed) / len(timed) + ex:.3f} ms")
    print(f"   bench.py kernels['{key}'].avg_ms (HIP events)                      : {b['kernels'][key]['avg_ms']:.3f} ms")
    print(f"   --stats average over all {len(ms)} launches                            : {sum(ms) / len(ms):.3f} ms\n")
print(f"bench line of that run: value {b['value']} E-steps/s, ms_per_step {b['ms_per_step']} (under the tracer); roofline {b['roofline']}")
