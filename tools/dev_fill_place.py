#!/usr/bin/env python3
"""Where the K(X, Z) fill of a shard-sized step goes: beside the M x M prelude on the side stream (default) or in line right in front
of the moments kernel (overlap_fill=False), where it is the last thing the chip does before the N-pass (the clock: profiles/
r05_clock_lab.txt).  Replayed steps, alternating on one box.   usage: dev_fill_place.py [workload] [rows]"""
import gc, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
pkg = importlib.import_module("t-svgp_amd")
wl = sys.argv[1] if len(sys.argv) > 1 else "ns"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 125000
w = dict(bench.WORKLOADS[wl]); w["N"] = rows
X, Y, Z = bench.make_data(w)
dt = torch.float64 if w["dtype"] == "f64" else torch.float32
Xd, Yd = torch.as_tensor(X, dtype=dt, device="cuda:0"), torch.as_tensor(Y, dtype=dt, device="cuda:0")
lik = pkg.Gaussian(variance=w.get("noise", 0.1)) if w["lik"] == "gaussian" else pkg.Bernoulli()
models = {}
MODES = ("side stream", "all in line", "fill in line")
for mode in MODES:
    kernel, wrap = bench.make_kernel(pkg, w)
    m = pkg.t_SVGP(kernel, lik, wrap(Z), num_latent_gps=w["P"], num_data=w["N"], compute_dtype=dt)
    m.overlap_fill = mode != "all in line"
    m.FILL_INLINE_MAX_NM = 10 ** 12 if mode == "fill in line" else 0
    models[mode] = m
for rep in range(3):
    for mode in MODES:
        m = models[mode]
        for _ in range(5): m.natgrad_step((Xd, Yd), lr=0.8)
        torch.cuda.synchronize(); gc.collect(); gc.disable(); t0 = time.perf_counter()
        for _ in range(40): m.natgrad_step((Xd, Yd), lr=0.8)
        torch.cuda.synchronize(); dt_ = (time.perf_counter() - t0) / 40; gc.enable()
        print(f"{wl} rows {rows} {mode}: {dt_ * 1e3:.3f} ms per step", flush=True)
