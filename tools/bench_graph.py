#!/usr/bin/env python3
"""E-steps/s of the launch-bound config C1 (N=1000, M=32) and of a mid-size problem, eager vs hipGraph replay
(t_SVGP(use_graph=True)).  GPU box."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
pkg = importlib.import_module("t-svgp_amd")
SIZES = [("c1 (N=1000, M=32)", bench.WORKLOADS["c1"])] + [
    (f"N={n}, M={m}, D=8", dict(bench.WORKLOADS["ns"], N=n, M=m))
    for n, m in ((2000, 64), (5000, 128), (20000, 128), (5000, 256), (20000, 256), (100000, 128), (50000, 512), (20000, 1024), (62500, 1024))]
for name, w in SIZES:
    X, Y, Z = bench.make_data(w)
    Xd, Yd = torch.as_tensor(X, device="cuda:0"), torch.as_tensor(Y, device="cuda:0")
    for use_graph in (False, True):
        m = pkg.t_SVGP(pkg.SquaredExponential(w.get("variance", 1.0), w.get("lengthscales", 1.0)), pkg.Gaussian(w.get("noise", 0.1)), Z,
                       num_data=w["N"], use_graph=use_graph)
        for _ in range(5): m.natgrad_step((Xd, Yd), lr=0.8)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200): m.natgrad_step((Xd, Yd), lr=0.8)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        print(f"{name:24s} {'hipGraph replay' if use_graph else 'eager':16s} {1 / dt:9.1f} E-steps/s  ({dt * 1e3:.3f} ms)  elbo {float(m.elbo((Xd, Yd))):.10f}", flush=True)
