#!/usr/bin/env python3
"""The headline workload (ns, N = 1e6) eagerly (what use_graph="auto" does at that size) against a replay of the captured step
(use_graph=True), alternating on one box."""
import gc, importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
pkg = importlib.import_module("t-svgp_amd")
w = bench.WORKLOADS["ns"]
X, Y, Z = bench.make_data(w)
Xd, Yd = torch.as_tensor(X, device="cuda:0"), torch.as_tensor(Y, device="cuda:0")
res = {}
models = {}
for mode in (False, True):
    models[mode] = pkg.t_SVGP(pkg.SquaredExponential(1.0, 1.0), pkg.Gaussian(0.1), Z, num_data=w["N"], use_graph=mode)
for rep in range(3):
    for mode in (False, True):
        m = models[mode]
        for _ in range(4): m.natgrad_step((Xd, Yd), lr=0.8)
        torch.cuda.synchronize(); gc.collect(); gc.disable(); t0 = time.perf_counter()
        for _ in range(20): m.natgrad_step((Xd, Yd), lr=0.8)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20; gc.enable()
        print(f"use_graph={mode}: {dt * 1e3:.3f} ms per step", flush=True)
