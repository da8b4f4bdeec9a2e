#!/usr/bin/env python3
"""Per-launch kernel times over consecutive E-steps (GPU box): shows drift / context effects.  usage: steptrace.py [rows] [steps]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
pkg = importlib.import_module("t-svgp_amd")
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w = dict(bench.WORKLOADS["ns"], N=rows)
X, Y, Z = bench.make_data(w)
dev = torch.device("cuda:0")
Xd = torch.as_tensor(X).to(dev); Yd = torch.as_tensor(Y).to(dev)
m = pkg.t_SVGP(pkg.SquaredExponential(1.0, 1.0), pkg.Gaussian(0.1), Z, num_data=rows, device=dev)
eng = m._get_engine()
for _ in range(2): m.natgrad_step((Xd, Yd), lr=0.8)
torch.cuda.synchronize()
eng.profile = {}
import time
t0 = time.perf_counter()
for _ in range(steps): m.natgrad_step((Xd, Yd), lr=0.8)
torch.cuda.synchronize()
print("ms/step", (time.perf_counter() - t0) / steps * 1e3)
for k, evs in eng.profile.items():
    print(f"{k:22s}", " ".join(f"{a.elapsed_time(b):7.3f}" for a, b in evs))
