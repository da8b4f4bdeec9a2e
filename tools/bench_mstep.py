#!/usr/bin/env python3
"""Times one M-step gradient evaluation (t_SVGP.elbo_and_grads) on the `ns` workload (N=1e6, M=1024, D=8, fp64) and
lists its kernels (GPU box)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
pkg = importlib.import_module("t-svgp_amd")
w = bench.WORKLOADS["ns"]
X, Y, Z = bench.make_data(w)
Xd, Yd = torch.as_tensor(X, device="cuda:0"), torch.as_tensor(Y, device="cuda:0")
m = pkg.t_SVGP(pkg.SquaredExponential(1.0, 1.0), pkg.Gaussian(0.1), Z, num_data=w["N"])
for _ in range(3): m.natgrad_step((Xd, Yd), lr=0.8)
for _ in range(2): m.elbo_and_grads((Xd, Yd))
eng = m._get_engine()
import gc
gc.collect(); gc.disable()  # as timeit does: one full collection of this process is 30-40 ms, i.e. +7 ms per evaluation when it falls into the loop
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): e, g = m.elbo_and_grads((Xd, Yd))
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"elbo_and_grads: {dt * 1e3:.2f} ms per evaluation; elbo {float(e):.6f}")
eng.profile = {}  # a second loop with per-launch HIP events for the kernel list (the brackets cost time: not the figure above)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): e, g = m.elbo_and_grads((Xd, Yd))
torch.cuda.synchronize(); dtp = (time.perf_counter() - t0) / 5
prof = eng.profile_summary(); eng.profile = None
print(f"(with per-launch events: {dtp * 1e3:.2f} ms per evaluation)")
print({k: float(v) if v.dim() == 0 else [round(float(x), 4) for x in v.reshape(-1)[:4]] for k, v in g.items()})
for k, (n, ms, *_) in prof.items():
    print(f"  {k:22s} x{n / 5:.0f}  {ms:8.3f} ms")
