#!/usr/bin/env python3
"""Where one M-step gradient evaluation (t_SVGP.elbo_and_grads, ns workload) spends its wall time: the engine / model calls
wrapped with a synchronisation and a host clock each (so phases do not overlap: their sum is an upper bound of the evaluation)."""
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
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): e, g = m.elbo_and_grads((Xd, Yd))
torch.cuda.synchronize(); print(f"plain: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per evaluation")
eng = m._get_engine()
acc = {}
def wrap(obj, name, label=None):
    f = getattr(obj, name)
    def g(*a, **k):
        torch.cuda.synchronize(); t = time.perf_counter()
        r = f(*a, **k)
        torch.cuda.synchronize(); acc[label or name] = acc.get(label or name, 0.0) + time.perf_counter() - t
        return r
    setattr(obj, name, g)
for n in ("trmm", "kernel_grad", "se_fill", "_site_sums"):
    wrap(eng, n)
wrap(eng, "run", "run (incl. fill, trmm, mv, norm, lik_map, site sums)")
wrap(m, "_site_operands"); wrap(m, "_check_step")
tm = importlib.import_module("t-svgp_amd.models.tsvgp")
wrap(torch, "mv"); wrap(torch.linalg, "vector_norm", "vector_norm")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): e, g = m.elbo_and_grads((Xd, Yd))
torch.cuda.synchronize(); tot = (time.perf_counter() - t0) / 5
print(f"wrapped: {tot * 1e3:.2f} ms per evaluation")
for k, v in acc.items():
    print(f"  {k:60s} {v / 5 * 1e3:8.2f} ms")
