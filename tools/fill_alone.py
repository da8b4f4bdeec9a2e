#!/usr/bin/env python3
"""The K(X, Z) fill alone at the `ns` sizes (N=1e6, M=1024, D=8, fp64), 20 launches back to back; TSVGP_HIP_LIB picks the
build (tools/ab_builds.sh).  GPU box."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
pkg = importlib.import_module("t-svgp_amd")
w = bench.WORKLOADS["ns"]
X, Y, Z = bench.make_data(w)
Xd = torch.as_tensor(X, device="cuda:0"); Zd = torch.as_tensor(Z, device="cuda:0")
m = pkg.t_SVGP(pkg.SquaredExponential(1.0, 1.0), pkg.Gaussian(0.1), Z, num_data=w["N"])
eng = m._get_engine()
il = torch.ones(8, dtype=torch.float64, device="cuda:0")
K = torch.empty((1000064, 1024), dtype=torch.float64, device="cuda:0")
for _ in range(3): eng.se_fill(Xd, Zd, il, 1.0, K)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): eng.se_fill(Xd, Zd, il, 1.0, K)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
print(os.environ.get("TSVGP_HIP_LIB", "default"), f"fill alone {dt*1e3:.3f} ms  {K.numel()*8/dt/1e12:.2f} TB/s")
