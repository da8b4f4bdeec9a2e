#!/usr/bin/env python3
"""Host-side profile of the replayed E-step at a launch-bound size (BASELINE configs[0]: N = 1000, M = 32): cProfile over 500
steps, top entries by cumulative and by own time.  GPU box.  usage: python tools/prof_small_step.py [poll|block]"""
import cProfile, importlib, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
pkg = importlib.import_module("t-svgp_amd")
w = bench.WORKLOADS["c1"]
X, Y, Z = bench.make_data(w)
Xd, Yd = torch.as_tensor(X, device="cuda:0"), torch.as_tensor(Y, device="cuda:0")
m = pkg.t_SVGP(pkg.SquaredExponential(w.get("variance", 1.0), w.get("lengthscales", 1.0)), pkg.Gaussian(w.get("noise", 0.1)), Z,
               num_data=w["N"], use_graph=True, projection=os.environ.get("PROJECTION", "auto"))
print("routes:", m._routes(1e-9))
if len(sys.argv) > 1 and sys.argv[1] == "block":
    m.poll_status = False
for _ in range(10):
    m.natgrad_step((Xd, Yd), lr=0.8)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(500):
    m.natgrad_step((Xd, Yd), lr=0.8)
torch.cuda.synchronize()
print(f"plain loop: {(time.perf_counter() - t0) / 500 * 1e3:.3f} ms per step")
t0 = time.perf_counter()
for _ in range(500):
    m._graphs[next(k for k, v in m._graphs.items() if isinstance(v, dict))]["graph"].replay()
torch.cuda.synchronize()
print(f"replay only, no status read: {(time.perf_counter() - t0) / 500 * 1e3:.3f} ms per step")
pr = cProfile.Profile()
pr.enable()
for _ in range(500):
    m.natgrad_step((Xd, Yd), lr=0.8)
torch.cuda.synchronize()
pr.disable()
for key in ("cumulative", "tottime"):
    pstats.Stats(pr).sort_stats(key).print_stats(18)
