#!/usr/bin/env python3
"""Host-side time of the sections of one eager E-step (where does the CPU block?): wraps the model's / engine's methods with
perf_counter stamps.  usage: host_sections.py [workload] [steps]   (GPU box)"""
import importlib, os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
pkg = importlib.import_module("t-svgp_amd")
wname = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
w = dict(bench.WORKLOADS[wname])
dt = torch.float64 if w["dtype"] == "f64" else torch.float32
X, Y, Z = bench.make_data(w)
dev = torch.device("cuda", 0)
Xd, Yd = torch.as_tensor(X, dtype=dt).to(dev), torch.as_tensor(Y, dtype=dt).to(dev)
lik = pkg.Gaussian(0.1) if w["lik"] == "gaussian" else pkg.Bernoulli()
m = pkg.t_SVGP(pkg.SquaredExponential(1.0, 1.0), lik, Z, num_data=w["N"], compute_dtype=dt, device=dev)
eng = m._get_engine()
acc = collections.defaultdict(list)
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); acc[label].append(time.perf_counter() - t0); return r
    setattr(obj, name, g)
wrap(eng, "start_fill", "start_fill"); wrap(m, "_site_operands", "site_operands"); wrap(eng, "run", "engine.run")
wrap(m, "_apply_site_update", "apply_site_update"); wrap(m, "_read_flags", "read_flags (waits for the GPU)"); wrap(m, "_routes", "routes")
for _ in range(5): m.natgrad_step((Xd, Yd), lr=0.8)
torch.cuda.synchronize(); acc.clear()
t0 = time.perf_counter()
per = []
for _ in range(steps):
    t1 = time.perf_counter(); m.natgrad_step((Xd, Yd), lr=0.8); per.append(time.perf_counter() - t1)
torch.cuda.synchronize()
tot = (time.perf_counter() - t0) / steps * 1e3
print(f"{wname}: {tot:.3f} ms per step; host time per section (mean / max over {steps} steps, ms):")
for k, v in acc.items():
    print(f"   {k:34s} {np.mean(v) * 1e3:8.3f} {np.max(v) * 1e3:8.3f}")
print(f"   natgrad_step total                 {np.mean(per) * 1e3:8.3f} {np.max(per) * 1e3:8.3f}")
