#!/usr/bin/env python3
"""Eager against hipGraph replay of the whole E-step at shard sizes from one eighth of the headline workload to all of it
(N = 125 000 ... 1 000 000 rows, M = 1024, D = 8, fp64): what `use_graph="auto"` should do at those sizes.  (GPU box)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
pkg = importlib.import_module("t-svgp_amd")
dev = torch.device("cuda", 0)
argv = sys.argv[1:]
wl = argv.pop(0) if argv and argv[0] in bench.WORKLOADS else "ns"  # usage: bench_graph_sizes.py [workload] [rows ...]
for rows in [int(a) for a in argv] or [125_000, 250_000, 500_000, 1_000_000]:
    w = dict(bench.WORKLOADS[wl], N=rows)
    X, Y, Z = bench.make_data(w)
    dtype = torch.float64 if w["dtype"] == "f64" else torch.float32
    Xd, Yd = torch.as_tensor(X, dtype=dtype).to(dev), torch.as_tensor(Y, dtype=dtype).to(dev)
    res = {}
    for mode in (False, True):
        kernel, wrap = bench.make_kernel(pkg, w)
        lik = pkg.Gaussian(variance=w.get("noise", 0.1)) if w["lik"] == "gaussian" else pkg.Bernoulli()
        m = pkg.t_SVGP(kernel, lik, wrap(Z), num_latent_gps=w["P"], num_data=rows, compute_dtype=dtype, device=dev, use_graph=mode)
        for _ in range(4):
            m.natgrad_step((Xd, Yd), lr=0.8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            m.natgrad_step((Xd, Yd), lr=0.8)
        torch.cuda.synchronize()
        res[mode] = ((time.perf_counter() - t0) / 20 * 1e3, float(m.elbo((Xd, Yd))))
        del m
        torch.cuda.empty_cache()
    print(f"{wl} rows {rows:8d}: eager {res[False][0]:7.3f} ms   hipGraph replay {res[True][0]:7.3f} ms   ({res[False][0] - res[True][0]:+.3f} ms)   "
          f"elbo equal: {abs(res[False][1] - res[True][1]) <= 1e-12 * abs(res[False][1])}", flush=True)
