#!/usr/bin/env python3
"""Round 5: the Bernoulli likelihood map inside the moments kernel's tail against the moments alone (TSVGP_LIK_NONE: mean and
variance out) followed by tsvgp_lik_map on their own -- C3's shape (fp32) and the fp64 one.  Alternating, HIP events, one box.
usage: lik_split_lab.py [rows] [M]     (GPU box)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
M = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
B = importlib.import_module("t-svgp_amd._backend")
lib = B.lib()
dev = "cuda:0"
Np = (rows + 127) // 128 * 128
nwg = Np // 128
for dt, sfx in ((torch.float32, "f32"), (torch.float64, "f64")):
    A = (torch.randn(Np, M, dtype=torch.float64, device=dev) / 32).to(dt)
    T = (torch.triu(torch.randn(1, M, M, dtype=torch.float64, device=dev)) / 32).to(dt)
    gam = torch.randn(M, 1, dtype=torch.float64, device=dev).to(dt)
    Y = (torch.rand(rows, 1, device=dev) < 0.5).to(dt)
    mean = torch.empty(Np, 1, dtype=dt, device=dev); var = torch.empty_like(mean)
    g0 = torch.empty(Np, 1, dtype=dt, device=dev); g1 = torch.empty_like(g0)
    g0b = torch.empty_like(g0); g1b = torch.empty_like(g0)
    vep = torch.empty(nwg, dtype=torch.float64, device=dev); npp = torch.empty(nwg, dtype=torch.int32, device=dev)
    vepb = torch.empty_like(vep); nppb = torch.empty_like(npp)
    mom, lmap = getattr(lib, "tsvgp_moments_" + sfx), getattr(lib, "tsvgp_lik_map_" + sfx)
    def fused(lik):
        assert mom(A.data_ptr(), T.data_ptr(), gam.data_ptr(), Y.data_ptr(), 4.0, lik, 0.1, None, None, g0.data_ptr(),
                   g1.data_ptr(), vep.data_ptr(), npp.data_ptr(), rows, Np, M, 1, 1, None) == 0
    def split(lik):
        assert mom(A.data_ptr(), T.data_ptr(), gam.data_ptr(), None, 4.0, 0, 0.1, mean.data_ptr(), var.data_ptr(), None,
                   None, vepb.data_ptr(), nppb.data_ptr(), rows, Np, M, 1, 1, None) == 0
        assert lmap(mean.data_ptr(), var.data_ptr(), Y.data_ptr(), lik, 0.1, g0b.data_ptr(), g1b.data_ptr(), vepb.data_ptr(),
                    nppb.data_ptr(), rows, Np, 1, None) == 0
    def only_map(lik):
        assert lmap(mean.data_ptr(), var.data_ptr(), Y.data_ptr(), lik, 0.1, g0b.data_ptr(), g1b.data_ptr(), vepb.data_ptr(),
                    nppb.data_ptr(), rows, Np, 1, None) == 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    def timed(fn, lik, n=8):
        for _ in range(2): fn(lik)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): fn(lik)
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    for lik, name in ((2, "Bernoulli"), (1, "Gaussian")):
        res = []
        for rep in range(3):
            res.append((timed(fused, lik), timed(split, lik), timed(only_map, lik, 50)))
        fused(lik); split(lik); torch.cuda.synchronize()
        d0 = float((g0[:rows] - g0b[:rows]).abs().max()); d1 = float((g1[:rows] - g1b[:rows]).abs().max())
        dv = abs(float(vep.sum() - vepb.sum()))
        print(f"{sfx} rows {rows} M {M} {name}: fused " + " ".join(f"{r[0]:.3f}" for r in res) + " ms | moments(NONE) + lik_map "
              + " ".join(f"{r[1]:.3f}" for r in res) + " ms | lik_map alone " + " ".join(f"{r[2]:.4f}" for r in res)
              + f" ms | max |g0 - g0'| {d0:.2e} |g1 - g1'| {d1:.2e} |ve - ve'| {dv:.2e}", flush=True)
