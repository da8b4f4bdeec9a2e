#!/usr/bin/env python3
"""The K(X, Z) fill alone in fp32 at C3's sizes (N = 1e6, M = 1024, D = 16) and D = 8, 20 launches back to back, with a check
against torch on a row sample.  GPU box."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
estep = importlib.import_module("t-svgp_amd.estep")
eng = estep.EStepEngine(torch.float32, "cuda:0")
N, M = 1_000_000, 1024
for D in (16, 8):
    g = torch.Generator(device="cpu").manual_seed(0)
    X = torch.randn(N, D, generator=g).to("cuda:0"); Z = X[:M].clone()
    il = torch.full((D,), 0.7, dtype=torch.float32, device="cuda:0")
    K = torch.empty((N + 64, M), dtype=torch.float32, device="cuda:0")
    for _ in range(3): eng.se_fill(X, Z, il, 1.3, K)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): eng.se_fill(X, Z, il, 1.3, K)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    idx = torch.randint(0, N, (2000,), generator=g).to("cuda:0")
    ref = 1.3 * torch.exp(-0.5 * (((X[idx].double() * 0.7)[:, None, :] - (Z.double() * 0.7)[None]) ** 2).sum(-1))
    err = (K[idx].double() - ref).abs().max().item()
    print(f"fp32 D={D}: fill alone {dt*1e3:.3f} ms  {K.numel()*4/dt/1e12:.2f} TB/s  (of 8: {K.numel()*4/dt/8e12:.2f})  max abs err on 2000 rows {err:.2e}")
