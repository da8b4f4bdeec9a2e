#!/usr/bin/env python3
"""Times tsvgp_potrf_f64 (batch 1, 2, 4) against torch.linalg.cholesky at M = 1024 (GPU box)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
estep = importlib.import_module("t-svgp_amd.estep")
eng = estep.EStepEngine(torch.float64, "cuda:0")
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
def timeit(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for batch in (1, 2, 4):
    A = torch.randn(batch, M, M, dtype=torch.float64, device="cuda:0")
    A = A @ A.transpose(-1, -2) / M + torch.eye(M, dtype=torch.float64, device="cuda:0")
    print(f"M={M} batch={batch}: tsvgp_potrf {timeit(lambda: eng.cholesky(A)):.3f} ms   torch.cholesky_ex {timeit(lambda: torch.linalg.cholesky_ex(A)):.3f} ms")
