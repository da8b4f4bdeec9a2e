#!/usr/bin/env python3
"""A known-good fp64 MFMA reference on the same chip: rocBLAS DGEMM through torch.mm, at a large square size and at the
E-step's own shape ([N, M] x [M, M], dense).  What fraction of the 78.6 TFLOP/s peak does the vendor's tuned kernel hold?
The ceiling the hand-written panel / site kernels are judged against (cdna_hip_programming.md 5.4 rule 10).  GPU box."""
import sys
import torch

dev = "cuda:0"
def bench(m, n, k, reps=5):
    a = torch.randn(m, k, dtype=torch.float64, device=dev)
    b = torch.randn(k, n, dtype=torch.float64, device=dev)
    c = torch.empty(m, n, dtype=torch.float64, device=dev)
    for _ in range(2): torch.mm(a, b, out=c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): torch.mm(a, b, out=c)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    tf = 2.0 * m * n * k / ms / 1e9
    print(f"rocBLAS dgemm {m} x {n} x {k}: {ms:.3f} ms  {tf:.2f} TFLOP/s  ({tf / 78.6:.3f} of 78.6)", flush=True)

bench(8192, 8192, 8192)
bench(16384, 16384, 4096)
bench(1_000_000, 1024, 1024)
bench(125_000, 1024, 1024)
