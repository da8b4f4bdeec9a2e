#!/usr/bin/env python3
"""How much does a short stretch of low activity in front of the moments kernel cost it?  Times tsvgp_moments_f64 at
the headline sizes (a) back to back, (b) after `gap_us` of a spinning single-workgroup kernel (torch.cuda._sleep), (c) after
the K(X,Z) fill.   usage: exp_gap.py [gap_us ...]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
p = importlib.import_module("t-svgp_amd")
E = importlib.import_module("t-svgp_amd.estep"); K_ = importlib.import_module("t-svgp_amd.kernels")
dev = torch.device("cuda:0")
N, M, D = 1_000_000, 1024, 8
eng = E.EStepEngine(torch.float64, dev)
g = torch.Generator().manual_seed(0)
X = torch.randn(N, D, generator=g, dtype=torch.float64).to(dev); Z = X[:M].clone()
kern = K_.SquaredExponential(variance=1.0, lengthscales=1.0)
inv_ls = kern.inv_lengthscales(D, torch.float64, dev)
Np = (N + 127) // 128 * 128
Kfu = torch.empty(Np, M, dtype=torch.float64, device=dev)
eng.se_fill(X, Z, inv_ls, 1.0, Kfu)
T = torch.triu(torch.randn(M, M, dtype=torch.float64, device=dev)) / 32
gam = torch.randn(M, 1, dtype=torch.float64, device=dev); Y = torch.randn(N, 1, dtype=torch.float64, device=dev)
g0 = torch.empty(Np, 1, dtype=torch.float64, device=dev); g1 = torch.empty_like(g0)
vep = torch.empty(Np // 128, dtype=torch.float64, device=dev); npp = torch.empty(Np // 128, dtype=torch.int32, device=dev)
fn = eng._fn("tsvgp_moments")
def moments():
    assert fn(Kfu.data_ptr(), T.data_ptr(), gam.data_ptr(), Y.data_ptr(), 1e9, 1, 0.1, None, None, g0.data_ptr(), g1.data_ptr(),
              vep.data_ptr(), npp.data_ptr(), N, Np, M, 1, 1, eng._stream()) == 0
def timed(pre, reps=8):
    for _ in range(2): pre(); moments()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(reps):
        pre()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); moments(); e1.record(); torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / reps
cyc_per_us = 2100  # torch.cuda._sleep counts device clock cycles (roughly)
print(f"back to back (the timing sync is the only gap): {timed(lambda: moments()):.3f} ms")
for gap in [int(a) for a in sys.argv[1:]] or [100, 300, 1000, 3000]:
    print(f"after {gap:5d} us of a one-workgroup spin:          {timed(lambda: (moments(), torch.cuda._sleep(gap * cyc_per_us))):.3f} ms")
print(f"after the K(X,Z) fill:                          {timed(lambda: (moments(), eng.se_fill(X, Z, inv_ls, 1.0, Kfu))):.3f} ms")
print(f"after the fill + 300 us spin:                   {timed(lambda: (moments(), eng.se_fill(X, Z, inv_ls, 1.0, Kfu), torch.cuda._sleep(300 * cyc_per_us))):.3f} ms")
