#!/usr/bin/env python3
"""Is the trmm kernel's speed data dependent?  Times it with random vs real (Linv9, Kfu) operands."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
pkg = importlib.import_module("t-svgp_amd"); estep = importlib.import_module("t-svgp_amd.estep"); B = pkg._backend
rows = 500000
w = dict(bench.WORKLOADS["ns"], N=rows)
X, Y, Z = bench.make_data(w)
dev = torch.device("cuda:0")
eng = estep.EStepEngine(torch.float64, dev)
Xd = torch.as_tensor(X).to(dev); Zd = torch.as_tensor(Z).to(dev)
k = pkg.SquaredExponential(1.0, 1.0)
Kzz = eng.kuu(Zd, k)
M = 1024; Np = B.round_up(rows)
L9 = torch.linalg.cholesky(Kzz + 1e-9 * torch.eye(M, device=dev, dtype=torch.float64))
Linv = torch.linalg.solve_triangular(L9, torch.eye(M, device=dev, dtype=torch.float64), upper=False).contiguous()
print("Linv |x| stats: min nonzero", float(Linv[Linv != 0].abs().min()), "frac |x|<1e-100:", float((Linv.abs() < 1e-100).double().mean()), "frac denormal:", float(((Linv != 0) & (Linv.abs() < 2.3e-308)).double().mean()))
Kfu = torch.empty(Np, M, dtype=torch.float64, device=dev)
eng.se_fill(Xd, Zd, torch.ones(8, dtype=torch.float64, device=dev), 1.0, Kfu)
print("Kfu min nonzero", float(Kfu[Kfu != 0].min()))
Bw = torch.empty_like(Kfu)
g = torch.Generator(device=dev).manual_seed(0)
Trand = torch.tril(torch.randn(M, M, generator=g, device=dev, dtype=torch.float64)) / 32
Arand = torch.randn(Np, M, generator=g, device=dev, dtype=torch.float64)
Lflush = torch.where(Linv.abs() < 1e-30, torch.zeros_like(Linv), Linv)
def t(name, A, T):
    for _ in range(2): eng.trmm(A, T, Bw, 0)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): eng.trmm(A, T, Bw, 0)
    e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1)/5:8.3f} ms")
t("A=Kfu  T=Linv (real)", Kfu, Linv)
t("A=Kfu  T=rand", Kfu, Trand)
t("A=rand T=Linv", Arand, Linv)
t("A=rand T=rand", Arand, Trand)
t("A=Kfu  T=Linv flushed<1e-30", Kfu, Lflush)
t("A=zeros T=zeros", torch.zeros_like(Kfu), torch.zeros_like(Linv))

print("--- in sequence: fill -> trmm (x6), events around trmm only")
inv = torch.ones(8, dtype=torch.float64, device=dev)
def seq(pre):
    ts = []
    for _ in range(6):
        pre()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); eng.trmm(Kfu, Linv, Bw, 0); e1.record()
        ts.append((e0, e1))
    torch.cuda.synchronize()
    return " ".join(f"{a.elapsed_time(b):7.3f}" for a, b in ts)
print("pre = nothing          ", seq(lambda: None))
print("pre = se_fill(Kfu)     ", seq(lambda: eng.se_fill(Xd, Zd, inv, 1.0, Kfu)))
other = torch.empty_like(Kfu)
print("pre = se_fill(other)   ", seq(lambda: eng.se_fill(Xd, Zd, inv, 1.0, other)))
print("pre = sync+sleep 20ms  ", seq(lambda: (torch.cuda.synchronize(), __import__('time').sleep(0.02))))
print("pre = fill + sync      ", seq(lambda: (eng.se_fill(Xd, Zd, inv, 1.0, Kfu), torch.cuda.synchronize())))
