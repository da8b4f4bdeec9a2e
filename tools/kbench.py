#!/usr/bin/env python3
"""Kernel-level timing harness (GPU box): times each C-ABI kernel on random data with HIP events.
usage: python tools/kbench.py [--rows 262144] [--M 1024] [--P 1] [--dtype f64] [--reps 5]"""
import argparse, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=262144)
ap.add_argument("--M", type=int, default=1024)
ap.add_argument("--D", type=int, default=8)
ap.add_argument("--P", type=int, default=1)
ap.add_argument("--dtype", default="f64")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--nsplits", default="")
a = ap.parse_args()
pkg = importlib.import_module("t-svgp_amd")
estep = importlib.import_module("t-svgp_amd.estep")
B = pkg._backend
dt = torch.float64 if a.dtype == "f64" else torch.float32
eng = estep.EStepEngine(dt, "cuda:0")
N, M, P, D = a.rows, a.M, a.P, a.D
Np, Mp = B.round_up(N), B.round_up(M)
g = torch.Generator(device="cuda:0").manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g, device="cuda:0", dtype=torch.float64).to(dt)
X, Z = rnd(N, D), rnd(M, D)
inv_ls = torch.ones(D, dtype=dt, device="cuda:0")
Kfu = torch.empty(Np, Mp, dtype=dt, device="cuda:0")
Bw = torch.empty(Np, Mp, dtype=dt, device="cuda:0")
Tl = torch.tril(rnd(Mp, Mp)) / Mp ** 0.5
Tu = torch.triu(rnd(P, Mp, Mp)) / Mp ** 0.5
gam = rnd(Mp, P)
Y = rnd(N, P)
g0 = torch.empty(Np, P, dtype=dt, device="cuda:0"); g1 = torch.empty(Np, P, dtype=dt, device="cuda:0")
vep = torch.empty(Np // 128, dtype=torch.float64, device="cuda:0"); npp = torch.empty(Np // 128, dtype=torch.int32, device="cuda:0")
nsplit = min(eng.choose_nsplit(Mp, P, Np), Np // 16)
work = torch.empty(int(eng._fn("tsvgp_site_accum_work_bytes")(Mp, P, nsplit)), dtype=torch.uint8, device="cuda:0")
acc2 = torch.empty(P, Mp, Mp, dtype=torch.float64, device="cuda:0"); acc1 = torch.empty(P, Mp, dtype=torch.float64, device="cuda:0")
st = eng._stream()
esz = 8 if a.dtype == "f64" else 4

def timeit(name, fn, flops=None, byts=None):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    extra = ""
    if flops: extra += f"  {flops / ms / 1e9:7.2f} TFLOP/s (algorithmic)"
    if byts: extra += f"  {byts / ms / 1e6:7.1f} GB/s"
    print(f"{name:28s} {ms:9.3f} ms{extra}", flush=True)

timeit("se_fill", lambda: eng.se_fill(X, Z, inv_ls, 1.0, Kfu), byts=N * M * esz)
timeit("trmm lower", lambda: eng.trmm(Kfu, Tl, Bw, B.TRI_LOWER), flops=N * M * (M + 1))
timeit("trmm upper", lambda: eng.trmm(Kfu, Tu[0], Bw, B.TRI_UPPER), flops=N * M * (M + 1))
for lik, nm in ((0, "none"), (1, "gauss"), (2, "bern")):
    timeit(f"moments upper lik={nm}", lambda: B.check(eng._fn("tsvgp_moments")(Bw.data_ptr(), Tu.data_ptr(), gam.data_ptr(), Y.data_ptr(), 1e9, lik, 0.1, None, None, g0.data_ptr(), g1.data_ptr(), vep.data_ptr(), npp.data_ptr(), N, Np, Mp, P, B.TRI_UPPER, st), "m"), flops=N * M * (M + 1) * P)
timeit("moments lower lik=gauss", lambda: B.check(eng._fn("tsvgp_moments")(Bw.data_ptr(), Tl.data_ptr(), gam.data_ptr(), Y.data_ptr(), 1e9, 1, 0.1, None, None, g0.data_ptr(), g1.data_ptr(), vep.data_ptr(), npp.data_ptr(), N, Np, Mp, 1, B.TRI_LOWER, st), "m"), flops=N * M * (M + 1))
g1.uniform_(-1.0, -0.1); g0.normal_()
for ns in [nsplit] + [int(x) for x in a.nsplits.split(",") if x]:
    work = torch.empty(int(eng._fn("tsvgp_site_accum_work_bytes")(Mp, P, ns)), dtype=torch.uint8, device="cuda:0")
    timeit(f"site_accum nsplit={ns}", lambda: B.check(eng._fn("tsvgp_site_accum")(Bw.data_ptr(), g0.data_ptr(), g1.data_ptr(), acc2.data_ptr(), acc1.data_ptr(), work.data_ptr(), Np, Mp, P, ns, st), "s"), flops=N * M * (M + 1) * P)
