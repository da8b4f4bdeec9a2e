#!/usr/bin/env python3
"""Round 5: the block step of tsvgp_potrf_f64 -- new (inverted diagonal tiles + substitution panels), old (TSVGP_POTRF_DIAG_V1:
assembled inverse + product panels) and the tile-dataflow diagonal kernel (TSVGP_POTRF_DIAG_V2) -- against torch.linalg.cholesky:
factor, factor-and-solve, failure index, timing.  GPU box.  usage: dev_diag2.py [M ...]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
B = importlib.import_module("t-svgp_amd._backend")
estep = importlib.import_module("t-svgp_amd.estep")
dev = "cuda:0"
eng = estep.EStepEngine(torch.float64, dev)
Ms = [int(a) for a in sys.argv[1:]] or [128, 256, 384, 1024]
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
torch.manual_seed(0)
ok = True
for M in Ms:
    for batch in (1, 2, 3):
        A = torch.randn(batch, M, M, dtype=torch.float64, device=dev)
        A = A @ A.transpose(-1, -2) / M + torch.eye(M, dtype=torch.float64, device=dev)
        ref = torch.linalg.cholesky(A)
        out = {}
        for name, fl in (("new", 0), ("fused", B.POTRF_FUSE), ("old", B.POTRF_DIAG_V1), ("dv2", B.POTRF_DIAG_V2)):
            eng.potrf_flags = fl
            L, info = eng.cholesky(A)
            err = ((L - ref).abs().max() / ref.abs().max()).item()
            Lr = torch.tril(torch.randn(batch, M, M, dtype=torch.float64, device=dev))
            U, info2, Dm = eng.cholesky_solve_upper(A, Lr)
            # A = U U^T, D = U^-1 Lr^T
            e_u = ((U @ U.transpose(-1, -2) - A).abs().max() / A.abs().max()).item()
            Dref = torch.linalg.solve_triangular(U, Lr.transpose(-1, -2), upper=True)
            e_d = ((Dm - Dref).abs().max() / Dref.abs().max()).item()
            t = timeit(lambda: eng.cholesky(A))
            t2 = timeit(lambda: eng.cholesky_solve_upper(A, Lr))
            out[name] = (err, e_u, e_d, t, t2, int(info.abs().sum()), int(info2.abs().sum()))
            good = err < 1e-13 and e_u < 1e-13 and e_d < 1e-11 and out[name][5] == 0
            ok &= good
            print(f"M={M} batch={batch} {name}: |L-ref| {err:.1e} |UU^T-A| {e_u:.1e} |D-ref| {e_d:.1e} info {out[name][5]},{out[name][6]}  "
                  f"cholesky {t:.1f} us  solve_upper {t2:.1f} us {'ok' if good else 'FAIL'}", flush=True)
    # failure index: a matrix that stops being definite at column c (1-based info = c)
    for c in (1, 17, 70, M - 5):
        A = torch.eye(M, dtype=torch.float64, device=dev)[None].clone() * 2.0
        A[0, c - 1, c - 1] = -1.0
        infos = []
        for fl in (0, B.POTRF_DIAG_V1, B.POTRF_DIAG_V2):
            eng.potrf_flags = fl
            _, info = eng.cholesky(A)
            infos.append(int(info[0]))
        good = infos[0] == infos[1] == infos[2] == c
        ok &= good
        print(f"M={M} first bad pivot at column {c}: info new {infos[0]} old {infos[1]} dv2 {infos[2]} {'ok' if good else 'FAIL'}", flush=True)
print("ALL OK" if ok else "FAILURES")
sys.exit(0 if ok else 1)
