#!/usr/bin/env python3
"""Bitwise repeatability and large-size correctness of the hand-laid kernels (GPU box).  A timing-dependent fault -- the class
met while building panel1_kernel: wrong values in some lanes, different from run to run (inline-asm memory operations the
compiler neither pads nor counts; round 3 blamed an MFMA operand hazard, round 4 measured that none exists:
profiles/r04_hazard_probe.txt) -- shows up as two launches on the same input disagreeing.
Every product (moments / trmm in the upper and lower form, site sums; fp64 and fp32) runs REPS times on one input: all outputs
must be bit-identical, and the first one must match a torch fp64 reference (chunked over rows).
    python tools/determinism_check.py [--rows 262144] [--M 1024] [--reps 12]"""
import argparse, importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

pkg = importlib.import_module("t-svgp_amd")
estep = importlib.import_module("t-svgp_amd.estep")
B = pkg._backend


def check(rows=262144, M_=1024, reps=12, out=print):
    """Returns the number of failed (type, product) combinations."""
    import types
    a = types.SimpleNamespace(rows=rows, M=M_, reps=reps)
    N, M = a.rows, a.M
    Np, Mp = B.round_up(N), B.round_up(M)
    bad = 0
    for dt, tol in ((torch.float64, 1e-12), (torch.float32, 2e-5)):
        eng = estep.EStepEngine(dt, "cuda:0")
        g = torch.Generator(device="cuda:0").manual_seed(1)
        rnd = lambda *s: torch.randn(*s, generator=g, device="cuda:0", dtype=torch.float64)
        A64 = torch.zeros(Np, Mp, dtype=torch.float64, device="cuda:0")
        A64[:N, :M] = rnd(N, M)
        A = A64.to(dt)
        A64 = A.double()
        Y = rnd(N, 1).to(dt)
        gam = rnd(Mp, 1).to(dt)
        st = eng._stream()
        for mode, name in ((B.TRI_UPPER, "upper"), (B.TRI_LOWER, "lower")):
            T64 = rnd(Mp, Mp) / Mp ** 0.5
            T64 = (torch.triu(T64) if mode == B.TRI_UPPER else torch.tril(T64)).to(dt).double()
            Tm = T64.to(dt).contiguous()
            # reference: C = A T^T (the triangle is explicit in T), q = row sums of squares, mean = A gamma
            ref_q = torch.empty(N, dtype=torch.float64, device="cuda:0")
            ref_C = torch.empty(Np, Mp, dtype=torch.float64, device="cuda:0")
            for lo in range(0, Np, 32768):
                ref_C[lo:lo + 32768] = A64[lo:lo + 32768] @ T64.T
            ref_q = (ref_C[:N] ** 2).sum(1)
            ref_mean = (A64[:N] @ gam.double())[:, 0]
            outs = []
            for r in range(a.reps):
                C = torch.empty(Np, Mp, dtype=dt, device="cuda:0")
                eng.trmm(A, Tm, C, mode)
                mean = torch.empty(N, 1, dtype=dt, device="cuda:0"); var = torch.empty(N, 1, dtype=dt, device="cuda:0")
                g0 = torch.empty(Np, 1, dtype=dt, device="cuda:0"); g1 = torch.empty(Np, 1, dtype=dt, device="cuda:0")
                vep = torch.empty(Np // 128, dtype=torch.float64, device="cuda:0"); npp = torch.empty(Np // 128, dtype=torch.int32, device="cuda:0")
                B.check(eng._fn("tsvgp_moments")(A.data_ptr(), Tm.data_ptr(), gam.data_ptr(), Y.data_ptr(), 0.0, 1, 0.1, mean.data_ptr(),
                                                 var.data_ptr(), g0.data_ptr(), g1.data_ptr(), vep.data_ptr(), npp.data_ptr(), N, Np, Mp, 1,
                                                 mode, st), "moments")
                torch.cuda.synchronize()
                outs.append((C, mean, var, g0, vep))
            same = all(all(torch.equal(x, y) for x, y in zip(outs[0], o)) for o in outs[1:])
            e_C = float((outs[0][0].double() - ref_C).abs().max() / ref_C.abs().max())
            e_q = float(((-outs[0][2].double()[:, 0]) - ref_q).abs().max() / ref_q.abs().max())  # var = kdiag (0) - q
            e_m = float((outs[0][1].double()[:, 0] - ref_mean).abs().max() / ref_mean.abs().max())
            ok = same and max(e_C, e_q, e_m) < tol
            bad += not ok
            out(f"{str(dt):14s} {name}: {a.reps} launches bit-identical: {same};  max rel err  trmm {e_C:.2e}  q {e_q:.2e}  mean {e_m:.2e}   {'ok' if ok else 'FAIL'}")
            del outs, ref_C
        # site sums
        P = 1
        g0 = rnd(Np, P).to(dt); g1 = (-torch.rand(Np, P, generator=g, device="cuda:0", dtype=torch.float64) - 0.1).to(dt)
        g0[N:] = 0; g1[N:] = 0
        ref2 = torch.zeros(Mp, Mp, dtype=torch.float64, device="cuda:0")
        for lo in range(0, Np, 32768):
            blk = A64[lo:lo + 32768]
            ref2 += blk.T @ (blk * g1[lo:lo + 32768].double())
        ref1 = (A64 * g0.double()).sum(0)
        nsplit = eng.choose_nsplit(Mp, P, Np)
        work = torch.empty(int(eng._fn("tsvgp_site_accum_work_bytes")(Mp, P, nsplit)), dtype=torch.uint8, device="cuda:0")
        outs = []
        for r in range(a.reps):
            acc2 = torch.empty(P, Mp, Mp, dtype=torch.float64, device="cuda:0"); acc1 = torch.empty(P, Mp, dtype=torch.float64, device="cuda:0")
            B.check(eng._fn("tsvgp_site_accum")(A.data_ptr(), g0.data_ptr(), g1.data_ptr(), acc2.data_ptr(), acc1.data_ptr(), work.data_ptr(),
                                                Np, Mp, P, nsplit, st), "site_accum")
            torch.cuda.synchronize()
            outs.append((acc2, acc1))
        same = all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs[1:])
        e2 = float((outs[0][0][0] - ref2).abs().max() / ref2.abs().max()); e1 = float((outs[0][1][0] - ref1).abs().max() / ref1.abs().max())
        ok = same and max(e2, e1) < tol * 10
        bad += not ok
        out(f"{str(dt):14s} site sums ({nsplit} slices): {a.reps} launches bit-identical: {same};  max rel err  acc2 {e2:.2e}  acc1 {e1:.2e}   {'ok' if ok else 'FAIL'}")
        del outs, A, A64
        eng.release()
        torch.cuda.empty_cache()
    return bad


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=262144)
    ap.add_argument("--M", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=12)
    args = ap.parse_args()
    sys.exit(1 if check(args.rows, args.M, args.reps, out=lambda m: print(m, flush=True)) else 0)
