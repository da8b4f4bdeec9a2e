"""is torch.cholesky_solve (ROCm) intermittently wrong for small batched systems? + timing at M=2000."""
import torch, time
torch.manual_seed(0)
dev = "cuda"
def chol_solve_trsm(Bm, L):
    X = torch.linalg.solve_triangular(L, Bm, upper=False)
    return torch.linalg.solve_triangular(L.transpose(-1, -2), X, upper=True)
for M, P in ((12, 2), (12, 1), (50, 2), (2000, 1)):
    A = torch.randn(M, M, dtype=torch.float64, device=dev); K = A @ A.T / M + torch.eye(M, dtype=torch.float64, device=dev)
    L = torch.linalg.cholesky(K)
    bad = 0; worst = 0.0
    reps = 300 if M < 100 else 20
    for it in range(reps):
        Bm = torch.randn(P, M, M, dtype=torch.float64, device=dev)
        Bm = Bm @ Bm.transpose(-1, -2)        # produced by a kernel right before the solve
        X1 = torch.cholesky_solve(Bm.contiguous(), L)
        G1 = torch.cholesky_solve(X1.transpose(-1, -2).contiguous(), L)
        R = chol_solve_trsm(chol_solve_trsm(Bm, L).transpose(-1, -2), L)
        e = float((G1 - R).abs().max() / R.abs().max())
        worst = max(worst, e); bad += e > 1e-10
    print(f"M={M} P={P}: {bad}/{reps} mismatches, worst rel diff {worst:.2e}")
    for name, fn in (("cholesky_solve", lambda: torch.cholesky_solve(torch.cholesky_solve(Bm, L).transpose(-1, -2).contiguous(), L)),
                     ("2x2 trsm", lambda: chol_solve_trsm(chol_solve_trsm(Bm, L).transpose(-1, -2), L))):
        fn(); torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize(); print(f"   {name}: {(time.perf_counter() - t) * 100:.3f} ms")
