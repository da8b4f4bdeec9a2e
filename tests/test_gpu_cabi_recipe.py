"""The C-ABI seam exercised the way INTEGRATION.md section B describes it: an E-step assembled from raw ``ctypes`` calls into
``libtsvgp_hip.so`` plus M x M torch algebra written straight from the reference's formulas -- no ``t-svgp_amd`` model or
engine code in between -- against the oracle.  The recipe uses the plain lower Cholesky factors a reference maintainer would
have at hand (so the moments run in TSVGP_TRI_DENSE mode) and the direct route (tsvgp.py:246-304)."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import tsvgp_oracle as O
from tests.helpers import pkg, relerr, synthetic

pytestmark = pytest.mark.gpu
vp, i64, i32, f64 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double
SE, GAUSSIAN, BERNOULLI, TRI_DENSE = 0, 1, 2, 2


def _lib():
    lib = ctypes.CDLL(pkg()._backend.LIB_PATH)
    lib.tsvgp_kernel_fill_f64.argtypes = [i32, vp, vp, vp, f64, vp, i64, i32, i32, i64, vp]
    lib.tsvgp_moments_f64.argtypes = [vp, vp, vp, vp, f64, i32, f64, vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, vp]
    lib.tsvgp_site_accum_work_bytes_f64.argtypes, lib.tsvgp_site_accum_work_bytes_f64.restype = [i32, i32, i32], i64
    lib.tsvgp_site_accum_f64.argtypes = [vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, vp]
    lib.tsvgp_potrf_inv_f64.argtypes = [vp, i32, i32, i32, i64, vp, vp, vp, vp, vp, i32, vp]
    lib.tsvgp_potrf_f64.argtypes = [vp, i32, i32, i32, i64, vp, vp, i32, vp]
    return lib


def _chol_with_inverse(lib, A):
    b, M, _ = A.shape
    L = A.clone()
    X = torch.empty(2, b, M, M, dtype=A.dtype, device=A.device)
    info = torch.empty(b, dtype=torch.int32, device=A.device)
    work = torch.empty(b, 128 * 128, dtype=A.dtype, device=A.device)
    T = torch.empty_like(L)
    assert lib.tsvgp_potrf_inv_f64(L.data_ptr(), M, M, b, M * M, info.data_ptr(), work.data_ptr(), X[0].data_ptr(),
                                   X[1].data_ptr(), T.data_ptr(), 0, None) == 0
    assert int(info.abs().sum()) == 0
    return torch.tril(L), X[0]


@pytest.mark.parametrize("lik", ["gaussian", "bernoulli"])
def test_estep_from_raw_cabi_calls_matches_oracle(lik):
    lib = _lib()
    dev, dt = "cuda:0", torch.float64
    N, M, Din, P = 1000, 128, 8, 1  # M a multiple of 128: no padding to write out; D = 8: cond(K_uu) ~ 1e1, where the
    # direct route of the recipe is accurate (it needs cond(K_uu + jitter I) <= 6e4)
    Xn, Yn, Zn = synthetic(N=N, M=M, D=Din, lik=lik, seed=4)
    X, Y, Z = (torch.as_tensor(a, device=dev) for a in (Xn, Yn, Zn))
    variance, noise, lr, jitter = 1.0, 0.1, 0.8, 1e-9
    inv_ls = torch.ones(Din, dtype=dt, device=dev)
    Np = -(-N // 128) * 128
    Id = torch.eye(M, dtype=dt, device=dev)
    l1 = torch.zeros(M, P, dtype=dt, device=dev)
    Ls = (-1e-10 * Id)[None].clone()  # tsvgp.py:174-180
    ora = O.t_SVGP(O.SquaredExponential(variance, 1.0), O.Gaussian(noise) if lik == "gaussian" else O.Bernoulli(), Zn)
    lik_id = GAUSSIAN if lik == "gaussian" else BERNOULLI
    for _ in range(2):
        # M x M prelude, from the reference's formulas (util.py:168-175, tsvgp.py:209-211, 268-270)
        Kzz = torch.empty(M, M, dtype=dt, device=dev)
        assert lib.tsvgp_kernel_fill_f64(SE, Z.data_ptr(), Z.data_ptr(), inv_ls.data_ptr(), variance, Kzz.data_ptr(), M, M, Din, M, None) == 0
        K6, K9 = Kzz + 1e-6 * Id, Kzz + jitter * Id
        W = Id + Ls[0].T @ K6 @ Ls[0]
        (cW, c9), (cWinv, c9inv) = [t for t in zip(*[_chol_with_inverse(lib, A[None]) for A in (W, K9)])]
        D = (cWinv[0] @ Ls[0].T)[None].contiguous()  # chol(W)^-1 L^T, dense
        beta = l1 - D[0].T @ (D[0] @ (K6 @ l1))  # K6^-1 m
        K9inv = c9inv[0].T @ c9inv[0]
        # N pass: fill, moments + likelihood gradients, site sums
        Kfu = torch.empty(Np, M, dtype=dt, device=dev)
        assert lib.tsvgp_kernel_fill_f64(SE, X.data_ptr(), Z.data_ptr(), inv_ls.data_ptr(), variance, Kfu.data_ptr(), N, M, Din, M, None) == 0
        g0 = torch.empty(Np, P, dtype=dt, device=dev)
        g1 = torch.empty_like(g0)
        ve = torch.empty(Np // 128, dtype=dt, device=dev)
        nonpos = torch.empty(Np // 128, dtype=torch.int32, device=dev)
        assert lib.tsvgp_moments_f64(Kfu.data_ptr(), D.data_ptr(), beta.contiguous().data_ptr(), Y.data_ptr(), variance, lik_id, noise,
                                     None, None, g0.data_ptr(), g1.data_ptr(), ve.data_ptr(), nonpos.data_ptr(), N, Np, M, P,
                                     TRI_DENSE, None) == 0
        assert int(nonpos.sum()) == 0
        nsplit = 4
        work = torch.empty(lib.tsvgp_site_accum_work_bytes_f64(M, P, nsplit), dtype=torch.uint8, device=dev)
        acc2 = torch.empty(P, M, M, dtype=dt, device=dev)
        acc1 = torch.empty(P, M, dtype=dt, device=dev)
        assert lib.tsvgp_site_accum_f64(Kfu.data_ptr(), g0.data_ptr(), g1.data_ptr(), acc2.data_ptr(), acc1.data_ptr(),
                                        work.data_ptr(), Np, M, P, nsplit, None) == 0
        # epilogue (tsvgp.py:278-303): direct projection, chain rule, convex update, -chol
        G1 = K9inv @ acc2[0] @ K9inv
        G0 = K9inv @ acc1.T
        meanZ = Kzz @ beta
        l1 = (1 - lr) * l1 + lr * (G0 - 2.0 * G1 @ meanZ)
        target = (1 - lr) * (Ls[0] @ Ls[0].T) - 2.0 * lr * 0.5 * (G1 + G1.T) + jitter * Id
        Lnew = target[None].clone()
        info = torch.empty(1, dtype=torch.int32, device=dev)
        pw = torch.empty(1, 128 * 128, dtype=dt, device=dev)
        assert lib.tsvgp_potrf_f64(Lnew.data_ptr(), M, M, 1, M * M, info.data_ptr(), pw.data_ptr(), 0, None) == 0
        assert int(info[0]) == 0
        Ls = -torch.tril(Lnew)
        ora.natgrad_step((Xn, Yn), lr=lr)
        assert relerr(l1.cpu().numpy(), ora.lambda_1) < 1e-8
        assert relerr((Ls @ Ls.transpose(-1, -2)).cpu().numpy(), ora.lambda_2) < 1e-8


def _factor_and_solve(lib, W, L):
    """INTEGRATION.md's ``factor_and_solve``: chol_W and D = chol_W^-1 L^T of src/util.py:171-175 in one pass, upper form
    (tsvgp_potrf_solve_f64 + tsvgp_flip_transpose_f64); W, L [b, M, M], M a multiple of 128."""
    b, M, _ = W.shape
    tall = torch.empty(b, 2 * M, M, dtype=W.dtype, device=W.device)
    tall[:, :M] = torch.flip(W, (-2, -1))
    tall[:, M:] = torch.flip(torch.tril(L), (-2, -1))
    info = torch.empty(b, dtype=torch.int32, device=W.device)
    work = torch.empty(b, 128 * 128, dtype=W.dtype, device=W.device)
    lib.tsvgp_potrf_solve_f64.argtypes = [vp, i32, i32, i32, i64, vp, vp, i32, i32, vp]
    assert lib.tsvgp_potrf_solve_f64(tall.data_ptr(), M, M, b, 2 * M * M, info.data_ptr(), work.data_ptr(), M, 2, None) == 0
    assert int(info.abs().sum()) == 0
    D = torch.empty(b, M, M, dtype=W.dtype, device=W.device)
    lib.tsvgp_flip_transpose_f64.argtypes = [vp, i32, i64, vp, i32, i64, i32, i32, vp]
    assert lib.tsvgp_flip_transpose_f64(tall[:, M:].data_ptr(), M, 2 * M * M, D.data_ptr(), M, M * M, M, b, None) == 0
    return D


@pytest.mark.parametrize("lik", ["gaussian", "bernoulli"])
def test_estep_from_raw_cabi_calls_with_the_round4_entry_points(lik):
    """The same E-step through the entry points round 4 added (``factor_and_solve``): the upper-form D and the
    inverse factor of K_uu + jitter I out of ONE batched factor-and-solve, beta by ``tsvgp_site_beta_f64``, the whole site update
    by ``tsvgp_site_update_f64``; the moments in TSVGP_TRI_UPPER mode.  Against the oracle (tsvgp.py:246-304)."""
    lib = _lib()
    lib.tsvgp_site_beta_f64.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp]
    lib.tsvgp_site_update_f64.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, f64, f64, vp, f64, vp]
    dev, dt = "cuda:0", torch.float64
    N, M, Din, P, TRI_UPPER = 1000, 256, 8, 1, 1
    Xn, Yn, Zn = synthetic(N=N, M=M, D=Din, lik=lik, seed=6)
    X, Y, Z = (torch.as_tensor(a, device=dev) for a in (Xn, Yn, Zn))
    variance, noise, lr, jitter = 1.0, 0.1, 0.8, 1e-9
    inv_ls = torch.ones(Din, dtype=dt, device=dev)
    Np = -(-N // 128) * 128
    Id = torch.eye(M, dtype=dt, device=dev)
    l1 = torch.zeros(M, P, dtype=dt, device=dev)
    Ls = (-1e-10 * Id)[None].clone()
    ora = O.t_SVGP(O.SquaredExponential(variance, 1.0), O.Gaussian(noise) if lik == "gaussian" else O.Bernoulli(), Zn)
    lik_id = GAUSSIAN if lik == "gaussian" else BERNOULLI
    for _ in range(2):
        Kzz = torch.empty(M, M, dtype=dt, device=dev)
        assert lib.tsvgp_kernel_fill_f64(SE, Z.data_ptr(), Z.data_ptr(), inv_ls.data_ptr(), variance, Kzz.data_ptr(), M, M, Din, M, None) == 0
        K6, K9 = Kzz + 1e-6 * Id, Kzz + jitter * Id
        W = Id + Ls[0].T @ K6 @ Ls[0]
        sol = _factor_and_solve(lib, torch.stack([W, K9]), torch.stack([Ls[0], Id]))  # [D ; U9^-1], both upper triangular
        D, U9inv = sol[:1].contiguous(), sol[1]
        K9inv = U9inv.T @ U9inv
        beta = torch.empty_like(l1)
        bw = torch.empty(P * M * (1 + (M + 63) // 64), dtype=dt, device=dev)
        assert lib.tsvgp_site_beta_f64(D.data_ptr(), (K6 @ l1).contiguous().data_ptr(), l1.data_ptr(), bw.data_ptr(), beta.data_ptr(),
                                       M, P, None) == 0
        Kfu = torch.empty(Np, M, dtype=dt, device=dev)
        assert lib.tsvgp_kernel_fill_f64(SE, X.data_ptr(), Z.data_ptr(), inv_ls.data_ptr(), variance, Kfu.data_ptr(), N, M, Din, M, None) == 0
        g0 = torch.empty(Np, P, dtype=dt, device=dev)
        g1 = torch.empty_like(g0)
        ve = torch.empty(Np // 128, dtype=dt, device=dev)
        nonpos = torch.empty(Np // 128, dtype=torch.int32, device=dev)
        assert lib.tsvgp_moments_f64(Kfu.data_ptr(), D.data_ptr(), beta.data_ptr(), Y.data_ptr(), variance, lik_id, noise, None, None,
                                     g0.data_ptr(), g1.data_ptr(), ve.data_ptr(), nonpos.data_ptr(), N, Np, M, P, TRI_UPPER, None) == 0
        assert int(nonpos.sum()) == 0
        nsplit = 4
        work = torch.empty(lib.tsvgp_site_accum_work_bytes_f64(M, P, nsplit), dtype=torch.uint8, device=dev)
        acc2 = torch.empty(P, M, M, dtype=dt, device=dev)
        acc1 = torch.empty(P, M, dtype=dt, device=dev)
        assert lib.tsvgp_site_accum_f64(Kfu.data_ptr(), g0.data_ptr(), g1.data_ptr(), acc2.data_ptr(), acc1.data_ptr(),
                                        work.data_ptr(), Np, M, P, nsplit, None) == 0
        G1 = (K9inv @ acc2[0] @ K9inv)[None].contiguous()  # direct route, tsvgp.py:279-280
        G0 = (K9inv @ acc1.T).contiguous()
        meanZ = (Kzz @ beta).contiguous()
        LLt = (Ls @ Ls.transpose(-1, -2)).contiguous()
        target, l1_new = torch.empty_like(G1), torch.empty_like(l1)
        uw = torch.empty(P * ((M + 31) // 32) * M, dtype=dt, device=dev)
        assert lib.tsvgp_site_update_f64(G1.data_ptr(), G0.data_ptr(), LLt.data_ptr(), meanZ.data_ptr(), l1.data_ptr(), target.data_ptr(),
                                         l1_new.data_ptr(), uw.data_ptr(), M, P, lr, jitter, None, 0.0, None) == 0
        l1 = l1_new
        info = torch.empty(1, dtype=torch.int32, device=dev)
        pw = torch.empty(1, 128 * 128, dtype=dt, device=dev)
        assert lib.tsvgp_potrf_f64(target.data_ptr(), M, M, 1, M * M, info.data_ptr(), pw.data_ptr(), 0, None) == 0
        assert int(info[0]) == 0
        Ls = -torch.tril(target)
        ora.natgrad_step((Xn, Yn), lr=lr)
        assert relerr(l1.cpu().numpy(), ora.lambda_1) < 1e-8
        assert relerr((Ls @ Ls.transpose(-1, -2)).cpu().numpy(), ora.lambda_2) < 1e-8
