"""Kernel-level parity on the GPU: every C-ABI entry point against NumPy / the oracle on the same seeded inputs.

Tolerances: fp64 <= 1e-11 relative (pure rounding); fp32 stated per test (inputs rounded to fp32, fp32 MFMA).
"""
import numpy as np
import pytest
import torch

from oracle import tsvgp_oracle as O
from tests.helpers import pkg, relerr

pytestmark = pytest.mark.gpu

DTYPES = [(torch.float64, 1e-11), (torch.float32, 2e-5)]


@pytest.fixture(scope="module")
def engines():
    from importlib import import_module

    estep = import_module("t-svgp_amd.estep")
    return {dt: estep.EStepEngine(dt, "cuda:0") for dt, _ in DTYPES}


@pytest.mark.parametrize("dtype,tol", DTYPES)
def test_mfma_fragment_maps(engines, dtype, tol):
    """A (16x4) @ B (4x16) with asymmetric random operands: catches swapped row/col maps."""
    a, b, c = engines[dtype].selftest_mfma(dtype)
    ref = a.double().cpu().numpy() @ b.double().cpu().numpy()
    assert relerr(c.cpu().numpy(), ref) < tol


@pytest.mark.parametrize("dtype,tol", DTYPES)
@pytest.mark.parametrize("N,M,D", [(1, 1, 1), (130, 32, 1), (1000, 200, 8), (257, 513, 16), (64, 1024, 19)])
def test_se_fill(engines, dtype, tol, N, M, D):
    eng = engines[dtype]
    rng = np.random.RandomState(1)
    X, Z = rng.randn(N, D), rng.randn(M, D)
    ls = 0.7 + rng.rand(D)
    var = 1.3
    B = pkg()._backend
    Np, Mp = B.round_up(N), B.round_up(M)
    out = torch.full((Np, Mp), float("nan"), dtype=dtype, device="cuda:0")
    t = lambda a: torch.as_tensor(a, dtype=dtype, device="cuda:0").contiguous()
    eng.se_fill(t(X), t(Z), t(1.0 / ls), var, out)
    K = out.double().cpu().numpy()
    ref = O.SquaredExponential(variance=var, lengthscales=ls).K(X, Z)
    assert relerr(K[:N, :M], ref) < tol * 10
    assert np.all(K[N:, :] == 0) and np.all(K[:, M:] == 0)  # zero-filled padding


def test_se_fill_exp_is_the_library_exp_bit_for_bit(engines):
    """The fill's own fp64 exp (same reduction and polynomial as the device library's, fewer instructions) against
    torch.exp on the device, D = 1 and Z = 0 so that the exponent's argument is formed identically on both sides (the expanded
    form x^2 + z^2 - 2 x z is then x^2 exactly): exact equality over arguments from 0 down into the underflow range, and zeros
    (not NaN) far below it."""
    eng = engines[torch.float64]
    B = pkg()._backend
    N, M = 4096, 128
    g = torch.Generator(device="cpu").manual_seed(5)
    X = (torch.rand(N, 1, generator=g, dtype=torch.float64) * 80.0).to("cuda:0")  # -0.5 s down to -3200
    X[:8, 0] = torch.tensor([0.0, 1e-9, 1.0, 38.6, 38.7, 46.4, 46.6, 1e6], dtype=torch.float64)
    Z = torch.zeros(M, 1, dtype=torch.float64, device="cuda:0")
    out = torch.empty((B.round_up(N), B.round_up(M)), dtype=torch.float64, device="cuda:0")
    eng.se_fill(X, Z, torch.ones(1, dtype=torch.float64, device="cuda:0"), 1.0, out)
    ref = torch.exp(-0.5 * (X * X)).expand(N, M)
    assert torch.equal(out[:N, :M], ref)
    assert float(out[7, 0]) == 0.0 and int((ref == 0).sum()) > 0 and int((ref > 0.5).sum()) > 0
    # the product's own form, sum_d ((x - z) / l)^2 with NON-ZERO Z (unit lengthscale: the difference and its square are the
    # same fp64 operations on either side): bit for bit again, down into the underflow range
    Zr = (torch.rand(M, 1, generator=g, dtype=torch.float64) * 40.0).to("cuda:0")
    eng.se_fill(X, Zr, torch.ones(1, dtype=torch.float64, device="cuda:0"), 1.0, out)
    d = X - Zr.reshape(1, M)
    ref2 = torch.exp(-0.5 * (d * d))
    assert torch.equal(out[:N, :M], ref2)
    assert int((ref2 == 0).sum()) > 0 and int((ref2 > 0.5).sum()) > 0


@pytest.mark.parametrize("dtype,tol", DTYPES)
@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("Np,Mp", [(128, 128), (384, 256), (256, 640), (256, 1024)])
def test_trmm(engines, dtype, tol, mode, Np, Mp):
    eng = engines[dtype]
    rng = np.random.RandomState(2)
    A = rng.randn(Np, Mp)
    Tm = rng.randn(Mp, Mp)
    mask = {0: np.tril(np.ones((Mp, Mp))), 1: np.triu(np.ones((Mp, Mp))), 2: np.ones((Mp, Mp))}[mode]
    # the kernel's k-range is tile-granular: feed a matrix that is zero outside the intended triangle
    Tm_m = Tm * mask
    t = lambda a: torch.as_tensor(a, dtype=dtype, device="cuda:0").contiguous()
    C = torch.empty((Np, Mp), dtype=dtype, device="cuda:0")
    eng.trmm(t(A), t(Tm_m), C, mode)
    ref = t(A).double().cpu().numpy() @ t(Tm_m).double().cpu().numpy().T
    assert relerr(C.cpu().numpy(), ref) < tol * 20


def _moments_ref(A, Tm, gamma, kdiag):
    C = np.einsum("nj,pij->pni", A, Tm)
    q = np.sum(C * C, axis=-1).T  # [N, P]
    return A @ gamma, kdiag - q


@pytest.mark.parametrize("dtype,tol", DTYPES)
@pytest.mark.parametrize("lik", ["none", "gaussian", "bernoulli"])
@pytest.mark.parametrize("N,M,P,mode", [(100, 128, 1, 2), (300, 256, 2, 1), (129, 384, 3, 1),
                                        (600, 1024, 1, 1), (300, 1024, 8, 1),  # these two: the benchmark's 8 x 8 tile grid
                                        (300, 256, 2, 0), (129, 384, 3, 0), (600, 1024, 1, 0)])  # the lower form
def test_moments_and_likelihood_map(engines, dtype, tol, lik, N, M, P, mode):
    eng = engines[dtype]
    B = pkg()._backend
    rng = np.random.RandomState(3)
    Np = B.round_up(N)
    A = np.zeros((Np, M))
    A[:N] = rng.randn(N, M) / np.sqrt(M)
    Tm = rng.randn(P, M, M) * 0.5
    if mode == 1:
        Tm = np.triu(Tm)
    elif mode == 0:
        Tm = np.tril(Tm)
    gamma = rng.randn(M, P)
    kdiag = 2.5
    Y = (rng.rand(N, P) > 0.5).astype(float) if lik == "bernoulli" else rng.randn(N, P)
    t = lambda a: torch.as_tensor(a, dtype=dtype, device="cuda:0").contiguous()
    At, Tmt, gt, Yt = t(A), t(Tm), t(gamma), t(Y)
    mean = torch.empty((N, P), dtype=dtype, device="cuda:0")
    var = torch.empty((N, P), dtype=dtype, device="cuda:0")
    g0 = torch.full((Np, P), float("nan"), dtype=dtype, device="cuda:0")
    g1 = torch.full((Np, P), float("nan"), dtype=dtype, device="cuda:0")
    vep = torch.zeros(Np // 128, dtype=torch.float64, device="cuda:0")
    npp = torch.zeros(Np // 128, dtype=torch.int32, device="cuda:0")
    lik_id = {"none": 0, "gaussian": 1, "bernoulli": 2}[lik]
    fn = eng._fn("tsvgp_moments")
    B.check(fn(At.data_ptr(), Tmt.data_ptr(), gt.data_ptr(), Yt.data_ptr(), kdiag, lik_id, 0.3, mean.data_ptr(),
               var.data_ptr(), g0.data_ptr(), g1.data_ptr(), vep.data_ptr(), npp.data_ptr(), N, Np, M, P, mode,
               eng._stream()), "moments")
    torch.cuda.synchronize()
    mref, vref = _moments_ref(At.double().cpu().numpy()[:N], Tmt.double().cpu().numpy(), gt.double().cpu().numpy(), kdiag)
    assert relerr(mean.cpu().numpy(), mref) < tol * 20
    assert np.max(np.abs(var.double().cpu().numpy() - vref)) < tol * 20 * kdiag
    assert int(npp.sum()) == int(np.sum(vref <= 0))
    if lik == "none":
        return
    ok = vref > 0  # the likelihood map is only defined for positive variance
    mu_k, var_k = mean.double().cpu().numpy(), var.double().cpu().numpy()
    olik = O.Gaussian(variance=0.3) if lik == "gaussian" else O.Bernoulli()
    vs = np.where(ok, var_k, 1.0)
    r0, r1 = olik.variational_expectations_grads(mu_k, vs, Y)
    r1 = np.minimum(r1, -1e-8)
    k0, k1 = g0.double().cpu().numpy(), g1.double().cpu().numpy()
    ltol = 1e-10 if dtype == torch.float64 else 1e-5
    np.testing.assert_allclose(k0[:N][ok], r0[ok], rtol=ltol, atol=ltol)
    np.testing.assert_allclose(k1[:N][ok], r1[ok], rtol=ltol, atol=ltol)
    assert np.all(k0[N:] == 0) and np.all(k1[N:] == 0)
    if ok.all():
        ve_ref = np.sum(olik.variational_expectations(mu_k, var_k, Y))
        assert abs(float(vep.sum()) - ve_ref) < 1e-9 * max(1.0, abs(ve_ref))


@pytest.mark.parametrize("dtype,tol", DTYPES)
@pytest.mark.parametrize("lik", ["none", "gaussian"])
@pytest.mark.parametrize("N,M,P", [(100, 128, 1), (300, 256, 2), (129, 384, 3), (1000, 1024, 1), (150, 1152, 8)])
def test_moments_mean_only(engines, dtype, tol, lik, N, M, P):
    """TSVGP_LIK_MEANONLY: mean and the Gaussian gradient map without the variance product; Tm is not read."""
    eng = engines[dtype]
    B = pkg()._backend
    rng = np.random.RandomState(5)
    Np = B.round_up(N)
    A = np.zeros((Np, M))
    A[:N] = rng.randn(N, M) / np.sqrt(M)
    gamma, Y = rng.randn(M, P), rng.randn(N, P)
    t = lambda a: torch.as_tensor(a, dtype=dtype, device="cuda:0").contiguous()
    At, gt, Yt = t(A), t(gamma), t(Y)
    mean = torch.empty((N, P), dtype=dtype, device="cuda:0")
    g0 = torch.full((Np, P), float("nan"), dtype=dtype, device="cuda:0")
    g1 = torch.full((Np, P), float("nan"), dtype=dtype, device="cuda:0")
    vep = torch.zeros(Np // 128, dtype=torch.float64, device="cuda:0")
    npp = torch.full((Np // 128,), 7, dtype=torch.int32, device="cuda:0")
    lik_id = {"none": 0, "gaussian": 1}[lik] | B.LIK_MEANONLY
    fn = eng._fn("tsvgp_moments")
    call = lambda a_ptr, lid, var_ptr: fn(a_ptr, None, gt.data_ptr(), Yt.data_ptr(), 2.5, lid, 0.3, mean.data_ptr(), var_ptr,
                                          g0.data_ptr(), g1.data_ptr(), vep.data_ptr(), npp.data_ptr(), N, Np, M, P, 1,
                                          eng._stream())
    B.check(call(At.data_ptr(), lik_id, None), "moments (mean only)")
    torch.cuda.synchronize()
    mref = At.double().cpu().numpy()[:N] @ gt.double().cpu().numpy()
    assert relerr(mean.cpu().numpy(), mref) < tol * 20
    assert int(npp.sum()) == 0 and torch.isnan(vep).all()
    if lik == "gaussian":
        mu = mean.double().cpu().numpy()
        ltol = 1e-12 if dtype == torch.float64 else 1e-5
        np.testing.assert_allclose(g0.double().cpu().numpy()[:N], (Yt.double().cpu().numpy() - mu) / 0.3, rtol=ltol, atol=ltol)
        np.testing.assert_allclose(g1.double().cpu().numpy()[:N], -0.5 / 0.3, rtol=ltol)
        assert np.all(g0.cpu().numpy()[N:] == 0) and np.all(g1.cpu().numpy()[N:] == 0)
    # a non-finite row is counted (the step's status check then raises as for a non-positive variance)
    Abad = At.clone()
    Abad[N // 2, 3] = float("nan")
    B.check(call(Abad.data_ptr(), lik_id, None), "moments (mean only)")
    assert int(npp.sum()) == P
    # rejected: Bernoulli (its gradients need the variance), a variance output
    assert call(At.data_ptr(), 2 | B.LIK_MEANONLY, None) == 1
    assert call(At.data_ptr(), lik_id, mean.data_ptr()) == 1


@pytest.mark.parametrize("dtype,tol", DTYPES)
@pytest.mark.parametrize("Np,Mp,P,nsplit", [(128, 128, 1, 1), (1024, 256, 2, 3), (640, 384, 1, 7), (4096, 128, 3, 64),
                                            (1024, 1024, 1, 5), (640, 1024, 8, 3),  # the benchmark's 36 lower tiles
                                            (128, 256, 1, 8), (256, 128, 2, 5)])  # slices of one chunk, and empty ones
def test_site_accum(engines, dtype, tol, Np, Mp, P, nsplit):
    eng = engines[dtype]
    B = pkg()._backend
    rng = np.random.RandomState(4)
    Bm = rng.randn(Np, Mp)
    g0 = rng.randn(Np, P)
    g1 = -rng.rand(Np, P) - 0.1
    g0[-5:] = 0
    g1[-5:] = 0
    t = lambda a: torch.as_tensor(a, dtype=dtype, device="cuda:0").contiguous()
    Bt, g0t, g1t = t(Bm), t(g0), t(g1)
    nbytes = int(eng._fn("tsvgp_site_accum_work_bytes")(Mp, P, nsplit))
    work = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
    acc2 = torch.full((P, Mp, Mp), float("nan"), dtype=torch.float64, device="cuda:0")
    acc1 = torch.full((P, Mp), float("nan"), dtype=torch.float64, device="cuda:0")
    B.check(eng._fn("tsvgp_site_accum")(Bt.data_ptr(), g0t.data_ptr(), g1t.data_ptr(), acc2.data_ptr(), acc1.data_ptr(),
                                        work.data_ptr(), Np, Mp, P, nsplit, eng._stream()), "site_accum")
    torch.cuda.synchronize()
    Bd, g0d, g1d = Bt.double().cpu().numpy(), g0t.double().cpu().numpy(), g1t.double().cpu().numpy()
    ref2 = np.einsum("nm,no,nl->lmo", Bd, Bd, g1d)
    ref1 = np.einsum("nm,nl->lm", Bd, g0d)
    a2 = acc2.cpu().numpy()
    assert relerr(a2, ref2) < tol * 50
    assert relerr(acc1.cpu().numpy(), ref1) < tol * 50
    assert np.array_equal(a2, np.swapaxes(a2, -1, -2))  # exactly symmetric
    # fixed-order reduction: bitwise reproducible
    acc2b = torch.empty_like(acc2)
    acc1b = torch.empty_like(acc1)
    B.check(eng._fn("tsvgp_site_accum")(Bt.data_ptr(), g0t.data_ptr(), g1t.data_ptr(), acc2b.data_ptr(), acc1b.data_ptr(),
                                        work.data_ptr(), Np, Mp, P, nsplit, eng._stream()), "site_accum")
    torch.cuda.synchronize()
    assert torch.equal(acc2, acc2b) and torch.equal(acc1, acc1b)


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-12), (torch.float32, 1e-5)])
@pytest.mark.parametrize("P", [1, 9])
def test_site_accum_more_slices_than_chunks(engines, dtype, tol, P):
    """``nsplit`` above Np / chunk (16 rows in fp64, 32 in fp32 with P <= 8, 16 in the P > 8 kernel): workgroups whose slice is
    EMPTY contribute exact zeros -- correct, only wasted (include/tsvgp_hip.h: any nsplit >= 1 is valid)."""
    eng = engines[dtype]
    B = pkg()._backend
    Np, Mp = 256, 256
    rng = np.random.RandomState(8)
    t = lambda a: torch.as_tensor(a, dtype=dtype, device="cuda:0").contiguous()
    Bt, g0t, g1t = t(rng.randn(Np, Mp)), t(rng.randn(Np, P)), t(-rng.rand(Np, P) - 0.1)
    Bd, g0d, g1d = Bt.double().cpu().numpy(), g0t.double().cpu().numpy(), g1t.double().cpu().numpy()
    ref2, ref1 = np.einsum("nm,no,nl->lmo", Bd, Bd, g1d), np.einsum("nm,nl->lm", Bd, g0d)
    for nsplit in (Np // 32 + 3, Np // 16 + 5, 40):  # beyond the fp32 chunk count, beyond the fp64 one, far beyond both
        nbytes = int(eng._fn("tsvgp_site_accum_work_bytes")(Mp, P, nsplit))
        work = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
        acc2 = torch.full((P, Mp, Mp), float("nan"), dtype=torch.float64, device="cuda:0")
        acc1 = torch.full((P, Mp), float("nan"), dtype=torch.float64, device="cuda:0")
        B.check(eng._fn("tsvgp_site_accum")(Bt.data_ptr(), g0t.data_ptr(), g1t.data_ptr(), acc2.data_ptr(), acc1.data_ptr(),
                                            work.data_ptr(), Np, Mp, P, nsplit, eng._stream()), "site_accum")
        torch.cuda.synchronize()
        assert relerr(acc2.cpu().numpy(), ref2) < tol * 50 and relerr(acc1.cpu().numpy(), ref1) < tol * 50, nsplit


def test_site_accum_rejects_misaligned_operands(engines):
    """B, g0, g1 travel by 16-byte LDS-DMA pieces: a view that starts off a 16-byte boundary is an invalid argument, not a
    memory fault (include/tsvgp_hip.h (5))."""
    eng = engines[torch.float64]
    Np, Mp, P = 128, 128, 1
    Bt = torch.zeros(Np * Mp + 2, dtype=torch.float64, device="cuda:0")
    g = torch.zeros(Np * P + 2, dtype=torch.float64, device="cuda:0")
    acc2 = torch.zeros((P, Mp, Mp), dtype=torch.float64, device="cuda:0")
    acc1 = torch.zeros((P, Mp), dtype=torch.float64, device="cuda:0")
    work = torch.empty(int(eng.lib.tsvgp_site_accum_work_bytes_f64(Mp, P, 1)), dtype=torch.uint8, device="cuda:0")
    call = lambda b, g0, g1: eng.lib.tsvgp_site_accum_f64(b, g0, g1, acc2.data_ptr(), acc1.data_ptr(), work.data_ptr(), Np, Mp,
                                                           P, 1, eng._stream())
    assert call(Bt.data_ptr(), g.data_ptr(), g.data_ptr()) == 0
    assert call(Bt.data_ptr() + 8, g.data_ptr(), g.data_ptr()) == 1
    assert call(Bt.data_ptr(), g.data_ptr() + 8, g.data_ptr()) == 1
    assert call(Bt.data_ptr(), g.data_ptr(), g.data_ptr() + 8) == 1
    torch.cuda.synchronize()


@pytest.mark.parametrize("M,P,num_data", [(96, 1, None), (200, 3, 5000.0), (1024, 1, 1.0e6), (33, 2, 77.0)])
def test_site_update_and_target_kernels(engines, M, P, num_data):
    """``tsvgp_site_update_f64`` (one launch: symmetrised G1 -> the matrix of the final factorisation, the chain rule of reference
    src/util.py:429-438 and the convex update of lambda_1, src/models/tsvgp.py:284-297) and ``tsvgp_site_target_f64`` against
    NumPy; repeated launches bit-identical (fixed-order row sums)."""
    eng = engines[torch.float64]
    rng = np.random.RandomState(M + P)
    G1, LLt = rng.randn(P, M, M), rng.randn(P, M, M)
    LLt = LLt @ np.swapaxes(LLt, -1, -2)
    G0, meanZ, l1 = rng.randn(M, P), rng.randn(M, P), rng.randn(M, P)
    lr, jitter, rows_n = 0.7, 1e-9, 1234.0
    t = lambda a: torch.as_tensor(a, device="cuda:0")
    rows = torch.full((), rows_n, dtype=torch.float64, device="cuda:0")
    target, l1_new = eng.site_update(t(G1), t(G0), t(LLt), t(meanZ), t(l1), lr, jitter, rows, num_data)
    s = 1.0 if num_data is None else num_data / rows_n
    Gs = 0.5 * (G1 + np.swapaxes(G1, -1, -2))
    want_t = (1 - lr) * LLt - 2 * lr * s * Gs + jitter * np.eye(M)
    want_l = (1 - lr) * l1 + lr * s * (G0 - 2.0 * np.einsum("pmo,op->mp", Gs, meanZ))
    assert relerr(target.cpu().numpy(), want_t) < 1e-14
    assert relerr(l1_new.cpu().numpy(), want_l) < 1e-13
    target2, l1_new2 = eng.site_update(t(G1), t(G0), t(LLt), t(meanZ), t(l1), lr, jitter, rows, num_data)
    assert torch.equal(target, target2) and torch.equal(l1_new, l1_new2)
    tt, Gsym = eng.site_target(t(G1), t(LLt), 1 - lr, -2 * lr, jitter, rows, num_data)
    assert relerr(tt.cpu().numpy(), want_t) < 1e-14 and relerr(Gsym.cpu().numpy(), Gs) < 1e-15


@pytest.mark.parametrize("M,P", [(96, 1), (200, 3), (1024, 1), (33, 2), (1, 1)])
def test_site_beta_kernel(engines, M, P):
    """``tsvgp_site_beta_f64``: beta = l1 - D^T (D v) with D upper triangular (what lies below the diagonal is never read) --
    K^-1 m of reference src/util.py:176-179 -- against NumPy; repeated launches bit-identical."""
    eng = engines[torch.float64]
    rng = np.random.RandomState(3 * M + P)
    D = np.triu(rng.randn(P, M, M)) / np.sqrt(M)
    junk = D + np.tril(rng.randn(P, M, M), -1)
    v, l1 = rng.randn(M, P), rng.randn(M, P)
    t = lambda a: torch.as_tensor(a, device="cuda:0")
    beta = eng.site_beta(t(junk), t(v), t(l1))
    want = l1 - np.einsum("pij,pi->jp", D, np.einsum("pij,jp->pi", D, v))
    assert relerr(beta.cpu().numpy(), want) < 1e-13
    assert torch.equal(beta, eng.site_beta(t(junk), t(v), t(l1)))


def test_invalid_arguments_are_rejected(engines):
    eng = engines[torch.float64]
    lib = eng.lib
    x = torch.zeros(128 * 128, dtype=torch.float64, device="cuda:0")
    assert lib.tsvgp_trmm_f64(x.data_ptr(), x.data_ptr(), x.data_ptr(), 100, 128, 0, None) == 1  # Np not padded
    assert lib.tsvgp_trmm_f64(None, x.data_ptr(), x.data_ptr(), 128, 128, 0, None) == 1
    assert lib.tsvgp_se_fill_f64(x.data_ptr(), x.data_ptr(), x.data_ptr(), 1.0, x.data_ptr(), 10, 4, 2, 100, None) == 1
    assert lib.tsvgp_site_accum_work_bytes_f64(100, 1, 1) == -1


@pytest.mark.parametrize("robust", [False, True])
@pytest.mark.parametrize("M,batch", [(128, 1), (256, 3), (1024, 1), (200, 2), (33, 1)])
def test_potrf(engines, M, batch, robust):
    """Blocked Cholesky (tsvgp_potrf_f64) vs LAPACK, with the panels solved by the inverted diagonal block (default) and
    by substitution (TSVGP_POTRF_SUBST); non-positive-definite input reports info like potrf."""
    eng = engines[torch.float64]
    rng = np.random.RandomState(5)
    A = rng.randn(batch, M, M)
    A = A @ np.swapaxes(A, -1, -2) / M + 0.5 * np.eye(M)
    L, info = eng.cholesky(torch.as_tensor(A, device="cuda:0"), robust=robust)
    torch.cuda.synchronize()
    assert int(info.abs().sum()) == 0
    ref = np.linalg.cholesky(A)
    assert relerr(L.cpu().numpy(), ref) < 1e-12
    assert np.array_equal(np.triu(L.cpu().numpy(), 1), np.zeros_like(A))
    bad = A.copy()
    bad[0, M // 2, M // 2] = -1.0
    _, info = eng.cholesky(torch.as_tensor(bad, device="cuda:0"), robust=robust)
    assert int(info[0]) == M // 2 + 1  # 1-based index of the first non-positive pivot


@pytest.mark.parametrize("M", [256, 640])
def test_potrf_substitution_panels_on_a_barely_definite_matrix(engines, M):
    """cond ~ 1e14 with lambda_min a few tens of eps * lambda_max (the new Lambda_2 on an ill-conditioned K_uu): wherever
    LAPACK's factorisation goes through, the one with substitution panels does too, with a backward error at rounding
    level, and so does its inverse factor."""
    eng = engines[torch.float64]
    rng = np.random.RandomState(8)
    Q, _ = np.linalg.qr(rng.randn(M, M))
    lam = np.logspace(0, -14.2, M)
    A = (Q * lam) @ Q.T
    A = 0.5 * (A + A.T)
    ref = np.linalg.cholesky(A)  # LAPACK goes through (raises otherwise)
    L, info, Linv = eng.cholesky(torch.as_tensor(A, device="cuda:0"), inverse=True, robust=True)
    assert int(info.abs().sum()) == 0
    Ln = L.cpu().numpy()
    assert np.max(np.abs(Ln @ Ln.T - A)) < 50 * np.finfo(float).eps * np.max(np.abs(A))
    assert np.max(np.abs(Ln - ref)) < 1e-6 * np.max(np.abs(ref))  # the factor itself is only determined to ~cond * eps
    X = Linv.cpu().numpy()
    assert np.max(np.abs(X @ Ln - np.eye(M))) < 1e-6


@pytest.mark.parametrize("M,batch", [(128, 1), (256, 3), (1024, 1), (640, 2), (1536, 1), (200, 2), (33, 1)])
def test_potrf_inverse(engines, M, batch):
    """tsvgp_potrf_inv_f64: the factor and its inverse (2x2 block recursion on the inverted diagonal blocks), also for
    block counts that are not powers of two."""
    eng = engines[torch.float64]
    rng = np.random.RandomState(6)
    A = rng.randn(batch, M, M)
    A = A @ np.swapaxes(A, -1, -2) / M + 0.5 * np.eye(M)
    L, info, Linv = eng.cholesky(torch.as_tensor(A, device="cuda:0"), inverse=True, robust=(M % 3 == 0))
    torch.cuda.synchronize()
    assert int(info.abs().sum()) == 0
    ref = np.linalg.cholesky(A)
    assert relerr(L.cpu().numpy(), ref) < 1e-12
    X = Linv.cpu().numpy()
    assert np.array_equal(np.triu(X, 1), np.zeros_like(A))
    assert relerr(X, np.linalg.inv(ref)) < 1e-11
    assert np.max(np.abs(X @ ref - np.eye(M))) < 1e-11


@pytest.mark.parametrize("robust", [False, True])
@pytest.mark.parametrize("M,batch", [(128, 1), (256, 3), (1024, 1), (640, 2), (200, 2), (33, 1)])
def test_potrf_solve_upper(engines, M, batch, robust):
    """tsvgp_potrf_solve_f64 + tsvgp_flip_transpose_f64 (``EStepEngine.cholesky_solve_upper``): the upper-form factor A = U U^T and
    D = U^-1 L^T for a lower triangular L in ONE pass (the right-hand side rides through the factorisation as panel rows) --
    reference src/util.py:168-175 (chol_W, then triangular_solve(chol_W, L^T)) -- against NumPy; D exactly upper triangular;
    the upper part of the L it is handed is ignored (sites.py:63: only the lower triangle is the parameter); a non-positive
    pivot reports info like potrf."""
    eng = engines[torch.float64]
    rng = np.random.RandomState(7 + M)
    A = rng.randn(batch, M, M)
    A = A @ np.swapaxes(A, -1, -2) / M + 0.5 * np.eye(M)
    L = np.tril(rng.randn(batch, M, M)) / np.sqrt(M)
    junk = L + np.triu(rng.randn(batch, M, M), 1)  # what sits above the diagonal must not matter
    U, info, Dm = eng.cholesky_solve_upper(torch.as_tensor(A, device="cuda:0"), torch.as_tensor(junk, device="cuda:0"), robust=robust)
    torch.cuda.synchronize()
    assert int(info.abs().sum()) == 0
    J = np.eye(M)[::-1]
    C = np.linalg.cholesky(J @ A @ J)
    U_ref = J @ C @ J  # A = U U^T, U upper
    assert relerr(U.cpu().numpy(), U_ref) < 1e-12
    D_ref = np.linalg.solve(U_ref, np.swapaxes(L, -1, -2))
    Dn = Dm.cpu().numpy()
    assert relerr(Dn, D_ref) < 1e-11
    assert np.array_equal(np.tril(Dn, -1), np.zeros_like(Dn))
    # the variance identity the moments kernel relies on: D^T D = L W^-1 L^T (util.py:176-184)
    assert relerr(np.swapaxes(Dn, -1, -2) @ Dn, L @ np.linalg.solve(A, np.swapaxes(L, -1, -2))) < 1e-10
    bad = A.copy()
    bad[0, M // 2, M // 2] = -1.0
    _, info, _ = eng.cholesky_solve_upper(torch.as_tensor(bad, device="cuda:0"), torch.as_tensor(L, device="cuda:0"), robust=robust)
    assert int(info[0]) != 0 and (batch == 1 or int(info[1:].abs().sum()) == 0)


@pytest.mark.parametrize("dtype,tol", DTYPES)
@pytest.mark.parametrize("name,kind", [("Matern32", 2), ("Matern52", 3)])
def test_matern_fill(engines, dtype, tol, name, kind):
    """tsvgp_kernel_fill_*: the Matern profiles (GPflow [ext] definitions, r = sqrt(max(r2, 1e-36))) against the oracle,
    including K(Z, Z) whose diagonal sits at r = 0."""
    eng = engines[dtype]
    rng = np.random.RandomState(3)
    N, M, D = 300, 70, 4
    X, Z = rng.randn(N, D), rng.randn(M, D)
    ls = 0.7 + rng.rand(D)
    B = pkg()._backend
    t = lambda a: torch.as_tensor(a, dtype=dtype, device="cuda:0").contiguous()
    ker = getattr(O, name)(variance=1.3, lengthscales=ls)
    for A in (X, Z):
        out = torch.full((B.round_up(A.shape[0]), B.round_up(M)), float("nan"), dtype=dtype, device="cuda:0")
        eng.se_fill(t(A), t(Z), t(1.0 / ls), 1.3, out, kind)
        K = out.double().cpu().numpy()
        assert relerr(K[:A.shape[0], :M], ker.K(A, Z)) < tol * 10
        assert np.all(K[A.shape[0]:, :] == 0) and np.all(K[:, M:] == 0)


# ---------------------------------------------------------------------------------------------------------------------
# latent-batched entry points (BASELINE configs[4]): one launch over P latents with one kernel each
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype,tol", DTYPES)
def test_kernel_fill_batched(engines, dtype, tol):
    import ctypes

    eng = engines[dtype]
    B = pkg()._backend
    rng = np.random.RandomState(11)
    N, M, D, P = 300, 200, 5, 3
    X, Z = rng.randn(N, D), rng.randn(M, D)
    ls = 0.7 + rng.rand(P, D)
    var = [1.3, 0.6, 2.0]
    Np, Mp = B.round_up(N), B.round_up(M)
    t = lambda a: torch.as_tensor(a, dtype=dtype, device="cuda:0").contiguous()
    out = torch.full((P, Np, Mp), float("nan"), dtype=dtype, device="cuda:0")
    ctype = ctypes.c_double if dtype == torch.float64 else ctypes.c_float
    Xt, Zt, il = t(X), t(Z), t(1.0 / ls)
    B.check(eng._fn("tsvgp_kernel_fill_batched")(B.KERNEL_SE, Xt.data_ptr(), Zt.data_ptr(), il.data_ptr(), (ctype * P)(*var),
                                                 out.data_ptr(), Np * Mp, N, M, D, Mp, P, eng._stream()), "fill batched")
    K = out.double().cpu().numpy()
    for p in range(P):
        ref = O.SquaredExponential(variance=var[p], lengthscales=ls[p]).K(X, Z)
        assert relerr(K[p, :N, :M], ref) < tol * 10
        assert np.all(K[p, N:, :] == 0) and np.all(K[p, :, M:] == 0)
    # rejected: overlapping latent slices, too many latents
    fn = eng._fn("tsvgp_kernel_fill_batched")
    assert fn(B.KERNEL_SE, Xt.data_ptr(), Zt.data_ptr(), il.data_ptr(), (ctype * P)(*var), out.data_ptr(), Np * Mp - 2, N, M, D,
              Mp, P, eng._stream()) == 1
    assert fn(B.KERNEL_SE, Xt.data_ptr(), Zt.data_ptr(), il.data_ptr(), (ctype * P)(*var), out.data_ptr(), Np * Mp, N, M, D,
              Mp, B.MAX_BATCH + 1, eng._stream()) == 1


@pytest.mark.parametrize("dtype,tol", DTYPES)
@pytest.mark.parametrize("mode,inplace", [(0, False), (1, False), (1, True), (2, False)])
def test_trmm_batched(engines, dtype, tol, mode, inplace):
    """tsvgp_trmm_batched_*: per-latent operands and factors, a shared operand (strideA = 0), and the in-place upper form."""
    eng = engines[dtype]
    B = pkg()._backend
    rng = np.random.RandomState(12)
    Np, Mp, P = 384, 256, 3
    A = rng.randn(P, Np, Mp)
    mask = {0: np.tril(np.ones((Mp, Mp))), 1: np.triu(np.ones((Mp, Mp))), 2: np.ones((Mp, Mp))}[mode]
    Tm = rng.randn(P, Mp, Mp) * mask
    t = lambda a: torch.as_tensor(a, dtype=dtype, device="cuda:0").contiguous()
    At, Tt = t(A), t(Tm)
    ref = np.einsum("pnj,pij->pni", At.double().cpu().numpy(), Tt.double().cpu().numpy())
    fn = eng._fn("tsvgp_trmm_batched")
    if inplace:
        C = At.clone()
        B.check(fn(C.data_ptr(), Np * Mp, Tt.data_ptr(), Mp * Mp, C.data_ptr(), Np * Mp, Np, Mp, mode, P, eng._stream()), "trmm")
    else:
        C = torch.full((P, Np, Mp), float("nan"), dtype=dtype, device="cuda:0")
        B.check(fn(At.data_ptr(), Np * Mp, Tt.data_ptr(), Mp * Mp, C.data_ptr(), Np * Mp, Np, Mp, mode, P, eng._stream()), "trmm")
    assert relerr(C.cpu().numpy(), ref) < tol * 20
    # one shared operand for all latents
    C2 = torch.empty((P, Np, Mp), dtype=dtype, device="cuda:0")
    B.check(fn(At[1].data_ptr(), 0, Tt.data_ptr(), Mp * Mp, C2.data_ptr(), Np * Mp, Np, Mp, mode, P, eng._stream()), "trmm")
    ref2 = np.einsum("nj,pij->pni", At[1].double().cpu().numpy(), Tt.double().cpu().numpy())
    assert relerr(C2.cpu().numpy(), ref2) < tol * 20
    # in place is refused for the lower and dense forms (a later tile would read what an earlier one overwrote)
    if mode != 1:
        assert fn(At.data_ptr(), Np * Mp, Tt.data_ptr(), Mp * Mp, At.data_ptr(), Np * Mp, Np, Mp, mode, P, eng._stream()) == 1


@pytest.mark.parametrize("dtype,tol", DTYPES)
@pytest.mark.parametrize("lik", ["gaussian", "bernoulli"])
def test_moments_and_site_accum_batched(engines, dtype, tol, lik):
    """One operand and one prior variance per latent: tsvgp_moments_batched_* + tsvgp_site_accum_batched_* against NumPy /
    the oracle's likelihood maps."""
    import ctypes

    eng = engines[dtype]
    B = pkg()._backend
    rng = np.random.RandomState(13)
    N, M, P = 333, 256, 3
    Np = B.round_up(N)
    A = np.zeros((P, Np, M))
    A[:, :N] = rng.randn(P, N, M) / np.sqrt(M)
    Tm = np.triu(rng.randn(P, M, M) * 0.1)  # q = |Tm a|^2 ~ 1.3 < kdiag: positive variances
    gamma = rng.randn(M, P)
    kd = [2.5, 3.0, 2.2]
    Y = (rng.rand(N, P) > 0.5).astype(float) if lik == "bernoulli" else rng.randn(N, P)
    t = lambda a: torch.as_tensor(a, dtype=dtype, device="cuda:0").contiguous()
    At, Tmt, gt, Yt = t(A), t(Tm), t(gamma), t(Y)
    mean = torch.empty((N, P), dtype=dtype, device="cuda:0")
    var = torch.empty((N, P), dtype=dtype, device="cuda:0")
    g0 = torch.full((Np, P), float("nan"), dtype=dtype, device="cuda:0")
    g1 = torch.full((Np, P), float("nan"), dtype=dtype, device="cuda:0")
    vep = torch.zeros(Np // 128, dtype=torch.float64, device="cuda:0")
    npp = torch.zeros(Np // 128, dtype=torch.int32, device="cuda:0")
    lik_id = {"gaussian": 1, "bernoulli": 2}[lik]
    B.check(eng._fn("tsvgp_moments_batched")(At.data_ptr(), Np * M, Tmt.data_ptr(), gt.data_ptr(), Yt.data_ptr(),
                                             (ctypes.c_double * P)(*kd), lik_id, 0.3, mean.data_ptr(), var.data_ptr(),
                                             g0.data_ptr(), g1.data_ptr(), vep.data_ptr(), npp.data_ptr(), N, Np, M, P, 1,
                                             eng._stream()), "moments batched")
    Ad, Td, gd = At.double().cpu().numpy(), Tmt.double().cpu().numpy(), gt.double().cpu().numpy()
    C = np.einsum("pnj,pij->pni", Ad[:, :N], Td)
    vref = (np.asarray(kd)[:, None] - np.sum(C * C, axis=-1)).T
    mref = np.einsum("pnj,jp->np", Ad[:, :N], gd)
    assert relerr(mean.cpu().numpy(), mref) < tol * 20
    assert np.max(np.abs(var.double().cpu().numpy() - vref)) < tol * 20 * max(kd)
    assert int(npp.sum()) == int(np.sum(vref <= 0)) == 0
    olik = O.Gaussian(variance=0.3) if lik == "gaussian" else O.Bernoulli()
    mu_k, var_k = mean.double().cpu().numpy(), var.double().cpu().numpy()
    r0, r1 = olik.variational_expectations_grads(mu_k, var_k, Y)
    r1 = np.minimum(r1, -1e-8)
    ltol = 1e-10 if dtype == torch.float64 else 1e-5
    np.testing.assert_allclose(g0.double().cpu().numpy()[:N], r0, rtol=ltol, atol=ltol)
    np.testing.assert_allclose(g1.double().cpu().numpy()[:N], r1, rtol=ltol, atol=ltol)
    assert np.all(g0.cpu().numpy()[N:] == 0) and np.all(g1.cpu().numpy()[N:] == 0)
    # site sums over the per-latent operands
    nsplit = 3
    work = torch.empty(int(eng._fn("tsvgp_site_accum_work_bytes")(M, P, nsplit)), dtype=torch.uint8, device="cuda:0")
    acc2 = torch.full((P, M, M), float("nan"), dtype=torch.float64, device="cuda:0")
    acc1 = torch.full((P, M), float("nan"), dtype=torch.float64, device="cuda:0")
    B.check(eng._fn("tsvgp_site_accum_batched")(At.data_ptr(), Np * M, g0.data_ptr(), g1.data_ptr(), acc2.data_ptr(),
                                                acc1.data_ptr(), work.data_ptr(), Np, M, P, nsplit, eng._stream()), "site_accum")
    g0d, g1d = g0.double().cpu().numpy(), g1.double().cpu().numpy()
    assert relerr(acc2.cpu().numpy(), np.einsum("pnm,pno,np->pmo", Ad, Ad, g1d)) < tol * 50
    assert relerr(acc1.cpu().numpy(), np.einsum("pnm,np->pm", Ad, g0d)) < tol * 50


@pytest.mark.parametrize("dtype,tol", DTYPES)
@pytest.mark.parametrize("kind,name", [(0, "SquaredExponential"), (3, "Matern52")])
@pytest.mark.parametrize("N,M,D", [(300, 70, 40), (129, 200, 784)])
def test_fill_large_input_dimension(engines, dtype, tol, kind, name, N, M, D):
    """D > 32 (the reference's MNIST notebook has 784 inputs): distance by a library GEMM, tsvgp_gram_to_kernel_* in place."""
    eng = engines[dtype]
    B = pkg()._backend
    rng = np.random.RandomState(6)
    X, Z = rng.randn(N, D) / np.sqrt(D), rng.randn(M, D) / np.sqrt(D)
    ls = 0.7 + rng.rand(D)
    t = lambda a: torch.as_tensor(a, dtype=dtype, device="cuda:0").contiguous()
    out = torch.full((B.round_up(N), B.round_up(M)), float("nan"), dtype=dtype, device="cuda:0")
    eng.se_fill(t(X), t(Z), t(1.0 / ls), 1.3, out, kind)
    K = out.double().cpu().numpy()
    ref = getattr(O, name)(variance=1.3, lengthscales=ls).K(X, Z)
    assert relerr(K[:N, :M], ref) < (tol * 10 if dtype == torch.float64 else 2e-4)
    assert np.all(K[N:, :] == 0) and np.all(K[:, M:] == 0)


@pytest.mark.parametrize("variant", ["default", "DIAG_V1", "DIAG_V2", "FUSE"])
@pytest.mark.parametrize("M,batch", [(128, 1), (256, 2), (384, 3), (1024, 1)])
def test_potrf_block_step_variants(engines, variant, M, batch):
    """Round 5: the block step of tsvgp_potrf_f64 / tsvgp_potrf_solve_f64 in its four forms -- the default (inverted 16 x 16
    diagonal tiles + substitution panels on MFMA tile registers, tsvgp_chol.hip), round 4's (TSVGP_POTRF_DIAG_V1: assembled
    128 x 128 inverse + product panels; what a call beside a long fill takes), the tile-dataflow diagonal kernel
    (TSVGP_POTRF_DIAG_V2) and the fused diagonal + panel launch (TSVGP_POTRF_FUSE) -- against NumPy: factor, solve and
    the LAPACK index of the first non-positive pivot (reference src/util.py:376-389, src/models/tsvgp.py:270, :300)."""
    eng = engines[torch.float64]
    B = pkg()._backend
    saved = eng.potrf_flags
    eng.potrf_flags = {"default": 0, "DIAG_V1": B.POTRF_DIAG_V1, "DIAG_V2": B.POTRF_DIAG_V2, "FUSE": B.POTRF_FUSE}[variant]
    try:
        rng = np.random.RandomState(11 + M)
        A = rng.randn(batch, M, M)
        A = A @ np.swapaxes(A, -1, -2) / M + np.eye(M)
        Ad = torch.as_tensor(A, device="cuda:0")
        L, info = eng.cholesky(Ad)
        assert int(info.abs().sum()) == 0
        assert relerr(L.cpu().numpy(), np.linalg.cholesky(A)) < 1e-13
        Lr = np.tril(rng.randn(batch, M, M)) / np.sqrt(M)
        U, info, Dm = eng.cholesky_solve_upper(Ad, torch.as_tensor(Lr, device="cuda:0"))
        assert int(info.abs().sum()) == 0
        Un = U.cpu().numpy()
        assert relerr(Un @ np.swapaxes(Un, -1, -2), A) < 1e-13
        assert relerr(Dm.cpu().numpy(), np.linalg.solve(Un, np.swapaxes(Lr, -1, -2))) < 1e-11
        for c in sorted({1, 17, min(70, M), M - 5}):  # first non-positive pivot at column c (1-based, LAPACK's info)
            bad = np.tile(2.0 * np.eye(M), (batch, 1, 1))
            bad[0, c - 1, c - 1] = -1.0
            _, info = eng.cholesky(torch.as_tensor(bad, device="cuda:0"))
            assert int(info[0]) == c and int(info[1:].abs().sum()) == 0
    finally:
        eng.potrf_flags = saved


@pytest.mark.parametrize("M,P,shared", [(64, 1, True), (200, 3, True), (1024, 2, False)])
def test_gemv_rows(engines, M, P, shared):
    """tsvgp_gemv_f64 (``EStepEngine.gemv``): y[:, p] = A_p v[:, p], one matrix for every latent or one per latent -- the
    matrix-vector products of the replicated chain ((K_uu + 1e-6 I) lambda_1, K_uu beta: reference src/util.py:176-179,
    src/models/tsvgp.py:249-254) -- against NumPy, odd M included (the scalar path)."""
    eng = engines[torch.float64]
    rng = np.random.RandomState(3 + M)
    for Mx in (M, M + 1):
        A = rng.randn(Mx, Mx) if shared else rng.randn(P, Mx, Mx)
        v = rng.randn(Mx, P)
        y = eng.gemv(torch.as_tensor(A, device="cuda:0"), torch.as_tensor(v, device="cuda:0")).cpu().numpy()
        ref = A @ v if shared else np.einsum("pmk,kp->mp", A, v)
        assert relerr(y, ref) < 1e-13


def test_cond2_estimate_through_the_engine(engines):
    """util.cond2_estimate with the engine's factorisation (the route gate of the models, tsvgp.py ``_routes``): the fast path
    -- upper-form factorisation with the identity riding along, row GEMVs, a normalisation every fourth step -- against the exact
    2-norm condition number on kernel matrices from cond 1e1 to 1e9 (a lower bound within ten percent, as on the CPU path),
    a batch of matrices, M not a multiple of 128, and inf for a matrix that is not positive definite."""
    from importlib import import_module

    U = import_module("t-svgp_amd.util")
    eng = engines[torch.float64]
    rng = np.random.RandomState(0)
    for M in (200, 1024):
        X = rng.randn(M, 6)
        mats = []
        for ell in (0.6, 1.0, 1.6, 2.5):
            Z = X / ell
            d2 = ((Z[:, None, :] - Z[None]) ** 2).sum(-1)
            mats.append(np.exp(-0.5 * d2) + 1e-9 * np.eye(M))
        A = torch.as_tensor(np.stack(mats), device="cuda:0")
        est = U.cond2_estimate(A, eng.cholesky).cpu().numpy()
        slow = U.cond2_estimate(A).cpu().numpy()  # torch's factorisation and GEMVs: the same iteration
        for a, e, s in zip(mats, est, slow):
            ev = np.linalg.eigvalsh(a)
            c = ev[-1] / ev[0]
            assert 0.9 * c <= e <= 1.0001 * c, (M, c, e)
            assert abs(e - s) <= 1e-6 * s, (M, e, s)
    bad = torch.as_tensor(np.diag([1.0, -1.0, 2.0] + [1.0] * 125), device="cuda:0")
    assert np.isinf(float(U.cond2_estimate(bad, eng.cholesky)[0]))


def test_clock_keeper_leaves_on_the_flag_and_on_its_bound(engines):
    """tsvgp_keeper_run / tsvgp_keeper_signal (the opt-in clock keeper beside the M x M sections): its waves
    leave when the flag is raised -- at once when it already is, within microseconds of a signal from another stream -- and
    after max_us on their own when nobody raises it (the exit condition every wave reaches)."""
    eng = engines[torch.float64]
    dev = eng.device
    lib = eng.lib
    flag = torch.zeros(16, dtype=torch.int32, device=dev)
    main = torch.cuda.current_stream(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn):
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    assert lib.tsvgp_keeper_signal(flag.data_ptr(), 1, main.cuda_stream) == 0
    timed(lambda: lib.tsvgp_keeper_run(flag.data_ptr(), 50000.0, 0, main.cuda_stream))  # (first launch: code load)
    raised = timed(lambda: lib.tsvgp_keeper_run(flag.data_ptr(), 50000.0, 0, main.cuda_stream))
    assert int(flag[0]) == 1 and raised < 5.0, raised
    assert lib.tsvgp_keeper_signal(flag.data_ptr(), 0, main.cuda_stream) == 0
    bound = timed(lambda: lib.tsvgp_keeper_run(flag.data_ptr(), 3000.0, 0, main.cuda_stream))
    assert 2.9 < bound < 8.0, bound
    # raised from ANOTHER stream while the keeper runs: it leaves far below its bound.  (Streams and events exist before the launch:
    # anything that synchronises the device behind a running keeper -- an allocation, a first library call loading code objects --
    # waits for its bound, which is why the engine's default bound is 4 ms.)
    side = torch.cuda.Stream(dev)
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    assert lib.tsvgp_keeper_signal(flag.data_ptr(), 0, main.cuda_stream) == 0
    side.wait_stream(main)
    s0.record(side)
    assert lib.tsvgp_keeper_run(flag.data_ptr(), 200000.0, 0, side.cuda_stream) == 0
    s1.record(side)
    assert lib.tsvgp_keeper_signal(flag.data_ptr(), 1, main.cuda_stream) == 0
    torch.cuda.synchronize()
    assert s0.elapsed_time(s1) < 50.0, s0.elapsed_time(s1)
    # through the engine: begin / end return, the keeper is gone behind end (its bound here: 20 ms)
    old = eng.clock_keeper, eng.keeper_max_us
    eng.clock_keeper, eng.keeper_max_us = -1, 20000.0
    try:
        t = eng.keeper_begin()
        assert t is not None
        eng.keeper_end(t)
        torch.cuda.synchronize()
        assert t.query()
        eng.clock_keeper = 0
        assert eng.keeper_begin() is None
    finally:
        eng.clock_keeper, eng.keeper_max_us = old
    for bad in (lambda: lib.tsvgp_keeper_run(None, 10.0, 0, None), lambda: lib.tsvgp_keeper_run(flag.data_ptr(), 0.0, 0, None),
                lambda: lib.tsvgp_keeper_run(flag.data_ptr(), 2.0e6, 0, None), lambda: lib.tsvgp_keeper_signal(None, 1, None)):
        assert bad() == 1  # TSVGP_EINVAL


def test_tri_copy_shift_and_the_runs_form_of_factor_and_solve(engines):
    """tsvgp_tri_copy_shift_f64 (the copy / triangle / index reversal of tsvgp_tri_copy_f64 with a shift of the diagonal, flip = 3: a
    plain copy) against NumPy, and ``EStepEngine.cholesky_solve_upper`` fed with runs (matrices, shift) -- K_uu + jitter I of reference
    src/models/tsvgp.py:270 and I + L^T K L of src/util.py:171-172 without an assembled batch -- against the same call on the
    batch assembled by hand: bit for bit (the shift is the same addition, made in another kernel)."""
    eng = engines[torch.float64]
    rng = np.random.RandomState(5)
    for M in (96, 128, 200):
        A = rng.randn(2, M, M)
        At = torch.as_tensor(A, device="cuda:0")
        J = np.eye(M)[::-1]
        for flip, ref in ((0, np.tril(A) * 0.5), (1, np.triu(J @ A @ J) * 0.5), (2, (J @ A @ J) * 0.5), (3, A * 0.5)):
            got = eng.tri_copy(At, M, 0.5, flip, diag_add=0.25).cpu().numpy()
            want = ref + 0.25 * np.eye(M)
            assert np.array_equal(got, want), (M, flip)
        # factor and solve: runs against the assembled batch
        X = rng.randn(3, M, M + 5)
        S = X @ X.transpose(0, 2, 1) / M
        L = np.tril(rng.randn(3, M, M))
        St, Lt_ = torch.as_tensor(S, device="cuda:0"), torch.as_tensor(L, device="cuda:0")
        batch = St.clone()
        batch[:2].diagonal(dim1=-2, dim2=-1).add_(1.0)
        batch[2:].diagonal(dim1=-2, dim2=-1).add_(1e-3)
        U0, i0, D0 = eng.cholesky_solve_upper(batch, Lt_)
        U1, i1, D1 = eng.cholesky_solve_upper([(St[:2], 1.0), (St[2:], 1e-3)], [Lt_[:1], Lt_[1:]])
        assert int(i0.abs().sum()) == 0 and int(i1.abs().sum()) == 0
        assert torch.equal(U0, U1) and torch.equal(D0, D1)
        Ur = U1.cpu().numpy()
        Sref = S + np.stack([np.eye(M), np.eye(M), 1e-3 * np.eye(M)])
        assert relerr(Ur @ Ur.transpose(0, 2, 1), Sref) < 1e-13
        assert relerr(Ur @ D1.cpu().numpy(), L.transpose(0, 2, 1)) < 1e-10
    lib = eng.lib
    assert lib.tsvgp_tri_copy_shift_f64(At.data_ptr(), M, M * M, At.data_ptr(), M, M * M, M, 2, 1.0, 0.0, 4, None) == 1  # TSVGP_EINVAL
    assert lib.tsvgp_tri_copy_f64(At.data_ptr(), M, M * M, At.data_ptr(), M, M * M, M, 2, 1.0, 3, None) == 1
