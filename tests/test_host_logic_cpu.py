"""Host logic on CPU (no GPU, no HIP compute): the M x M site algebra in torch against the oracle, the GPflow-style
containers, and the sharded E-step over ``gloo`` with world_size 2 (rank-sharded rows, one all-reduce of the packed
accumulators, replicated epilogue) against the oracle's single-process full-data step."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import tsvgp_oracle as O
from tests.cpu_engine import NumpyShardEngine
from tests.helpers import free_port, pkg, relerr, synthetic


def _pair(Z, lik, P=1, num_data=None, kind="plain"):
    p = pkg()
    if kind == "white":  # t_SVGP_white (one latent)
        mk = lambda mod, **kw: mod.t_SVGP_white(mod.SquaredExponential(1.0, 1.0),
                                                mod.Gaussian(0.1) if lik == "gaussian" else mod.Bernoulli(), Z,
                                                num_data=num_data, **kw)
        hip = mk(p, device="cpu")
        hip._engine = NumpyShardEngine()
        return hip, mk(O)
    if kind == "separate":  # one kernel per latent on shared inducing points
        ls = np.linspace(0.8, 1.4, P)
        mk = lambda mod, **kw: mod.t_SVGP(mod.SeparateIndependent([mod.SquaredExponential(1.0, float(l)) for l in ls]),
                                          mod.Gaussian(0.1) if lik == "gaussian" else mod.Bernoulli(),
                                          mod.SharedIndependentInducingVariables(Z), num_latent_gps=P, num_data=num_data, **kw)
        hip = mk(p, device="cpu")
        hip._engine = NumpyShardEngine()
        return hip, mk(O)
    hip = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Gaussian(0.1) if lik == "gaussian" else p.Bernoulli(), Z,
                   num_latent_gps=P, num_data=num_data, device="cpu")
    hip._engine = NumpyShardEngine()  # test double: the HIP engine cannot exist without a GPU
    ora = O.t_SVGP(O.SquaredExponential(1.0, 1.0), O.Gaussian(0.1) if lik == "gaussian" else O.Bernoulli(), Z,
                   num_latent_gps=P, num_data=num_data)
    return hip, ora


@pytest.mark.parametrize("lik,P", [("gaussian", 1), ("bernoulli", 2)])
def test_host_prelude_and_epilogue_match_oracle(lik, P):
    """The torch M x M algebra (site operands, whitened back-solves, natural-parameter update, -chol) is exact
    reference algebra: with a NumPy N-pass plugged in, the model reproduces the oracle step for step."""
    X, Y, Z = synthetic(N=300, M=24, D=2, P=P, lik=lik, seed=2)
    hip, ora = _pair(Z, lik, P, num_data=450)
    for _ in range(4):
        hip.natgrad_step((X, Y), lr=0.7)
        ora.natgrad_step((X, Y), lr=0.7)
        assert relerr(hip.lambda_1.numpy(), ora.lambda_1) < 1e-9
        assert relerr(hip.lambda_2.numpy(), ora.lambda_2) < 1e-9
    assert abs(float(hip.elbo((X, Y))) - ora.elbo((X, Y))) < 1e-9 * abs(ora.elbo((X, Y)))
    assert abs(float(hip.prior_kl()) - ora.prior_kl()) < 1e-8 * max(1.0, abs(ora.prior_kl()))


def test_skip_unused_variance_host_logic():
    """skip_unused_variance asks the N-pass for the mean only under a Gaussian likelihood (and leaves other
    likelihoods alone); the step is the oracle's, which computes the variance as the reference does."""
    X, Y, Z = synthetic(N=300, M=24, D=2, P=2, lik="gaussian", seed=2)
    hip, ora = _pair(Z, "gaussian", 2)
    hip.skip_unused_variance = True
    seen = []
    inner = hip._engine.run
    hip._engine.run = lambda *a, **k: (seen.append(k.get("mean_only")), inner(*a, **k))[1]
    for _ in range(3):
        hip.natgrad_step((X, Y), lr=0.7)
        ora.natgrad_step((X, Y), lr=0.7)
        assert relerr(hip.lambda_1.numpy(), ora.lambda_1) < 1e-9
        assert relerr(hip.lambda_2.numpy(), ora.lambda_2) < 1e-9
    assert seen == [True] * 3
    assert abs(float(hip.elbo((X, Y))) - ora.elbo((X, Y))) < 1e-9 * abs(ora.elbo((X, Y)))  # elbo keeps the variance
    Xb, Yb, Zb = synthetic(N=300, M=24, D=2, P=1, lik="bernoulli", seed=2)
    hb, ob = _pair(Zb, "bernoulli", 1)
    hb.skip_unused_variance = True
    seen_b = []
    inner_b = hb._engine.run
    hb._engine.run = lambda *a, **k: (seen_b.append(k.get("mean_only")), inner_b(*a, **k))[1]
    hb.natgrad_step((Xb, Yb), lr=0.5)
    ob.natgrad_step((Xb, Yb), lr=0.5)
    assert seen_b == [False] and relerr(hb.lambda_1.numpy(), ob.lambda_1) < 1e-9


def test_white_predict_f_extra_data_host_logic():
    """t_SVGP_white.predict_f_extra_data (tsvgp_white.py:134-160): the M x M algebra of the conditioned sites against
    the oracle with the NumPy N-pass plugged in, Gaussian and Bernoulli, both jitter settings of the reference's test."""
    for lik in ("gaussian", "bernoulli"):
        X, Y, Z = synthetic(N=200, M=16, D=2, P=1, lik=lik, seed=3)
        Xe, Ye, _ = synthetic(N=120, M=16, D=2, P=1, lik=lik, seed=4)
        hip, ora = _pair(Z, lik, 1, kind="white")
        for _ in range(2):
            hip.natgrad_step((X, Y), lr=0.8)
            ora.natgrad_step((X, Y), lr=0.8)
        l1 = hip.lambda_1.numpy().copy()
        # the oracle forms K9^-1 products explicitly (tsvgp_white.py:196-206): its own rounding is cond(K9) eps times a modest
        # factor, and the order of its BLAS sums moves the comparison by that much (the whitened model's tolerance everywhere)
        tol = max(1e-8, 1000 * np.linalg.cond(O.Kuu(ora.inducing_variable, ora.kernel) + 1e-9 * np.eye(16)) * 2.2e-16)
        for kw in ({}, dict(jitter=1e-4)):
            mh, vh = hip.predict_f_extra_data(X[:50] + 0.1, (Xe, Ye), **kw)
            mo, vo = ora.predict_f_extra_data(X[:50] + 0.1, (Xe, Ye), **kw)
            assert relerr(mh.numpy(), mo) < tol and relerr(vh.numpy(), vo) < tol
        assert np.array_equal(hip.lambda_1.numpy(), l1)


def test_white_compute_data_natural_params_host_logic():
    """t_SVGP_white.compute_data_natural_params (tsvgp_white.py:181-209): [G0 - 2 G1 meanZ, G1] in the reference's shapes,
    on every projection route, against the oracle; the state is not touched and ``nat_params`` is ignored."""
    for lik in ("gaussian", "bernoulli"):
        X, Y, Z = synthetic(N=200, M=16, D=2, P=1, lik=lik, seed=3)
        hip, ora = _pair(Z, lik, 1, num_data=200, kind="white")
        for _ in range(2):
            hip.natgrad_step((X, Y), lr=0.8)
            ora.natgrad_step((X, Y), lr=0.8)
        l1, L2 = hip.lambda_1.numpy().copy(), hip.lambda_2.numpy().copy()
        want = ora.compute_data_natural_params((X, Y))
        assert want[0].shape == (16, 1) and want[1].shape == (1, 16, 16)
        for projection in ("auto", "whitened", "direct"):
            hip.projection = projection
            got = hip.compute_data_natural_params((X, Y), nat_params="ignored")
            assert tuple(got[0].shape) == (16, 1) and tuple(got[1].shape) == (1, 16, 16)
            assert relerr(got[0].numpy(), want[0]) < 1e-8 and relerr(got[1].numpy(), want[1]) < 1e-8
        want5 = ora.compute_data_natural_params((X, Y), jitter=1e-5)
        got5 = hip.compute_data_natural_params((X, Y), jitter=1e-5)
        assert relerr(got5[0].numpy(), want5[0]) < 1e-8 and relerr(got5[1].numpy(), want5[1]) < 1e-8
        assert relerr(want5[1], want[1]) > 1e-7  # the jitter argument reaches the projection (:198)
        assert np.array_equal(hip.lambda_1.numpy(), l1) and np.array_equal(hip.lambda_2.numpy(), L2)


@pytest.mark.parametrize("lik", ["gaussian", "bernoulli"])
def test_white_direct_route_host_logic(lik):
    """t_SVGP_white(projection="direct"): moments on k with the factor of Q = K6^-1 - R^-1, sums over k k^T -- the same
    step as the oracle's, with the NumPy N-pass plugged in; "auto" picks it on a well-conditioned K_uu."""
    rng = np.random.RandomState(5)
    X, Y, _ = synthetic(N=250, M=20, D=3, P=1, lik=lik, seed=8)
    Z = rng.randn(20, 3) * 1.6  # spread inducing points: cond(K_uu) of order 10
    hip, ora = _pair(Z, lik, 1, num_data=400, kind="white")
    assert hip._use_direct()
    seen = []
    inner = hip._engine.run
    hip._engine.run = lambda *a, **k: (seen.append(k.get("whiten_T") is None), inner(*a, **k))[1]
    for _ in range(4):
        hip.natgrad_step((X, Y), lr=0.7)
        ora.natgrad_step((X, Y), lr=0.7)
        assert relerr(hip.lambda_1.numpy(), ora.lambda_1) < 1e-9
        assert relerr(hip.lambda_2.numpy(), ora.lambda_2) < 1e-9
    assert abs(float(hip.elbo((X, Y))) - ora.elbo((X, Y))) < 1e-9 * abs(ora.elbo((X, Y)))
    mh, vh = hip.predict_f(X[:40] + 0.1)
    mo, vo = ora.predict_f(X[:40] + 0.1)
    assert relerr(mh.numpy(), mo) < 1e-9 and relerr(vh.numpy(), vo) < 1e-9
    assert all(seen)  # no N-sized whitening anywhere
    hip.projection = "whitened"
    hip.natgrad_step((X, Y), lr=0.7)
    ora.natgrad_step((X, Y), lr=0.7)
    assert relerr(hip.lambda_2.numpy(), ora.lambda_2) < 1e-9 and seen[-1] is False


def test_util_functions_match_oracle():
    p = pkg()
    rng = np.random.RandomState(0)
    M, P, N = 7, 2, 5
    A = rng.randn(M, M)
    K = A @ A.T + M * np.eye(M)
    l1 = rng.randn(M, P)
    L = -np.tril(rng.randn(P, M, M)) * 0.3 - np.eye(M)
    t = torch.as_tensor
    m, cS = p.util.posterior_from_dense_site(t(K), t(l1), t(L))
    mo, cSo = O.posterior_from_dense_site(K, l1, L)
    assert relerr(m.numpy(), mo) < 1e-11 and relerr(cS.numpy(), cSo) < 1e-11
    Kuf = rng.randn(M, N)
    Kff = np.full((N, 1), 50.0)
    mu, cov = p.util.conditional_from_precision_sites(t(K), t(Kff), t(Kuf), t(l1), L=t(L))
    muo, covo = O.conditional_from_precision_sites(K, Kff, Kuf, l1, L=L)
    assert relerr(mu.numpy(), muo) < 1e-11 and relerr(cov.numpy(), covo) < 1e-11
    g = [rng.randn(M, P), rng.randn(P, M, M)]
    a0, a1 = p.util.gradient_transformation_mean_var_to_expectation(t(l1), [t(g[0]), t(g[1])])
    b0, b1 = O.gradient_transformation_mean_var_to_expectation(l1, g)
    assert relerr(a0.numpy(), b0) < 1e-13 and relerr(a1.numpy(), b1) < 1e-13
    # site-form KL == gauss_kl of the explicit posterior
    D, cW = p.util.site_projection_D(t(K), t(L), return_chol=True)
    DKl = torch.einsum("pmk,kp->pm", D @ t(K), t(l1))
    beta = t(l1) - torch.einsum("pkm,pk->mp", D, DKl)
    kl = p.util.kl_from_dense_site(t(K), t(l1), D, cW, beta)
    assert abs(float(kl) - O.gauss_kl(mo, cSo, K)) < 1e-10 * abs(O.gauss_kl(mo, cSo, K))
    with pytest.raises(FloatingPointError):
        p.util.cholesky(t(-np.eye(3)))
    with pytest.raises(ValueError):
        p.util.posterior_from_dense_site(t(K), t(l1[:, :1]), t(L))  # shape mismatch (util.py:368-372)
    # the forms for sites stored pre-multiplied by K_uu (util.py:11-88, 239-291, 294-346, 394-426), one latent
    lw = rng.randn(M, 1)
    Lw = np.tril(rng.randn(1, M, M)) * 0.3 + np.eye(M)
    L2w = Lw @ np.swapaxes(Lw, -1, -2)
    for kw_t, kw_o in ((dict(L=t(Lw)), dict(L=Lw)), (dict(L2=t(L2w)), dict(L2=L2w))):
        mu, cov = p.util.conditional_from_precision_sites_white(t(K), t(Kff), t(Kuf), t(lw), **kw_t)
        muo, covo = O.conditional_from_precision_sites_white(K, Kff, Kuf, lw, **kw_o)
        assert mu.shape == (N, 1) and cov.shape == (N, 1)
        assert relerr(mu.numpy(), muo) < 1e-11 and relerr(cov.numpy(), covo) < 1e-11
        for fn in (p.util.kl_from_precision_sites_white, p.util.kl_from_precision_sites):
            assert abs(float(fn(t(K), t(lw), **kw_t)) - O.kl_from_precision_sites_white(K, lw, **kw_o)) < 1e-11
    mu5, cov5 = p.util.conditional_from_precision_sites_white(t(K), t(Kff), t(Kuf), t(lw), L2=t(L2w), jitter=1e-2)
    muo5, covo5 = O.conditional_from_precision_sites_white(K, Kff, Kuf, lw, L2=L2w, jitter=1e-2)
    assert relerr(mu5.numpy(), muo5) < 1e-11 and relerr(cov5.numpy(), covo5) < 1e-11 and relerr(muo5, muo) > 1e-6
    mw, cSw = p.util.posterior_from_dense_site_white(t(K), t(lw), t(L2w))
    mwo, cSwo = O.posterior_from_dense_site_white(K, lw, L2w)
    assert relerr(mw.numpy(), mwo) < 1e-11 and relerr(cSw.numpy(), cSwo) < 1e-11
    # the same posterior in the other parameterisation: sites K^-1 lw, K^-1 L2w K^-1 (tests/models/test_condit.py:27-66)
    Ki = np.linalg.inv(K)
    m_p, cS_p = O.posterior_from_dense_site(K, Ki @ lw, np.linalg.cholesky(Ki @ L2w[0] @ Ki)[None])
    assert relerr(mw.numpy(), m_p) < 1e-9 and relerr((cSw @ cSw.transpose(-1, -2)).numpy(), cS_p @ np.swapaxes(cS_p, -1, -2)) < 1e-9
    with pytest.raises(ValueError):
        p.util.kl_from_precision_sites_white(t(K), t(lw))  # neither L nor L2
    # per-datum sites projected onto the inducing points (util.py:188-236)
    d1, d2 = rng.randn(12, P), rng.rand(12, P) + 0.5  # 12 data points > M: the projected site has a factor
    Kuf7 = rng.randn(M, 12)
    for kuu in (None, K):
        for chol in (False, True):
            lo, Lo = O.project_diag_sites(Kuf7, d1, d2, Kuu_=kuu, cholesky=chol)
            lh, Lh = p.util.project_diag_sites(t(Kuf7), t(d1), t(d2), Kuu=None if kuu is None else t(kuu), cholesky=chol)
            assert relerr(lh.numpy(), lo) < 1e-11 and relerr(Lh.numpy(), Lo) < 1e-11


def test_containers_follow_the_reference():
    p = pkg()
    s = p.DenseSites(np.zeros((4, 2)), lambda_2_sqrt=np.ones((2, 4, 4)))
    assert np.array_equal(s.lambda_2_sqrt.numpy(), np.tril(np.ones((2, 4, 4))))  # triangular() transform
    assert s.num_latent_gps == 4  # reference src/sites.py:57 (sic)
    assert s.lambda_2.shape == (2, 4, 4)
    s2 = p.DenseSites(np.zeros((3, 1)), lambda_2=2.0 * np.eye(3)[None])
    np.testing.assert_allclose(s2.lambda_2_sqrt.numpy()[0], np.sqrt(2.0) * np.eye(3))
    k = p.SquaredExponential(variance=2.0, lengthscales=[1.0, 2.0])
    assert k.ard and np.allclose(k.inv_lengthscales(2).numpy(), [1.0, 0.5])
    with pytest.raises(ValueError):
        k.inv_lengthscales(3)
    par = p.Parameter(np.arange(3.0))
    par.assign([1.0, 2.0, 3.0])
    with pytest.raises(ValueError):
        par.assign(np.zeros(4))
    assert p.inducingpoint_wrapper(np.zeros((5, 2))).num_inducing == 5
    assert p.default_jitter() == 1e-6 and p.default_float() == torch.float64
    lo_hi = [p.distributed.shard_bounds(10, 3, r) for r in range(3)]
    assert lo_hi == [(0, 4), (4, 7), (7, 10)]


def _worker(rank, world, port, lik, P, out, kind="plain"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = pkg()
        X, Y, Z = synthetic(N=401, M=20, D=2, P=P, lik=lik, seed=4)  # 401 rows: uneven shards (201 + 200)
        Xs, Ys = p.distributed.shard_rows(X, Y)
        hip, _ = _pair(Z, lik, P, num_data=401, kind=kind)
        assert hip._reduce()
        if kind != "white":  # one kernel per latent: the M x M work is split over the ranks
            assert hip._latent_split(hip._routes(1e-9)) == (kind == "separate")
        for _ in range(3):
            hip.natgrad_step((Xs, Ys), lr=0.8)
        elbo = float(hip.elbo((Xs, Ys)))
        extra = {}
        if kind == "white":  # compute_data_natural_params of the rank's shard: sums all-reduced, every rank gets the full-data result
            g = hip.compute_data_natural_params((Xs, Ys))
            extra = dict(gm0=g[0].numpy(), gm1=g[1].numpy())
        if rank == 0:
            L2 = hip.lambda_2
            np.savez(out, l1=hip.lambda_1.numpy(), L2=L2.numpy(), elbo=elbo, **extra)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("lik,P", [("gaussian", 1), ("bernoulli", 2)])
def test_sharded_estep_gloo_world2_matches_single_process_oracle(tmp_path, lik, P):
    out = str(tmp_path / "r0.npz")
    port = free_port()
    mp.spawn(_worker, args=(2, port, lik, P, out), nprocs=2, join=True)
    got = np.load(out)
    X, Y, Z = synthetic(N=401, M=20, D=2, P=P, lik=lik, seed=4)
    _, ora = _pair(Z, lik, P, num_data=401)
    for _ in range(3):
        ora.natgrad_step((X, Y), lr=0.8)
    assert relerr(got["l1"], ora.lambda_1) < 1e-9
    assert relerr(got["L2"], ora.lambda_2) < 1e-9
    assert abs(float(got["elbo"]) - ora.elbo((X, Y))) < 1e-9 * abs(ora.elbo((X, Y)))


def test_sharded_estep_gloo_world8_matches_single_process_oracle(tmp_path):
    """The rank count of the metric's largest point (8): 401 rows over eight ranks (shards of 51 and 50 rows), one packed
    all-reduce per step, against the oracle's single-process steps -- rehearsed here on CPU over gloo because no 8-GPU node has
    been available (reference src/models/tsvgp.py:278-281, :95 summed over the ranks)."""
    out = str(tmp_path / "r0.npz")
    mp.spawn(_worker, args=(8, free_port(), "bernoulli", 1, out), nprocs=8, join=True)
    got = np.load(out)
    X, Y, Z = synthetic(N=401, M=20, D=2, P=1, lik="bernoulli", seed=4)
    _, ora = _pair(Z, "bernoulli", 1, num_data=401)
    for _ in range(3):
        ora.natgrad_step((X, Y), lr=0.8)
    assert relerr(got["l1"], ora.lambda_1) < 1e-9
    assert relerr(got["L2"], ora.lambda_2) < 1e-9
    assert abs(float(got["elbo"]) - ora.elbo((X, Y))) < 1e-9 * abs(ora.elbo((X, Y)))


@pytest.mark.parametrize("kind,lik,P", [("white", "bernoulli", 1), ("separate", "gaussian", 2), ("separate", "bernoulli", 3)])
def test_sharded_variants_gloo_world2(tmp_path, kind, lik, P):
    """The same world_size-2 shard + all-reduce path for t_SVGP_white, and separate per-latent kernels, whose M x M algebra is
    SPLIT over the ranks by latent (``t_SVGP._step_device_split``: all-gather of the N-pass operands, reduce-scatter of the sums
    by latent, all-gather of the new state; P = 3 on two ranks leaves rank 1 a padded slot)."""
    out = str(tmp_path / "r0.npz")
    port = free_port()
    mp.spawn(_worker, args=(2, port, lik, P, out, kind), nprocs=2, join=True)
    got = np.load(out)
    X, Y, Z = synthetic(N=401, M=20, D=2, P=P, lik=lik, seed=4)
    _, ora = _pair(Z, lik, P, num_data=401, kind=kind)
    for _ in range(3):
        ora.natgrad_step((X, Y), lr=0.8)
    assert relerr(got["l1"], ora.lambda_1) < 1e-9
    assert relerr(got["L2"], ora.lambda_2) < 1e-9
    assert abs(float(got["elbo"]) - ora.elbo((X, Y))) < 1e-9 * abs(ora.elbo((X, Y)))
    if kind == "white":
        # products with K9^-1 themselves (as in the reference): both sides carry cond(K9) times their rounding (4.8e5 here;
        # 1000 cond eps is the whitened model's tolerance in tools/fuzz_parity.py too)
        g = ora.compute_data_natural_params((X, Y))
        tol = max(1e-8, 1000 * np.linalg.cond(O.Kuu(ora.inducing_variable, ora.kernel) + 1e-9 * np.eye(20)) * 2.2e-16)
        assert relerr(got["gm0"], g[0]) < tol and relerr(got["gm1"], g[1]) < tol


def _worker_split3(rank, world, port, out, inject_failure):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = pkg()
        P = 2
        X, Y, Z = synthetic(N=401, M=20, D=2, P=P, lik="gaussian", seed=4)
        Xs, Ys = p.distributed.shard_rows(X, Y)
        hip, _ = _pair(Z, "gaussian", P, num_data=401, kind="separate")
        routes = hip._routes(1e-9)
        assert hip._latent_split(routes)
        own = [q for q in range(P) if q % world == rank]
        assert own == ([rank] if rank < P else [])  # rank 2 owns no latent: zero slots, flags from the scalars only
        demoted = []
        if inject_failure:
            # the owner of latent 0 reports a failed FINAL factorisation on the first step (what a direct route that lost
            # definiteness does): the max-reduced status words must make EVERY rank restore the state, demote and retry
            real = hip._apply_site_update
            calls = {"n": 0}

            def flaky(*a, **k):
                res = real(*a, **k)
                calls["n"] += 1
                if rank == 0 and calls["n"] == 1:
                    flags = res[0].clone()
                    flags[2] = 1.0
                    return (flags,) + tuple(res[1:])
                return res

            hip._apply_site_update = flaky
            real_demote = dict(hip._DEMOTE)

            class Spy(dict):
                def get(self, k, d=None):
                    demoted.append(k)
                    return real_demote.get(k, d)

            hip._DEMOTE = Spy(real_demote)
        for _ in range(3):
            hip.natgrad_step((Xs, Ys), lr=0.8)
        elbo = float(hip.elbo((Xs, Ys)))
        np.savez(out + f".{rank}.npz", l1=hip.lambda_1.numpy(), L2=hip.lambda_2.numpy(), elbo=elbo,
                 demoted=len(demoted), routes_before=np.array(routes), routes_after=np.array(hip._routes(1e-9)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("inject_failure", [False, True])
def test_latent_split_with_a_rank_that_owns_no_latent(tmp_path, inject_failure):
    """P = 2 latents on THREE ranks (what P < world does on an 8-GPU node, e.g. P = 3): rank 2 owns no latent, sends zero
    slots into both all-gathers and takes its status words from the all-reduced scalars alone
    (``t_SVGP._step_device_split``, the ``own == []`` branch).  With ``inject_failure`` the owner of latent 0 reports a failed
    final factorisation once: all three ranks -- also the one without a latent -- must see it through the max-reduce, put the
    state back, move one route down and redo the step together.  Every rank's state against the oracle."""
    out = str(tmp_path / "r")
    mp.spawn(_worker_split3, args=(3, free_port(), out, inject_failure), nprocs=3, join=True)
    X, Y, Z = synthetic(N=401, M=20, D=2, P=2, lik="gaussian", seed=4)
    _, ora = _pair(Z, "gaussian", 2, num_data=401, kind="separate")
    for _ in range(3):
        ora.natgrad_step((X, Y), lr=0.8)
    for r in range(3):
        got = np.load(out + f".{r}.npz")
        assert relerr(got["l1"], ora.lambda_1) < 1e-9
        assert relerr(got["L2"], ora.lambda_2) < 1e-9
        assert abs(float(got["elbo"]) - ora.elbo((X, Y))) < 1e-9 * abs(ora.elbo((X, Y)))
        if inject_failure:
            assert int(got["demoted"]) == 2  # both latents moved one route down, on every rank
            down = {"direct": "whitened", "whitened": "projected"}
            assert list(got["routes_after"]) == [down[r] for r in got["routes_before"]]  # remembered until (theta, Z) change
        else:
            assert int(got["demoted"]) == 0


def test_cond2_estimate_tracks_the_eigenvalue_ratio():
    """util.cond2_estimate (one factorisation + power iterations: the route gate's replacement for eigvalsh) against the
    exact 2-norm condition number on kernel matrices from cond 1e1 to 1e11; a lower bound within a few percent."""
    p = pkg()
    from importlib import import_module

    U = import_module("t-svgp_amd.util")
    rng = np.random.RandomState(0)
    X = rng.randn(200, 6)
    mats = []
    for ell in (0.6, 1.0, 1.6, 2.5):
        Z = X / ell
        d2 = ((Z[:, None, :] - Z[None]) ** 2).sum(-1)
        mats.append(np.exp(-0.5 * d2) + 1e-9 * np.eye(200))
    x = np.linspace(-1, 1, 120)[:, None]
    mats.append(np.pad(np.exp(-0.5 * ((x - x.T) / 0.5) ** 2) + 1e-9 * np.eye(120), ((0, 80), (0, 80))) + np.diag([0.0] * 120 + [1.0] * 80))
    A = torch.as_tensor(np.stack(mats))
    est = U.cond2_estimate(A).numpy()
    for a, e in zip(mats, est):
        ev = np.linalg.eigvalsh(a)
        c = ev[-1] / ev[0]
        assert 0.9 * c <= e <= 1.0001 * c, (c, e)
    bad = torch.as_tensor(np.diag([1.0, -1.0, 2.0]))
    assert np.isinf(float(U.cond2_estimate(bad)[0]))


def test_bmv_and_triangular_packing():
    """util.bmv (per-latent matrix-vector products) and the lower-triangle packing of the all-reduce payload."""
    from importlib import import_module

    U, Dm = import_module("t-svgp_amd.util"), import_module("t-svgp_amd.distributed")
    g = torch.Generator().manual_seed(0)
    for P in (1, 3, 6):
        A3 = torch.randn(P, 9, 9, generator=g, dtype=torch.float64)
        A2 = torch.randn(9, 9, generator=g, dtype=torch.float64)
        v = torch.randn(9, P, generator=g, dtype=torch.float64)
        assert torch.allclose(U.bmv(A3, v), torch.einsum("pmk,kp->mp", A3, v))
        assert torch.allclose(U.bmv(A3, v, True), torch.einsum("pkm,kp->mp", A3, v))
        assert torch.allclose(U.bmv(A2, v), A2 @ v) and torch.allclose(U.bmv(A2, v, True), A2.T @ v)

    class S:
        pass

    st = S()
    P, M = 2, 7
    a = torch.randn(P, M, M, generator=g, dtype=torch.float64)
    st.acc2, st.acc1 = a + a.transpose(-1, -2), torch.randn(P, M, generator=g, dtype=torch.float64)
    st.ve_sum, st.nonpos, st.n_rows = torch.tensor(1.5, dtype=torch.float64), torch.tensor(0.0, dtype=torch.float64), 11
    packed = Dm.pack_stats(st, True)
    assert packed.numel() == Dm.packed_size(P, M, True) == P * (M * (M + 1) // 2) + P * M + 3
    acc2, acc1, ve, nonpos, rows = Dm.unpack_stats(packed, P, M, True)
    assert torch.equal(acc2, st.acc2) and torch.equal(acc1, st.acc1) and float(ve) == 1.5 and float(rows) == 11.0
    out = Dm.reduce_stats(st, P, M, True, reduce=False)
    assert out[0] is st.acc2 and float(out[4]) == 11.0


def test_cholesky_workgroup_fits_beside_the_fill(product_asm):
    """A scheduling property, checked at compile time: the 8-wave workgroup of the diagonal-block Cholesky runs beside the
    K(X, Z) fill of the same E-step (three waves per SIMD, 96 VGPRs or fewer); above 112 VGPRs it no longer fits on a CU the fill
    occupies and every one of its 16 launches per step waits for a fill workgroup to retire (measured: +0.2 ms per call)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("isa_hazards", os.path.join(os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))), "tools", "isa_hazards.py"))
    lint = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lint)
    vgprs = lint.vgpr_counts(product_asm)
    # (the <false> instantiation: what a call beside the fill launches -- round 5's fused diagonal + panel variant <true>, an
    # experiment flag, holds a strip of the panel in registers on top and is never launched there)
    diag = [v for k, v in vgprs.items() if "potrf_diag_kernelILb0E" in k]
    beside = [v for k, v in vgprs.items() if "chol_tile_kernel" in k]
    # (256-thread workgroups, one wave per SIMD: the 224 registers per lane the fill leaves free are theirs alone)
    assert beside and max(beside) <= 224, f"chol_tile_kernel uses {max(beside) if beside else None} VGPRs"
    fill = [v for k, v in vgprs.items() if "se_fill_kernelIdLi0ELi8E" in k]
    assert diag and fill, "kernel names not found in the assembly's metadata"
    assert max(diag) <= 112, f"potrf_diag_kernel uses {max(diag)} VGPRs"
    assert max(fill) <= 96, f"se_fill_kernel<double, SE, 8> uses {max(fill)} VGPRs"


def test_site_sum_slices_fill_whole_rounds():
    """EStepEngine.choose_nsplit's rule as a pure function: the launch of the site sums runs in rounds of `slots` equal workgroups
    (n_off * ns + nt * ns_diag(ns) per latent), so the chosen count leaves the last round (nearly) full and prefers few rounds;
    the measured optima of round 3 (profiles/r03_second_half_kernel_ab.txt) are what it returns."""
    import importlib

    slices = importlib.import_module("t-svgp_amd.estep").site_sum_slices
    nd = lambda ns, num: max(1, (num * ns + 31) // 32)  # syrk_ns_diag of the kernel source
    for Mp, P, Np, slots, f64 in [(1024, 1, 1000064, 256, True), (512, 1, 1000064, 256, True), (1024, 1, 125056, 256, True),
                                  (1024, 1, 1000064, 512, False), (1024, 8, 1000064, 256, True), (768, 1, 1000064, 256, True),
                                  (2048, 1, 250112, 256, True), (256, 1, 1000064, 256, True)]:
        ns = slices(Mp, P, Np, slots, f64)
        nt = Mp // 128
        wgs = P * (nt * (nt - 1) // 2 * ns + nt * nd(ns, 20 if f64 else 22))
        rounds = -(-wgs // slots)
        assert ns >= 1 and wgs / (rounds * slots) > 0.9, (Mp, P, Np, ns, wgs)  # the last round is not left half empty
        assert Np // (16 if f64 else 32) // ns >= 32  # slices stay long against the per-workgroup overhead
    assert slices(1024, 1, 1000064, 256, True) == 62  # 2048 workgroups = 8 rounds exactly (15.19 ms; 224 slices: 15.62)
    assert slices(1024, 1, 1000064, 512, False) == 61  # fp32, two workgroups per CU (7.65 ms; 120 slices: 7.99)
    assert slices(512, 1, 1000064, 256, True) in (60, 120)  # 512 / 1020 workgroups (4.19 / 4.15 ms; 651 slices: 4.72)
    assert slices(128, 1, 1024, 256, True) >= 1 and slices(128, 1, 128, 256, True) == 1  # tiny problems: at least one slice
    assert slices(1024, 1, None, 256, True) == 7 * 8 and slices(1024, 1, 1000064, 256, True, oversubscribe=4) == 28
