"""Model-level parity on the GPU: t_SVGP (HIP path, through the C-ABI) against the CPU oracle, on identical inputs.

Stated tolerances:
  fp64: max rel err <= 1e-8 on lambda_1, Lambda_2 = L L^T, mean, var;  |dELBO| / |ELBO| <= 1e-9
  fp32 (against the fp64 oracle): mean, var atol 1e-4 + rtol 1e-3;  |dELBO| / |ELBO| <= 1e-4
The reference-shaped tests restate reference tests/models/test_tsvgp.py on the HIP model (decimal=4 as there).
"""
import numpy as np
import pytest
import torch

from oracle import tsvgp_oracle as O
from tests.helpers import c1_problem, make_pair, pkg, relerr, synthetic

pytestmark = pytest.mark.gpu


def _compare_state(hip, ora, tol):
    assert relerr(hip.lambda_1.numpy(), ora.lambda_1) < tol
    assert relerr(hip.lambda_2.cpu().numpy(), ora.lambda_2) < tol


@pytest.mark.parametrize("lik,P", [("gaussian", 1), ("gaussian", 2), ("bernoulli", 1), ("bernoulli", 2)])
def test_natgrad_steps_match_oracle_fp64(lik, P):
    """8 E-steps (lr 0.8) from the default init on a D=3 problem; state, moments, gradients and ELBO vs the oracle."""
    X, Y, Z = synthetic(N=700, M=48, D=3, P=P, lik=lik, seed=0)
    hip, ora = make_pair(Z, lik=lik, P=P)
    for step in range(8):
        hip.natgrad_step((X, Y), lr=0.8)
        ora.natgrad_step((X, Y), lr=0.8)
        _compare_state(hip, ora, 1e-8)
    e_h, e_o = float(hip.elbo((X, Y))), float(ora.elbo((X, Y)))
    assert abs(e_h - e_o) / abs(e_o) < 1e-9
    Xs = X[:200] + 0.05
    for fn in ("predict_f", "new_predict_f"):
        mu_h, var_h = getattr(hip, fn)(Xs)
        mu_o, var_o = getattr(ora, fn)(Xs)
        assert relerr(mu_h.cpu().numpy(), mu_o) < 1e-8
        assert relerr(var_h.cpu().numpy(), var_o) < 1e-8


def test_c1_config_trajectory_fp64():
    """Config C1 (N=1000, M=32, D=1, Gaussian): 8 E-steps lr=0.8; ELBO after each step matches the oracle."""
    X, Y, Z, hp = c1_problem()
    hip, ora = make_pair(Z, lik="gaussian", noise=hp["noise"], lengthscales=hp["lengthscales"], variance=hp["variance"])
    for _ in range(8):
        hip.natgrad_step((X, Y), lr=0.8)
        ora.natgrad_step((X, Y), lr=0.8)
        e_h, e_o = float(hip.elbo((X, Y))), float(ora.elbo((X, Y)))
        assert abs(e_h - e_o) / abs(e_o) < 1e-9
    _compare_state(hip, ora, 1e-8)


def test_intermediates_match_oracle_fp64():
    """mean, var, g0, g1 of one step and the whitened accumulators mapped back to the reference's G0, G1."""
    X, Y, Z = synthetic(N=500, M=40, D=2, P=1, lik="bernoulli", seed=3)
    hip, ora = make_pair(Z, lik="bernoulli")
    for _ in range(2):
        hip.natgrad_step((X, Y), lr=0.5)
        ora.natgrad_step((X, Y), lr=0.5)
    ops = hip._site_operands(whiten_jitter=1e-9)
    B = pkg()._backend
    st = hip._get_engine().run(hip._as_device(X), hip._as_device(Y), ops["Z"], hip.kernel, moment_Tm=ops["moment_Tm"],
                               moment_mode=ops["moment_mode"], gamma=ops["gamma"], lik_id=B.LIK_BERNOULLI,
                               whiten_T=ops["whiten_T"], whiten_mode=ops["whiten_mode"], sites=True, want_moments=True, want_grads=True)
    ora.natgrad_step((X, Y), lr=0.5)  # fills ora.last with the intermediates of the same state
    last = ora.last
    assert relerr(st.mean.cpu().numpy(), last["mean"]) < 1e-8
    assert relerr(st.var.cpu().numpy(), last["var"]) < 1e-8
    assert relerr(st.g0.cpu().numpy(), last["g0"]) < 1e-8
    assert relerr(st.g1.cpu().numpy(), last["g1"]) < 1e-8
    U9 = ops["U9"].cpu().numpy()
    G1 = np.stack([np.linalg.solve(U9.T, np.linalg.solve(U9.T, a).T).T for a in st.acc2.cpu().numpy()])
    G0 = np.linalg.solve(U9.T, st.acc1.cpu().numpy().T)
    assert relerr(G1, last["G1"]) < 1e-8
    assert relerr(G0, last["G0"]) < 1e-8


@pytest.mark.parametrize("lik", ["gaussian", "bernoulli"])
def test_natgrad_steps_fp32_against_fp64_oracle(lik):
    X, Y, Z = synthetic(N=2000, M=64, D=4, P=1, lik=lik, seed=1)
    hip, ora = make_pair(Z, lik=lik, compute_dtype=torch.float32)
    for _ in range(4):
        hip.natgrad_step((X, Y), lr=0.8)
        ora.natgrad_step((X, Y), lr=0.8)
    e_h, e_o = float(hip.elbo((X, Y))), float(ora.elbo((X, Y)))
    assert abs(e_h - e_o) / abs(e_o) < 1e-4
    mu_h, var_h = hip.predict_f(X[:300])
    mu_o, var_o = ora.predict_f(X[:300])
    np.testing.assert_allclose(mu_h.cpu().numpy(), mu_o, rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(var_h.cpu().numpy(), var_o, rtol=1e-3, atol=1e-4)


# ---------------------------------------------------------------------------------------------------------------------
# the reference's own tests, restated on the HIP model (reference tests/models/test_tsvgp.py)
# ---------------------------------------------------------------------------------------------------------------------
def _ref_setup():
    rng = np.random.RandomState(123)
    func = lambda x: np.sin(x * 3 * 3.14) + 0.3 * np.cos(x * 9 * 3.14) + 0.5 * np.sin(x * 7 * 3.14)
    X = rng.rand(8, 1) * 2 - 1
    Y = func(X) + 0.2 * rng.randn(8, 1)
    return X, Y, rng


@pytest.fixture(name="tsvgp_gpr_optim_setup")
def _tsvgp_gpr_optim_setup():
    p = pkg()
    X, Y, _ = _ref_setup()
    kernel = p.SquaredExponential(lengthscales=2.0, variance=2.25)
    tsvgp = p.t_SVGP(kernel=kernel, likelihood=p.Gaussian(variance=0.3), inducing_variable=p.InducingPoints(X))
    for _ in range(10):
        tsvgp.natgrad_step((X, Y), lr=0.9)
    return tsvgp, (X, Y, O.SquaredExponential(lengthscales=2.0, variance=2.25), 0.3)


def test_tsvgp_elbo_optimal(tsvgp_gpr_optim_setup):
    """reference tests/models/test_tsvgp.py:106-110."""
    tsvgp, (X, Y, kern, noise) = tsvgp_gpr_optim_setup
    np.testing.assert_almost_equal(float(tsvgp.elbo((X, Y))), O.gpr_log_marginal_likelihood(kern, X, Y, noise), decimal=4)


def test_predictions_match_tsvgp_gpr_optimal(tsvgp_gpr_optim_setup):
    """reference tests/models/test_tsvgp.py:113-120."""
    tsvgp, (X, Y, kern, noise) = tsvgp_gpr_optim_setup
    mu, var = tsvgp.predict_f(X + 1.0)
    mu_gpr, var_gpr = O.gpr_predict_f(kern, X, Y, noise, X + 1.0)
    np.testing.assert_array_almost_equal(mu.cpu().numpy(), mu_gpr, decimal=4)
    np.testing.assert_array_almost_equal(var.cpu().numpy(), var_gpr, decimal=4)


def test_tsvgp_unchanged_at_optimum(tsvgp_gpr_optim_setup):
    """reference tests/models/test_tsvgp.py:134-145."""
    tsvgp, (X, Y, _, _) = tsvgp_gpr_optim_setup
    e0 = float(tsvgp.elbo((X, Y)))
    tsvgp.natgrad_step((X, Y), lr=0.9)
    np.testing.assert_almost_equal(e0, float(tsvgp.elbo((X, Y))), decimal=4)


def test_tsvgp_minibatch_same_elbo(tsvgp_gpr_optim_setup):
    """reference tests/models/test_tsvgp.py:148-165."""
    tsvgp, (X, Y, _, _) = tsvgp_gpr_optim_setup
    tsvgp.num_data = 8
    x = X[0].repeat(8)[:, None]
    y = Y[0].repeat(8)[:, None]
    e2 = float(tsvgp.elbo((x, y)))
    e1 = float(tsvgp.elbo((X[0][:, None], Y[0][:, None])))
    np.testing.assert_almost_equal(e2, e1, decimal=4)


@pytest.mark.parametrize("num_latent_gps", [1, 2])
def test_bernoulli_fixture_matches_oracle(num_latent_gps):
    """reference tests/models/test_tsvgp.py:46-88,123-131 shape: 20 steps lr=1.0, Bernoulli, Z = X (ill-conditioned
    K_uu, cond ~ 1e16 before jitter): new_predict_f of the HIP model vs the oracle's, decimal=4 as the reference."""
    p = pkg()
    X, Yr, rng = _ref_setup()
    Y = np.tile((Yr > 0).astype(float), [1, num_latent_gps]) * rng.rand(1, num_latent_gps)
    hip = p.t_SVGP(p.SquaredExponential(lengthscales=2.0, variance=2.25), p.Bernoulli(), p.InducingPoints(X),
                   num_latent_gps=num_latent_gps)
    ora = O.t_SVGP(O.SquaredExponential(lengthscales=2.0, variance=2.25), O.Bernoulli(), O.InducingPoints(X),
                   num_latent_gps=num_latent_gps)
    for _ in range(20):
        hip.natgrad_step((X, Y), lr=1.0)
        ora.natgrad_step((X, Y), lr=1.0)
    mu_h, var_h = hip.new_predict_f(X + 0.1)
    mu_o, var_o = ora.new_predict_f(X + 0.1)
    np.testing.assert_array_almost_equal(mu_h.cpu().numpy(), mu_o, decimal=4)
    np.testing.assert_array_almost_equal(var_h.cpu().numpy(), var_o, decimal=4)


# ---------------------------------------------------------------------------------------------------------------------
# API / error behaviour
# ---------------------------------------------------------------------------------------------------------------------
def test_api_surface_and_mutation():
    p = pkg()
    X, Y, Z = synthetic(N=300, M=20, D=2)
    m = p.t_SVGP(p.SquaredExponential(), p.Gaussian(0.1), Z)  # raw array is wrapped (tsvgp.py:150)
    assert m.lambda_1.shape == (20, 1) and m.lambda_2_sqrt.shape == (1, 20, 20)
    np.testing.assert_allclose(np.diagonal(m.lambda_2_sqrt.numpy()[0]), -1e-10)  # negative-diagonal convention
    assert m.natgrad_step((X, Y), 0.5) is None  # positional lr (docs/notebooks/classification_1D.py:108)
    L = m.lambda_2_sqrt.numpy()[0]
    assert np.all(np.diagonal(L) < 0) and np.allclose(L, np.tril(L))
    e0 = float(m.elbo((X, Y)))
    m.kernel.lengthscales.assign(0.7)  # parameters are read fresh on every call
    assert float(m.elbo((X, Y))) != e0
    q_mu, q_sqrt = m.get_mean_chol_cov_inducing_posterior()
    assert q_mu.shape == (20, 1) and q_sqrt.shape == (1, 20, 20)
    mu, var = m.predict_y(X[:5])
    assert mu.shape == (5, 1) and bool((var > 0).all())
    assert m.predict_log_density((X[:5], Y[:5])).shape == (5,)
    # GPflow mixin surface the reference's driver uses (experiments/uci_regression.py:114,160)
    loss = m.training_loss_closure((X, Y), compile=True)
    assert abs(float(loss()) + float(m.elbo((X, Y)))) < 1e-12 * abs(float(m.elbo((X, Y))))
    batches = iter([(X[:100], Y[:100]), (X[100:], Y[100:])])
    it_loss = m.training_loss_closure(batches)
    assert float(it_loss()) != float(it_loss())
    names = {id(v) for v in m.trainable_variables}
    assert names == {id(m.kernel.variance), id(m.kernel.lengthscales), id(m.likelihood.variance), id(m.inducing_variable.Z)}
    assert id(m.lambda_1) not in names  # the sites are not trainable (src/sites.py:56-63)


def test_error_behaviour():
    p = pkg()
    X, Y, Z = synthetic(N=100, M=10, D=2)
    m = p.t_SVGP(p.SquaredExponential(), p.Gaussian(0.1), Z)
    with pytest.raises(ValueError):
        m.natgrad_step((X[:, :1], Y))  # D mismatch
    with pytest.raises(ValueError):
        m.natgrad_step((X, Y[:50]))  # row mismatch
    with pytest.raises(AssertionError):
        p.t_SVGP(p.SquaredExponential(), p.Gaussian(0.1), Z, lambda_2_sqrt=np.eye(10))  # ndim != 3 (tsvgp.py:182)
    dup = p.t_SVGP(p.SquaredExponential(), p.Gaussian(0.1), np.zeros((4, 2)))  # duplicate inducing points
    with pytest.raises(FloatingPointError):
        dup.natgrad_step((X, Y), jitter=0.0)  # singular K_uu: Cholesky fails as in TF


def test_warm_cache_matches_cold_and_invalidates():
    """cache_whitened=True (reuse chol(K_uu), its inverse and B between E-steps) gives the same trajectory as the default
    rebuild-everything path, and notices kernel-parameter assigns, a changed jitter and other data."""
    p = pkg()
    X, Y, Z = synthetic(N=900, M=40, D=3, lik="bernoulli", seed=5)
    Xd, Yd = torch.as_tensor(X, device="cuda:0"), torch.as_tensor(Y, device="cuda:0")
    cold = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Bernoulli(), Z)
    warm = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Bernoulli(), Z, cache_whitened=True)
    for i in range(6):
        if i == 3:  # external hyperparameter change between E-steps (docs/notebooks/regression_1D.py:186)
            cold.kernel.lengthscales.assign(0.8)
            warm.kernel.lengthscales.assign(0.8)
        jit = 1e-9 if i != 4 else 1e-8
        cold.natgrad_step((Xd, Yd), lr=0.7, jitter=jit)
        warm.natgrad_step((Xd, Yd), lr=0.7, jitter=jit)
        if i == 1:
            warm.predict_f(Xd[:100])  # overwrites the work buffers: the cache must notice
        # Bit for bit (measured in round 5, tools/dev_warm_cold.py: 0.0 on every step): a warm step reuses K_uu, its factor and
        # inverse factor and the N x M operand as the cold step of the same state computes them, the lambda-dependent
        # factorisation of W is the same batched call with one matrix fewer in the batch (batch members do not interact), and
        # every kernel sums in a fixed order.  A difference here is a stale or mismatched cache entry, not rounding.
        assert relerr(warm.lambda_1.numpy(), cold.lambda_1.numpy()) == 0.0
        assert relerr(warm.lambda_2.cpu().numpy(), cold.lambda_2.cpu().numpy()) == 0.0
    assert warm._get_engine()._b_tag is not None


@pytest.mark.parametrize("lik,P", [("gaussian", 2), ("bernoulli", 1)])
@pytest.mark.parametrize("projection", ["whitened", "direct", "projected"])
def test_both_projection_routes_match_oracle(lik, P, projection):
    """The whitened route (N-sized triangular product), the direct route (sums on K_fu, K_uu^-1 applied afterwards) and
    the projected route (a = K_uu^-1 k by two triangular products, sums over a a^T as the reference) are the same
    algebra; on a well-conditioned K_uu all meet the fp64 tolerance against the oracle."""
    p = pkg()
    rng = np.random.RandomState(11)
    X, Y, _ = synthetic(N=1200, M=64, D=6, P=P, lik=lik, seed=6)
    Z = rng.randn(64, 6) * 1.5  # spread inducing points: cond(K_uu) of order 10
    hip = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Gaussian(0.1) if lik == "gaussian" else p.Bernoulli(), Z,
                   num_latent_gps=P, projection=projection)
    ora = O.t_SVGP(O.SquaredExponential(1.0, 1.0), O.Gaussian(0.1) if lik == "gaussian" else O.Bernoulli(), Z,
                   num_latent_gps=P)
    for _ in range(5):
        hip.natgrad_step((X, Y), lr=0.8)
        ora.natgrad_step((X, Y), lr=0.8)
        _compare_state(hip, ora, 1e-8)
    e_h, e_o = float(hip.elbo((X, Y))), float(ora.elbo((X, Y)))
    assert abs(e_h - e_o) / abs(e_o) < 1e-9


@pytest.mark.parametrize("P,separate", [(1, False), (2, False), (2, True)])
@pytest.mark.parametrize("projection", ["whitened", "direct", "projected"])
def test_skip_unused_variance_matches_oracle(P, separate, projection):
    """Gaussian likelihood: d ve/d mean and d ve/d var do not depend on the predictive variance, so natgrad_step may skip
    the variance product (TSVGP_LIK_MEANONLY): same sites as the oracle, which computes it as the reference does."""
    p = pkg()
    rng = np.random.RandomState(12)
    X, Y, _ = synthetic(N=1500, M=70, D=5, P=P, lik="gaussian", seed=8)
    Z = rng.randn(70, 5) * 1.5
    mk = lambda m, ls: m.SquaredExponential(1.0, ls)
    kh = p.SeparateIndependent([mk(p, 1.0), mk(p, 1.3)]) if separate else mk(p, 1.0)
    ko = O.SeparateIndependent([mk(O, 1.0), mk(O, 1.3)]) if separate else mk(O, 1.0)
    wrap = (lambda m: m.SharedIndependentInducingVariables(Z)) if separate else (lambda m: Z)
    hip = p.t_SVGP(kh, p.Gaussian(0.1), wrap(p), num_latent_gps=P, projection=projection, skip_unused_variance=True)
    ora = O.t_SVGP(ko, O.Gaussian(0.1), wrap(O), num_latent_gps=P)
    for _ in range(4):
        hip.natgrad_step((X, Y), lr=0.7)
        ora.natgrad_step((X, Y), lr=0.7)
        _compare_state(hip, ora, 1e-8)
    assert abs(float(hip.elbo((X, Y))) - float(ora.elbo((X, Y)))) < 1e-9 * abs(float(ora.elbo((X, Y))))
    # the flag is inert for a likelihood whose gradients need the variance
    Xb, Yb, _ = synthetic(N=600, M=70, D=5, P=1, lik="bernoulli", seed=9)
    hb = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Bernoulli(), Z, skip_unused_variance=True)
    ob = O.t_SVGP(O.SquaredExponential(1.0, 1.0), O.Bernoulli(), Z)
    hb.natgrad_step((Xb, Yb), lr=0.5)
    ob.natgrad_step((Xb, Yb), lr=0.5)
    _compare_state(hb, ob, 1e-8)
    # a NaN target no longer shows in a variance: the non-finite mean / gradient count raises instead
    Ybad = Y.copy()
    Ybad[7, 0] = np.nan
    with pytest.raises(FloatingPointError):
        hip.natgrad_step((X, Ybad), lr=0.7)


def test_auto_projection_gate():
    """"auto" picks the direct route only when cond(K_uu + jitter I) is small; ill-conditioned K_uu stays whitened."""
    p = pkg()
    rng = np.random.RandomState(12)
    well = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Gaussian(0.1), rng.randn(50, 8) * 2.0)
    ill = p.t_SVGP(p.SquaredExponential(1.0, 2.0), p.Gaussian(0.1), rng.rand(50, 1) * 2 - 1)  # test_tsvgp.py geometry
    assert well._use_direct(1e-9) == [True] and ill._use_direct(1e-9) == [False]
    assert well._cond_cache[1][0] < 1e3 and ill._cond_cache[1][0] > 1e6
    well.kernel.lengthscales.assign(50.0)  # nearly constant kernel: the cached decision must be re-evaluated
    assert well._use_direct(1e-9) == [False]


def test_separate_kernels_mixed_routes():
    """With separate kernels "auto" decides per latent: a short-lengthscale latent takes the direct route, a
    long-lengthscale one (ill-conditioned K_uu) the whitened route, inside the same step; result against the oracle."""
    p = pkg()
    rng = np.random.RandomState(22)
    P, M, D = 3, 40, 4
    X, Y, _ = synthetic(N=800, M=M, D=D, P=P, lik="gaussian", seed=9)
    Z = rng.randn(M, D) * 1.5
    ls = [0.7, 6.0, 1.0]
    hip = p.t_SVGP(p.SeparateIndependent([p.SquaredExponential(1.0, l) for l in ls]), p.Gaussian(0.1),
                   p.SharedIndependentInducingVariables(Z), num_latent_gps=P)
    ora = O.t_SVGP(O.SeparateIndependent([O.SquaredExponential(1.0, l) for l in ls]), O.Gaussian(0.1),
                   O.SharedIndependentInducingVariables(Z), num_latent_gps=P)
    assert hip._use_direct(1e-9) == [True, False, True]
    for _ in range(4):
        hip.natgrad_step((X, Y), lr=0.8)
        ora.natgrad_step((X, Y), lr=0.8)
        _compare_state(hip, ora, 1e-8)
    e_h, e_o = float(hip.elbo((X, Y))), float(ora.elbo((X, Y)))
    assert abs(e_h - e_o) / abs(e_o) < 1e-9


@pytest.mark.parametrize("projection", ["whitened", "direct"])
@pytest.mark.parametrize("lik", ["gaussian", "bernoulli"])
def test_separate_kernels_match_oracle(lik, projection):
    """One SE kernel per latent on shared inducing points (SeparateIndependent + SharedIndependentInducingVariables,
    reference docs/notebooks/heteroskedastic.py:62-76; K_uu [P, M, M], rank-3 A at tsvgp.py:271-277): state, ELBO, KL and
    predictions against the oracle, both projection routes."""
    p = pkg()
    rng = np.random.RandomState(21)
    P, M, D = 3, 48, 5
    X, Y, _ = synthetic(N=900, M=M, D=D, P=P, lik=lik, seed=8)
    Z = rng.randn(M, D) * 1.5
    ls = [0.8, 1.0, 1.4]
    var = [1.0, 0.7, 1.3]
    hip = p.t_SVGP(p.SeparateIndependent([p.SquaredExponential(v, l) for v, l in zip(var, ls)]),
                   p.Gaussian(0.1) if lik == "gaussian" else p.Bernoulli(), p.SharedIndependentInducingVariables(Z),
                   num_latent_gps=P, projection=projection)
    ora = O.t_SVGP(O.SeparateIndependent([O.SquaredExponential(v, l) for v, l in zip(var, ls)]),
                   O.Gaussian(0.1) if lik == "gaussian" else O.Bernoulli(), O.SharedIndependentInducingVariables(Z),
                   num_latent_gps=P)
    for _ in range(5):
        hip.natgrad_step((X, Y), lr=0.8)
        ora.natgrad_step((X, Y), lr=0.8)
        _compare_state(hip, ora, 1e-8)
    e_h, e_o = float(hip.elbo((X, Y))), float(ora.elbo((X, Y)))
    assert abs(e_h - e_o) / abs(e_o) < 1e-9
    assert abs(float(hip.prior_kl()) - ora.prior_kl()) < 1e-8 * abs(ora.prior_kl())
    mu_h, var_h = hip.predict_f(X[:100] + 0.05)
    mu_o, var_o = ora.predict_f(X[:100] + 0.05)
    assert relerr(mu_h.cpu().numpy(), mu_o) < 1e-8 and relerr(var_h.cpu().numpy(), var_o) < 1e-8
    m_h, cS_h = hip.get_mean_chol_cov_inducing_posterior()
    m_o, cS_o = ora.get_mean_chol_cov_inducing_posterior()
    assert relerr(m_h.cpu().numpy(), m_o) < 1e-8 and relerr(cS_h.cpu().numpy(), cS_o) < 1e-7
    with pytest.raises(NotImplementedError):
        hip.new_predict_f(X[:10])  # not broadcastable in the reference either (tsvgp.py:214)
    # a kernel parameter of ONE latent changes: the cached route decision and K_uu follow
    hip.kernel.kernels[1].lengthscales.assign(1.2)
    ora.kernel.kernels[1].lengthscales = np.asarray(1.2)
    hip.natgrad_step((X, Y), lr=0.8)
    ora.natgrad_step((X, Y), lr=0.8)
    _compare_state(hip, ora, 1e-8)


@pytest.mark.parametrize("name", ["Matern52", "Matern32"])
def test_matern_kernels_match_oracle(name):
    """The reference's UCI experiment runs Matern-5/2 (experiments/uci_regression.py:42-44): E-steps, ELBO and
    predictions with the Matern family against the oracle."""
    p = pkg()
    X, Y, Z = synthetic(N=600, M=40, D=3, P=1, lik="gaussian", seed=4)
    hip = p.t_SVGP(getattr(p, name)(1.2, 0.9), p.Gaussian(0.1), Z)
    ora = O.t_SVGP(getattr(O, name)(1.2, 0.9), O.Gaussian(0.1), Z)
    for _ in range(4):
        hip.natgrad_step((X, Y), lr=0.8)
        ora.natgrad_step((X, Y), lr=0.8)
        _compare_state(hip, ora, 1e-8)
    e_h, e_o = float(hip.elbo((X, Y))), float(ora.elbo((X, Y)))
    assert abs(e_h - e_o) / abs(e_o) < 1e-9
    mu_h, var_h = hip.predict_f(X[:100] + 0.05)
    mu_o, var_o = ora.predict_f(X[:100] + 0.05)
    assert relerr(mu_h.cpu().numpy(), mu_o) < 1e-8 and relerr(var_h.cpu().numpy(), var_o) < 1e-8


@pytest.mark.parametrize("lik", ["gaussian", "bernoulli"])
def test_graph_replay_matches_eager(lik):
    """use_graph=True: the step is captured into a hipGraph on the second call with the same (data, parameters, lr) and
    replayed afterwards; states, ELBO and the invalidation on a parameter change must equal the eager path's."""
    p = pkg()
    X, Y, Z = synthetic(N=500, M=32, D=3, P=1, lik=lik, seed=3)
    Xd, Yd = torch.as_tensor(X, device="cuda:0"), torch.as_tensor(Y, device="cuda:0")
    mk = lambda **kw: p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Gaussian(0.1) if lik == "gaussian" else p.Bernoulli(), Z, **kw)
    eager, graph = mk(), mk(use_graph=True)
    for step in range(6):
        eager.natgrad_step((Xd, Yd), lr=0.6)
        graph.natgrad_step((Xd, Yd), lr=0.6)
        assert relerr(graph.lambda_1.numpy(), eager.lambda_1.numpy()) < 1e-13, step
        assert relerr(graph.lambda_2.cpu().numpy(), eager.lambda_2.cpu().numpy()) < 1e-13, step
    captured = [e for e in graph._graphs.values() if isinstance(e, dict)]
    assert len(captured) == 1  # steps 3.. were replays of one graph
    assert abs(float(graph.elbo((Xd, Yd))) - float(eager.elbo((Xd, Yd)))) < 1e-12 * abs(float(eager.elbo((Xd, Yd))))
    # a hyperparameter change is a new key: eager once, then a new capture; a user assign to the state is picked up
    for m in (eager, graph):
        m.kernel.lengthscales.assign(0.8)
        m.lambda_1.assign(m.lambda_1.value * 0.5)
    for step in range(3):
        eager.natgrad_step((Xd, Yd), lr=0.6)
        graph.natgrad_step((Xd, Yd), lr=0.6)
        assert relerr(graph.lambda_1.numpy(), eager.lambda_1.numpy()) < 1e-13
    assert len([e for e in graph._graphs.values() if isinstance(e, dict)]) == 2


def test_graph_replay_sees_in_place_parameter_edits():
    """Scalars are baked into a captured step as kernel ARGUMENTS and Z into a capture-owned buffer, so the graph key must see
    edits that never pass through ``assign``: ``likelihood.variance.value.mul_()``, ``kernel.variance.value.add_()``,
    ``Z.value.add_()`` (round-3 advisor finding: only the assign counters were keyed for the likelihood and Z, and a replay
    then silently used the stale value).  After every edit the graph model must equal an eager model AND the oracle."""
    p = pkg()
    X, Y, Z = synthetic(N=600, M=32, D=3, P=1, lik="gaussian", seed=5)
    Xd, Yd = torch.as_tensor(X, device="cuda:0"), torch.as_tensor(Y, device="cuda:0")
    mk = lambda **kw: p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Gaussian(0.1), Z.copy(), **kw)
    eager, graph = mk(use_graph=False), mk(use_graph=True)
    ora = O.t_SVGP(O.SquaredExponential(1.0, 1.0), O.Gaussian(0.1), Z.copy())
    noise, kvar, Zo = 0.1, 1.0, Z.copy()

    def steps(n):
        for _ in range(n):
            eager.natgrad_step((Xd, Yd), lr=0.6)
            graph.natgrad_step((Xd, Yd), lr=0.6)
            ora.natgrad_step((X, Y), lr=0.6)
        assert relerr(graph.lambda_1.numpy(), eager.lambda_1.numpy()) < 1e-12
        assert relerr(graph.lambda_2.cpu().numpy(), eager.lambda_2.cpu().numpy()) < 1e-12
        assert relerr(graph.lambda_1.numpy(), ora.lambda_1) < 1e-8 and relerr(graph.lambda_2.cpu().numpy(), ora.lambda_2) < 1e-8

    steps(4)  # eager, capture, two replays
    assert sum(isinstance(e, dict) for e in graph._graphs.values()) == 1
    for m in (eager, graph):
        m.likelihood.variance.value.mul_(2.0)  # in place: no assign(), the version counter does not move
    noise *= 2.0
    ora.likelihood = O.Gaussian(noise)
    steps(4)
    for m in (eager, graph):
        m.kernel.variance.value.add_(0.25)
    kvar += 0.25
    ora.kernel = O.SquaredExponential(kvar, 1.0)
    steps(4)
    shift = 0.05 * np.random.RandomState(1).randn(*Z.shape)
    for m in (eager, graph):
        m.inducing_variable.Z.value.add_(torch.as_tensor(shift, device=m.inducing_variable.Z.value.device))
    Zo = Zo + shift
    ora.inducing_variable = O.inducingpoint_wrapper(Zo)
    steps(4)
    e_o = ora.elbo((X, Y))
    assert abs(float(graph.elbo((Xd, Yd))) - e_o) < 1e-9 * abs(e_o)


def test_graph_replay_error_path_restores_state():
    """A replayed step that fails its status check leaves the state untouched and raises like the eager path."""
    p = pkg()
    X, Y, Z = synthetic(N=300, M=16, D=2, P=1, lik="gaussian", seed=2)
    Xd, Yd = torch.as_tensor(X, device="cuda:0"), torch.as_tensor(Y, device="cuda:0")
    m = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Gaussian(0.1), Z, use_graph=True)
    for _ in range(3):
        m.natgrad_step((Xd, Yd), lr=0.5)
    assert any(isinstance(e, dict) for e in m._graphs.values())
    bad = m.lambda_2_sqrt.value.clone()
    bad[0, 3, 3] = float("nan")  # poisons W: the prelude factorisation reports failure
    m.sites.assign_lambda_2_sqrt(bad)
    l1 = m.lambda_1.numpy().copy()
    with pytest.raises(FloatingPointError):
        m.natgrad_step((Xd, Yd), lr=0.5)
    assert np.array_equal(m.lambda_1.numpy(), l1)
    assert np.isnan(m.lambda_2_sqrt.numpy()[0, 3, 3])


@pytest.mark.parametrize("lik", ["gaussian", "bernoulli"])
@pytest.mark.parametrize("projection", ["auto", "whitened"])
def test_fp32_direct_route_against_fp64_oracle(lik, projection):
    """The configuration of BASELINE configs[2] in small: fp32 N-sized arrays with a well-conditioned K_uu (D = 16,
    cond ~ 3), where "auto" takes the direct route (gate cond <= 30 in fp32).  Both routes meet the fp32 tolerances."""
    p = pkg()
    rng = np.random.RandomState(51)
    N, M, D = 4000, 128, 16
    X, Y, _ = synthetic(N=N, M=M, D=D, P=1, lik=lik, seed=11)
    Z = X[:M].copy()
    mk = lambda mod, **kw: mod.t_SVGP(mod.SquaredExponential(1.0, 1.0), mod.Gaussian(0.1) if lik == "gaussian" else mod.Bernoulli(),
                                      Z, num_data=N, **kw)
    hip, ora = mk(p, compute_dtype=torch.float32, projection=projection), mk(O)
    assert hip._use_direct(1e-9) == [projection == "auto"]
    for _ in range(4):
        hip.natgrad_step((X, Y), lr=0.8)
        ora.natgrad_step((X, Y), lr=0.8)
    e_h, e_o = float(hip.elbo((X, Y))), float(ora.elbo((X, Y)))
    assert abs(e_h - e_o) / abs(e_o) < 1e-4
    mu_h, var_h = hip.predict_f(X[:300] + 0.05)
    mu_o, var_o = ora.predict_f(X[:300] + 0.05)
    np.testing.assert_allclose(mu_h.cpu().numpy(), mu_o, rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(var_h.cpu().numpy(), var_o, rtol=1e-3, atol=1e-4)
    assert relerr(hip.lambda_1.numpy(), ora.lambda_1) < 1e-3


def test_ill_conditioned_kuu_takes_the_projected_route():
    """cond(K_uu + 1e-9 I) ~ 7e10 (1-D inputs, 219 redundant inducing points: a typical 1-D demo geometry).  The cheaper
    routes lose the definiteness of -2 lambda_2 + jitter I there (their M x M back-mapping amplifies rounding by
    |K_uu^-1|); "auto" takes the projected route, which like the reference forms G1 as a sum of outer products.  The
    natural parameters themselves are only determined to ~cond * eps in this regime (for ANY arithmetic order), so the
    comparison is on what is well posed: ELBO and predictions."""
    p = pkg()
    rng = np.random.RandomState(3)
    N, M = 1500, 219
    X = rng.randn(N, 1)
    Y = (np.sin(3 * X) + 0.3 * rng.randn(N, 1) > 0).astype(float)
    Z = rng.randn(M, 1) * 1.5
    mk = lambda mod, **kw: mod.t_SVGP(mod.SquaredExponential(0.58, 1.17), mod.Bernoulli(), Z, num_data=N, **kw)
    hip, ora = mk(p), mk(O)
    assert hip._routes(1e-9) == ["projected"] and hip._cond_cache[1][0] > 1e10
    for _ in range(3):
        hip.natgrad_step((X, Y), lr=0.7)
        ora.natgrad_step((X, Y), lr=0.7)
    e_h, e_o = float(hip.elbo((X, Y))), float(ora.elbo((X, Y)))
    assert abs(e_h - e_o) / abs(e_o) < 1e-8
    mu_h, var_h = hip.predict_f(X[:200])
    mu_o, var_o = ora.predict_f(X[:200])
    assert relerr(mu_h.cpu().numpy(), mu_o) < 1e-6 and relerr(var_h.cpu().numpy(), var_o) < 1e-6
    # the ladder: forced onto the whitened route the final factorisation fails and the step is redone projected
    forced = mk(p)
    forced.WHITENED_MAX_COND = {torch.float64: float("inf"), torch.float32: float("inf")}
    assert forced._routes(1e-9) == ["whitened"]
    forced.natgrad_step((X, Y), lr=0.7)
    assert forced._routes(1e-9) == ["projected"]  # remembered until the parameters change
    one = mk(O)
    one.natgrad_step((X, Y), lr=0.7)
    assert abs(float(forced.elbo((X, Y))) - one.elbo((X, Y))) < 1e-8 * abs(one.elbo((X, Y)))


@pytest.mark.parametrize("lik", ["gaussian", "bernoulli"])
def test_execution_options_do_not_change_the_step(lik):
    """overlap_fill (K(X, Z) fill on a side stream beside the M x M prelude), use_graph, cache_whitened and (Gaussian)
    skip_unused_variance are execution options: the state after the same steps is the same to rounding, on device tensors
    that are reused from call to call (the side stream must order itself against the previous step's readers of the
    K(X, Z) buffer and against predict_f / elbo calls in between)."""
    p = pkg()
    rng = np.random.RandomState(17)
    X, Y, _ = synthetic(N=3000, M=130, D=5, P=1, lik=lik, seed=10)
    Z = rng.randn(130, 5) * 1.4
    Xd, Yd = torch.as_tensor(X, device="cuda:0"), torch.as_tensor(Y, device="cuda:0")
    mk = lambda **kw: p.t_SVGP(p.SquaredExponential(1.0, 1.1), p.Gaussian(0.15) if lik == "gaussian" else p.Bernoulli(), Z,
                               num_data=3000, **kw)
    variants = {"plain": mk(overlap_fill=False), "overlap": mk(), "graph": mk(use_graph=True), "warm": mk(cache_whitened=True),
                "whitened+overlap": mk(projection="whitened"), "graph+fork": mk(use_graph=True),
                "whitened+graph+fork": mk(projection="whitened", use_graph=True)}
    for name in ("graph+fork", "whitened+graph+fork"):  # capture as a shard of N * M >= 1e8 is captured: fill forked inside
        variants[name].GRAPH_FORK_MIN_NM = 0
    if lik == "gaussian":
        variants["skip"] = mk(skip_unused_variance=True)
        variants["skip+graph+warm-off"] = mk(skip_unused_variance=True, use_graph=True)
    elbos = {}
    for name, m in variants.items():
        for i in range(6):
            m.natgrad_step((Xd, Yd), lr=0.6)
            if i == 2:
                m.predict_f(Xd[:300])  # overwrites the work buffers between two steps
        elbos[name] = float(m.elbo((Xd, Yd)))
    ref = variants["plain"]
    for name, m in variants.items():
        tol = 1e-9 if "whitened" in name else 1e-12
        assert relerr(m.lambda_1.numpy(), ref.lambda_1.numpy()) < tol, name
        assert relerr(m.lambda_2.cpu().numpy(), ref.lambda_2.cpu().numpy()) < tol, name
        assert abs(elbos[name] - elbos["plain"]) < 1e-10 * abs(elbos["plain"]), name


def test_auto_graph_with_changing_minibatches():
    """use_graph="auto" (the default) at a launch-bound size, fed a DIFFERENT minibatch on every call as fresh host
    arrays: the device copies tend to land on recycled addresses, so a graph captured for one batch is replayed for
    another -- it must read the data that is there now.  State against the oracle after every step."""
    p = pkg()
    rng = np.random.RandomState(23)
    X, Y, _ = synthetic(N=2400, M=48, D=3, P=1, lik="bernoulli", seed=12)
    Z = rng.randn(48, 3) * 1.3
    hip = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Bernoulli(), Z, num_data=2400)
    ora = O.t_SVGP(O.SquaredExponential(1.0, 1.0), O.Bernoulli(), Z, num_data=2400)
    assert hip.use_graph == "auto"
    for i in range(8):
        sl = slice(300 * i, 300 * (i + 1))
        hip.natgrad_step((X[sl].copy(), Y[sl].copy()), lr=0.3)
        ora.natgrad_step((X[sl], Y[sl]), lr=0.3)
        _compare_state(hip, ora, 1e-8)
    # and with device tensors that stay put: captured on the second call, replayed afterwards
    Xd, Yd = torch.as_tensor(X[:300], device="cuda:0"), torch.as_tensor(Y[:300], device="cuda:0")
    for i in range(5):
        hip.natgrad_step((Xd, Yd), lr=0.3)
        ora.natgrad_step((X[:300], Y[:300]), lr=0.3)
        _compare_state(hip, ora, 1e-8)
    assert any(isinstance(e, dict) for e in hip._graphs.values())


@pytest.mark.parametrize("separate", [False, True])
@pytest.mark.parametrize("lik", ["gaussian", "bernoulli"])
def test_predict_y_and_log_density_match_oracle(lik, separate):
    """``predict_y`` / ``predict_log_density`` (GPflow GPModel [ext]; the NLPD the reference's drivers log,
    experiments/uci_regression.py:157, uci_classification.py:139) on held-out points against the oracle, Gaussian and
    Bernoulli, one shared kernel and one kernel per latent."""
    p = pkg()
    P, M, D = 2, 40, 3
    X, Y, Z = synthetic(N=700, M=M, D=D, P=P, lik=lik, seed=12)
    Xt, Yt, _ = synthetic(N=150, M=M, D=D, P=P, lik=lik, seed=13)
    mk = lambda mod: (mod.SeparateIndependent([mod.SquaredExponential(1.0, 0.9), mod.SquaredExponential(0.8, 1.3)])
                      if separate else mod.SquaredExponential(1.1, 1.2))
    iv = lambda mod: mod.SharedIndependentInducingVariables(Z) if separate else Z
    hip = p.t_SVGP(mk(p), p.Gaussian(0.1) if lik == "gaussian" else p.Bernoulli(), iv(p), num_latent_gps=P)
    ora = O.t_SVGP(mk(O), O.Gaussian(0.1) if lik == "gaussian" else O.Bernoulli(), iv(O), num_latent_gps=P)
    for _ in range(3):
        hip.natgrad_step((X, Y), lr=0.8)
        ora.natgrad_step((X, Y), lr=0.8)
    ymu_h, yvar_h = hip.predict_y(Xt)
    ymu_o, yvar_o = ora.predict_y(Xt)
    assert relerr(ymu_h.cpu().numpy(), ymu_o) < 1e-8 and relerr(yvar_h.cpu().numpy(), yvar_o) < 1e-8
    ld_h = hip.predict_log_density((Xt, Yt)).cpu().numpy()
    ld_o = ora.predict_log_density((Xt, Yt))
    assert ld_h.shape == ld_o.shape == (150,)
    assert relerr(ld_h, ld_o) < 1e-8
    nlpd_h, nlpd_o = -float(np.mean(ld_h)), -float(np.mean(ld_o))
    assert abs(nlpd_h - nlpd_o) < 1e-9 * abs(nlpd_o)


@pytest.mark.parametrize("batched", [True, False])
def test_p8_separate_kernels_match_oracle(batched):
    """BASELINE configs[4] at small N: P = 8 latents, one SE kernel per latent with l_p = linspace(0.8, 1.5, 8) on shared
    inducing points, M = 192.  `batched`: the latent-batched launches (fill / in-place whitening / moments / site sums,
    one each: tsvgp_*_batched_*) against the one-pass-per-latent path; both against the oracle.  Long lengthscales are
    whitened, short ones direct ("auto" decides per latent), so the batched whitening runs on a sub-range of the latents."""
    p = pkg()
    P, M, D = 8, 192, 8
    X, Y, Z = synthetic(N=1500, M=M, D=D, P=P, lik="gaussian", seed=5)
    ls = np.linspace(0.8, 1.5, P)
    hip = p.t_SVGP(p.SeparateIndependent([p.SquaredExponential(1.0, float(l)) for l in ls]), p.Gaussian(0.1),
                   p.SharedIndependentInducingVariables(Z), num_latent_gps=P)
    ora = O.t_SVGP(O.SeparateIndependent([O.SquaredExponential(1.0, float(l)) for l in ls]), O.Gaussian(0.1),
                   O.SharedIndependentInducingVariables(Z), num_latent_gps=P)
    eng = hip._get_engine()
    eng.batch_separate = batched
    hip._routes(1e-9)
    conds = sorted(hip._cond_cache[1])
    hip.DIRECT_MAX_COND = {torch.float64: 0.5 * (conds[2] + conds[3])}  # this model only: the five longest lengthscales whiten
    routes = hip._routes(1e-9)
    assert routes == ["direct"] * 3 + ["whitened"] * 5
    for _ in range(3):
        hip.natgrad_step((X, Y), lr=0.8)
        ora.natgrad_step((X, Y), lr=0.8)
        _compare_state(hip, ora, 1e-8)
    assert eng.last_batched == batched
    e_h, e_o = float(hip.elbo((X, Y))), float(ora.elbo((X, Y)))
    assert abs(e_h - e_o) / abs(e_o) < 1e-9
    mu_h, var_h = hip.predict_f(X[:200] + 0.05)
    mu_o, var_o = ora.predict_f(X[:200] + 0.05)
    assert relerr(mu_h.cpu().numpy(), mu_o) < 1e-8 and relerr(var_h.cpu().numpy(), var_o) < 1e-8


def test_large_input_dimension_matches_oracle():
    """D = 48 inputs (beyond the fused fill kernel's 32): E-steps, ELBO and predictions against the oracle."""
    X, Y, Z = synthetic(N=600, M=40, D=48, P=1, lik="bernoulli", seed=14)
    X, Z = X / np.sqrt(48.0), Z / np.sqrt(48.0)
    hip, ora = make_pair(Z, lik="bernoulli", lengthscales=0.8)
    for _ in range(3):
        hip.natgrad_step((X, Y), lr=0.8)
        ora.natgrad_step((X, Y), lr=0.8)
        _compare_state(hip, ora, 1e-8)
    e_h, e_o = float(hip.elbo((X, Y))), float(ora.elbo((X, Y)))
    assert abs(e_h - e_o) / abs(e_o) < 1e-9
    mu_h, var_h = hip.predict_f(X[:100] + 0.01)
    mu_o, var_o = ora.predict_f(X[:100] + 0.01)
    assert relerr(mu_h.cpu().numpy(), mu_o) < 1e-8 and relerr(var_h.cpu().numpy(), var_o) < 1e-8


def test_clock_keeper_does_not_change_a_step():
    """The opt-in clock keeper (EStepEngine.keeper_begin / keeper_end around the M x M prelude and epilogue) only runs register
    arithmetic on a side stream: a trajectory with it is the trajectory without it, bit for bit -- eagerly and replayed from a
    hipGraph (the side stream joins the capture)."""
    p = pkg()
    X, Y, Z = synthetic(N=3000, M=130, D=3, lik="gaussian", seed=11)
    Xd, Yd = torch.as_tensor(X, device="cuda:0"), torch.as_tensor(Y, device="cuda:0")
    for use_graph in (False, True):
        plain = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Gaussian(0.1), Z, use_graph=use_graph)
        kept = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Gaussian(0.1), Z, use_graph=use_graph)
        kept.KEEPER_MIN_NM = 0
        kept._get_engine().clock_keeper = -1  # (the default, TSVGP_CLOCK_KEEPER unset, is 0: off)
        assert plain._get_engine().clock_keeper == 0
        for i in range(4):
            plain.natgrad_step((Xd, Yd), lr=0.6)
            kept.natgrad_step((Xd, Yd), lr=0.6)
            assert kept._keep_clock and not plain._keep_clock
            assert relerr(kept.lambda_1.numpy(), plain.lambda_1.numpy()) == 0.0
            assert relerr(kept.lambda_2.cpu().numpy(), plain.lambda_2.cpu().numpy()) == 0.0
