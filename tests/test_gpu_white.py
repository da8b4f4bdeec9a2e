"""t_SVGP_white (reference src/models/tsvgp_white.py) on the GPU: the HIP path against the oracle and
the reference's own relational tests restated on the HIP model.  Tolerances as tests/test_gpu_model.py."""
import os

import numpy as np
import pytest

from oracle import tsvgp_oracle as O
from tests.helpers import pkg, relerr, synthetic
from tests.test_golden_cpu import WHITE_FIXTURES, load_white_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("lik", ["gaussian", "bernoulli"])
def test_white_steps_match_oracle_fp64(lik):
    p = pkg()
    rng = np.random.RandomState(31)
    X, Y, _ = synthetic(N=900, M=48, D=4, P=1, lik=lik, seed=5)
    Z = rng.randn(48, 4) * 1.3
    mk = lambda mod: mod.t_SVGP_white(mod.SquaredExponential(1.2, 0.9), mod.Gaussian(0.2) if lik == "gaussian" else mod.Bernoulli(),
                                      Z, num_data=900)
    hip, ora = mk(p), mk(O)
    assert abs(float(hip.elbo((X, Y))) - ora.elbo((X, Y))) < 1e-9 * abs(ora.elbo((X, Y)))
    for _ in range(6):
        hip.natgrad_step((X, Y), lr=0.7)
        ora.natgrad_step((X, Y), lr=0.7)
        assert relerr(hip.lambda_1.numpy(), ora.lambda_1) < 1e-8
        assert relerr(hip.lambda_2.numpy(), ora.lambda_2) < 1e-8
    assert abs(float(hip.elbo((X, Y))) - ora.elbo((X, Y))) < 1e-9 * abs(ora.elbo((X, Y)))
    assert abs(float(hip.prior_kl()) - ora.prior_kl()) < 1e-8 * abs(ora.prior_kl())
    mu_h, var_h = hip.predict_f(X[:150] + 0.05)
    mu_o, var_o = ora.predict_f(X[:150] + 0.05)
    assert relerr(mu_h.cpu().numpy(), mu_o) < 1e-8 and relerr(var_h.cpu().numpy(), var_o) < 1e-8
    m_h, cS_h = hip.get_mean_chol_cov_inducing_posterior()
    m_o, cS_o = ora.get_mean_chol_cov_inducing_posterior()
    assert relerr(m_h.cpu().numpy(), m_o) < 1e-8 and relerr(cS_h.cpu().numpy(), cS_o) < 1e-7


@pytest.mark.parametrize("path", WHITE_FIXTURES, ids=[os.path.basename(p)[:-4] for p in WHITE_FIXTURES])
def test_white_hip_matches_fixture(path):
    fx = np.load(path)
    model = load_white_model(fx, pkg())
    X, Y, lr = fx["X"], fx["Y"], float(fx["lr"])
    steps = [int(s) for s in fx["steps"]]
    for step in range(1, max(steps) + 1):
        if step in steps:  # moments of THIS step, from the pre-step state like the fixture's
            mu, var = model.predict_f(X)
            assert relerr(mu.cpu().numpy(), fx[f"s{step}_mean"]) < 1e-8 and relerr(var.cpu().numpy(), fx[f"s{step}_var"]) < 1e-8
        model.natgrad_step((X, Y), lr=lr)
        if step in steps:
            assert relerr(model.lambda_1.numpy(), fx[f"s{step}_lambda_1"]) < 1e-8
            assert relerr(model.lambda_2.numpy(), fx[f"s{step}_lambda_2"]) < 1e-8
            e = float(model.elbo((X, Y)))
            assert abs(e - float(fx[f"s{step}_elbo"])) < 1e-9 * abs(float(fx[f"s{step}_elbo"]))
            mu, var = model.predict_f(fx["Xs"])
            assert relerr(mu.cpu().numpy(), fx[f"s{step}_pred_mean"]) < 1e-8
            assert relerr(var.cpu().numpy(), fx[f"s{step}_pred_var"]) < 1e-8


def test_white_reference_relations_on_hip():
    """reference tests/models/test_tsvgp_white.py:64-115 and tests/models/test_condit.py:69-83 on the HIP models
    (decimal=4 as there): white == unwhitened t-SVGP before and after a step; exact-GP optimum; SGPR equality."""
    p = pkg()
    rng = np.random.RandomState(123)
    func = lambda x: np.sin(x * 3 * 3.14) + 0.3 * np.cos(x * 9 * 3.14) + 0.5 * np.sin(x * 7 * 3.14)
    X = rng.rand(8, 1) * 2 - 1
    Y = func(X) + 0.2 * rng.randn(8, 1)
    k = lambda mod: mod.SquaredExponential(variance=2.25, lengthscales=2.0)
    plain, white = p.t_SVGP(k(p), p.Gaussian(0.3), X.copy()), p.t_SVGP_white(k(p), p.Gaussian(0.3), X.copy())
    np.testing.assert_almost_equal(float(plain.elbo((X, Y))), float(white.elbo((X, Y))), decimal=4)
    plain.natgrad_step((X, Y), lr=0.9)
    white.natgrad_step((X, Y), lr=0.9)
    for a, b in zip(plain.predict_f(X), white.predict_f(X)):
        np.testing.assert_array_almost_equal(a.cpu().numpy(), b.cpu().numpy(), decimal=4)
    opt = p.t_SVGP_white(k(p), p.Gaussian(0.3), X.copy())
    opt.natgrad_step((X, Y * 0), lr=1.0)
    np.testing.assert_almost_equal(float(opt.elbo((X, Y * 0))), O.gpr_log_marginal_likelihood(k(O), X, Y * 0, 0.3), decimal=4)
    rs = np.random.RandomState(0)
    Xs, Ys, Zs = rs.randn(10, 1), rs.randn(10, 1), rs.randn(3, 1)
    w = p.t_SVGP_white(p.SquaredExponential(1.0, 1.0), p.Gaussian(0.3), Zs)
    w.natgrad_step((Xs, Ys), lr=1.0)
    m1, v1 = w.predict_f(Ys)
    m2, v2 = O.sgpr_predict_f(O.SquaredExponential(1.0, 1.0), Xs, Ys, Zs, 0.3, Ys)
    np.testing.assert_array_almost_equal(m1.cpu().numpy(), m2, decimal=4)
    np.testing.assert_array_almost_equal(v1.cpu().numpy(), v2, decimal=4)


def test_white_api_and_errors():
    p = pkg()
    X, Y, Z = synthetic(N=300, M=16, D=2, P=1, lik="gaussian", seed=2)
    m = p.t_SVGP_white(p.SquaredExponential(1.0, 1.0), p.Gaussian(0.1), Z)
    assert m.lambda_1.shape == (16, 1) and m.lambda_2.shape == (1, 16, 16)
    assert m.natgrad_step((X, Y), lr=0.5) is None  # mutates in place
    assert float(m.lambda_2.value.abs().max()) > 1e-3
    with pytest.raises(NotImplementedError):
        p.t_SVGP_white(p.SquaredExponential(1.0, 1.0), p.Gaussian(0.1), Z, num_latent_gps=2)
    l1, L2 = m.lambda_1.numpy().copy(), m.lambda_2.numpy().copy()
    m.kernel.lengthscales.assign(1e-9)  # K(X, Z) = 0: var = kff > 0 still; make the likelihood parameter invalid instead
    m.kernel.lengthscales.assign(1.0)
    with pytest.raises(ValueError):
        m.natgrad_step((X, Y[:, :0]), lr=0.5)  # Y must be [N, 1]
    assert np.array_equal(m.lambda_1.numpy(), l1) and np.array_equal(m.lambda_2.numpy(), L2)


@pytest.mark.parametrize("lik", ["gaussian", "bernoulli"])
def test_white_predict_f_extra_data(lik):
    """predict_f_extra_data (tsvgp_white.py:134-160) on the HIP path: against the oracle, and the reference's own
    relation (tests/models/test_condit.py:84-104: stepping on data + extra == conditioning on extra, Gaussian, lr=1)."""
    p = pkg()
    rng = np.random.RandomState(41)
    X, Y, _ = synthetic(N=700, M=40, D=3, P=1, lik=lik, seed=6)
    Xe, Ye, _ = synthetic(N=450, M=40, D=3, P=1, lik=lik, seed=7)
    Z = rng.randn(40, 3) * 1.3
    mk = lambda mod: mod.t_SVGP_white(mod.SquaredExponential(1.1, 1.0), mod.Gaussian(0.2) if lik == "gaussian" else mod.Bernoulli(), Z)
    hip, ora = mk(p), mk(O)
    for _ in range(3):
        hip.natgrad_step((X, Y), lr=0.8)
        ora.natgrad_step((X, Y), lr=0.8)
    l1, L2 = hip.lambda_1.numpy().copy(), hip.lambda_2.numpy().copy()
    Xs = X[:200] + 0.05
    for kw in ({}, dict(jitter=1e-5)):
        mh, vh = hip.predict_f_extra_data(Xs, (Xe, Ye), **kw)
        mo, vo = ora.predict_f_extra_data(Xs, (Xe, Ye), **kw)
        assert relerr(mh.cpu().numpy(), mo) < 1e-8 and relerr(vh.cpu().numpy(), vo) < 1e-8
    assert np.array_equal(hip.lambda_1.numpy(), l1) and np.array_equal(hip.lambda_2.numpy(), L2)
    if lik == "gaussian":
        one, both = mk(p), mk(p)
        one.natgrad_step((X, Y), lr=1.0)
        both.natgrad_step((np.vstack([X, Xe]), np.vstack([Y, Ye])), lr=1.0)
        m1, v1 = both.predict_f(Xs)
        m2, v2 = one.predict_f_extra_data(Xs, extra_data=(Xe, Ye))
        np.testing.assert_array_almost_equal(m1.cpu().numpy(), m2.cpu().numpy(), decimal=4)
        np.testing.assert_array_almost_equal(v1.cpu().numpy(), v2.cpu().numpy(), decimal=4)


@pytest.mark.parametrize("lik", ["gaussian", "bernoulli"])
def test_white_compute_data_natural_params(lik):
    """compute_data_natural_params (tsvgp_white.py:181-209) on the HIP path, every projection route, against the oracle:
    [G0 - 2 G1 meanZ [M, 1], G1 [1, M, M]]; K_uu applied to it is what natgrad_step adds to the sites (:244-245)."""
    p = pkg()
    rng = np.random.RandomState(43)
    X, Y, _ = synthetic(N=800, M=40, D=3, P=1, lik=lik, seed=8)
    Z = rng.randn(40, 3) * 1.3
    mk = lambda mod: mod.t_SVGP_white(mod.SquaredExponential(1.1, 1.0), mod.Gaussian(0.2) if lik == "gaussian" else mod.Bernoulli(), Z)
    hip, ora = mk(p), mk(O)
    for _ in range(2):
        hip.natgrad_step((X, Y), lr=0.8)
        ora.natgrad_step((X, Y), lr=0.8)
    l1, L2 = hip.lambda_1.numpy().copy(), hip.lambda_2.numpy().copy()
    want = ora.compute_data_natural_params((X, Y))
    K = O.Kuu(ora.inducing_variable, ora.kernel)
    tol = max(1e-8, 1000 * np.linalg.cond(K + 1e-9 * np.eye(40)) * 2.2e-16)  # explicit K9^-1 products on both sides
    for projection in ("auto", "whitened", "direct"):
        hip.projection = projection
        got = hip.compute_data_natural_params((X, Y))
        assert tuple(got[0].shape) == (40, 1) and tuple(got[1].shape) == (1, 40, 40)
        assert relerr(got[0].cpu().numpy(), want[0]) < tol and relerr(got[1].cpu().numpy(), want[1]) < tol
    assert np.array_equal(hip.lambda_1.numpy(), l1) and np.array_equal(hip.lambda_2.numpy(), L2)
    # a full step (lr = 1) from these sums is the state natgrad_step itself reaches
    hip.projection = "auto"
    got = hip.compute_data_natural_params((X, Y))
    hip.natgrad_step((X, Y), lr=1.0)
    Kt = hip.lambda_1.value.new_tensor(K)
    assert relerr(hip.lambda_1.numpy(), (Kt @ got[0]).cpu().numpy()) < tol
    assert relerr(hip.lambda_2.numpy(), (-2.0 * Kt @ got[1] @ Kt).cpu().numpy()) < tol


@pytest.mark.parametrize("lik", ["gaussian", "bernoulli"])
@pytest.mark.parametrize("projection", ["direct", "whitened", "auto"])
def test_white_projection_routes_match_oracle(lik, projection):
    """The direct route of t_SVGP_white (no N-sized whitening: moments on K(X, Z) with the factor of Q = K6^-1 - R^-1,
    sums over k k^T) and the whitened one are the same algebra; on a well-conditioned K_uu both meet the fp64 tolerance
    against the oracle and "auto" takes the direct one."""
    p = pkg()
    rng = np.random.RandomState(51)
    X, Y, _ = synthetic(N=1100, M=60, D=5, P=1, lik=lik, seed=9)
    Z = rng.randn(60, 5) * 1.5
    mk = lambda mod, **kw: mod.t_SVGP_white(mod.SquaredExponential(1.1, 1.0), mod.Gaussian(0.2) if lik == "gaussian" else mod.Bernoulli(),
                                            Z, num_data=1100, **kw)
    hip, ora = mk(p, projection=projection), mk(O)
    assert hip._use_direct() == (projection != "whitened")
    for _ in range(5):
        hip.natgrad_step((X, Y), lr=0.7)
        ora.natgrad_step((X, Y), lr=0.7)
        assert relerr(hip.lambda_1.numpy(), ora.lambda_1) < 1e-8
        assert relerr(hip.lambda_2.numpy(), ora.lambda_2) < 1e-8
    assert abs(float(hip.elbo((X, Y))) - ora.elbo((X, Y))) < 1e-9 * abs(ora.elbo((X, Y)))
    mu_h, var_h = hip.predict_f(X[:150] + 0.05)
    mu_o, var_o = ora.predict_f(X[:150] + 0.05)
    assert relerr(mu_h.cpu().numpy(), mu_o) < 1e-8 and relerr(var_h.cpu().numpy(), var_o) < 1e-8


def test_white_auto_route_gate_and_fallback():
    """"auto" stays whitened on an ill-conditioned K_uu; a forced direct route on such a K_uu either matches or reports
    its own failure, and "auto" after a failure keeps working on the whitened route."""
    p = pkg()
    X, Y, Z = synthetic(N=800, M=64, D=1, P=1, lik="gaussian", seed=3)  # Z = X[:M] in 1-D: cond(K_uu) huge
    hip = p.t_SVGP_white(p.SquaredExponential(1.0, 1.0), p.Gaussian(0.1), Z)
    ora = O.t_SVGP_white(O.SquaredExponential(1.0, 1.0), O.Gaussian(0.1), Z)
    assert not hip._use_direct()
    hip.natgrad_step((X, Y), lr=0.8)
    ora.natgrad_step((X, Y), lr=0.8)
    assert relerr(hip.lambda_2.numpy(), ora.lambda_2) < 1e-7
    # simulate a direct-route failure on a model that would choose it
    rng = np.random.RandomState(4)
    X2, Y2, _ = synthetic(N=500, M=32, D=4, P=1, lik="gaussian", seed=5)
    Z2 = rng.randn(32, 4) * 1.5
    h2 = p.t_SVGP_white(p.SquaredExponential(1.0, 1.0), p.Gaussian(0.1), Z2)
    o2 = O.t_SVGP_white(O.SquaredExponential(1.0, 1.0), O.Gaussian(0.1), Z2)
    assert h2._use_direct()
    h2._direct_failed = True
    assert not h2._use_direct()
    h2.natgrad_step((X2, Y2), lr=0.8)
    o2.natgrad_step((X2, Y2), lr=0.8)
    assert relerr(h2.lambda_2.numpy(), o2.lambda_2) < 1e-8


@pytest.mark.parametrize("lik", ["gaussian", "bernoulli"])
def test_white_indefinite_lambda_2_takes_the_two_product_variance(lik):
    """Lambda_2 + 1e-9 I NOT positive definite while K + Lambda_2 + 1e-9 I is: the reference goes on (its conditional only
    factors the latter, src/util.py:73-86; the class does not crop d ve / d var, src/models/tsvgp_white.py:188-191, so the
    probit link's far tails can drive Lambda_2 there), and so must the mirror -- round 2 raised FloatingPointError here because
    its single-product variance needs the factor of Lambda_2 + 1e-9 I.  Sites with Lambda_2 = -0.45 K_uu (negative definite)
    and a random lambda_1: predictions, ELBO, conditioning on extra data and natural-gradient steps against the oracle."""
    p = pkg()
    rng = np.random.RandomState(32)
    N, M, D = 700, 40, 4
    X, Y, _ = synthetic(N=N, M=M, D=D, P=1, lik=lik, seed=6)
    Z = rng.randn(M, D) * 1.3
    kern = O.SquaredExponential(1.2, 0.9)
    Kuu = kern.K(Z)
    lam2 = (-0.45 * Kuu)[None]
    lam1 = 0.3 * rng.randn(M, 1)
    assert np.linalg.eigvalsh(lam2[0] + 1e-9 * np.eye(M)).max() < 0  # no Cholesky factor of Lambda_2 + 1e-9 I
    assert np.linalg.eigvalsh(lam2[0] + Kuu + (1e-6 + 1e-9) * np.eye(M)).min() > 0  # the reference's R is fine
    mk = lambda mod: mod.t_SVGP_white(mod.SquaredExponential(1.2, 0.9), mod.Gaussian(0.2) if lik == "gaussian" else mod.Bernoulli(),
                                      Z, num_data=N, lambda_1=lam1.copy(), lambda_2=lam2.copy())
    hip, ora = mk(p), mk(O)
    mu_h, var_h = hip.predict_f(X[:200] + 0.05)
    assert hip._two_product  # the single-product form was tried, failed in its own factorisation, and the model moved over
    mu_o, var_o = ora.predict_f(X[:200] + 0.05)
    assert relerr(mu_h.cpu().numpy(), mu_o) < 1e-8 and relerr(var_h.cpu().numpy(), var_o) < 1e-8
    assert abs(float(hip.elbo((X, Y))) - ora.elbo((X, Y))) < 1e-9 * abs(ora.elbo((X, Y)))
    mu_h, var_h = hip.predict_f_extra_data(X[:100] - 0.1, (X[300:380], Y[300:380]))
    mu_o, var_o = ora.predict_f_extra_data(X[:100] - 0.1, (X[300:380], Y[300:380]))
    assert relerr(mu_h.cpu().numpy(), mu_o) < 1e-8 and relerr(var_h.cpu().numpy(), var_o) < 1e-8
    for _ in range(3):
        hip.natgrad_step((X, Y), lr=0.3)
        ora.natgrad_step((X, Y), lr=0.3)
        assert relerr(hip.lambda_1.numpy(), ora.lambda_1) < 1e-8
        assert relerr(hip.lambda_2.numpy(), ora.lambda_2) < 1e-8
    assert abs(float(hip.elbo((X, Y))) - ora.elbo((X, Y))) < 1e-9 * abs(ora.elbo((X, Y)))
    # a fresh model on definite sites never leaves the single-product form
    fresh = p.t_SVGP_white(p.SquaredExponential(1.2, 0.9), p.Gaussian(0.2), Z, num_data=N)
    fresh.natgrad_step((X, Y if lik == "gaussian" else Y + 0.0), lr=0.5)
    assert not fresh._two_product
