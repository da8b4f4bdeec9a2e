"""Pins the CPU oracle (oracle/tsvgp_oracle.py) with the reference's own relational tests.

The reference holds no golden vectors; its tests compare t_SVGP against closed forms or
against GPflow models.  The ones below are restated with GPflow-independent closed forms:

* reference ``tests/models/test_tsvgp.py:106-165`` -- exact-GP equalities at the Gaussian fixed point
* reference ``tests/models/test_tsvgp.py:123-131`` -- t-SVGP E-step == SVGP natural gradient (gamma=1);
  the SVGP side is an independent torch-autograd implementation written here (GPflow is not installed)
* reference ``tests/test_utils.py:44-137``         -- site-form algebra identities
"""
import numpy as np
import pytest
import torch

from oracle import tsvgp_oracle as O

LENGTH_SCALE = 2.0
VARIANCE = 2.25
NUM_DATA = 8
NOISE_VARIANCE = 0.3


def _setup(rng):
    """reference tests/models/test_tsvgp.py:91-103."""

    def func(x):
        return np.sin(x * 3 * 3.14) + 0.3 * np.cos(x * 9 * 3.14) + 0.5 * np.sin(x * 7 * 3.14)

    input_points = rng.rand(NUM_DATA, 1) * 2 - 1
    observations = func(input_points) + 0.2 * rng.randn(NUM_DATA, 1)
    kernel = O.SquaredExponential(lengthscales=LENGTH_SCALE, variance=VARIANCE)
    return input_points, observations, kernel, NOISE_VARIANCE


@pytest.fixture(name="tsvgp_gpr_optim_setup")
def _tsvgp_gpr_optim_setup():
    """reference tests/models/test_tsvgp.py:19-43 (GPR replaced by its closed form)."""
    rng = np.random.RandomState(123)
    X, Y, kernel, noise = _setup(rng)
    tsvgp = O.t_SVGP(kernel=kernel, likelihood=O.Gaussian(variance=noise), inducing_variable=O.InducingPoints(X))
    for _ in range(10):
        tsvgp.natgrad_step((X, Y), lr=0.9)
    return tsvgp, (X, Y, kernel, noise)


def test_tsvgp_elbo_optimal(tsvgp_gpr_optim_setup):
    """reference tests/models/test_tsvgp.py:106-110."""
    tsvgp, (X, Y, kernel, noise) = tsvgp_gpr_optim_setup
    np.testing.assert_almost_equal(tsvgp.elbo((X, Y)), O.gpr_log_marginal_likelihood(kernel, X, Y, noise), decimal=4)


def test_predictions_match_tsvgp_gpr_optimal(tsvgp_gpr_optim_setup):
    """reference tests/models/test_tsvgp.py:113-120."""
    tsvgp, (X, Y, kernel, noise) = tsvgp_gpr_optim_setup
    Xs = X + 1.0
    mu, var = tsvgp.predict_f(Xs)
    mu_gpr, var_gpr = O.gpr_predict_f(kernel, X, Y, noise, Xs)
    np.testing.assert_array_almost_equal(mu, mu_gpr, decimal=4)
    np.testing.assert_array_almost_equal(var, var_gpr, decimal=4)
    # and the alternative predictive of tsvgp.py:215-232 agrees with predict_f
    mu2, var2 = tsvgp.new_predict_f(Xs)
    np.testing.assert_array_almost_equal(mu, mu2, decimal=4)
    np.testing.assert_array_almost_equal(var, var2, decimal=4)


def test_tsvgp_unchanged_at_optimum(tsvgp_gpr_optim_setup):
    """reference tests/models/test_tsvgp.py:134-145."""
    tsvgp, (X, Y, _, _) = tsvgp_gpr_optim_setup
    optim_elbo = tsvgp.elbo((X, Y))
    tsvgp.natgrad_step((X, Y), lr=0.9)
    np.testing.assert_almost_equal(optim_elbo, tsvgp.elbo((X, Y)), decimal=4)


def test_tsvgp_minibatch_same_elbo(tsvgp_gpr_optim_setup):
    """reference tests/models/test_tsvgp.py:148-165 (both fixtures are the same object there)."""
    tsvgp, (X, Y, _, _) = tsvgp_gpr_optim_setup
    x = X[0].repeat(NUM_DATA)[:, None]
    y = Y[0].repeat(NUM_DATA)[:, None]
    elbo2 = tsvgp.elbo((x, y))  # num_data still None here, as in the reference (same object, set below)
    tsvgp.num_data = NUM_DATA
    elbo2 = tsvgp.elbo((x, y))
    elbo1 = tsvgp.elbo((X[0][:, None], Y[0][:, None]))
    np.testing.assert_almost_equal(elbo2, elbo1, decimal=4)


def test_optimal_gaussian_sites_closed_form(tsvgp_gpr_optim_setup):
    """With Z=X the fixed point of the dense site is lambda_1 = K^-1 A^T Y/s2 ... equivalently the
    posterior q(u) equals the exact GP posterior at X (closed form, independent of the E-step code)."""
    tsvgp, (X, Y, kernel, noise) = tsvgp_gpr_optim_setup
    m, cholS = tsvgp.get_mean_chol_cov_inducing_posterior()
    mu_gpr, _ = O.gpr_predict_f(kernel, X, Y, noise, X)
    np.testing.assert_array_almost_equal(m, mu_gpr, decimal=4)
    K = kernel.K(X)
    S_exact = K - K @ np.linalg.solve(K + noise * np.eye(NUM_DATA), K)
    np.testing.assert_array_almost_equal(cholS[0] @ cholS[0].T, S_exact, decimal=4)


# ---------------------------------------------------------------------------
# Bernoulli: independent SVGP + natural gradient (torch autograd, fp64, CPU)
# ---------------------------------------------------------------------------
def _torch_svgp_natgrad(X, Y, Z, lengthscale, variance, P, steps, gamma=1.0):
    """GPflow SVGP(whiten=False) + NaturalGradient(gamma) restated with autograd:
    theta <- theta - gamma * dLoss/d eta, (theta natural, eta expectation parameters)."""
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float64)
    X, Y, Z = t(X), t(Y), t(Z)
    M = Z.shape[0]

    def kern(A, B):
        d = (A[:, None, :] - B[None, :, :]) / lengthscale
        return variance * torch.exp(-0.5 * (d * d).sum(-1))

    Kmm = kern(Z, Z) + 1e-6 * torch.eye(M, dtype=torch.float64)
    Lm = torch.linalg.cholesky(Kmm)
    gh_x, gh_w = np.polynomial.hermite.hermgauss(20)
    gh_x, gh_w = t(gh_x * np.sqrt(2.0)), t(gh_w / np.sqrt(np.pi))

    def predict(Xn, q_mu, S):
        Kmn = kern(Z, Xn)
        A = torch.cholesky_solve(Kmn, Lm)  # Kmm^-1 Kmn [M,N]
        mean = A.T @ q_mu
        var = variance - (Kmn * A).sum(0)[:, None] + torch.einsum("mn,pmo,on->np", A, S, A)
        return mean, var

    def loss(eta1, eta2):
        q_mu = eta1  # [M,P]
        S = eta2 - torch.einsum("mp,op->pmo", eta1, eta1)  # [P,M,M]
        mean, var = predict(X, q_mu, S)
        F = mean[..., None] + torch.sqrt(var)[..., None] * gh_x
        p = 0.5 * (1 + torch.erf(F / np.sqrt(2.0))) * (1 - 2e-3) + 1e-3
        logp = torch.log(torch.where(Y[..., None] == 1, p, 1 - p))
        ve = (logp * gh_w).sum(-1).sum()
        Ls = torch.linalg.cholesky(S)
        alpha = torch.linalg.solve_triangular(Lm, q_mu, upper=False)
        LpiLq = torch.linalg.solve_triangular(Lm.expand(P, M, M), Ls, upper=False)
        kl = 0.5 * ((alpha**2).sum() - M * P - torch.log(torch.diagonal(Ls, dim1=-2, dim2=-1) ** 2).sum()
                    + (LpiLq**2).sum() + P * torch.log(torch.diagonal(Lm) ** 2).sum())
        return -(ve - kl)

    q_mu = torch.zeros(M, P, dtype=torch.float64)
    S = torch.eye(M, dtype=torch.float64).expand(P, M, M).clone()  # gpflow SVGP init q_sqrt = I
    for _ in range(steps):
        eta1 = q_mu.clone().requires_grad_(True)
        eta2 = (S + torch.einsum("mp,op->pmo", q_mu, q_mu)).clone().requires_grad_(True)
        g1, g2 = torch.autograd.grad(loss(eta1, eta2), [eta1, eta2])
        g2 = 0.5 * (g2 + g2.transpose(-1, -2))
        Sinv = torch.linalg.inv(S)
        th1 = torch.einsum("pmo,op->mp", Sinv, q_mu) - gamma * g1
        th2 = -0.5 * Sinv - gamma * g2
        S = torch.linalg.inv(-2.0 * th2)
        S = 0.5 * (S + S.transpose(-1, -2))
        q_mu = torch.einsum("pmo,op->mp", S, th1)
    return lambda Xn: tuple(a.numpy() for a in predict(t(Xn), q_mu, S))


@pytest.mark.parametrize("num_latent_gps", [1, 2])
@pytest.mark.parametrize("labels", ["reference", "binary"])
def test_predictions_match_tsvgp_qsvgp_optimal(num_latent_gps, labels):
    """reference tests/models/test_tsvgp.py:46-88,123-131.  'reference' label mode reproduces the
    fixture's (odd) ``observations *= rand`` rescale; 'binary' keeps true 0/1 labels."""
    rng = np.random.RandomState(123)
    X, obs, kernel, _ = _setup(rng)
    Y = np.tile((obs > 0.0).astype(float), [1, num_latent_gps])
    if labels == "reference":
        Y = Y * rng.rand(1, num_latent_gps)
    tsvgp = O.t_SVGP(kernel=kernel, likelihood=O.Bernoulli(), inducing_variable=O.InducingPoints(X),
                     num_latent_gps=num_latent_gps)
    for _ in range(20):
        tsvgp.natgrad_step((X, Y), lr=1.0)
    svgp_predict = _torch_svgp_natgrad(X, Y, X, LENGTH_SCALE, VARIANCE, num_latent_gps, steps=20)
    Xs = X + 0.1
    mu_t, var_t = tsvgp.new_predict_f(Xs)
    mu_q, var_q = svgp_predict(Xs)
    np.testing.assert_array_almost_equal(mu_t, mu_q, decimal=4)
    np.testing.assert_array_almost_equal(var_t, var_q, decimal=4)


def test_bernoulli_quadrature_against_adaptive_integration():
    """20-pt Gauss-Hermite of log p(y|f) vs scipy adaptive quadrature; gradients vs central differences."""
    from scipy.integrate import quad

    lik = O.Bernoulli()
    m = np.array([[-1.3], [0.2], [2.0]])
    v = np.array([[0.4], [1.5], [0.05]])
    y = np.array([[1.0], [0.0], [1.0]])
    ve = np.array([lik.variational_expectations(m[i:i + 1], v[i:i + 1], y[i:i + 1])[0] for i in range(3)])
    for i in range(3):
        f = lambda x: lik._logp(np.array(x), y[i, 0]) * np.exp(-0.5 * (x - m[i, 0]) ** 2 / v[i, 0]) / np.sqrt(2 * np.pi * v[i, 0])
        ref, _ = quad(f, m[i, 0] - 12 * np.sqrt(v[i, 0]), m[i, 0] + 12 * np.sqrt(v[i, 0]), epsabs=1e-12)
        assert abs(ve[i] - ref) < 1e-4  # 20-pt GH truncation error, not a bug
    g0, g1 = lik.variational_expectations_grads(m, v, y)
    h = 1e-6
    fd0 = (lik.variational_expectations(m + h, v, y) - lik.variational_expectations(m - h, v, y)) / (2 * h)
    fd1 = (lik.variational_expectations(m, v + h, y) - lik.variational_expectations(m, v - h, y)) / (2 * h)
    np.testing.assert_allclose(g0[:, 0], fd0, rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(g1[:, 0], fd1, rtol=1e-6, atol=1e-9)


# ---------------------------------------------------------------------------
# util identities  (reference tests/test_utils.py)
# ---------------------------------------------------------------------------
def _util_setup(num_latent_gps=1, seed=123):
    """reference tests/test_utils.py:26-41."""
    rng = np.random.RandomState(seed)
    NUM_DATA_U, NUM_INDUCING = 3, 2
    input_points = rng.rand(NUM_DATA_U, 1) * 2 - 1
    inducing_points = rng.rand(NUM_INDUCING, 1) * 2 - 1
    kernel = O.SquaredExponential(lengthscales=LENGTH_SCALE, variance=VARIANCE)
    Kuu = kernel.K(inducing_points)
    Kuf = kernel.K(inducing_points, input_points)
    lambda_1 = rng.randn(NUM_DATA_U, num_latent_gps)
    lambda_2 = np.ones((NUM_DATA_U, num_latent_gps))
    return Kuu, Kuf, kernel, lambda_1, lambda_2, input_points, inducing_points


@pytest.mark.parametrize("num_latent_gps", [1, 2])
def test_site_conditionals(num_latent_gps):
    """reference tests/test_utils.py:44-77."""
    Kuu, Kuf, kernel, l1, l2, X, Z = _util_setup(num_latent_gps)
    l, L = O.project_diag_sites(Kuf, l1, l2, Kuu_=None)
    l_white, L_white = O.project_diag_sites(Kuf, l1, l2, Kuu_=Kuu)
    q_mu, cov = O.mean_cov_from_precision_site(Kuu, l, L)
    q_sqrt = O._chol(cov)
    # GPflow conditional with jitter-free Kuu (the reference passes raw arrays; default jitter applies)
    mu, var = O.conditional(X, O.InducingPoints(Z), kernel, q_mu, q_sqrt=q_sqrt, white=False)
    Kff = kernel.K_diag(X)[..., None]
    mu2, var2 = O.conditional_from_precision_sites_white(Kuu, Kff, Kuf, l, L=L)
    mu3, var3 = O.conditional_from_precision_sites(Kuu, Kff, Kuf, l_white, L=L_white)
    np.testing.assert_array_almost_equal(mu, mu2, decimal=3)
    np.testing.assert_array_almost_equal(var, var2, decimal=3)
    np.testing.assert_array_almost_equal(mu, mu3, decimal=3)
    np.testing.assert_array_almost_equal(var, var3, decimal=3)


def test_mean_cov_from_precision_site():
    """reference tests/test_utils.py:80-102 (Woodbury == naive inverse)."""
    Kuu, Kuf, _, l1, l2, _, _ = _util_setup()
    m = Kuu.shape[-1]
    l, L = O.project_diag_sites(Kuf, l1, l2, cholesky=True)
    Luu = O._chol(Kuu)
    iKuu = O._chol_solve(Luu, np.eye(m))
    iKuuLB = O._chol_solve(Luu, L)
    B = iKuuLB @ O._T(iKuuLB)
    Lprec = O._chol(iKuu + B)
    inv2 = O._chol_solve(Lprec, np.eye(m))
    mean2 = O._chol_solve(Lprec[0], iKuu @ l)
    mean1, inv1 = O.mean_cov_from_precision_site(Kuu, l, L)
    np.testing.assert_array_almost_equal(inv1, inv2, decimal=5)
    np.testing.assert_array_almost_equal(mean1, mean2, decimal=5)


def test_kl_from_precision_sites():
    """reference tests/test_utils.py:105-118."""
    Kuu, Kuf, _, l1, l2, _, _ = _util_setup()
    l, L = O.project_diag_sites(Kuf, l1, l2, Kuu_=None)
    q_mu, q_cov = O.mean_cov_from_precision_site(Kuu, l, L)
    q_sqrt = O._chol(q_cov)
    np.testing.assert_almost_equal(O.kl_from_precision_sites_white(Kuu, l, L=L), O.gauss_kl(q_mu, q_sqrt, Kuu))


def test_posteriors_from_dense_sites():
    """reference tests/test_utils.py:124-137."""
    Kuu, Kuf, _, l1, l2, _, _ = _util_setup()
    l, L = O.project_diag_sites(Kuf, l1, l2, Kuu_=Kuu)
    l_white, L_white = O.project_diag_sites(Kuf, l1, l2, Kuu_=None)
    m, v = O.posterior_from_dense_site(Kuu, l, L)
    m_white, v_white = O.posterior_from_dense_site_white(Kuu, l_white, L_white @ O._T(L_white))
    np.testing.assert_array_almost_equal(m, m_white, decimal=3)
    np.testing.assert_array_almost_equal(v, v_white, decimal=3)


def test_gauss_kl_against_closed_form():
    """gauss_kl restatement vs the textbook KL between Gaussians."""
    rng = np.random.RandomState(0)
    M, P = 5, 2
    A = rng.randn(M, M)
    K = A @ A.T + M * np.eye(M)
    q_mu = rng.randn(M, P)
    q_sqrt = np.tril(rng.randn(P, M, M)) + 2 * np.eye(M)
    kl = 0.0
    Kinv = np.linalg.inv(K)
    for p in range(P):
        S = q_sqrt[p] @ q_sqrt[p].T
        kl += 0.5 * (np.trace(Kinv @ S) + q_mu[:, p] @ Kinv @ q_mu[:, p] - M
                     + np.linalg.slogdet(K)[1] - np.linalg.slogdet(S)[1])
    np.testing.assert_allclose(O.gauss_kl(q_mu, q_sqrt, K), kl, rtol=1e-12)


def test_separate_kernels_are_independent_models():
    """[ext] restatement check for (SharedIndependentInducingVariables, SeparateIndependent), the layout of
    docs/notebooks/heteroskedastic.py:62-76: with a likelihood that factorises over outputs, P separate kernels on
    shared inducing points are P independent single-output t-SVGPs (natgrad_step :234-304 is per latent: rank-3 A
    at :271-277, batched K_uu in util.py:367), and with identical kernels they equal the shared-kernel model."""
    rng = np.random.RandomState(3)
    N, M, D, P = 150, 12, 2, 3
    X = rng.randn(N, D)
    Z = X[:M].copy()
    Y = np.sin(X @ rng.randn(D, P)) + 0.3 * rng.randn(N, P)
    ls = [0.8, 1.0, 1.3]

    def sep(lengths):
        return O.t_SVGP(O.SeparateIndependent([O.SquaredExponential(1.0, l) for l in lengths]), O.Gaussian(0.1),
                        O.SharedIndependentInducingVariables(Z), num_latent_gps=P, num_data=N)

    mm = sep(ls)
    for _ in range(3):
        mm.natgrad_step((X, Y), lr=0.8)
    total = 0.0
    for p in range(P):
        m1 = O.t_SVGP(O.SquaredExponential(1.0, ls[p]), O.Gaussian(0.1), Z, num_data=N)
        for _ in range(3):
            m1.natgrad_step((X, Y[:, p:p + 1]), lr=0.8)
        np.testing.assert_allclose(mm.lambda_1[:, p], m1.lambda_1[:, 0], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(mm.lambda_2[p], m1.lambda_2[0], rtol=1e-7, atol=1e-9)
        mu, var = m1.predict_f(X[:10])
        mum, varm = mm.predict_f(X[:10])
        np.testing.assert_allclose(mum[:, p], mu[:, 0], rtol=1e-7, atol=1e-9)
        np.testing.assert_allclose(varm[:, p], var[:, 0], rtol=1e-7, atol=1e-9)
        total += m1.elbo((X, Y[:, p:p + 1]))
    assert abs(total - mm.elbo((X, Y))) < 1e-7 * abs(total)
    shared = O.t_SVGP(O.SquaredExponential(1.0, 1.0), O.Gaussian(0.1), Z, num_latent_gps=P, num_data=N)
    same = sep([1.0] * P)
    for _ in range(3):
        shared.natgrad_step((X, Y), lr=0.8)
        same.natgrad_step((X, Y), lr=0.8)
    np.testing.assert_allclose(same.lambda_1, shared.lambda_1, rtol=1e-7, atol=1e-9)
    assert abs(same.elbo((X, Y)) - shared.elbo((X, Y))) < 1e-8 * abs(shared.elbo((X, Y)))


def test_white_model_pins():
    """t_SVGP_white restated (oracle) against the reference's own relational tests, all computable without GPflow:
    tests/models/test_tsvgp_white.py:64-91 (white == unwhitened t-SVGP: initial ELBO, initial predictions, predictions
    after one lr=0.9 step; decimal=4 there), :94-115 (one lr=1 step with Z = X reaches the exact-GP optimum) and
    tests/models/test_condit.py:69-83 (one lr=1 step == Titsias' collapsed SGPR posterior; closed form restated in
    oracle.sgpr_predict_f)."""
    rng = np.random.RandomState(123)

    def func(x):
        return np.sin(x * 3 * 3.14) + 0.3 * np.cos(x * 9 * 3.14) + 0.5 * np.sin(x * 7 * 3.14)

    X = rng.rand(8, 1) * 2 - 1
    Y = func(X) + 0.2 * rng.randn(8, 1)
    k = O.SquaredExponential(variance=2.25, lengthscales=2.0)
    plain, white = O.t_SVGP(k, O.Gaussian(0.3), X.copy()), O.t_SVGP_white(k, O.Gaussian(0.3), X.copy())
    np.testing.assert_almost_equal(plain.elbo((X, Y)), white.elbo((X, Y)), decimal=4)
    for a, b in zip(plain.predict_f(X), white.predict_f(X)):
        np.testing.assert_array_almost_equal(a, b, decimal=4)
    plain.natgrad_step((X, Y), lr=0.9)
    white.natgrad_step((X, Y), lr=0.9)
    for a, b in zip(plain.predict_f(X), white.predict_f(X)):
        np.testing.assert_array_almost_equal(a, b, decimal=4)
    opt = O.t_SVGP_white(k, O.Gaussian(0.3), X.copy())
    opt.natgrad_step((X, Y * 0), lr=1.0)
    np.testing.assert_almost_equal(opt.elbo((X, Y * 0)), O.gpr_log_marginal_likelihood(k, X, Y * 0, 0.3), decimal=4)
    k1 = O.SquaredExponential(1.0, 1.0)
    for seed in range(4):  # seeds with well separated inducing points (cond K_uu < 100)
        rs = np.random.RandomState(seed)
        Xs, Ys, Zs = rs.randn(10, 1), rs.randn(10, 1), rs.randn(3, 1)
        w = O.t_SVGP_white(k1, O.Gaussian(0.3), Zs)
        w.natgrad_step((Xs, Ys), lr=1.0)
        m1, v1 = w.predict_f(Ys)
        m2, v2 = O.sgpr_predict_f(k1, Xs, Ys, Zs, 0.3, Ys)
        np.testing.assert_array_almost_equal(m1, m2, decimal=6)
        np.testing.assert_array_almost_equal(v1, v2, decimal=6)


def test_white_extra_data_conditioning_pin():
    """Reference tests/models/test_condit.py:84-104 on the oracle: a model stepped (lr=1) on data + extra data predicts
    what a model stepped on the data alone predicts through predict_f_extra_data(extra) (decimal=4 there), with the
    call's jitter=0.0 as in the test and with its default; and an independent closed form: for a Gaussian likelihood
    both equal Titsias' collapsed SGPR posterior on the concatenated data."""
    k1 = O.SquaredExponential(1.0, 1.0)
    for seed in range(4):
        rs = np.random.RandomState(seed)
        X, Y, Z = rs.randn(10, 1), rs.randn(10, 1), rs.randn(3, 1)
        Xe, Ye = rs.randn(10, 1), rs.randn(10, 1)
        Xc, Yc = np.vstack([X, Xe]), np.vstack([Y, Ye])
        both, one = O.t_SVGP_white(k1, O.Gaussian(0.3), Z.copy()), O.t_SVGP_white(k1, O.Gaussian(0.3), Z.copy())
        both.natgrad_step((Xc, Yc), lr=1.0)
        one.natgrad_step((X, Y), lr=1.0)
        l1, L2 = one.lambda_1.copy(), one.lambda_2.copy()
        m, v = both.predict_f(X)
        for jit in (0.0, None):
            kw = {} if jit is None else dict(jitter=jit)
            m_, v_ = one.predict_f_extra_data(X, extra_data=(Xe, Ye), **kw)
            np.testing.assert_array_almost_equal(m, m_, decimal=4)
            np.testing.assert_array_almost_equal(v, v_, decimal=4)
        assert np.array_equal(one.lambda_1, l1) and np.array_equal(one.lambda_2, L2)  # the state is not touched
        m2, v2 = O.sgpr_predict_f(k1, Xc, Yc, Z, 0.3, X)
        np.testing.assert_array_almost_equal(m_, m2, decimal=4)
        np.testing.assert_array_almost_equal(v_, v2, decimal=4)


def test_row_blocked_evaluators_equal_the_single_call_forms():
    """oracle.elbo_chunked / predict_f_chunked (what bench.py's `elbo_match` evaluates at N = 1e6) give the numbers of
    base_SVGP.elbo / predict_f (src/models/tsvgp.py:79-114) for any block size, Gaussian and Bernoulli, P = 2."""
    rng = np.random.RandomState(5)
    N, M, D, P = 900, 24, 3, 2
    X = rng.randn(N, D)
    F = np.sin(X @ rng.randn(D, P))
    Z = X[:M].copy()
    for lik, Y in ((O.Gaussian(0.1), F + 0.3 * rng.randn(N, P)), (O.Bernoulli(), (F + 0.3 * rng.randn(N, P) > 0).astype(float))):
        m = O.t_SVGP(O.SquaredExponential(1.2, 0.9), lik, Z, num_latent_gps=P, num_data=4 * N)
        for _ in range(3):
            m.natgrad_step((X, Y), lr=0.7)
        e = m.elbo((X, Y))
        for chunk in (N, 250, 77):
            assert abs(O.elbo_chunked(m, (X, Y), chunk_rows=chunk) - e) < 1e-12 * abs(e)
        mu, var = m.predict_f(X[:300])
        mu_c, var_c = O.predict_f_chunked(m, X[:300], chunk_rows=64)
        assert np.max(np.abs(mu - mu_c)) < 1e-13 and np.max(np.abs(var - var_c)) < 1e-13


@pytest.mark.parametrize("lik,P,separate", [("gaussian", 1, False), ("bernoulli", 2, False), ("gaussian", 3, True)])
def test_row_blocked_step_is_the_step(lik, P, separate):
    """``natgrad_step_chunked`` -- the reference's E-step (src/models/tsvgp.py:234-304) with its N-sized ops taken per row block
    and the block sums of G0 / G1 compensated: what is run at N = 1e6 (full-size state match, measured CPU baseline) -- is
    ``natgrad_step``: same state, same intermediates, to rounding, over three steps, with a block size that does not divide N;
    the ELBO it reports for the state a step starts from is ``elbo`` of that state."""
    from tests.helpers import synthetic

    X, Y, Z = synthetic(N=1203, M=24, D=3, P=P, lik=lik, seed=1)  # cond(K_uu) ~ 1e3: block order shows at ~1e-13 only

    def mk():
        k = O.SeparateIndependent([O.SquaredExponential(1.0, l) for l in (0.9, 1.1, 1.3)]) if separate else O.SquaredExponential(1.0, 1.0)
        iv = O.SharedIndependentInducingVariables(Z) if separate else Z
        return O.t_SVGP(k, O.Gaussian(0.1) if lik == "gaussian" else O.Bernoulli(), iv, num_latent_gps=P, num_data=2000)

    a, b = mk(), mk()
    rel = lambda x, y: np.max(np.abs(x - y)) / max(np.max(np.abs(y)), 1e-300)  # (the first step's mean is exactly zero)
    for _ in range(3):
        # both from the SAME state (bit for bit), so that what is compared is the row blocking alone: near the fixed point
        # G0 is a small difference of large terms and would amplify a last-digit difference of the states to ~1e-10
        b.sites.lambda_1, b.sites._lambda_2_sqrt = a.sites.lambda_1.copy(), a.sites._lambda_2_sqrt.copy()
        e = a.elbo((X, Y))
        a.natgrad_step((X, Y), lr=0.8)
        O.natgrad_step_chunked(b, (X, Y), lr=0.8, chunk_rows=100)
        assert abs(b.last["elbo_before"] - e) < 1e-12 * abs(e)
        for k in ("mean", "var", "g0", "g1", "G0", "G1", "meanZ"):
            assert rel(b.last[k], a.last[k]) < 1e-11, k
        assert rel(b.lambda_1, a.lambda_1) < 1e-11 and rel(b.lambda_2, a.lambda_2) < 1e-11


def test_compensated_sum_recovers_what_plain_addition_loses():
    s = O._CompensatedSum()
    parts = [np.array([1.0, 1e16]), np.array([1e-3, 1.0]), np.array([-1.0, -1e16]), np.array([1e-3, 1.0])]
    for x in parts:
        s.add(x)
    assert np.allclose(s.value(), [2e-3, 2.0], rtol=1e-15, atol=0)


def test_site_sum_contractions_equal_the_einsums_they_lower():
    """src/models/tsvgp.py:279-280: the oracle evaluates the two tf.einsum contractions as per-latent BLAS products (what
    TensorFlow lowers them to); np.einsum's own loop for the same subscripts is the definition they are held to."""
    rng = np.random.RandomState(11)
    for n, m, P in ((37, 9, 1), (64, 12, 3)):
        A = rng.randn(n, m, P)
        g0, g1 = rng.randn(n, P), -rng.rand(n, P)
        np.testing.assert_allclose(O._einsum_nml_nl(A, g0), np.einsum("nml,nl->ml", A, g0), rtol=1e-13, atol=1e-13)
        want = np.einsum("nml,nol,nl->lmo", A, A, g1)
        got = O._einsum_nml_nol_nl(A, g1)
        assert got.shape == (P, m, m)
        np.testing.assert_allclose(got, want, rtol=1e-13, atol=1e-13)
