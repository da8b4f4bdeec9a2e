"""M-step gradient: d ELBO / d (kernel variance, lengthscales, Z, noise variance) with the sites fixed,
HIP path against CENTRAL FINITE DIFFERENCES OF THE ORACLE'S ELBO (an independent derivation: the oracle has no gradient
code).  The reference's own pin (tests/models/test_tsvgp.py:168-188) compares against GPflow's SVGP and cannot be
reproduced here, so absolute parity of the gradient is pinned by these differences only.
Tolerance: the difference quotient (h = 1e-5 relative, fp64) is good to ~1e-7 relative; asserted at 2e-6."""
import numpy as np
import pytest
import torch

from oracle import tsvgp_oracle as O
from tests.helpers import pkg, relerr, synthetic

pytestmark = pytest.mark.gpu


def _models(kname, lik, P, Z, ls, var, noise, N):
    p = pkg()
    mk = lambda mod, **kw: mod.t_SVGP(getattr(mod, kname)(var, ls), mod.Gaussian(noise) if lik == "gaussian" else mod.Bernoulli(),
                                      Z, num_latent_gps=P, num_data=N, **kw)
    return mk(p), mk(O)


def _oracle_elbo(kname, lik, P, Z, ls, var, noise, N, state, data):
    m = O.t_SVGP(getattr(O, kname)(var, ls), O.Gaussian(noise) if lik == "gaussian" else O.Bernoulli(), Z,
                 num_latent_gps=P, num_data=N, lambda_1=state[0], lambda_2_sqrt=state[1])
    return m.elbo(data)


@pytest.mark.parametrize("kname,lik,P", [("SquaredExponential", "gaussian", 1), ("SquaredExponential", "bernoulli", 2),
                                         ("Matern52", "gaussian", 2), ("Matern32", "bernoulli", 1)])
def test_elbo_gradients_match_oracle_finite_differences(kname, lik, P):
    rng = np.random.RandomState(41)
    N, M, D = 500, 24, 3
    X, Y, _ = synthetic(N=N, M=M, D=D, P=P, lik=lik, seed=7)
    Z = rng.randn(M, D) * 1.2
    ls, var, noise = np.array([0.9, 1.2, 1.5]), 1.3, 0.2
    hip, ora = _models(kname, lik, P, Z, ls, var, noise, N)
    for _ in range(3):
        hip.natgrad_step((X, Y), lr=0.7)
        ora.natgrad_step((X, Y), lr=0.7)
    state = (ora.lambda_1.copy(), ora.lambda_2_sqrt.copy())
    elbo, grads = hip.elbo_and_grads((X, Y))
    f = lambda **kw: _oracle_elbo(kname, lik, P, kw.get("Z", Z), kw.get("ls", ls), kw.get("var", var), kw.get("noise", noise), N,
                                  state, (X, Y))
    assert abs(float(elbo) - f()) < 1e-9 * abs(f())

    def fd(key, base, idx=None, h=1e-5):
        step = h * max(1.0, abs(float(base if idx is None else base[idx])))
        up, dn = np.array(base, dtype=np.float64, copy=True), np.array(base, dtype=np.float64, copy=True)
        if idx is None:
            up, dn = up + step, dn - step
        else:
            up[idx] += step
            dn[idx] -= step
        return (f(**{key: up if idx is not None else float(up)}) - f(**{key: dn if idx is not None else float(dn)})) / (2 * step)

    scale = max(abs(fd("var", var)), 1.0)
    assert abs(float(grads["variance"]) - fd("var", var)) < 2e-6 * scale
    g_ls = grads["lengthscales"].cpu().numpy()
    for d in range(D):
        ref = fd("ls", ls, d)
        assert abs(g_ls[d] - ref) < 2e-6 * max(abs(ref), scale), (d, g_ls[d], ref)
    g_Z = grads["Z"].cpu().numpy()
    for (m_, d) in [(0, 0), (5, 1), (M - 1, 2), (11, 0)]:
        ref = fd("Z", Z, (m_, d))
        assert abs(g_Z[m_, d] - ref) < 2e-6 * max(abs(ref), scale), (m_, d, g_Z[m_, d], ref)
    if lik == "gaussian":
        ref = fd("noise", noise)
        assert abs(float(grads["likelihood_variance"]) - ref) < 2e-6 * max(abs(ref), scale)
    else:
        assert "likelihood_variance" not in grads


def test_elbo_gradients_isotropic_lengthscale_and_larger_problem():
    """Scalar lengthscale (the reference's default kernels) at a size where every kernel runs several workgroups."""
    rng = np.random.RandomState(42)
    N, M, D = 3000, 200, 8
    X, Y, _ = synthetic(N=N, M=M, D=D, P=1, lik="gaussian", seed=8)
    Z = rng.randn(M, D) * 1.2
    hip, ora = _models("SquaredExponential", "gaussian", 1, Z, 1.1, 0.9, 0.15, N)
    for _ in range(2):
        hip.natgrad_step((X, Y), lr=0.8)
        ora.natgrad_step((X, Y), lr=0.8)
    state = (ora.lambda_1.copy(), ora.lambda_2_sqrt.copy())
    elbo, grads = hip.elbo_and_grads((X, Y))
    f = lambda ls=1.1, var=0.9: _oracle_elbo("SquaredExponential", "gaussian", 1, Z, ls, var, 0.15, N, state, (X, Y))
    assert abs(float(elbo) - f()) < 1e-9 * abs(f())
    h = 1e-5
    ref_ls = (f(ls=1.1 + h) - f(ls=1.1 - h)) / (2 * h)
    ref_var = (f(var=0.9 + h) - f(var=0.9 - h)) / (2 * h)
    assert grads["lengthscales"].dim() == 0
    assert abs(float(grads["lengthscales"]) - ref_ls) < 2e-6 * abs(ref_ls)
    assert abs(float(grads["variance"]) - ref_var) < 2e-6 * abs(ref_var)


def test_em_loop_matches_oracle_driven_loop():
    """The E/M loop of the reference's driver (experiments/uci_regression.py:132-160) on the C1-style 1-D problem:
    the HIP loop (analytic gradients) against the same loop driven by the oracle with finite-difference gradients and the
    same Adam.  Hyperparameters after two iterations agree to 1e-5; the ELBO log is increasing."""
    p = pkg()
    T = p.training
    rng = np.random.RandomState(0)
    N, M = 300, 12
    X = rng.rand(N, 1) * 2 - 1
    Y = np.sin(15 * X) + 0.5 * rng.randn(N, 1)
    Z0 = np.linspace(X.min(), X.max(), M)[:, None]
    hip = p.t_SVGP(p.SquaredExponential(0.3, 0.1), p.Gaussian(1.0), Z0.copy(), num_data=N)
    logf, _ = T.em_fit(hip, (X, Y), iterations=2, n_e_steps=3, n_m_steps=4, nat_lr=0.8, adam_lr=0.01)

    # oracle-driven replica: same schedule, gradients by central differences of the oracle's ELBO
    theta = dict(variance=np.array(0.3), lengthscales=np.array(0.1), likelihood_variance=np.array(1.0), Z=Z0.copy())
    state = [None, None]

    def ora_model():
        return O.t_SVGP(O.SquaredExponential(float(theta["variance"]), float(theta["lengthscales"])),
                        O.Gaussian(float(theta["likelihood_variance"])), theta["Z"], num_data=N,
                        lambda_1=state[0], lambda_2_sqrt=state[1])

    opt = T.Adam(0.01)
    sp_inv = lambda x: x + np.log(-np.expm1(-x))
    for _ in range(2):
        m = ora_model()
        for _ in range(3):
            m.natgrad_step((X, Y), lr=0.8)
        state = [m.lambda_1.copy(), m.lambda_2_sqrt.copy()]
        for _ in range(4):
            grads = {}
            for name in theta:
                base = np.array(theta[name], dtype=np.float64)
                g = np.zeros_like(base)
                for idx in np.ndindex(*base.shape) if base.ndim else [()]:
                    h = 1e-6 * max(1.0, abs(float(base[idx])))
                    up, dn = np.array(base), np.array(base)
                    up[idx] += h
                    dn[idx] -= h
                    theta[name] = up
                    e_up = ora_model().elbo((X, Y))
                    theta[name] = dn
                    e_dn = ora_model().elbo((X, Y))
                    g[idx] = (e_up - e_dn) / (2 * h)
                theta[name] = base
                grads[name] = g
            u, gu = {}, {}
            low = {"likelihood_variance": 1e-6}  # gpflow.likelihoods.Gaussian: positive(lower=1e-6) [ext]; others lower 0
            for name in theta:
                if name == "Z":
                    u[name], gu[name] = torch.as_tensor(theta[name]), torch.as_tensor(-grads[name])
                else:
                    un = sp_inv(theta[name] - low.get(name, 0.0))
                    u[name] = torch.as_tensor(un)
                    gu[name] = torch.as_tensor(-grads[name] / (1.0 + np.exp(-un)))
            opt.step(u, gu)
            for name in theta:
                theta[name] = u[name].numpy() if name == "Z" else np.asarray(low.get(name, 0.0) + np.log1p(np.exp(u[name].numpy())))
    assert abs(float(hip.kernel.variance.value) - float(theta["variance"])) < 1e-5
    assert abs(float(hip.kernel.lengthscales.value) - float(theta["lengthscales"])) < 1e-5
    assert abs(float(hip.likelihood.variance.value) - float(theta["likelihood_variance"])) < 1e-5
    assert np.max(np.abs(hip.inducing_variable.Z.numpy() - theta["Z"])) < 1e-5
    assert logf[1] > logf[0]


@pytest.mark.parametrize("lik", ["gaussian", "bernoulli"])
def test_elbo_gradients_separate_kernels(lik):
    """One kernel per latent on shared inducing points (SeparateIndependent, docs/notebooks/heteroskedastic.py:62-76): the
    gradient entries "kernels.<p>.variance" / "kernels.<p>.lengthscales", Z and the noise against central differences of
    the oracle's ELBO; then one Adam M-step moves every kernel's parameters."""
    p = pkg()
    rng = np.random.RandomState(43)
    N, M, D, P = 400, 20, 2, 3
    X, Y, _ = synthetic(N=N, M=M, D=D, P=P, lik=lik, seed=8)
    Z = rng.randn(M, D) * 1.2
    names = ["SquaredExponential", "Matern52", "SquaredExponential"]
    var0, ls0, noise = [1.3, 0.8, 1.1], [np.array([0.9, 1.4]), np.array([1.2, 1.2]), np.array([1.6, 0.7])], 0.2

    def mk(mod, var=var0, ls=ls0, Zv=Z, nz=noise, **kw):
        kern = mod.SeparateIndependent([getattr(mod, n)(v, l) for n, v, l in zip(names, var, ls)])
        return mod.t_SVGP(kern, mod.Gaussian(nz) if lik == "gaussian" else mod.Bernoulli(),
                          mod.SharedIndependentInducingVariables(Zv), num_latent_gps=P, num_data=N, **kw)

    hip, ora = mk(p), mk(O)
    for _ in range(3):
        hip.natgrad_step((X, Y), lr=0.7)
        ora.natgrad_step((X, Y), lr=0.7)
    state = dict(lambda_1=ora.lambda_1.copy(), lambda_2_sqrt=ora.lambda_2_sqrt.copy())
    f = lambda **kw: mk(O, **kw, **state).elbo((X, Y))
    elbo, grads = hip.elbo_and_grads((X, Y))
    assert abs(float(elbo) - f()) < 1e-9 * abs(f())
    h = 1e-5
    scale = 1.0
    for k in range(P):
        up, dn = list(var0), list(var0)
        up[k], dn[k] = var0[k] + h, var0[k] - h
        ref = (f(var=up) - f(var=dn)) / (2 * h)
        scale = max(scale, abs(ref))
        assert abs(float(grads[f"kernels.{k}.variance"]) - ref) < 2e-6 * max(abs(ref), scale), (k, ref)
        g_ls = grads[f"kernels.{k}.lengthscales"].cpu().numpy()
        for d in range(D):
            up, dn = [l.copy() for l in ls0], [l.copy() for l in ls0]
            up[k][d] += h
            dn[k][d] -= h
            ref = (f(ls=up) - f(ls=dn)) / (2 * h)
            assert abs(g_ls[d] - ref) < 2e-6 * max(abs(ref), scale), (k, d, g_ls[d], ref)
    g_Z = grads["Z"].cpu().numpy()
    for (m_, d) in [(0, 0), (7, 1), (M - 1, 0)]:
        up, dn = Z.copy(), Z.copy()
        up[m_, d] += h
        dn[m_, d] -= h
        ref = (f(Zv=up) - f(Zv=dn)) / (2 * h)
        assert abs(g_Z[m_, d] - ref) < 2e-6 * max(abs(ref), scale), (m_, d, g_Z[m_, d], ref)
    if lik == "gaussian":
        ref = (f(nz=noise + h) - f(nz=noise - h)) / (2 * h)
        assert abs(float(grads["likelihood_variance"]) - ref) < 2e-6 * max(abs(ref), scale)
    # the M-step driver sees every kernel's parameters
    T = p.training
    before = float(hip.elbo((X, Y)))
    T.m_step(hip, (X, Y), T.Adam(0.01), steps=3)
    assert set(T.trainable_parameters(hip)) >= {f"kernels.{k}.{n}" for k in range(P) for n in ("variance", "lengthscales")}
    for k in range(P):
        assert abs(float(hip.kernel.kernels[k].variance.value) - var0[k]) > 1e-3
    assert float(hip.elbo((X, Y))) > before


@pytest.mark.parametrize("kname,lik", [("SquaredExponential", "gaussian"), ("Matern52", "bernoulli")])
def test_elbo_gradients_large_input_dimension(kname, lik):
    """D = 784 (the reference's MNIST loop, docs/notebooks/mnist.py:117-192): beyond the fused gradient kernel's sizes the
    contraction runs in GEMM form (``tsvgp_gram_to_gradw_*``); d ELBO / d (variance, lengthscale, Z, noise) against central
    differences of the ORACLE's ELBO, same tolerance as at small D."""
    rng = np.random.RandomState(43)
    N, M, D = 400, 40, 784
    X = rng.randn(N, D) * 0.3
    w = rng.randn(D, 1) / np.sqrt(D)
    f_lat = np.sin(3.0 * X @ w)
    Y = f_lat + 0.3 * rng.randn(N, 1) if lik == "gaussian" else (f_lat + 0.3 * rng.randn(N, 1) > 0).astype(np.float64)
    Z = X[:M] + 0.05 * rng.randn(M, D)
    ls, var, noise = 6.0, 1.2, 0.2  # |x - z| ~ 0.3 sqrt(2 D) ~ 12: a lengthscale at which the kernel matrix is informative
    hip, ora = _models(kname, lik, 1, Z, ls, var, noise, N)
    for _ in range(2):
        hip.natgrad_step((X, Y), lr=0.7)
        ora.natgrad_step((X, Y), lr=0.7)
    state = (ora.lambda_1.copy(), ora.lambda_2_sqrt.copy())
    elbo, grads = hip.elbo_and_grads((X, Y))
    f = lambda **kw: _oracle_elbo(kname, lik, 1, kw.get("Z", Z), kw.get("ls", ls), kw.get("var", var), kw.get("noise", noise), N,
                                  state, (X, Y))
    assert abs(float(elbo) - f()) < 1e-9 * abs(f())
    h = 1e-5
    ref_var = (f(var=var + h) - f(var=var - h)) / (2 * h)
    ref_ls = (f(ls=ls + h * ls) - f(ls=ls - h * ls)) / (2 * h * ls)
    scale = max(abs(ref_var), 1.0)
    assert abs(float(grads["variance"]) - ref_var) < 2e-6 * scale
    assert grads["lengthscales"].dim() == 0
    assert abs(float(grads["lengthscales"]) - ref_ls) < 2e-6 * max(abs(ref_ls), scale)
    g_Z = grads["Z"].cpu().numpy()
    assert g_Z.shape == (M, D)
    for (m_, d) in [(0, 0), (7, 391), (M - 1, D - 1), (19, 100)]:
        Zu, Zd = Z.copy(), Z.copy()
        Zu[m_, d] += h
        Zd[m_, d] -= h
        ref = (f(Z=Zu) - f(Z=Zd)) / (2 * h)
        assert abs(g_Z[m_, d] - ref) < 2e-6 * max(abs(ref), scale), (m_, d, g_Z[m_, d], ref)
    if lik == "gaussian":
        ref = (f(noise=noise + h) - f(noise=noise - h)) / (2 * h)
        assert abs(float(grads["likelihood_variance"]) - ref) < 2e-6 * max(abs(ref), scale)


def test_gradient_pass_on_the_stored_tile_equals_the_fused_pass(monkeypatch):
    """Round 5: with one latent ``elbo_and_grads`` keeps the moments' triangular product (``EStepEngine.run(keep_tile=True)``:
    tsvgp_trmm, mean by a matrix-vector product, var by a row norm, tsvgp_lik_map) and takes Q k_n as a second triangular
    product of it; TSVGP_MSTEP_TILE=0 is round 3's pass (fused moments kernel + dense GEMM with Q).  Same ELBO and gradients,
    Gaussian and Bernoulli, fp64 to 1e-10 (reference experiments/uci_regression.py:159-160: what the M-step differentiates)."""
    p = pkg()
    for lik in ("gaussian", "bernoulli"):
        X, Y, Z = synthetic(N=1500, M=96, D=3, lik=lik, seed=4)
        Xd, Yd = torch.as_tensor(X, device="cuda:0"), torch.as_tensor(Y, device="cuda:0")
        out = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("TSVGP_MSTEP_TILE", mode)
            m = p.t_SVGP(p.SquaredExponential(1.3, 0.9), p.Gaussian(0.2) if lik == "gaussian" else p.Bernoulli(), Z, num_data=1500)
            for _ in range(3):
                m.natgrad_step((Xd, Yd), lr=0.7)
            e, g = m.elbo_and_grads((Xd, Yd))
            out[mode] = (float(e), {k: v.cpu().numpy() for k, v in g.items()})
        assert abs(out["1"][0] - out["0"][0]) < 1e-10 * abs(out["0"][0])
        for k in out["0"][1]:
            # (1e-8: the two passes round g0, g1 differently at the 1e-13 level, and the M x M part -- Q A2 Q against Q K Q -- carries
            #  that through a cancellation of ~1e4: 3.4e-9 of the largest entry on dZ, Bernoulli, measured)
            assert relerr(out["1"][1][k], out["0"][1][k]) < 1e-8, (lik, k)


@pytest.mark.parametrize("lik,P,sep", [("gaussian", 1, False), ("bernoulli", 2, False), ("gaussian", 2, True)])
def test_closed_form_kuu_gradient_equals_autograd_through_the_factorisation(monkeypatch, lik, P, sep):
    """The M x M part of elbo_and_grads: the closed form d surrogate / d K_uu (dQ = -Q dK Q, d beta = -Q dK beta, d log|W| =
    tr(Q dK); six GEMMs on the prelude's Q and beta, autograd only through K(Z, Z; theta)) against round 3's form, which
    differentiates through the factorisation of W and the triangular solve (TSVGP_MSTEP_AUTOGRAD=1): the same ELBO and the same
    gradients, for a shared kernel behind one and two latents and for one kernel per latent."""
    p = pkg()
    X, Y, Z = synthetic(N=1500, M=48, D=3, P=P, lik=lik, seed=21)
    Xd, Yd = torch.as_tensor(X, device="cuda:0"), torch.as_tensor(Y, device="cuda:0")
    likelihood = p.Gaussian(0.2) if lik == "gaussian" else p.Bernoulli()
    if sep:
        kern = p.SeparateIndependent([p.SquaredExponential(1.3, [0.9, 1.1, 1.4]), p.SquaredExponential(0.8, [1.2, 0.7, 1.0])])
    else:
        kern = p.SquaredExponential(1.3, [0.9, 1.1, 1.4])
    m = p.t_SVGP(kern, likelihood, Z, num_latent_gps=P, num_data=4000)
    for _ in range(3):
        m.natgrad_step((Xd, Yd), lr=0.5)
    monkeypatch.setenv("TSVGP_MSTEP_AUTOGRAD", "1")
    e_a, g_a = m.elbo_and_grads((Xd, Yd))
    monkeypatch.setenv("TSVGP_MSTEP_AUTOGRAD", "0")
    e_c, g_c = m.elbo_and_grads((Xd, Yd))
    assert abs(float(e_a) - float(e_c)) <= 1e-11 * abs(float(e_a))
    assert sorted(g_a) == sorted(g_c)
    for k in g_a:
        # 2e-7 of the largest entry: both forms subtract scale Q A2 Q and Q K Q / 2, terms some 1e6 times the gradient that is left
        # (1.2e-8 measured on dZ, Gaussian; the stored-tile / fused comparison above sees the same amplification)
        err = relerr(g_c[k].detach().cpu().numpy(), g_a[k].detach().cpu().numpy())
        assert err < 2e-7, (k, err)
