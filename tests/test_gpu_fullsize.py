"""Full-size checks on the GPU (BASELINE.json configs C2 / C3 sizes, where the oracle is too slow): size-independent
properties of the path.

* shard additivity   -- the per-shard accumulators (acc2, acc1, sum ve) of two half ranges add up to those of the whole
                        range: the exact property the N-sharded multi-GPU path relies on (fp64: 1e-11 relative).
* conjugate fixed point -- Gaussian likelihood, lr = 1: one step reaches the optimum, a second step leaves the site
                        parameters and the ELBO unchanged (reference tests/models/test_tsvgp.py:134-145 at N = 1e6).
* predictions against the oracle -- predict_f / new_predict_f of the N = 1e6 model on a row sample against the oracle's
                        ``conditional`` (tsvgp.py:97-114) and ``conditional_from_precision_sites`` (util.py:91-185)
                        evaluated on the HIP model's state.  (predict_f and new_predict_f run the same kernel call here,
                        so comparing them with each other says nothing.)
The headline shape (M = 1024, fp64) against the oracle: tests/test_gpu_benchshape.py.
"""
import numpy as np
import pytest
import torch

from tests.helpers import pkg, relerr, synthetic

pytestmark = pytest.mark.gpu


def _stats(model, X, Y, ops):
    B = pkg()._backend
    return model._get_engine().run(X, Y, ops["Z"], model.kernel, moment_Tm=ops["moment_Tm"], moment_mode=ops["moment_mode"],
                                   gamma=ops["gamma"], lik_id=model.likelihood.lik_id, lik_param=model.likelihood.lik_param,
                                   whiten_T=ops["whiten_T"], whiten_mode=ops["whiten_mode"], sites=True)


@pytest.mark.parametrize("cfg", ["c2_fp64", "c3_fp32", "ns_fp64"])
def test_shard_additivity_full_size(cfg):
    p = pkg()
    if cfg == "c2_fp64":
        N, M, D, lik, dt, tol = 1_000_000, 512, 8, "gaussian", torch.float64, 1e-11
    elif cfg == "ns_fp64":  # the metric's shape: 8 x 8 tiles, 7813 row panels
        N, M, D, lik, dt, tol = 1_000_000, 1024, 8, "gaussian", torch.float64, 1e-11
    else:
        N, M, D, lik, dt, tol = 1_000_000, 1024, 16, "bernoulli", torch.float32, 2e-4
    X, Y, Z = synthetic(N=N, M=M, D=D, lik=lik, seed=0)
    model = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Gaussian(0.1) if lik == "gaussian" else p.Bernoulli(), Z,
                     compute_dtype=dt)
    Xd = torch.as_tensor(X, dtype=dt, device="cuda:0")
    Yd = torch.as_tensor(Y, dtype=dt, device="cuda:0")
    model.natgrad_step((Xd, Yd), lr=0.8)  # a non-trivial state
    ops = model._site_operands(whiten_jitter=1e-9)
    whole = _stats(model, Xd, Yd, ops)
    w2, w1, wv = whole.acc2.clone(), whole.acc1.clone(), whole.ve_sum.clone()
    h = N // 2 + 37  # uneven cut, not a multiple of the 128-row tile
    a = _stats(model, Xd[:h], Yd[:h], ops)
    a2, a1, av = a.acc2.clone(), a.acc1.clone(), a.ve_sum.clone()
    b = _stats(model, Xd[h:], Yd[h:], ops)
    assert relerr((a2 + b.acc2).cpu().numpy(), w2.cpu().numpy()) < tol
    assert relerr((a1 + b.acc1).cpu().numpy(), w1.cpu().numpy()) < tol
    assert abs(float(av + b.ve_sum) - float(wv)) < max(tol, 1e-12) * abs(float(wv))
    assert float(whole.nonpos) == 0


def test_conjugate_fixed_point_full_size():
    p = pkg()
    N, M, D = 1_000_000, 512, 8
    X, Y, Z = synthetic(N=N, M=M, D=D, lik="gaussian", seed=1)
    model = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Gaussian(0.1), Z)
    Xd, Yd = torch.as_tensor(X, device="cuda:0"), torch.as_tensor(Y, device="cuda:0")
    model.natgrad_step((Xd, Yd), lr=1.0)
    l1, L2, e1 = model.lambda_1.numpy(), model.lambda_2.cpu().numpy(), float(model.elbo((Xd, Yd)))
    model.natgrad_step((Xd, Yd), lr=1.0)
    # the two jitters of the reference (1e-6 in the predictive, 1e-9 in the projection) make the conjugate step
    # contract by ~1e-6 per step instead of landing exactly: the second step moves lambda by O(1e-6) relative
    d1, d2 = relerr(model.lambda_1.numpy(), l1), relerr(model.lambda_2.cpu().numpy(), L2)
    de = abs(float(model.elbo((Xd, Yd))) - e1) / abs(e1)
    print(f"fixed point: d lambda_1 {d1:.2e}, d Lambda_2 {d2:.2e}, d ELBO {de:.2e}")
    assert d1 < 1e-4 and d2 < 1e-8 and de < 1e-8
    # predictions on a row sample against the oracle's two predictive forms, evaluated on this model's state
    from oracle import tsvgp_oracle as O

    ora = O.t_SVGP(O.SquaredExponential(1.0, 1.0), O.Gaussian(0.1), Z)
    ora.sites.lambda_1 = model.lambda_1.numpy()
    ora.sites._lambda_2_sqrt = np.tril(model.lambda_2_sqrt.numpy())
    Xs = X[::997][:1500] + 0.01
    mu_a, var_a = model.predict_f(Xs)
    mu_b, var_b = model.new_predict_f(Xs)
    mu_o, var_o = ora.predict_f(Xs)  # GPflow conditional on (m, chol S)
    mu_s, var_s = ora.new_predict_f(Xs)  # conditional_from_precision_sites
    assert relerr(mu_a.cpu().numpy(), mu_o) < 1e-8 and relerr(var_a.cpu().numpy(), var_o) < 1e-8
    assert relerr(mu_b.cpu().numpy(), mu_s) < 1e-8 and relerr(var_b.cpu().numpy(), var_s) < 1e-8


def test_mstep_gradient_full_size():
    """d ELBO / d (lengthscale, variance, noise) at N = 1e6 (977 row blocks of the gradient kernel, every partial buffer
    in play) against central differences of the HIP ELBO itself (the oracle cannot run this size)."""
    p = pkg()
    N, M, D = 1_000_000, 512, 8
    X, Y, Z = synthetic(N=N, M=M, D=D, lik="gaussian", seed=2)
    Xd, Yd = torch.as_tensor(X, device="cuda:0"), torch.as_tensor(Y, device="cuda:0")
    model = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Gaussian(0.1), Z, num_data=N)
    for _ in range(2):
        model.natgrad_step((Xd, Yd), lr=0.8)
    elbo, grads = model.elbo_and_grads((Xd, Yd))
    assert abs(float(elbo) - float(model.elbo((Xd, Yd)))) < 1e-11 * abs(float(elbo))
    for name, par in (("lengthscales", model.kernel.lengthscales), ("variance", model.kernel.variance),
                      ("likelihood_variance", model.likelihood.variance)):
        base, h = float(par.value), 1e-5
        par.assign(base + h)
        up = float(model.elbo((Xd, Yd)))
        par.assign(base - h)
        dn = float(model.elbo((Xd, Yd)))
        par.assign(base)
        ref = (up - dn) / (2 * h)
        assert abs(float(grads[name]) - ref) < 1e-5 * abs(ref), (name, float(grads[name]), ref)
