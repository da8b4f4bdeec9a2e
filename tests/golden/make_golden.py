#!/usr/bin/env python3
"""Generates the golden fixtures tests/golden/*.npz from the CPU oracle (oracle/tsvgp_oracle.py).

The reference itself cannot run here (TensorFlow / GPflow are not installed and nothing of it travels to the GPU
box), and its tests hold no golden vectors, so these fixtures are produced by the oracle -- which is pinned by the
reference's relational tests (tests/test_oracle_pins.py).  A fixture is DATA: inputs (X, Y, Z, hyperparameters) and
expected outputs (site parameters, predictive moments, likelihood gradients, site gradients, ELBO) after E-steps
1, 2 and 10.  Seeds follow reference tests/models/test_tsvgp.py:16 (RandomState(123)).

    python tests/golden/make_golden.py        # rewrites the .npz files next to this script
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import tsvgp_oracle as O  # noqa: E402

CASES = {
    # name: (N, M, D, P, likelihood, lengthscale, variance, noise, lr)
    "c1_gaussian_1d": (200, 16, 1, 1, "gaussian", 0.1, 0.3, 1.0, 0.8),
    "gaussian_d3_p2": (160, 12, 3, 2, "gaussian", 1.0, 1.0, 0.1, 0.8),
    "bernoulli_d2_p1": (160, 12, 2, 1, "bernoulli", 1.0, 1.5, None, 0.8),
    "bernoulli_d2_p2": (120, 10, 2, 2, "bernoulli", 0.8, 1.0, None, 1.0),
    # separate per-latent kernels on shared inducing points (SeparateIndependent + SharedIndependentInducingVariables,
    # reference docs/notebooks/heteroskedastic.py:62-74): the lengthscale entry is one value per latent
    "gaussian_d3_p3_separate": (160, 12, 3, 3, "gaussian", (0.8, 1.1, 1.5), 1.0, 0.1, 0.8),
}
STEPS = (1, 2, 10)


def make_case(name):
    N, M, D, P, lik, ls, var, noise, lr = CASES[name]
    rng = np.random.RandomState(123)
    if name.startswith("c1"):
        X = rng.rand(N, 1) * 2 - 1
        Y = np.sin(15 * X) + rng.randn(N, 1)
        Z = np.linspace(X.min(), X.max(), M)[:, None]
    else:
        X = rng.randn(N, D)
        f = np.sin(X @ rng.randn(D, P))
        eps = rng.randn(N, P)
        Y = f + np.sqrt(0.1) * eps if lik == "gaussian" else (f + np.sqrt(0.1) * eps > 0).astype(np.float64)
        Z = X[:M].copy()
    separate = isinstance(ls, tuple)
    if separate:
        kernel = O.SeparateIndependent([O.SquaredExponential(variance=var, lengthscales=l) for l in ls])
        iv = O.SharedIndependentInducingVariables(Z)
    else:
        kernel, iv = O.SquaredExponential(variance=var, lengthscales=ls), Z
    model = O.t_SVGP(kernel, O.Gaussian(variance=noise) if lik == "gaussian" else O.Bernoulli(), iv, num_latent_gps=P)
    out = dict(X=X, Y=Y, Z=Z, lengthscales=np.asarray(ls, dtype=np.float64), variance=var,
               noise=-1.0 if noise is None else noise, lr=lr, likelihood=lik, P=P, steps=np.array(STEPS),
               separate=int(separate))
    Xs = X[: min(N, 50)] + 0.05
    for step in range(1, max(STEPS) + 1):
        model.natgrad_step((X, Y), lr=lr)
        if step in STEPS:
            last = model.last
            mean_s, var_s = model.predict_f(Xs)
            out.update({f"s{step}_lambda_1": model.lambda_1.copy(), f"s{step}_lambda_2_sqrt": model.lambda_2_sqrt.copy(),
                        f"s{step}_mean": last["mean"], f"s{step}_var": last["var"], f"s{step}_g0": last["g0"],
                        f"s{step}_g1": last["g1"], f"s{step}_G0": last["G0"], f"s{step}_G1": last["G1"],
                        f"s{step}_elbo": model.elbo((X, Y)), f"s{step}_pred_mean": mean_s, f"s{step}_pred_var": var_s})
    out["Xs"] = Xs
    return out


WHITE_CASES = {
    # t_SVGP_white (reference src/models/tsvgp_white.py), one latent: name: (N, M, D, likelihood, lengthscale, variance, noise, lr)
    "white_gaussian_d3": (160, 12, 3, "gaussian", 1.0, 1.0, 0.1, 0.8),
    "white_bernoulli_d2": (160, 12, 2, "bernoulli", 1.0, 1.5, None, 0.8),
}


def make_white_case(name):
    N, M, D, lik, ls, var, noise, lr = WHITE_CASES[name]
    rng = np.random.RandomState(123)
    X = rng.randn(N, D)
    f = np.sin(X @ rng.randn(D, 1))
    eps = rng.randn(N, 1)
    Y = f + np.sqrt(0.1) * eps if lik == "gaussian" else (f + np.sqrt(0.1) * eps > 0).astype(np.float64)
    Z = X[:M].copy()
    model = O.t_SVGP_white(O.SquaredExponential(variance=var, lengthscales=ls),
                           O.Gaussian(variance=noise) if lik == "gaussian" else O.Bernoulli(), Z)
    out = dict(X=X, Y=Y, Z=Z, lengthscales=ls, variance=var, noise=-1.0 if noise is None else noise, lr=lr,
               likelihood=lik, steps=np.array(STEPS))
    Xs = X[:50] + 0.05
    for step in range(1, max(STEPS) + 1):
        model.natgrad_step((X, Y), lr=lr)
        if step in STEPS:
            mean_s, var_s = model.predict_f(Xs)
            out.update({f"s{step}_lambda_1": model.lambda_1.copy(), f"s{step}_lambda_2": model.lambda_2.copy(),
                        f"s{step}_mean": model.last["mean"], f"s{step}_var": model.last["var"],
                        f"s{step}_elbo": model.elbo((X, Y)), f"s{step}_pred_mean": mean_s, f"s{step}_pred_var": var_s})
    out["Xs"] = Xs
    return out


if __name__ == "__main__":
    for name in CASES:
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **make_case(name))
        print("wrote", name)
    os.makedirs(os.path.join(HERE, "white"), exist_ok=True)
    for name in WHITE_CASES:
        np.savez_compressed(os.path.join(HERE, "white", name + ".npz"), **make_white_case(name))
        print("wrote white/" + name)
