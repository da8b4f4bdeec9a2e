"""Shared problem builders for the parity tests (seeded; sizes the CPU oracle finishes in seconds)."""
import importlib

import numpy as np

from oracle import tsvgp_oracle as O


def free_port() -> int:
    """A rendezvous port nobody holds right now: bound on the loopback interface, read back and released (what bench.py's
    self_launch does).  A port derived from the pid can collide with a previous test's socket in TIME_WAIT, and a collision
    there is a hang, not a failure."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def pkg():
    return importlib.import_module("t-svgp_amd")


def synthetic(N, M, D, P=1, lik="gaussian", seed=0, noise=0.1):
    """SURVEY.md section 8(d) generator: X = randn(N, D); w = randn(D, P); eps = randn(N, P); f = sin(X w)."""
    rng = np.random.RandomState(seed)
    X = rng.randn(N, D)
    w = rng.randn(D, P)
    eps = rng.randn(N, P)
    f = np.sin(X @ w)
    if lik == "gaussian":
        Y = f + np.sqrt(noise) * eps
    else:
        Y = (f + np.sqrt(noise) * eps > 0).astype(np.float64)
    Z = X[:M].copy()
    return X, Y, Z


def c1_problem(N=1000, M=32, seed=0):
    """Config C1: 1-D regression, X in [-1, 1], Y = sin(15 X) + eps, Z on a grid, l = 0.1, s2 = 0.3."""
    rng = np.random.RandomState(seed)
    X = rng.rand(N, 1) * 2 - 1
    Y = np.sin(15 * X) + rng.randn(N, 1)
    Z = np.linspace(X.min(), X.max(), M)[:, None]
    return X, Y, Z, dict(lengthscales=0.1, variance=0.3, noise=1.0)


def make_pair(Z, lik="gaussian", noise=0.1, lengthscales=1.0, variance=1.0, P=1, compute_dtype=None, num_data=None):
    """(HIP model, oracle model) with identical hyperparameters."""
    p = pkg()
    import torch

    hip = p.t_SVGP(p.SquaredExponential(variance=variance, lengthscales=lengthscales),
                   p.Gaussian(variance=noise) if lik == "gaussian" else p.Bernoulli(), Z, num_latent_gps=P,
                   num_data=num_data, compute_dtype=compute_dtype or torch.float64)
    ora = O.t_SVGP(O.SquaredExponential(variance=variance, lengthscales=lengthscales),
                   O.Gaussian(variance=noise) if lik == "gaussian" else O.Bernoulli(), Z, num_latent_gps=P,
                   num_data=num_data)
    return hip, ora


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
