"""Likelihood objects of the hot path: ``Gaussian`` and ``Bernoulli`` (probit, 20-point Gauss-Hermite).

Their N-sized maps -- variational expectations and the (mean, var) gradients the E-step needs
(reference src/models/tsvgp.py:256-263) -- run inside the fused HIP moments kernel
(``tsvgp_moments_*`` with ``lik`` = GAUSSIAN / BERNOULLI); the classes here only carry parameters and
the small predictive helpers drivers call on test points (experiments/uci_regression.py:157).
"""
from __future__ import annotations

import math

import torch

from . import _backend as B
from .base import Parameter


class Gaussian:
    """gpflow.likelihoods.Gaussian [ext]."""

    lik_id = B.LIK_GAUSSIAN

    def __init__(self, variance=1.0):
        self.variance = Parameter(variance)

    @property
    def lik_param(self) -> float:
        return self.variance.item()

    def predict_mean_and_var(self, Fmu, Fvar):
        return Fmu, Fvar + self.variance.value

    def predict_log_density(self, Fmu, Fvar, Y):
        v = Fvar + self.variance.value
        return torch.sum(-0.5 * (math.log(2 * math.pi) + torch.log(v) + (Y - Fmu) ** 2 / v), dim=-1)


class Bernoulli:
    """gpflow.likelihoods.Bernoulli with the default probit link (1e-3 jitter) [ext]."""

    lik_id = B.LIK_BERNOULLI
    lik_param = 0.0
    num_gauss_hermite_points = 20

    @staticmethod
    def invlink(F):
        return 0.5 * (1.0 + torch.erf(F / math.sqrt(2.0))) * (1 - 2e-3) + 1e-3

    def predict_mean_and_var(self, Fmu, Fvar):
        p = self.invlink(Fmu / torch.sqrt(1 + Fvar))
        return p, p - torch.square(p)

    def predict_log_density(self, Fmu, Fvar, Y):
        p = self.invlink(Fmu / torch.sqrt(1 + Fvar))
        return torch.sum(torch.log(torch.where(Y == 1, p, 1 - p)), dim=-1)
