"""Gaussian exponential-family site containers (mirror of reference src/sites.py).

``DenseSites`` is the state the E-step mutates: lambda_1 [M, P] and the Cholesky-like factor of Lambda_2,
``lambda_2_sqrt`` [P, M, M], kept lower-triangular (the reference stores it under GPflow's ``triangular()``
transform, src/sites.py:63).  Note the reference's sign convention: the factor has a NEGATIVE diagonal
(src/models/tsvgp.py:176-179,300).
"""
from __future__ import annotations

import torch

from .base import Parameter, to_tensor


class Sites:
    """The base sites class (reference src/sites.py:14-23)."""

    def __init__(self, name=None):
        self.name = name


class DenseSites(Sites):
    """Sites with dense lambda_2 saved as a Cholesky factor (reference src/sites.py:43-80)."""

    def __init__(self, lambda_1, lambda_2_sqrt=None, lambda_2=None, name=None):
        super().__init__(name=name)
        self.lambda_1 = Parameter(lambda_1, trainable=False)  # [M, P]
        self.num_latent_gps = self.lambda_1.shape[0]  # sic, as reference src/sites.py:57
        assert (lambda_2_sqrt is not None) or (lambda_2 is not None)
        if lambda_2_sqrt is not None:
            self.factor = True
            self._lambda_2_sqrt = Parameter(torch.tril(to_tensor(lambda_2_sqrt)), trainable=False)  # [P, M, M]
        else:
            self.factor = False
            self._lambda_2 = Parameter(lambda_2, trainable=False)  # [P, M, M]

    @property
    def lambda_2(self) -> torch.Tensor:
        """second natural parameter"""
        if self.factor:
            L = self._lambda_2_sqrt.value
            return L @ L.transpose(-1, -2)
        return self._lambda_2.value

    @property
    def lambda_2_sqrt(self):
        """Cholesky factor of the second natural parameter (a Parameter when stored as a factor)."""
        if self.factor:
            return self._lambda_2_sqrt
        return Parameter(torch.linalg.cholesky(self._lambda_2.value), trainable=False)

    def assign_lambda_2(self, value):
        """Full second natural parameter (the whitened model's state, reference src/models/tsvgp_white.py:248)."""
        if self.factor:
            raise ValueError("sites are stored as a Cholesky factor")
        self._lambda_2.assign(value)

    def assign_lambda_2_sqrt(self, value, lower_and_owned: bool = False):
        """triangular() transform: only the lower triangle is kept (reference src/sites.py:63).
        ``lower_and_owned``: ``value`` is a fresh tensor with exact zeros above the diagonal (a factor straight out of the
        factorisation): adopted without the triangle pass and the copy."""
        if not self.factor:
            raise ValueError("sites are stored as a full lambda_2")
        if lower_and_owned and isinstance(value, torch.Tensor):
            self._lambda_2_sqrt.assign_owned(value)
        else:
            self._lambda_2_sqrt.assign(torch.tril(to_tensor(value, device=self._lambda_2_sqrt.device)))
