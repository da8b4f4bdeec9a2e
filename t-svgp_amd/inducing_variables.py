"""``gpflow.inducing_variables.InducingPoints`` / ``inducingpoint_wrapper`` mirror (reference tsvgp.py:22,150)."""
from __future__ import annotations

from .base import Parameter


class InducingPoints:
    def __init__(self, Z, name=None):
        self.Z = Parameter(Z)
        if self.Z.value.dim() != 2:
            raise ValueError("InducingPoints expects Z of shape [M, D]")
        self.name = name

    @property
    def num_inducing(self) -> int:
        return self.Z.shape[0]

    def __len__(self):
        return self.num_inducing


class SharedIndependentInducingVariables:
    """``gpflow.inducing_variables.SharedIndependentInducingVariables`` [ext]: one set of inducing points shared by
    all latent GPs (reference docs/notebooks/heteroskedastic.py:72-74; special-cased at tsvgp.py:249-252)."""

    def __init__(self, inducing_variable, name=None):
        self.inducing_variable = inducingpoint_wrapper(inducing_variable)
        self.inducing_variables = [self.inducing_variable]  # the attribute tsvgp.py:252 reads
        self.name = name

    @property
    def Z(self):
        return self.inducing_variable.Z

    @property
    def num_inducing(self) -> int:
        return self.inducing_variable.num_inducing

    def __len__(self):
        return self.num_inducing


def inducingpoint_wrapper(inducing_variable):
    """Accepts an InducingPoints (or shared multi-output wrapper) or a raw [M, D] array
    (reference docs/notebooks/regression_1D.py:79-84)."""
    if isinstance(inducing_variable, (InducingPoints, SharedIndependentInducingVariables)):
        return inducing_variable
    return InducingPoints(inducing_variable)
