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


def inducingpoint_wrapper(inducing_variable):
    """Accepts an InducingPoints or a raw [M, D] array (reference docs/notebooks/regression_1D.py:79-84)."""
    if isinstance(inducing_variable, InducingPoints):
        return inducing_variable
    return InducingPoints(inducing_variable)
