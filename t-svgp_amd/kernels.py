"""Covariance functions on the hot path.

``SquaredExponential`` mirrors ``gpflow.kernels.SquaredExponential`` (the kernel every reference test, notebook and
experiment on this path uses: reference tests/models/test_tsvgp.py:100, experiments/uci_regression.py:186-187).
The N-sized evaluations K(X, Z) run in the HIP fill kernel (``tsvgp_se_fill_*``); nothing here loops over data.
"""
from __future__ import annotations

import torch

from .base import Parameter, to_tensor


class SquaredExponential:
    """k(x, z) = variance * exp(-0.5 * sum_d ((x_d - z_d) / lengthscales_d)^2)."""

    kind = 0  # TSVGP_KERNEL_SE: selects the profile inside the HIP fill kernel (include/tsvgp_hip.h)

    def __init__(self, variance=1.0, lengthscales=1.0, name=None):
        self.variance = Parameter(variance)
        self.lengthscales = Parameter(lengthscales)
        self.name = name or type(self).__name__.lower()

    @property
    def ard(self) -> bool:
        return self.lengthscales.value.dim() > 0

    def inv_lengthscales(self, D: int, dtype=None, device=None) -> torch.Tensor:
        """[D] vector of 1/lengthscale (isotropic values are broadcast)."""
        # cached per parameter version: the E-step asks for it twice per step (K_uu, K(X, Z)) right behind the status
        # read of the previous step, where the GPU waits for the host
        # keyed on the Parameter object and on the tensor behind it as well: a replaced Parameter restarts its version at 0,
        # an in-place edit of .value bumps only the tensor's own counter.  The returned tensor is shared: read-only.
        par = self.lengthscales
        key = (id(par), par.version, par.value.data_ptr(), par.value._version, D, dtype, str(device))
        hit = self.__dict__.get("_inv_ls_cache")
        if hit is not None and hit[0] == key:
            return hit[1]
        ls = self.lengthscales.value
        if ls.dim() == 0:
            ls = ls.expand(D)
        if ls.shape[0] != D:
            raise ValueError(f"lengthscales has {ls.shape[0]} entries but the inputs have {D} columns")
        out = to_tensor(1.0 / ls, dtype, device).contiguous()
        self.__dict__["_inv_ls_cache"] = (key, out)
        return out

    def K_diag(self, X) -> torch.Tensor:
        X = to_tensor(X)
        return self.variance.value.expand(X.shape[0]).clone()

    @staticmethod
    def _profile(s: torch.Tensor) -> torch.Tensor:
        return torch.exp(-0.5 * s)

    def K_torch(self, Z: torch.Tensor, variance: torch.Tensor, lengthscales: torch.Tensor) -> torch.Tensor:
        """K(Z, Z) as a differentiable torch expression of (variance, lengthscales, Z), difference form like the HIP
        fill.  Only for the M x M part of the M-step gradient (``t_SVGP.elbo_and_grads``); N-sized work never comes here."""
        Zs = Z / lengthscales
        diff = Zs[:, None, :] - Zs[None, :, :]
        return variance * self._profile(torch.sum(diff * diff, dim=-1))


class Matern32(SquaredExponential):
    """``gpflow.kernels.Matern32`` [ext]: variance * (1 + sqrt(3) r) exp(-sqrt(3) r)."""

    kind = 2

    @staticmethod
    def _profile(s):
        a = torch.sqrt(3.0 * torch.clamp(s, min=1e-36))
        return (1.0 + a) * torch.exp(-a)


class Matern52(SquaredExponential):
    """``gpflow.kernels.Matern52`` [ext] (reference experiments/uci_regression.py:42-44):
    variance * (1 + sqrt(5) r + 5 r^2 / 3) exp(-sqrt(5) r)."""

    kind = 3

    @staticmethod
    def _profile(s):
        sc = torch.clamp(s, min=1e-36)
        a = torch.sqrt(5.0 * sc)
        return (1.0 + a + 5.0 / 3.0 * sc) * torch.exp(-a)


class SeparateIndependent:
    """``gpflow.kernels.SeparateIndependent`` [ext]: P independent latent GPs, one kernel each, no mixing matrix
    (reference docs/notebooks/heteroskedastic.py:62-67).  Used with ``SharedIndependentInducingVariables``; the
    E-step then carries K_uu as [P, M, M] (batched factorisations) and fills K(X, Z) once per latent."""

    def __init__(self, kernels, name=None):
        self.kernels = list(kernels)
        if not self.kernels:
            raise ValueError("SeparateIndependent needs at least one kernel")
        self.name = name or "separate_independent"

    @property
    def num_latent_gps(self) -> int:
        return len(self.kernels)

    def K_diag(self, X) -> torch.Tensor:
        return torch.stack([k.K_diag(X) for k in self.kernels], dim=1)  # [N, P]


def latent_kernels(kernel, P: int):
    """The list of per-latent kernels behind ``kernel``: P references to one shared kernel, or the separate ones."""
    if isinstance(kernel, SeparateIndependent):
        if len(kernel.kernels) != P:
            raise ValueError(f"SeparateIndependent has {len(kernel.kernels)} kernels but the model has {P} latent GPs")
        return kernel.kernels
    return [kernel] * P
