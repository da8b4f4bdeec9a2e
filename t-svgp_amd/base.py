"""Minimal GPflow-style parameter container on torch tensors.

Stands in for ``gpflow.base.Parameter`` / ``gpflow.config`` as used by the reference hot path
(reference src/sites.py:9-11,56,63; src/models/tsvgp.py:18,177,210).  Only what the E-step touches is
mirrored: ``assign``, ``numpy``, float64 default, ``default_jitter() == 1e-6``.
"""
from __future__ import annotations

import numpy as np
import torch

_DEFAULT_FLOAT = torch.float64
_DEFAULT_JITTER = 1e-6


def default_float():
    """gpflow.config.default_float() [ext]: the reference path is fp64-only (tsvgp.py:265)."""
    return _DEFAULT_FLOAT


def default_jitter():
    """gpflow.config.default_jitter() [ext] = 1e-6."""
    return _DEFAULT_JITTER


def default_device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def to_tensor(value, dtype=None, device=None) -> torch.Tensor:
    dtype = dtype or _DEFAULT_FLOAT
    device = device or default_device()
    if isinstance(value, Parameter):
        value = value.value
    if isinstance(value, torch.Tensor):
        return value.to(device=device, dtype=dtype)
    return torch.as_tensor(np.asarray(value), dtype=dtype).to(device)


def _host_scalar(value):
    """Python float of a scalar that already lives on the host; None for arrays and for device tensors."""
    if isinstance(value, Parameter):
        return value._host if value._host_version == value._stamp() else None
    if isinstance(value, torch.Tensor):
        return float(value) if (value.dim() == 0 and value.device.type == "cpu") else None
    if isinstance(value, (int, float, np.floating, np.integer)):
        return float(value)
    if isinstance(value, np.ndarray) and value.ndim == 0:
        return float(value)
    return None


class Parameter:
    """Holds one tensor; ``assign`` replaces its value in place (shape-checked)."""

    def __init__(self, value, dtype=None, trainable=True, name=None, device=None):
        self._value = to_tensor(value, dtype, device).clone()
        self.trainable = trainable
        self.name = name
        self.version = 0  # bumped by assign(): lets cached kernel factorisations notice external parameter changes
        self._host = _host_scalar(value)  # scalar parameters: the value as a Python float, without a device read
        self._host_version = self._stamp() if self._host is not None else None

    def _stamp(self):
        """(assign counter, the tensor's own edit counter): an in-place edit of ``.value`` changes the second."""
        return (self.version, self._value._version)

    def stamp(self):
        """Cache key of this parameter's CONTENTS as far as the host can tell without reading the device: identity, the
        ``assign`` counter and the tensor's own edit counter (an in-place edit of ``.value`` -- ``p.value.mul_(2)`` -- does
        not pass through ``assign``).  What a captured hipGraph or a cached factor that baked the value in is keyed on."""
        return (id(self), self.version, self._value._version)

    def item(self) -> float:
        """The scalar value as a Python float.  Kernel launches take scalars by value; reading them back from the
        device on every E-step would put a host synchronisation in the middle of the step, so the host copy is kept
        from the assignment (or read once per assignment when the value came as a device tensor)."""
        if self._host_version != self._stamp():
            self._host = float(self._value)
            self._host_version = self._stamp()
        return self._host

    @property
    def value(self) -> torch.Tensor:
        return self._value

    @property
    def shape(self):
        return tuple(self._value.shape)

    @property
    def dtype(self):
        return self._value.dtype

    @property
    def device(self):
        return self._value.device

    def assign(self, value):
        new = to_tensor(value, self._value.dtype, self._value.device)
        if new.dim() == 0 and self._value.dim() > 0:
            new = new.expand_as(self._value)
        if tuple(new.shape) != tuple(self._value.shape):
            raise ValueError(f"assign: shape {tuple(new.shape)} does not match {tuple(self._value.shape)}")
        self._value = new.clone()
        self.version += 1
        host = _host_scalar(value)
        if host is not None:
            self._host, self._host_version = host, self._stamp()
        return self

    def assign_owned(self, new: torch.Tensor):
        """``assign`` for a freshly computed tensor that nobody else holds: adopted as is (no copy)."""
        if tuple(new.shape) != tuple(self._value.shape) or new.dtype != self._value.dtype or new.device != self._value.device:
            return self.assign(new)
        self._value = new
        self.version += 1
        return self

    def numpy(self) -> np.ndarray:
        return self._value.detach().cpu().numpy()

    def __array__(self, dtype=None):
        a = self.numpy()
        return a.astype(dtype) if dtype is not None else a

    def __repr__(self):
        return f"Parameter(shape={self.shape}, dtype={self.dtype}, device={self.device})"
