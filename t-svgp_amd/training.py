"""E/M training loop around the HIP E-step: mirror of the t-SVGP branch of the reference's driver
(reference experiments/uci_regression.py:90-183): ``n_e_steps`` natural-gradient E-steps, then ``n_m_steps`` Adam steps on
the kernel variance / lengthscales, the Gaussian noise variance and the inducing inputs (``model.trainable_variables``
there; the sites are not trainable, src/sites.py:56-63).

GPflow optimises UNCONSTRAINED variables: positive parameters live behind a softplus transform [ext]
(``gpflow.utilities.positive()``, lower bound 0; the Gaussian likelihood's variance behind ``positive(lower=1e-6)``, a
softplus shifted by ``gpflow.likelihoods.Gaussian.DEFAULT_VARIANCE_LOWER_BOUND`` [ext]), the inducing inputs are
unconstrained.  ``Adam`` follows
``tf.optimizers.Adam`` [ext] (beta_1 = 0.9, beta_2 = 0.999, epsilon = 1e-7, bias-corrected step size).
"""
from __future__ import annotations

import math

import torch


def _softplus_inv(x: torch.Tensor) -> torch.Tensor:
    return x + torch.log(-torch.expm1(-x))


class Adam:
    """tf.optimizers.Adam [ext] on a dict of unconstrained tensors (minimises: pass gradients of the LOSS)."""

    def __init__(self, learning_rate=0.01, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.lr, self.b1, self.b2, self.eps = learning_rate, beta_1, beta_2, epsilon
        self.t = 0
        self.m, self.v = {}, {}

    def step(self, variables: dict, grads: dict) -> None:
        self.t += 1
        lr_t = self.lr * math.sqrt(1.0 - self.b2 ** self.t) / (1.0 - self.b1 ** self.t)
        for name, g in grads.items():
            m = self.m.setdefault(name, torch.zeros_like(g))
            v = self.v.setdefault(name, torch.zeros_like(g))
            m.mul_(self.b1).add_(g, alpha=1.0 - self.b1)
            v.mul_(self.b2).addcmul_(g, g, value=1.0 - self.b2)
            variables[name] = variables[name] - lr_t * m / (torch.sqrt(v) + self.eps)


VARIANCE_LOWER_BOUND = 1e-6  # gpflow.likelihoods.Gaussian: variance = Parameter(..., transform=positive(lower=1e-6)) [ext]


def trainable_parameters(model) -> dict:
    """name -> (Parameter, lower bound of its shifted-softplus transform, or None when unconstrained) for what the
    reference's M-step trains; the names are those of ``t_SVGP.elbo_and_grads`` ("kernels.<p>.variance" ... with one
    kernel per latent)."""
    kern = model.kernel
    if hasattr(kern, "kernels"):  # SeparateIndependent
        out = {}
        for p, k in enumerate(kern.kernels):
            out[f"kernels.{p}.variance"] = (k.variance, 0.0)
            out[f"kernels.{p}.lengthscales"] = (k.lengthscales, 0.0)
    else:
        out = {"variance": (kern.variance, 0.0), "lengthscales": (kern.lengthscales, 0.0)}
    out["Z"] = (model.inducing_variable.Z, None)
    if hasattr(model.likelihood, "variance"):
        out["likelihood_variance"] = (model.likelihood.variance, VARIANCE_LOWER_BOUND)
    return out


def m_step(model, data, optimizer: Adam, steps: int = 1):
    """``steps`` Adam steps on the negative ELBO with the sites fixed (experiments/uci_regression.py:159-160).
    Returns the ELBO seen at the last gradient evaluation."""
    params = trainable_parameters(model)
    elbo = None
    for _ in range(steps):
        elbo, grads = model.elbo_and_grads(data)
        u, gu = {}, {}
        for name, (par, lower) in params.items():
            theta = par.value.detach().to(torch.float64)
            g = -grads[name].to(theta.device)  # loss = -ELBO
            if lower is not None:  # theta = lower + softplus(u)
                u[name] = _softplus_inv(theta - lower)
                gu[name] = g.reshape(theta.shape) * torch.sigmoid(u[name])  # d theta / d u = sigmoid(u)
            else:
                u[name], gu[name] = theta, g
        optimizer.step(u, gu)
        for name, (par, lower) in params.items():
            par.assign(lower + torch.nn.functional.softplus(u[name]) if lower is not None else u[name])
    return elbo


def em_fit(model, data, iterations: int, n_e_steps: int = 8, n_m_steps: int = 20, nat_lr: float = 0.8,
           adam_lr: float = 0.1, test_data=None, optimizer: Adam = None):
    """The t-SVGP branch of the reference's training loop (experiments/uci_regression.py:132-160; defaults :17-21): per
    iteration ``n_e_steps`` E-steps, the ELBO (and test NLPD) logged, then ``n_m_steps`` M-steps.
    Returns (logf, nlpd)."""
    optimizer = optimizer or Adam(adam_lr)
    logf, nlpd = [], []
    for _ in range(iterations):
        for _ in range(n_e_steps):
            model.natgrad_step(data, lr=nat_lr)
        logf.append(float(model.elbo(data)))
        if test_data is not None:
            nlpd.append(-float(torch.mean(model.predict_log_density(test_data))))
        m_step(model, data, optimizer, n_m_steps)
    return logf, nlpd
