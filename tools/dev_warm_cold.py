#!/usr/bin/env python3
"""Warm (cache_whitened) against cold E-steps of tests/test_gpu_model.py's problem: the measured difference, per step."""
import importlib, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, torch
from helpers import pkg, synthetic, relerr
p = pkg()
X, Y, Z = synthetic(N=900, M=40, D=3, lik="bernoulli", seed=5)
Xd, Yd = torch.as_tensor(X, device="cuda:0"), torch.as_tensor(Y, device="cuda:0")
cold = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Bernoulli(), Z)
warm = p.t_SVGP(p.SquaredExponential(1.0, 1.0), p.Bernoulli(), Z, cache_whitened=True)
for i in range(6):
    if i == 3:
        cold.kernel.lengthscales.assign(0.8); warm.kernel.lengthscales.assign(0.8)
    jit = 1e-9 if i != 4 else 1e-8
    cold.natgrad_step((Xd, Yd), lr=0.7, jitter=jit); warm.natgrad_step((Xd, Yd), lr=0.7, jitter=jit)
    if i == 1:
        warm.predict_f(Xd[:100])
    print(i, "routes", cold._routes(jit), "lambda_1", relerr(warm.lambda_1.numpy(), cold.lambda_1.numpy()),
          "Lambda_2", relerr(warm.lambda_2.cpu().numpy(), cold.lambda_2.cpu().numpy()),
          "bitwise", bool(torch.equal(warm.lambda_1.value, cold.lambda_1.value)), "use_graph cold", cold._wants_graph(Xd))
