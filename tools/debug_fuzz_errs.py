#!/usr/bin/env python3
"""Replays one white-model trial of tools/fuzz_parity.py and prints every error component.  (GPU box)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import tsvgp_oracle as O
p = importlib.import_module("t-svgp_amd")
seed, want = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.RandomState(seed)
for trial in range(want + 1):
    N, M, D, P = int(rng.randint(1, 3000)), int(rng.randint(1, 300)), int(rng.randint(1, 9)), int(rng.randint(1, 4))
    lik = ["gaussian", "bernoulli"][rng.randint(2)]
    kname = ["SquaredExponential", "Matern52", "Matern32"][rng.randint(3)]
    route = ["auto", "whitened", "direct", "projected"][rng.randint(4)]
    white = P == 1 and rng.rand() < 0.25
    separate = P >= 2 and rng.rand() < 0.4
    X = rng.randn(N, D)
    f = np.sin(X @ rng.randn(D, P))
    Y = f + 0.3 * rng.randn(N, P) if lik == "gaussian" else (f + 0.3 * rng.randn(N, P) > 0).astype(float)
    Z = rng.randn(M, D) * 1.5
    ls, var, noise = 0.7 + rng.rand(), 0.5 + rng.rand(), 0.05 + rng.rand() * 0.5
    if separate:
        lss, vars_ = 0.7 + rng.rand(P), 0.5 + rng.rand(P)
mkl = lambda mod: mod.Gaussian(noise) if lik == "gaussian" else mod.Bernoulli()
hip, ora = (mod.t_SVGP_white(getattr(mod, kname)(var, ls), mkl(mod), Z, num_data=N) for mod in (p, O))
rel = lambda a, b: float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
print(f"trial {want}: N={N} M={M} D={D} {lik} {kname} white; cond_k6 {hip._cond_k6():.2e} direct {hip._use_direct()}")
for step in range(3):
    hip.natgrad_step((X, Y), lr=0.7); ora.natgrad_step((X, Y), lr=0.7)
    print(f"step {step}: l1 {rel(hip.lambda_1.numpy(), ora.lambda_1):.2e} L2 {rel(hip.lambda_2.numpy(), ora.lambda_2):.2e}")
e_h, e_o = float(hip.elbo((X, Y))), float(ora.elbo((X, Y)))
mu_h, var_h = hip.predict_f(X[:50] + 0.1); mu_o, var_o = ora.predict_f(X[:50] + 0.1)
print(f"elbo rel {abs(e_h - e_o) / abs(e_o):.2e}  pred mean {rel(mu_h.cpu().numpy(), mu_o):.2e} var {rel(var_h.cpu().numpy(), var_o):.2e}")
