"""Measured error of the projection routes against the oracle at the benchmark's M (1024, D = 8), as a function of
cond(K_uu + 1e-9 I): the evidence behind t_SVGP.DIRECT_MAX_COND.  Lengthscale sweeps the conditioning (Z = X[:M] of
randn inputs, as in bench.py); for every lengthscale EIGHT E-steps (lr 0.8: what the reference's loop runs per M-step,
experiments/uci_regression.py:17) on N rows with the route forced, Gaussian and Bernoulli likelihood; state and ELBO against
the oracle after every step (the worst step is reported beside the last).
usage: python tools/route_gate.py [N] [lengthscale ...]"""
import importlib
import sys

import numpy as np

sys.path.insert(0, "/root/repo")
from oracle import tsvgp_oracle as O  # noqa: E402
from tests.helpers import relerr, synthetic  # noqa: E402

p = importlib.import_module("t-svgp_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
ells = [float(a) for a in sys.argv[2:]] or [1.0, 1.2, 1.3, 1.4, 1.5, 1.7]
STEPS = 8
print(f"N = {N}, M = 1024, D = 8; max rel err vs the oracle over {STEPS} steps (tolerance 1e-8; ELBO 1e-9): "
      "worst step (l1, L2) | last step (l1, L2) | ELBO after the last", flush=True)
for lik in ("gaussian", "bernoulli"):
    X, Y, Z = synthetic(N=N, M=1024, D=8, lik=lik, seed=0)
    for ell in ells:
        mk = (lambda mod: mod.Gaussian(0.1)) if lik == "gaussian" else (lambda mod: mod.Bernoulli())
        ora = O.t_SVGP(O.SquaredExponential(1.0, ell), mk(O), Z)
        states = []
        for _ in range(STEPS):
            ora.natgrad_step((X, Y), lr=0.8)
            states.append((ora.lambda_1.copy(), ora.lambda_2.copy()))
        e_o = ora.elbo((X, Y))
        m2 = p.t_SVGP(p.SquaredExponential(1.0, ell), mk(p), Z)
        m2._routes(1e-9)
        cond = m2._cond_cache[1][0]
        for route in ("direct", "whitened"):
            m = p.t_SVGP(p.SquaredExponential(1.0, ell), mk(p), Z, projection=route)
            errs = []
            try:
                for s in range(STEPS):
                    m.natgrad_step((X, Y), lr=0.8)
                    errs.append((relerr(m.lambda_1.numpy(), states[s][0]), relerr(m.lambda_2.cpu().numpy(), states[s][1])))
                e_h = float(m.elbo((X, Y)))
                w1, w2 = max(e[0] for e in errs), max(e[1] for e in errs)
                msg = f"worst l1 {w1:.1e} L2 {w2:.1e} | last l1 {errs[-1][0]:.1e} L2 {errs[-1][1]:.1e} | elbo {abs(e_h - e_o) / abs(e_o):.1e}"
            except FloatingPointError as e:
                msg = f"FAILED after {len(errs)} steps ({e})"
            print(f"{lik:9s} l = {ell:4.2f} cond {cond:9.3g} {route:9s} {msg}", flush=True)
