"""Measured error of the projection routes against the oracle at the benchmark's M (1024, D = 8), as a function of
cond(K_uu + 1e-9 I): the evidence behind t_SVGP.DIRECT_MAX_COND.  Lengthscale sweeps the conditioning (Z = X[:M] of
randn inputs, as in bench.py); for every lengthscale two E-steps (lr 0.8) on N rows with the route forced, state and ELBO
against the oracle.  usage: python tools/route_gate.py [N] [lengthscale ...]"""
import importlib
import sys

import numpy as np

sys.path.insert(0, "/root/repo")
from oracle import tsvgp_oracle as O  # noqa: E402
from tests.helpers import relerr, synthetic  # noqa: E402

p = importlib.import_module("t-svgp_amd")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
ells = [float(a) for a in sys.argv[2:]] or [1.0, 1.2, 1.3, 1.4, 1.5, 1.7, 2.0]
X, Y, Z = synthetic(N=N, M=1024, D=8, lik="gaussian", seed=0)
print(f"N = {N}, M = 1024, D = 8, Gaussian; max rel err vs the oracle after steps 1 and 2 (tolerance 1e-8; ELBO 1e-9)")
for ell in ells:
    ora = O.t_SVGP(O.SquaredExponential(1.0, ell), O.Gaussian(0.1), Z)
    states = []
    for _ in range(2):
        ora.natgrad_step((X, Y), lr=0.8)
        states.append((ora.lambda_1.copy(), ora.lambda_2.copy()))
    e_o = ora.elbo((X, Y))
    cond = None
    for route in ("direct", "whitened"):
        m = p.t_SVGP(p.SquaredExponential(1.0, ell), p.Gaussian(0.1), Z, projection=route)
        errs = []
        try:
            for s in range(2):
                m.natgrad_step((X, Y), lr=0.8)
                errs.append((relerr(m.lambda_1.numpy(), states[s][0]), relerr(m.lambda_2.cpu().numpy(), states[s][1])))
            e_h = float(m.elbo((X, Y)))
            msg = " ".join(f"l1 {a:.1e} L2 {b:.1e}" for a, b in errs) + f" elbo {abs(e_h - e_o) / abs(e_o):.1e}"
        except FloatingPointError as e:
            msg = f"FAILED ({e})"
        if cond is None:
            m2 = p.t_SVGP(p.SquaredExponential(1.0, ell), p.Gaussian(0.1), Z)
            m2._routes(1e-9)
            cond = m2._cond_cache[1][0]
        print(f"l = {ell:4.2f} cond {cond:9.3g} {route:9s} {msg}", flush=True)
