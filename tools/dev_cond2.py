"""Times util.cond2_estimate (the route gate) on the headline's K_uu: engine path against the torch path."""
import os
import sys
import time
from importlib import import_module

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch

U = import_module("t-svgp_amd.util")
E = import_module("t-svgp_amd.estep")
eng = E.EStepEngine(torch.float64, "cuda:0")
rng = np.random.RandomState(0)
M = 1024
Z = rng.randn(M, 8)
d2 = ((Z[:, None, :] - Z[None]) ** 2).sum(-1)
A = torch.as_tensor(np.exp(-0.5 * d2 / 4.0) + 1e-9 * np.eye(M), device="cuda:0")
ev = np.linalg.eigvalsh(A.cpu().numpy())
print("exact cond %.4e" % (ev[-1] / ev[0]))
for name, potrf in (("engine", eng.cholesky), ("torch", None)):
    for _ in range(3):
        c = U.cond2_estimate(A, potrf)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        c = U.cond2_estimate(A, potrf)
    c = float(c[0])
    dt = (time.perf_counter() - t0) / 10
    print("%s: cond %.4e  %.3f ms per call (host clock, one read at the end)" % (name, c, dt * 1e3))
