#!/usr/bin/env python3
"""Diagnostic (round 4): the C4 test's own worker, repeated; which tag / rows of the predictions disagree with the oracle."""
import os, sys
import numpy as np, torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import free_port
from tests.test_gpu_sharded import _c4_problem, _worker_c4, C4_ROWS, C4_STEPS
from oracle import tsvgp_oracle as O
if __name__ == "__main__":
    X, Y, Z = _c4_problem()
    ora = O.t_SVGP(O.SquaredExponential(1.0, 1.0), O.Bernoulli(), Z, num_data=C4_ROWS)
    for _ in range(C4_STEPS):
        ora.natgrad_step((X, Y), lr=0.8)
    mu_o, var_o = ora.predict_f(X[:300])
    for rep in range(4):
        for world, backend in ((2, "gloo"), (1, "nccl")):
            out = f"/tmp/dbg_c4b_{world}_{rep}.npz"
            mp.spawn(_worker_c4, args=(world, free_port(), out, backend), nprocs=world, join=True)
            got = np.load(out)
            for tag in ("eager", "graph", "graphfork"):
                dmu, dvar = np.abs(got[tag + "_mu"] - mu_o)[:, 0], np.abs(got[tag + "_var"] - var_o)[:, 0]
                print(f"rep {rep} world {world} {backend} {tag}: mu max {dmu.max():.2e} rows>1e-3 {np.where(dmu > 1e-3)[0]} | var max {dvar.max():.2e} rows>1e-3 {np.where(dvar > 1e-3)[0]}", flush=True)
