#!/usr/bin/env python3
"""The t-SVGP branch of the reference's regression experiment (reference experiments/uci_regression.py:132-160) on a
synthetic problem, end to end on one MI355X: E-steps (natural gradients on the sites), ELBO / test NLPD log, Adam
M-steps on the kernel parameters, the noise variance and the inducing inputs.

    python examples/em_regression.py [--n 200000] [--m 256] [--d 8] [--iters 5]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tsvgp_amd as gp  # noqa: E402  (alias of the package directory t-svgp_amd/)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=200_000)
    ap.add_argument("--m", type=int, default=256)
    ap.add_argument("--d", type=int, default=8)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--kernel", default="Matern52", choices=["SquaredExponential", "Matern32", "Matern52"])
    args = ap.parse_args()
    rng = np.random.RandomState(0)
    X = rng.randn(args.n, args.d)
    Y = np.sin(X @ rng.randn(args.d, 1)) + np.sqrt(0.1) * rng.randn(args.n, 1)
    Xt, Yt = X[-2000:], Y[-2000:]
    X, Y = X[:-2000], Y[:-2000]
    dev = torch.device("cuda:0")
    Xd, Yd = torch.as_tensor(X, device=dev), torch.as_tensor(Y, device=dev)  # resident in HBM across the loop
    model = gp.t_SVGP(getattr(gp, args.kernel)(variance=1.0, lengthscales=np.ones(args.d)), gp.Gaussian(variance=0.5),
                      X[: args.m].copy(), num_data=X.shape[0])
    t0 = time.perf_counter()
    logf, nlpd = gp.training.em_fit(model, (Xd, Yd), iterations=args.iters, n_e_steps=8, n_m_steps=20, nat_lr=0.8,
                                    adam_lr=0.1, test_data=(Xt, Yt))  # experiments/uci_regression.py:20
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for i, (e, n) in enumerate(zip(logf, nlpd)):
        print(f"iteration {i}: ELBO {e:.3f}   test NLPD {n:.4f}")
    print(f"{args.iters} iterations (8 E-steps + 20 M-steps each) in {dt:.2f} s;  noise variance "
          f"{float(model.likelihood.variance.value):.4f} (true 0.1), kernel variance {float(model.kernel.variance.value):.3f}")


if __name__ == "__main__":
    main()
