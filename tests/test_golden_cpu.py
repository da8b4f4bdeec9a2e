"""The committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle) are
reproduced by the oracle here: a regression pin that any change to the oracle shows up against."""
import glob
import os

import numpy as np
import pytest

from oracle import tsvgp_oracle as O

FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def load_model(fx, module):
    lik = module.Gaussian(variance=float(fx["noise"])) if str(fx["likelihood"]) == "gaussian" else module.Bernoulli()
    if "separate" in fx.files and int(fx["separate"]):  # one kernel per latent on shared inducing points
        kernel = module.SeparateIndependent([module.SquaredExponential(variance=float(fx["variance"]), lengthscales=float(l))
                                             for l in fx["lengthscales"]])
        return module.t_SVGP(kernel, lik, module.SharedIndependentInducingVariables(fx["Z"]), num_latent_gps=int(fx["P"]))
    return module.t_SVGP(module.SquaredExponential(variance=float(fx["variance"]), lengthscales=float(fx["lengthscales"])),
                         lik, fx["Z"], num_latent_gps=int(fx["P"]))


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_reproduces_fixture(path):
    fx = np.load(path)
    model = load_model(fx, O)
    X, Y, lr = fx["X"], fx["Y"], float(fx["lr"])
    steps = [int(s) for s in fx["steps"]]
    for step in range(1, max(steps) + 1):
        model.natgrad_step((X, Y), lr=lr)
        if step in steps:
            np.testing.assert_allclose(model.lambda_1, fx[f"s{step}_lambda_1"], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(model.lambda_2, fx[f"s{step}_lambda_2_sqrt"] @ np.swapaxes(fx[f"s{step}_lambda_2_sqrt"], -1, -2),
                                       rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(model.elbo((X, Y)), fx[f"s{step}_elbo"], rtol=1e-10)
    assert len(FIXTURES) >= 5


WHITE_FIXTURES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "white", "*.npz")))


def load_white_model(fx, module, **kw):
    lik = module.Gaussian(variance=float(fx["noise"])) if str(fx["likelihood"]) == "gaussian" else module.Bernoulli()
    return module.t_SVGP_white(module.SquaredExponential(variance=float(fx["variance"]), lengthscales=float(fx["lengthscales"])),
                               lik, fx["Z"], **kw)


@pytest.mark.parametrize("path", WHITE_FIXTURES, ids=[os.path.basename(p)[:-4] for p in WHITE_FIXTURES])
def test_oracle_reproduces_white_fixture(path):
    fx = np.load(path)
    model = load_white_model(fx, O)
    X, Y, lr = fx["X"], fx["Y"], float(fx["lr"])
    steps = [int(s) for s in fx["steps"]]
    for step in range(1, max(steps) + 1):
        model.natgrad_step((X, Y), lr=lr)
        if step in steps:
            np.testing.assert_allclose(model.lambda_1, fx[f"s{step}_lambda_1"], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(model.lambda_2, fx[f"s{step}_lambda_2"], rtol=1e-9, atol=1e-12)
            np.testing.assert_allclose(model.elbo((X, Y)), fx[f"s{step}_elbo"], rtol=1e-10)
    assert len(WHITE_FIXTURES) >= 2
