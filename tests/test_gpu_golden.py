"""HIP path against the committed golden fixtures (inputs + expected outputs after E-steps 1, 2, 10).
fp64 tolerance: max rel err <= 1e-8 on lambda_1, Lambda_2, mean, var, g0, g1; |dELBO|/|ELBO| <= 1e-9.
The whitened accumulators are mapped back to the reference's G0 / G1 (tsvgp.py:279-280) before comparison."""
import os

import numpy as np
import pytest

from tests.helpers import pkg, relerr
from tests.test_golden_cpu import FIXTURES, load_model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_hip_matches_fixture(path):
    fx = np.load(path)
    p = pkg()
    B = p._backend
    model = load_model(fx, p)
    X, Y, lr = fx["X"], fx["Y"], float(fx["lr"])
    steps = [int(s) for s in fx["steps"]]
    lik_id = model.likelihood.lik_id
    for step in range(1, max(steps) + 1):
        if step in steps:  # intermediates of THIS step, computed from the pre-step state like the fixture's
            ops = model._site_operands(whiten_jitter=1e-9)
            st = model._get_engine().run(model._as_device(X), model._as_device(Y), ops["Z"], model.kernel,
                                         moment_Tm=ops["moment_Tm"], moment_mode=ops["moment_mode"], gamma=ops["gamma"],
                                         lik_id=lik_id, lik_param=model.likelihood.lik_param, whiten_T=ops["whiten_T"], whiten_mode=ops["whiten_mode"],
                                         sites=True, want_moments=True, want_grads=True)
            for key, val in (("mean", st.mean), ("var", st.var), ("g0", st.g0), ("g1", st.g1)):
                assert relerr(val.cpu().numpy(), fx[f"s{step}_{key}"]) < 1e-8, (step, key)
            acc2, acc1 = st.acc2.cpu().numpy(), st.acc1.cpu().numpy()
            U9 = np.broadcast_to(ops["U9"].cpu().numpy(), acc2.shape)  # one factor per latent for separate kernels
            G1 = np.stack([np.linalg.solve(u.T, np.linalg.solve(u.T, a).T).T for u, a in zip(U9, acc2)])
            G0 = np.stack([np.linalg.solve(u.T, a) for u, a in zip(U9, acc1)], axis=1)
            assert relerr(G1, fx[f"s{step}_G1"]) < 1e-8 and relerr(G0, fx[f"s{step}_G0"]) < 1e-8
        model.natgrad_step((X, Y), lr=lr)
        if step in steps:
            Ls = fx[f"s{step}_lambda_2_sqrt"]
            assert relerr(model.lambda_1.numpy(), fx[f"s{step}_lambda_1"]) < 1e-8
            assert relerr(model.lambda_2.cpu().numpy(), Ls @ np.swapaxes(Ls, -1, -2)) < 1e-8
            e = float(model.elbo((X, Y)))
            assert abs(e - float(fx[f"s{step}_elbo"])) < 1e-9 * abs(float(fx[f"s{step}_elbo"]))
            mu, var = model.predict_f(fx["Xs"])
            assert relerr(mu.cpu().numpy(), fx[f"s{step}_pred_mean"]) < 1e-8
            assert relerr(var.cpu().numpy(), fx[f"s{step}_pred_var"]) < 1e-8
