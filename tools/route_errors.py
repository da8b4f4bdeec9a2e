"""per-step error of both projection routes against the golden fixture + state-sensitivity floor."""
import numpy as np, importlib, sys, torch
sys.path.insert(0, "/root/repo")
from tests.test_golden_cpu import load_model
from oracle import tsvgp_oracle as O
p = importlib.import_module("t-svgp_amd")
def rel(a, b): return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))
name = sys.argv[1] if len(sys.argv) > 1 else "gaussian_d3_p2"
fx = np.load(f"/root/repo/tests/golden/{name}.npz")
X, Y, lr = fx["X"], fx["Y"], float(fx["lr"])
print("lr", lr, "N", X.shape, "M", fx["Z"].shape)
for route in ("whitened", "direct", "auto"):
    m = load_model(fx, p); m.projection = route
    o = load_model(fx, O)
    rng = np.random.default_rng(0)
    for step in range(1, 11):
        if step in (1, 2, 10):
            # what the test does: whitened-route intermediates from the current state
            ops = m._site_operands(whiten_jitter=1e-9)
            st = m._get_engine().run(m._as_device(X), m._as_device(Y), ops["Z"], m.kernel, moment_Tm=ops["moment_Tm"],
                                     moment_mode=ops["moment_mode"], gamma=ops["gamma"], lik_id=m.likelihood.lik_id,
                                     lik_param=m.likelihood.lik_param, whiten_T=ops["whiten_T"], whiten_mode=ops["whiten_mode"], sites=True,
                                     want_moments=True, want_grads=True)
            e_mean = rel(st.mean.cpu().numpy(), fx[f"s{step}_mean"]); e_var = rel(st.var.cpu().numpy(), fx[f"s{step}_var"])
            # sensitivity floor: oracle moments from an oracle state perturbed by 2e-16 relative noise
            o2 = load_model(fx, O)
            o2.sites.lambda_1 = o.lambda_1 * (1 + 2e-16 * rng.standard_normal(o.lambda_1.shape))
            Ls = o.lambda_2_sqrt * (1 + 2e-16 * rng.standard_normal(o.lambda_2_sqrt.shape))
            o2.sites._lambda_2_sqrt = np.tril(Ls)
            mu0, _ = o.predict_f(X); mu2, _ = o2.predict_f(X)
            # oracle evaluated AT the HIP state
            o3 = load_model(fx, O); o3.sites.lambda_1 = m.lambda_1.numpy(); o3.sites._lambda_2_sqrt = np.tril(m.lambda_2_sqrt.numpy())
            mu3, _ = o3.predict_f(X)
            print(f"{route} step {step}: test mean err {e_mean:.2e} var err {e_var:.2e} | eps-perturbed-state floor {rel(mu2, mu0):.2e}"
                  f" | oracle@HIPstate vs fixture {rel(mu3, fx[f's{step}_mean']):.2e} | HIP vs oracle@HIPstate {rel(st.mean.cpu().numpy(), mu3):.2e}")
        o.natgrad_step((X, Y), lr=lr); m.natgrad_step((X, Y), lr=lr); print("   cond cache", m._cond_cache[1] if m._cond_cache else None, "l1", rel(m.lambda_1.numpy(), o.lambda_1))
        if step in (1, 2, 10):
            Lf = fx[f"s{step}_lambda_2_sqrt"]
            mu, var = m.predict_f(fx["Xs"])
            print(f"   post-step state: l1 {rel(m.lambda_1.numpy(), fx[f's{step}_lambda_1']):.2e} L2 {rel(m.lambda_2.cpu().numpy(), Lf @ np.swapaxes(Lf, -1, -2)):.2e}"
                  f" elbo {abs(float(m.elbo((X, Y))) - float(fx[f's{step}_elbo'])) / abs(float(fx[f's{step}_elbo'])):.2e}"
                  f" pred {rel(mu.cpu().numpy(), fx[f's{step}_pred_mean']):.2e} {rel(var.cpu().numpy(), fx[f's{step}_pred_var']):.2e}")
