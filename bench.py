#!/usr/bin/env python3
"""bench.py -- natural-gradient E-steps/sec of the t-SVGP hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload ns|c2|c3|c5|c1]

A "step" is one full-batch ``t_SVGP.natgrad_step((X, Y), lr=0.8)`` (reference src/models/tsvgp.py:234-304) over all
N rows of a synthetic problem whose inputs are already resident in HBM; everything that depends on (theta, Z, lambda)
is rebuilt inside every step ("cold" E-step: kernel matrices, Choleskys, whitening, moments, site accumulation, site
update).  With N > 1 the driver starts one process per GPU (torch.distributed.run); the N rows are sharded
contiguously and each step performs one RCCL all-reduce of the packed dual accumulators ("strong" scaling: total
work is fixed).  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json):
  ns  (default) north_star target / the metric's sizes: Gaussian regression N=1e6, M=1024, D=8, P=1, fp64
  c2  configs[1]: Gaussian regression N=1e6, M=512,  D=8,  fp64
  c3  configs[2]: Bernoulli (probit, GH-20) N=1e6, M=1024, D=16, fp32 N-arrays (M x M algebra stays fp64)
  c5  configs[4]: P=8 latents, one SE kernel per latent (lengthscales linspace(0.8, 1.5, 8)) on shared inducing points:
      K_uu [P, M, M], batched factorisations, one K(X, Z) fill per latent; N=1e6, M=1024, D=8, fp64  
  c5s the shared-kernel special case of c5 (one K(X, Z) fill serves all latents)
  c1  configs[0]: N=1000, M=32, D=1 plumbing case
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "ns": dict(N=1_000_000, M=1024, D=8, P=1, lik="gaussian", dtype="f64",
               name="gaussian_regression_N1e6_M1024_D8_P1 (north_star target; the metric's N, M)"),
    "c2": dict(N=1_000_000, M=512, D=8, P=1, lik="gaussian", dtype="f64",
               name="gaussian_regression_N1e6_M512_D8_P1 (BASELINE configs[1])"),
    "c3": dict(N=1_000_000, M=1024, D=16, P=1, lik="bernoulli", dtype="f32",
               name="bernoulli_probit_N1e6_M1024_D16_P1 (BASELINE configs[2])"),
    "c5": dict(N=1_000_000, M=1024, D=8, P=8, lik="gaussian", dtype="f64", separate=True,
               name="gaussian_multioutput_N1e6_M1024_D8_P8_separate_kernels (BASELINE configs[4])"),
    "c5s": dict(N=1_000_000, M=1024, D=8, P=8, lik="gaussian", dtype="f64",
                name="gaussian_multioutput_N1e6_M1024_D8_P8_shared_kernel (special case of BASELINE configs[4])"),
    "c1": dict(N=1000, M=32, D=1, P=1, lik="gaussian", dtype="f64", name="gaussian_1d_N1000_M32 (BASELINE configs[0])",
               lengthscales=0.1, variance=0.3, noise=1.0),
}
PEAK_TFLOPS = {"f64": 78.6, "f32": 157.3}  # dense MFMA peaks (AMD datasheet; f32: MI355X_MICROARCH.md; f64 = f32/2)
HBM_PEAK_GBS = 8000.0


def make_data(w, seed=0):
    """X = randn(N, D); w = randn(D, P); eps = randn(N, P); f = sin(X w); Z = X[:M]."""
    rng = np.random.RandomState(seed)
    if w["D"] == 1 and w["M"] == 32:  # config C1: 1-D, X in [-1, 1], Y = sin(15 X) + eps, Z on a grid
        X = rng.rand(w["N"], 1) * 2 - 1
        Y = np.sin(15 * X) + rng.randn(w["N"], 1)
        return X, Y, np.linspace(X.min(), X.max(), w["M"])[:, None]
    X = rng.randn(w["N"], w["D"])
    W = rng.randn(w["D"], w["P"])
    eps = rng.randn(w["N"], w["P"])
    f = np.sin(X @ W)
    if w["lik"] == "gaussian":
        Y = f + np.sqrt(0.1) * eps
    else:
        Y = (f + np.sqrt(0.1) * eps > 0).astype(np.float64)
    return X, Y, X[: w["M"]].copy()


def kernel_flops(w, rows, batched=True, trmm_batch=1):
    """Algorithmic flops per launch of each MFMA kernel for `rows` data rows (triangular/symmetric counts, 2 per FMA)."""
    M, P = w["M"], (1 if (w.get("separate") and not batched) else w["P"])  # per-latent path: one launch per latent
    return {
        "tsvgp_trmm": rows * M * (M + 1) * trmm_batch,  # B = Kfu U^-T, triangular k-range (x latents per launch)
        "tsvgp_moments": rows * M * (M + 1) * P + 2 * rows * M * P,  # |F^T b|^2 (upper) + mean GEMV
        "tsvgp_site_accum": rows * M * (M + 1) * P + 2 * rows * M * P,  # lower half of sum g1 b b^T + sum g0 b
    }


def kernel_bytes(w, rows, esize, batched=False):
    """Algorithmic HBM bytes per launch of the fill kernel (the only HBM-bound kernel): write Kfu once, read X once."""
    P = w["P"] if (w.get("separate") and batched) else 1  # the batched fill writes every latent's K(X, Z) in one launch
    return {"tsvgp_se_fill": P * rows * w["M"] * esize + rows * w["D"] * esize}


def make_kernel(mod, w):
    """The workload's kernel (and inducing variable wrapper) in module `mod` (the package or the oracle)."""
    var = w.get("variance", 1.0)
    if w.get("separate"):  # l_p = linspace(0.8, 1.5, P)
        return mod.SeparateIndependent([mod.SquaredExponential(variance=var, lengthscales=float(l))
                                        for l in np.linspace(0.8, 1.5, w["P"])]), mod.SharedIndependentInducingVariables
    return mod.SquaredExponential(variance=var, lengthscales=w.get("lengthscales", 1.0)), (lambda Z: Z)


def _blas_info():
    """(threads, backend description) of the BLAS NumPy/SciPy run on."""
    try:
        from threadpoolctl import threadpool_info

        pools = [p for p in threadpool_info() if p.get("user_api") == "blas"] or threadpool_info()
        cores = max([p.get("num_threads", 1) for p in pools] or [os.cpu_count() or 1])
        desc = ", ".join(sorted({f"{p.get('internal_api', '?')} {p.get('version', '')}".strip() for p in pools})) or "unknown"
        return int(cores), desc
    except Exception:
        return int(os.cpu_count() or 1), "unknown"


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def _dgemm_gflops(n=4096, reps=3):
    """A plain NumPy DGEMM of this box on the BLAS pool the oracle runs on: the context for the port's own GFLOP/s."""
    a = np.random.RandomState(0).randn(n, n)
    a @ a  # pool ramp, page faults
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        a @ a
        best = min(best, time.perf_counter() - t0)
    return round(2.0 * n ** 3 / best / 1e9, 1)


class blas_threads:
    """The BLAS pools NumPy / SciPy run on, widened to every host core for the oracle's sections: a rank started by
    torch.distributed.run inherits OMP_NUM_THREADS=1 (its default for more than one process per node), which would time the
    CPU baseline on one core and stretch the oracle's ELBO over all N rows from 40 s to many minutes."""

    def __enter__(self):
        self._ctx = None
        try:
            from threadpoolctl import threadpool_limits

            self._ctx = threadpool_limits(limits=os.cpu_count() or 1, user_api="blas")
        except Exception:
            pass
        return self

    def __exit__(self, *exc):
        if self._ctx is not None:
            self._ctx.restore_original_limits()


def elbo_vs_1gpu(args, w, elbo, taken):
    """The ELBO after this run's steps against the committed ONE-GPU value of the same workload (profiles/elbo_1gpu.json,
    written by tools/make_elbo_1gpu.py from 1-GPU bench lines measured on MI355X): the multi-rank half of "ELBO match".
    Sharding only changes the order of the row sums, so the two agree to rounding (the 1-GPU tolerances)
    once both runs have taken the same number of steps -- or enough of them to sit at the fixed point."""
    path = os.path.join(ROOT, "profiles", "elbo_1gpu.json")
    key = args.workload if not args.rows else f"{args.workload}@{args.rows}"
    try:
        ref = json.load(open(path)).get("white" if args.model == "white" else "tsvgp", {}).get(key)
    except Exception:
        ref = None
    if not ref:
        return None
    return {"one_gpu": ref["elbo_after_steps"], "one_gpu_steps": ref["steps_taken"], "steps_taken": taken,
            "rel": abs(elbo - ref["elbo_after_steps"]) / abs(ref["elbo_after_steps"]),
            "same_steps": taken == ref["steps_taken"], "source": ref.get("source")}


def oracle_step_flops(w, n):
    """Flops the reference's op sequence executes per E-step on n rows: (4 + 4P) n M^2 + ~37 M^3 P."""
    M, P = w["M"], w["P"]
    return (4 + 4 * P) * n * M * M + 37.0 * M ** 3 * P


def cpu_baseline(w, budget_s=20.0, model_kind="tsvgp"):
    """The CPU oracle (op-for-op port of the reference sequence) timed on bounded row samples of the same workload.

    A step costs t(n) = a + b n: the M x M part (three posterior factorisations, five more Choleskys: ~37 M^3 flops)
    does not grow with the rows, the rest is linear in them.  Two sample sizes (N/50 and N/20, capped so that the
    [n, M, P] temporaries of the reference sequence stay below ~1 GB each) give a and b; the full-size figure is
    1 / (a + b N), labelled extrapolated.  `budget_s` bounds the timed work per sample size."""
    from oracle import tsvgp_oracle as O

    cores, blas = _blas_info()
    cap = max(2000, int(4.0e7 // (w["M"] * w["P"])))  # [n, M, P] fp64 temporaries <= ~0.3 GB: a step stays within seconds
    n1 = min(w["N"], max(2000, min(cap // 2, w["N"] // 50)))
    n2 = min(w["N"], max(2 * n1, min(cap, w["N"] // 20)))
    lik = O.Gaussian(variance=w.get("noise", 0.1)) if w["lik"] == "gaussian" else O.Bernoulli()
    kernel, wrap = make_kernel(O, w)
    cls = O.t_SVGP_white if model_kind == "white" else O.t_SVGP
    Xa, Ya, Z = make_data(dict(w, N=max(n1, n2)))
    med = {}
    for n_s in sorted({n1, n2}):
        X, Y = Xa[:n_s], Ya[:n_s]
        model = cls(kernel, lik, wrap(Z), num_latent_gps=w["P"])
        print(f"[cpu_baseline] oracle E-steps on {n_s} rows ...", file=sys.stderr, flush=True)
        model.natgrad_step((X, Y), lr=0.8)  # warm-up (BLAS thread pools, page faults)
        times = []
        t_all = time.perf_counter()
        while len(times) < 2 or (time.perf_counter() - t_all < budget_s and len(times) < 50):
            t0 = time.perf_counter()
            model.natgrad_step((X, Y), lr=0.8)
            times.append(time.perf_counter() - t0)
            print(f"[cpu_baseline]   step {len(times)}: {times[-1]:.2f} s", file=sys.stderr, flush=True)
            if time.perf_counter() - t_all > 2 * budget_s:
                break
        med[n_s] = (float(np.median(times)), len(times))
    if len(med) == 2:
        (na, (ta, ka)), (nb, (tb, kb)) = sorted(med.items())
        b = max((tb - ta) / (nb - na), 0.0)
        a = max(ta - b * na, 0.0)
        fit = f"t(n) = a + b n from n = {na} ({ta * 1e3:.0f} ms, median of {ka}) and n = {nb} ({tb * 1e3:.0f} ms, median of {kb}): " \
              f"a = {a * 1e3:.0f} ms (M x M part), b = {b * 1e6:.2f} us/row"
        n_big, t_big = nb, tb
    else:  # the whole problem fits one sample (configs[0]): measured, not extrapolated
        (n_big, (t_big, kb)), = med.items()
        a, b = t_big, 0.0
        fit = f"measured at the full n = {n_big} (median of {kb})"
    t_full = a + b * w["N"] if len(med) == 2 else t_big
    return {
        "value": 1.0 / t_full, "unit": "E-steps/s", "cores": cores, "threads": cores, "kind": "port",
        "blas": blas, "cpu_model": _cpu_model(), "host_logical_cpus": os.cpu_count(), "sample_gflops": round(oracle_step_flops(w, n_big) / t_big / 1e9, 1),
        "extrapolated_ms_per_step": round(t_full * 1e3, 1),
        "sample": f"oracle natgrad_step (NumPy/SciPy fp64, the reference's op sequence incl. its redundancies) on the first rows of "
                  f"the same workload (same M, D, P, likelihood); {fit}; "
                  + (f"extrapolated to N = {w['N']} as a + b N" if len(med) == 2 else "no extrapolation"),
    }


def elbo_match(model, w, X, Y, Z, budget_s, elbo_all_ranks=None):
    """The "ELBO match" half of the metric: the oracle's ELBO (row-blocked ``conditional`` + ``variational_expectations``,
    oracle.elbo_chunked) evaluated on the HIP model's state, against the HIP model's own ELBO of the same rows.
    Full N when the first block's timing says it fits `budget_s`, else a prefix (the HIP side is recomputed on it).
    With several ranks (`elbo_all_ranks`: the all-reduced ELBO of the sharded rows, every rank's kernels + RCCL) this runs on
    rank 0 alone: the state is replicated, the oracle takes all N rows on the host, and whatever the HIP side recomputes
    here (a row prefix, the row sample) runs on rank 0's GPU without collectives."""
    import torch
    from oracle import tsvgp_oracle as O

    lik = O.Gaussian(variance=w.get("noise", 0.1)) if w["lik"] == "gaussian" else O.Bernoulli()
    kernel, wrap = make_kernel(O, w)
    ora = O.t_SVGP(kernel, lik, wrap(Z), num_latent_gps=w["P"], num_data=w["N"],
                   lambda_1=model.lambda_1.numpy(), lambda_2_sqrt=model.lambda_2_sqrt.numpy())
    chunk = max(1000, int(2.0e7 // (w["M"] * w["P"])))
    t0 = time.perf_counter()
    O.elbo_chunked(ora, (X[:chunk], Y[:chunk]), chunk_rows=chunk)  # M x M part + one block: the cost model
    t_blk = time.perf_counter() - t0
    rows = w["N"] if t_blk * (w["N"] / chunk) <= budget_s else max(chunk, int(budget_s / t_blk) * chunk)
    rows = min(rows, w["N"])

    print(f"[elbo_match] oracle ELBO on {rows} of {w['N']} rows (first block: {t_blk:.1f} s) ...", file=sys.stderr, flush=True)

    def tick(done, total, last=[time.perf_counter()]):
        if time.perf_counter() - last[0] > 20.0:
            last[0] = time.perf_counter()
            print(f"[elbo_match] oracle: {done}/{total} rows", file=sys.stderr, flush=True)

    t0 = time.perf_counter()
    e_o = float(O.elbo_chunked(ora, (X[:rows], Y[:rows]), chunk_rows=chunk, progress=tick))
    t_o = time.perf_counter() - t0
    dev, dt = model.device, model.compute_dtype
    dp_saved, model.data_parallel = model.data_parallel, False  # rank 0 alone from here on: no collective
    if elbo_all_ranks is not None and rows == w["N"]:
        e_h = float(elbo_all_ranks)
    else:
        model._get_engine().release()
        Xd = torch.as_tensor(X[:rows], dtype=dt).to(dev)
        Yd = torch.as_tensor(Y[:rows], dtype=dt).to(dev)
        e_h = float(model.elbo((Xd, Yd)))
        del Xd, Yd
    # intermediates of tsvgp.py:246-263 on a row sample spread over the whole range
    idx = np.arange(0, w["N"], max(1, w["N"] // 20000))[:20000]
    Xs, Ys = X[idx], Y[idx]
    mu_o, var_o = O.predict_f_chunked(ora, Xs, chunk_rows=chunk)
    g0_o, g1_o = lik.variational_expectations_grads(mu_o, var_o, Ys)
    g1_o = np.minimum(g1_o, -1e-8)
    got = model.moments_and_gradients((torch.as_tensor(Xs, dtype=dt).to(dev), torch.as_tensor(Ys, dtype=dt).to(dev)))
    rel = lambda a, b: float(np.max(np.abs(a.cpu().numpy() - b)) / np.max(np.abs(b)))
    model.data_parallel = dp_saved
    return {"hip": e_h, "hip_is": "all-reduced ELBO of the sharded rows (every rank)" if (elbo_all_ranks is not None and rows == w["N"])
            else "one GPU", "oracle": e_o, "rel": abs(e_h - e_o) / abs(e_o), "rows": int(rows), "full_N": bool(rows == w["N"]),
            "oracle_seconds": round(t_o, 1),
            "sample_rows": int(len(idx)),
            "sample_max_rel_err": {k: rel(g, o) for k, g, o in zip(("mean", "var", "g0", "g1"), got, (mu_o, var_o, g0_o, g1_o))},
            "note": "oracle.elbo_chunked / predict_f_chunked (GPflow conditional + variational_expectations restated, "
                    "reference src/models/tsvgp.py:79-114) evaluated on the HIP model's (lambda_1, lambda_2_sqrt) after the "
                    "timed steps; rel = |hip - oracle| / |oracle|; tolerance stated in SURVEY 8(d): 1e-9 (fp64), 1e-4 (fp32)"}


def state_match(model, w, X, Y, Z, Xd, Yd, budget_s):
    """ONE oracle E-step over all N rows, from the HIP model's state (closes the gap "the oracle never takes a step at the
    metric's size"): ``oracle.natgrad_step_chunked`` -- the reference's op sequence (src/models/tsvgp.py:234-304) per row block,
    block sums of G0 / G1 added with compensation -- against the HIP model's next step from the same state:
      * before the step: ELBO (the `elbo_match` half of the metric), mean / var / g0 / g1 on ALL rows;
      * the step: G0, G1 (tsvgp.py:279-280) and the new (lambda_1, Lambda_2).
    The oracle's wall time for that step IS the measured CPU baseline (no extrapolation).  Returns None when the first block's
    timing projects beyond `budget_s` (a box with few host cores): the caller then falls back to `elbo_match` alone."""
    import torch
    from oracle import tsvgp_oracle as O

    lik = O.Gaussian(variance=w.get("noise", 0.1)) if w["lik"] == "gaussian" else O.Bernoulli()
    kernel, wrap = make_kernel(O, w)
    mk = lambda: O.t_SVGP(kernel, lik, wrap(Z), num_latent_gps=w["P"], num_data=w["N"],
                          lambda_1=model.lambda_1.numpy(), lambda_2_sqrt=model.lambda_2_sqrt.numpy())
    chunk = max(1000, int(2.0e7 // (w["M"] * w["P"])))
    N = w["N"]
    # the cost model: a step over TWO blocks; the first block pays for page faults and the BLAS pool's ramp (5.7 s against 3.8 s
    # in steady state on a 64-core box), so the projection is (M x M part + first block) + (blocks - 1) x the SECOND block's time
    stamps = [time.perf_counter()]
    O.natgrad_step_chunked(mk(), (X[:2 * chunk], Y[:2 * chunk]), lr=0.8, chunk_rows=chunk,
                           progress=lambda done, total: stamps.append(time.perf_counter()))
    t_two = time.perf_counter() - stamps[0]
    t_blk = (stamps[2] - stamps[1]) if len(stamps) >= 3 else t_two
    nblk = max(1.0, N / chunk)
    projected = (t_two - t_blk) + (nblk - 1.0) * t_blk if len(stamps) >= 3 else t_two * nblk
    if projected > budget_s:
        print(f"[state_match] skipped: a {chunk}-row block takes {t_blk:.1f} s in steady state -> {projected:.0f} s for {N} rows "
              f"(budget {budget_s:.0f} s)", file=sys.stderr, flush=True)
        return None
    print(f"[state_match] oracle E-step over all {N} rows in blocks of {chunk} ({t_blk:.1f} s per block in steady state, projected "
          f"{projected:.0f} s) ...", file=sys.stderr, flush=True)

    def tick(done, total, last=[time.perf_counter()]):
        if time.perf_counter() - last[0] > 20.0:
            last[0] = time.perf_counter()
            print(f"[state_match] oracle: {done}/{total} rows", file=sys.stderr, flush=True)

    ora = mk()
    t0 = time.perf_counter()
    O.natgrad_step_chunked(ora, (X, Y), lr=0.8, chunk_rows=chunk, progress=tick)
    t_o = time.perf_counter() - t0
    last = ora.last
    # the HIP side, from the same state: ELBO and intermediates before the step, the site sums, then the step itself
    rel = lambda a, b: float(np.max(np.abs(np.asarray(a) - b)) / np.max(np.abs(b)))
    e_h = float(model.elbo((Xd, Yd)))
    mean, var, g0, g1 = (t.cpu().numpy() for t in model.moments_and_gradients((Xd, Yd)))
    inter = {"mean": rel(mean, last["mean"]), "var": rel(var, last["var"]), "g0": rel(g0, last["g0"]), "g1": rel(g1, last["g1"])}
    del mean, var, g0, g1
    G0, G1 = (t.cpu().numpy() for t in model.site_sums((Xd, Yd)))
    # G0 = A^T (y - mean) / s2 is a small difference of N-sized terms once the sites fit the data: its relative error is set by
    # the error of g0 times sum_n |a_n| (oracle: A_abs_colsum), on either side; reported raw, in units of the tolerance its
    # inputs are stated to (1e-8 max|g0| sum_n |a_n| in fp64), and as the natural gradient it enters, G0 - 2 G1 meanZ
    tol_in = (1e-8 if w["dtype"] == "f64" else 1e-3) * np.max(np.abs(last["g0"])) * last["A_abs_colsum"]
    nat = lambda g0_, g1_: g0_ - 2.0 * np.einsum("lmo,ol->ml", g1_, last["meanZ"])
    sums = {"G0": rel(G0, last["G0"]), "G1": rel(G1, last["G1"]),
            "G0_in_units_of_its_input_tolerance": float(np.max(np.abs(G0 - last["G0"]) / tol_in)),
            "G0_minus_2_G1_meanZ": rel(nat(G0, G1), nat(last["G0"], last["G1"]))}
    del G0, G1
    model.natgrad_step((Xd, Yd), lr=0.8)
    state = {"lambda_1": rel(model.lambda_1.numpy(), ora.lambda_1), "Lambda_2": rel(model.lambda_2.cpu().numpy(), ora.lambda_2)}
    e_o = float(last["elbo_before"])
    cores, blas = _blas_info()
    t_extras = float(last.get("extras_seconds", 0.0))  # ELBO terms, |A| column sums, retained N-sized arrays: not the step's work
    t_step = t_o - t_extras
    return {
        "elbo_match": {"hip": e_h, "hip_is": "one GPU", "oracle": e_o, "rel": abs(e_h - e_o) / abs(e_o), "rows": int(N), "full_N": True,
                       "oracle_seconds": round(t_o, 1), "sample_rows": int(N), "sample_max_rel_err": inter,
                       "note": "from the oracle's row-blocked E-step (state_match): ELBO and mean / var / g0 / g1 of ALL rows at the "
                               "state the compared step starts from; rel = |hip - oracle| / |oracle|; tolerances of SURVEY 8(d): "
                               "1e-9 / 1e-8 (fp64), 1e-4 / (atol 1e-4 + rtol 1e-3) (fp32)"},
        "state_match": {"rows": int(N), "chunk_rows": int(chunk), "oracle_seconds": round(t_o, 1), "lr": 0.8,
                        "site_sums_max_rel_err": sums, "state_after_step_max_rel_err": state,
                        "note": "oracle.natgrad_step_chunked (reference src/models/tsvgp.py:234-304, row-blocked) and the HIP "
                                "natgrad_step, both from the HIP model's state after the timed steps, over all N rows: G0, G1 "
                                "(tsvgp.py:279-280) and the updated (lambda_1, Lambda_2 = L L^T)"},
        "cpu_measured": {"value": 1.0 / t_step, "unit": "E-steps/s", "cores": cores, "threads": cores, "kind": "port", "measured": True,
                         "blas": blas, "cpu_model": _cpu_model(), "host_logical_cpus": os.cpu_count(),
                         "seconds_per_step": round(t_step, 1), "seconds_incl_parity_extras": round(t_o, 1),
                         "parity_extras_seconds": round(t_extras, 1),
                         "gflops": round(oracle_step_flops(w, N) / t_step / 1e9, 1), "dgemm_gflops_same_box": _dgemm_gflops(),
                         "sample": f"ONE oracle natgrad_step over ALL N = {N} rows of the workload (NumPy/SciPy fp64, the reference's "
                                   f"op sequence incl. its redundancies, row-blocked by {chunk}; M x M parts once), wall time "
                                   f"measured on this box's host cores -- not extrapolated; the seconds the same pass spends on what "
                                   f"the reference's step does not do (ELBO terms, |A| column sums, retained N-sized arrays for the "
                                   f"parity checks) are timed apart and taken out",
                         "caveat": "a NumPy/SciPy PORT of the reference's op sequence: `gflops` against `dgemm_gflops_same_box` "
                                   "(a plain 4096^3 DGEMM on the same BLAS pool) says how far it is from what these cores deliver; "
                                   "the reference's TensorFlow/Eigen path would likely be several times faster.  A reported "
                                   "baseline, not the target"},
    }


def elbo_match_white(model, w, X, Y, Z, rows=20000):
    """ELBO match for `--model white`: the oracle's t_SVGP_white (reference src/models/tsvgp_white.py:23-246,
    src/util.py:11-88) on the HIP model's (lambda_1, lambda_2), ELBO and predictive moments of a row prefix through both (the
    minibatch scale num_data / rows applies on both sides)."""
    import torch
    from oracle import tsvgp_oracle as O

    lik = O.Gaussian(variance=w.get("noise", 0.1)) if w["lik"] == "gaussian" else O.Bernoulli()
    kernel, wrap = make_kernel(O, w)
    ora = O.t_SVGP_white(kernel, lik, wrap(Z), num_latent_gps=w["P"], num_data=w["N"],
                         lambda_1=model.lambda_1.numpy(), lambda_2=model.lambda_2.numpy())
    rows = min(rows, w["N"])
    t0 = time.perf_counter()
    e_o = float(ora.elbo((X[:rows], Y[:rows])))
    mu_o, var_o = ora.predict_f(X[:rows])
    t_o = time.perf_counter() - t0
    dev, dt = model.device, model.compute_dtype
    dp_saved, model.data_parallel = model.data_parallel, False  # rank 0 alone: no collective
    model._get_engine().release()
    Xd, Yd = torch.as_tensor(X[:rows], dtype=dt).to(dev), torch.as_tensor(Y[:rows], dtype=dt).to(dev)
    e_h = float(model.elbo((Xd, Yd)))
    mu_h, var_h = model.predict_f(Xd)
    model.data_parallel = dp_saved
    rel = lambda a, b: float(np.max(np.abs(np.asarray(a.cpu() if hasattr(a, "cpu") else a) - b)) / np.max(np.abs(b)))
    return {"hip": e_h, "hip_is": "one GPU", "oracle": e_o, "rel": abs(e_h - e_o) / abs(e_o), "rows": int(rows),
            "full_N": bool(rows == w["N"]), "oracle_seconds": round(t_o, 1), "sample_rows": int(rows),
            "sample_max_rel_err": {"mean": rel(mu_h, mu_o), "var": rel(var_h, var_o)},
            "note": "oracle t_SVGP_white.elbo / predict_f (reference src/models/tsvgp_white.py, src/util.py:11-88 restated) on the "
                    "HIP model's (lambda_1, lambda_2) after the timed steps, on a row prefix; rel = |hip - oracle| / |oracle|"}


def self_launch(n_ranks: int) -> int:
    """`python bench.py --gpus N` without a launcher: runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a child and returns its exit code."""
    import socket
    import subprocess

    with socket.socket() as s:  # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this pool
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n_ranks)))
    print(f"[bench] launching {n_ranks} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="ns", choices=sorted(WORKLOADS))
    ap.add_argument("--model", default="tsvgp", choices=["tsvgp", "white"],
                    help="white: t_SVGP_white (reference src/models/tsvgp_white.py) on the same workload; not the metric")
    ap.add_argument("--rows", type=int, default=None, help="override N (debugging only; the result is then not the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of timed oracle steps per sample size")
    ap.add_argument("--no-elbo-match", action="store_true")
    ap.add_argument("--elbo-budget", type=float, default=150.0,
                    help="seconds the oracle's ELBO evaluation may take; beyond it a row prefix is compared instead of all N")
    ap.add_argument("--no-side-lines", action="store_true", help="skip warm / forced-route / mean-only side measurements")
    ap.add_argument("--no-state-match", action="store_true",
                    help="skip the oracle's full-N E-step (state match + measured CPU baseline); elbo_match alone then runs")
    ap.add_argument("--loop", default=None, choices=["em"],
                    help="em: add the `em_loop` side line -- ONE iteration of the reference's training loop (8 E-steps + 20 Adam "
                         "M-steps, experiments/uci_regression.py:132-160) timed with its E / M split; never `value`")
    ap.add_argument("--state-budget", type=float, default=250.0,
                    help="seconds the oracle's full-N E-step may take (projected from its first row block); beyond it the step "
                         "is skipped and elbo_match / the extrapolated cpu_baseline stand alone")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # A plain `python bench.py --gpus N`: start one rank per GPU as a FRESH child process (torch.distributed.run), relay
        # its output (rank 0 prints the one JSON line) and exit with its code.  Nothing in this process has touched the GPU
        # yet (torch is not even imported), and it never replaces itself with another program.
        sys.exit(self_launch(args.gpus))

    import torch
    import torch.distributed as dist

    pkg = importlib.import_module("t-svgp_amd")
    w = dict(WORKLOADS[args.workload])
    if args.rows:
        w["N"] = args.rows
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
    if os.environ.get("TSVGP_BENCH_BACKEND", "nccl") != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)  # rehearsal: ranks may share a GPU
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("TSVGP_BENCH_BACKEND", "nccl")  # "gloo" only to rehearse the multi-rank path on one GPU
        # a finite rendezvous / collective timeout: a rank whose peers never arrive raises and exits non-zero (the launcher then
        # ends the job; the next attempt is a fresh process, never a re-exec of this one)
        import datetime

        pg_timeout = datetime.timedelta(seconds=int(os.environ.get("TSVGP_BENCH_PG_TIMEOUT", "600")))
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=device, timeout=pg_timeout)
            else:
                dist.init_process_group(backend, timeout=pg_timeout)
            dist.barrier()
        except Exception as exc:  # rendezvous or first collective failed / timed out
            print(f"[bench] rank {rank}: process group ({backend}, world {world}) did not come up within "
                  f"{pg_timeout.total_seconds():.0f} s: {exc}", file=sys.stderr, flush=True)
            sys.exit(3)

    collective = None
    if world > 1:
        # what the process group itself reports -- not what this script was asked for: the multi-GPU line proves by itself that
        # `world` ranks on `world` devices took part and which backend moved the bytes
        props = torch.cuda.get_device_properties(device)
        me = {"rank": dist.get_rank(), "local_rank": local_rank, "device_index": device.index, "device_name": props.name,
              "pci_bus_id": getattr(props, "pci_bus_id", None), "uuid": str(getattr(props, "uuid", "")) or None, "pid": os.getpid()}
        everyone = [None] * world
        dist.all_gather_object(everyone, me)
        try:
            rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            rccl = None
        collective = {"backend": dist.get_backend(), "world_size_seen": dist.get_world_size(), "device_per_rank": everyone,
                      "distinct_devices": len({(e["device_index"], e["pci_bus_id"], e["uuid"]) for e in everyone}),
                      "rccl_version": rccl, "hip_version": getattr(torch.version, "hip", None),
                      "pg_timeout_s": int(os.environ.get("TSVGP_BENCH_PG_TIMEOUT", "600"))}

    dtype = torch.float64 if w["dtype"] == "f64" else torch.float32
    esize = 8 if w["dtype"] == "f64" else 4
    X, Y, Z = make_data(w)
    lo, hi = pkg.distributed.shard_bounds(w["N"], world, rank)
    Xd = torch.as_tensor(X[lo:hi], dtype=dtype).to(device).contiguous()
    Yd = torch.as_tensor(Y[lo:hi], dtype=dtype).to(device).contiguous()
    rows = hi - lo
    if rank != 0:
        del X, Y  # rank 0 keeps the host copy: the oracle side of `elbo_match` runs there on all N rows

    lik = pkg.Gaussian(variance=w.get("noise", 0.1)) if w["lik"] == "gaussian" else pkg.Bernoulli()
    kernel, wrap = make_kernel(pkg, w)
    if args.model == "white":
        model = pkg.t_SVGP_white(kernel, lik, wrap(Z), num_latent_gps=w["P"], num_data=w["N"], compute_dtype=dtype, device=device)
        w["name"] += " [t_SVGP_white]"
    else:
        model = pkg.t_SVGP(kernel, lik, wrap(Z), num_latent_gps=w["P"], num_data=w["N"], compute_dtype=dtype, device=device)
    eng = model._get_engine()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    import gc

    class timed_region:
        """The cyclic garbage collector is paused inside a timed loop (as ``timeit`` does): a full collection of this
        process (torch, numpy and scipy loaded: ~1e6 tracked objects) takes 30-40 ms and, when one happened to fall into
        the 20-30 steps of a timed region, showed as +1.2 ms per step (measured: one 39 ms `_site_operands` call in 30,
        38.5 instead of 37.0 ms per step at N = 1e6, 12.6 instead of 11.3 ms at M = 512, kernel times unchanged)."""

        def __init__(self, collected=False):
            self.collected = collected  # the caller ran gc.collect() in front of its warm-up steps (see the headline region)

        def __enter__(self):
            if not self.collected:
                gc.collect()
            gc.disable()

        def __exit__(self, *exc):
            gc.enable()

    # Per-kernel HIP events are recorded in the timed region, except where the model's default replays the step from a
    # captured hipGraph (launch-bound sizes, configs[0]): events inside a replay would force the eager path.
    replayed = args.model == "tsvgp" and model._wants_graph(Xd) and not w.get("separate")
    if world > 1:
        # With several ranks a replayed step is two graphs around the all-reduce (t_SVGP._graph_step).  The mode changes the number
        # of steps this script takes (the eager pass for the per-kernel times), so the ranks AGREE on it: shards differ by a row and
        # "auto" could fall on either side of its size limit.
        flag = torch.tensor([1 if replayed else 0], dtype=torch.int32,
                            device=device if os.environ.get("TSVGP_BENCH_BACKEND", "nccl") == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        replayed = bool(int(flag.item()))
    # The projection route ("auto": by cond(K_uu + jitter I)) is decided once per change of (theta, Z, jitter) and cached,
    # so the timed steps below do not pay for it; its cost is measured here and reported as `route_gate_ms`.
    routes, conds, route_gate_ms = None, None, None
    if args.model == "tsvgp":
        model._routes(1e-9)  # library handles, first-call allocations
        model._cond_cache = None
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        routes = model._routes(1e-9)
        torch.cuda.synchronize(device)
        route_gate_ms = (time.perf_counter() - t0) * 1e3
        conds = [float(c) for c in model._cond_cache[1]] if model._cond_cache is not None else None
    n_warm = max(args.warmup, 3 if replayed else 0)  # a graph is captured on the second occurrence of a step
    # Housekeeping that takes the host tens of milliseconds -- the full garbage collection of `timed_region`, the pool of timing
    # events -- happens IN FRONT of the warm-up steps, not between them and the timed steps: the chip's clock follows its recent load
    # (profiles/r05_clock_lab.txt), and a 40 ms pause there made the first timed step 2.3 ms slower than the others.  The timed
    # region itself is unchanged: W untimed steps, barrier + synchronize, K steps, barrier + synchronize.
    gc.collect()
    if not replayed and hasattr(eng, "reserve_events"):
        eng.reserve_events(2 * 8 * args.steps * max(1, w["P"] if w.get("separate") else 1) + 64)
    if world > 1:
        pkg.distributed.reserve_timing(8 * args.steps + 16)
    for _ in range(n_warm):
        model.natgrad_step((Xd, Yd), lr=0.8)
    barrier()
    # Every C-ABI launch inside the timed region is bracketed with HIP events on its launch stream (pooled: no event is
    # created inside the region); they cost 0.05-0.1 ms per step (A/B with the collector paused: 11.38 vs 11.44 ms at
    # M = 512, 37.14 vs 37.14-37.4 ms at the headline size).
    eng.profile = None if replayed else {}
    D_ = pkg.distributed
    if world > 1:
        D_.TIMING = []  # HIP events around every collective of the timed steps (pooled; replayed steps take them too: the
        #                 all-reduce sits BETWEEN the two graphs of a step)
    with timed_region(collected=True):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            model.natgrad_step((Xd, Yd), lr=0.8)
        barrier()
        elapsed = time.perf_counter() - t0
    coll_times = None
    if world > 1:
        coll_times, D_.TIMING = D_.timing_summary(D_.TIMING, args.steps), None
    prof = {} if replayed else eng.profile_summary()
    eng.profile = None
    if replayed:
        # the headline ran from the captured graph (no events inside a replay): the per-kernel times and the roofline entry come
        # from an EAGER pass of the same steps taken right behind it (labelled in `kernel_timing`)
        model.use_graph = False
        gc.collect()  # (in front of the warm-up steps, as for the headline region)
        for _ in range(2):
            model.natgrad_step((Xd, Yd), lr=0.8)
        barrier()
        eng.profile = {}
        with timed_region(collected=True):
            for _ in range(args.steps):
                model.natgrad_step((Xd, Yd), lr=0.8)
            barrier()
        prof = eng.profile_summary()
        eng.profile = None
        model.use_graph = "auto"

    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t)
    elbo = float(model.elbo((Xd, Yd)))
    steps_before_elbo = n_warm + args.steps + ((2 + args.steps) if replayed else 0)

    def timed_steps(n_warm=2):
        """(seconds for args.steps steps, max over ranks) in the model's current mode, after n_warm untimed steps."""
        gc.collect()
        for _ in range(n_warm):
            model.natgrad_step((Xd, Yd), lr=0.8)
        barrier()
        with timed_region(collected=True):
            t0 = time.perf_counter()
            for _ in range(args.steps):
                model.natgrad_step((Xd, Yd), lr=0.8)
            barrier()
            tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt)

    # The other projection routes on the same workload, reported beside the headline and never as `value`: "auto" picks
    # the cheapest route the conditioning of K_uu allows (the `ns` geometry: direct); inducing points trained closer
    # together take the whitened (cond > DIRECT_MAX_COND) or projected (cond > WHITENED_MAX_COND) route, which cost
    # one / two more N M^2 products.
    forced = None
    if args.model == "tsvgp" and not args.no_side_lines and not w.get("separate") and w["N"] * w["M"] > 10_000_000:
        forced = {}
        for r in ("direct", "whitened", "projected"):
            if routes is not None and routes[0] == r:
                continue
            model.projection = r
            tr = timed_steps()
            forced[r] = {"value": round(args.steps / tr, 4), "unit": "E-steps/s", "ms_per_step": round(tr / args.steps * 1e3, 4)}
        model.projection = "auto"
        model._get_engine().release()  # the whitened operand (another N x M buffer) is not needed below

    # "warm" E-steps, reported beside the headline and never as `value`: consecutive E-steps with unchanged
    # hyperparameters (the reference's E/M loop runs 8 per M-step) reuse chol(K_uu), its inverse and the whitened B.
    model.cache_whitened = True
    warm_elapsed = timed_steps()
    model.cache_whitened = False

    # launch-bound sizes (configs[0]): t_SVGP(use_graph="auto"), the default, replays the step from a captured hipGraph
    # there, so `value` above is the replayed rate; the other mode (eager) is reported beside it.  At the metric's
    # sizes "auto" runs eagerly and this block is skipped.
    graph_line = None
    if world > 1 and replayed:
        graph_line = {"headline_mode": "hipGraph replay (two graphs around the all-reduce of the packed accumulators)",
                      "captured": any(isinstance(e, dict) for e in model._graphs.values()),
                      "note": "\"auto\" (default) replays where this rank's rows * M <= 2e8; the eager mode is timed beside it on one GPU only"}
    if world == 1 and args.model == "tsvgp" and w["N"] * w["M"] <= 200_000_000 and not w.get("separate"):
        auto_on = model._wants_graph(Xd)
        model.use_graph = not auto_on
        for _ in range(4):
            model.natgrad_step((Xd, Yd), lr=0.8)
        barrier()
        with timed_region():
            t0 = time.perf_counter()
            for _ in range(args.steps):
                model.natgrad_step((Xd, Yd), lr=0.8)
            barrier()
            tg = time.perf_counter() - t0
        model.use_graph = "auto"
        graph_line = {"mode": "eager (use_graph=False)" if auto_on else "hipGraph replay (use_graph=True)",
                      "headline_mode": "hipGraph replay" if auto_on else "eager",
                      "value": round(args.steps / tg, 4), "unit": "E-steps/s", "ms_per_step": round(tg / args.steps * 1e3, 4),
                      "captured": any(isinstance(e, dict) for e in model._graphs.values()),
                      "note": "use_graph: the whole step (about 120 dispatches) replayed from one captured hipGraph; "
                              "\"auto\" (default) turns it on where N * M <= 2e8"}

    # Gaussian likelihood only, reported beside the headline like `warm` and never as `value`: natgrad_step without the
    # predictive-variance product (t_SVGP(skip_unused_variance=True): under a Gaussian likelihood neither gradient
    # depends on it, so the updated sites are the same numbers); still cold, everything else rebuilt every step.
    skip_line = None
    if args.model == "tsvgp" and w["lik"] == "gaussian":
        model.skip_unused_variance = True
        for _ in range(2):
            model.natgrad_step((Xd, Yd), lr=0.8)
        barrier()
        eng.profile = {}
        with timed_region():
            t0 = time.perf_counter()
            for _ in range(args.steps):
                model.natgrad_step((Xd, Yd), lr=0.8)
            barrier()
            ts = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=device)
        skip_prof = eng.profile_summary()
        eng.profile = None
        if world > 1:
            dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        model.skip_unused_variance = False
        skip_line = {"value": round(args.steps / float(ts), 4), "unit": "E-steps/s",
                     "ms_per_step": round(float(ts) / args.steps * 1e3, 4),
                     "elbo_after_steps": float(model.elbo((Xd, Yd))),
                     "kernel_avg_ms": {k: round(v[1], 4) for k, v in sorted(skip_prof.items())},
                     "note": "skip_unused_variance=True (Gaussian likelihood): cold E-step without the predictive-variance "
                             "product, whose value the Gaussian site update does not use; not the headline"}

    # One iteration of the reference's E/M loop (experiments/uci_regression.py:132-160, defaults :17-21: 8 E-steps at lr 0.8, the
    # ELBO, 20 Adam steps at 0.1 on kernel variance / lengthscales / noise variance / Z), on a COPY of the state so that the parity
    # half below still sees the state the timed E-steps left.  A side line, never `value`.
    em_line = None
    if args.loop == "em" and args.model == "tsvgp" and not w.get("separate"):
        tr = pkg.training
        saved = (model.lambda_1.value.clone(), model.lambda_2_sqrt.value.clone(),
                 {n: p.value.clone() for n, (p, _) in tr.trainable_parameters(model).items()})
        opt = tr.Adam(0.1)
        for _ in range(2):  # library handles, buffers of the gradient pass
            model.elbo_and_grads((Xd, Yd))
        barrier()
        eng.profile = {}
        t0 = time.perf_counter()
        for _ in range(8):
            model.natgrad_step((Xd, Yd), lr=0.8)
        barrier()
        t_e = time.perf_counter() - t0
        e_kernels = eng.profile_summary()
        eng.profile = {}
        t0 = time.perf_counter()
        elbo_em = float(model.elbo((Xd, Yd)))
        barrier()
        t_l = time.perf_counter() - t0
        eng.profile = {}
        with timed_region():  # (the collector paused as for the headline: one full collection is 30-40 ms, +2 ms per gradient pass)
            t0 = time.perf_counter()
            tr.m_step(model, (Xd, Yd), opt, 20)
            barrier()
            t_m = time.perf_counter() - t0
        m_kernels = eng.profile_summary()
        eng.profile = None
        tm = torch.tensor([t_e, t_l, t_m], dtype=torch.float64, device=device)
        if world > 1:
            dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        t_e, t_l, t_m = (float(v) for v in tm)
        Mq, Pq = w["M"], w["P"]
        # algorithmic flops of one gradient evaluation on this rank's rows: the moments' triangular product and the site sums
        # (N M (M + 1) P each), and Q k_n for every row (2 N M^2 P as one dense product, or N M (M + 1) P when the moments'
        # product is reused: the figure below prices the CHEAPER form, whatever ran)
        g_flops = 3.0 * rows * Mq * (Mq + 1) * Pq
        em_line = {"iteration_s": round(t_e + t_l + t_m, 4), "e_block_ms": round(t_e * 1e3, 2), "elbo_log_ms": round(t_l * 1e3, 2),
                   "m_block_ms": round(t_m * 1e3, 2), "m_share": round(t_m / (t_e + t_l + t_m), 4),
                   "e_step_ms": round(t_e / 8 * 1e3, 3), "grad_eval_plus_adam_ms": round(t_m / 20 * 1e3, 3),
                   "elbo_after_e_block": elbo_em,
                   "m_step_roofline": {"bound": "mfma", "achieved": round(g_flops / (t_m / 20) / 1e12, 3), "peak": PEAK_TFLOPS[w["dtype"]],
                                       "unit": "TFLOP/s", "frac": round(g_flops / (t_m / 20) / 1e12 / PEAK_TFLOPS[w["dtype"]], 4),
                                       "algorithmic_flops_per_evaluation": g_flops,
                                       "note": "3 N M (M + 1) P: moments product, site sums, Q k_n from the moments' product"},
                   "m_step_kernel_avg_ms": {k: round(v[1], 4) for k, v in sorted(m_kernels.items())},
                   "m_step_kernel_ms_per_evaluation": round(sum(v[0] * v[1] for v in m_kernels.values()) / 20, 3),
                   "e_block_kernel_ms_per_step": round(sum(v[0] * v[1] for v in e_kernels.values()) / 8, 3),
                   "note": "ONE iteration of the reference's loop (experiments/uci_regression.py:132-160): 8 cold E-steps (lr 0.8), the "
                           "ELBO, 20 x (elbo_and_grads + Adam 0.1 on variance, lengthscales, noise variance, Z); every M-step changes "
                           "theta and Z, so each gradient pass rebuilds K_uu, K_uf and the factorisations; a side line, never `value`"}
        model.lambda_1.assign(saved[0])
        model.sites.assign_lambda_2_sqrt(saved[1])
        for n, (p_, _) in tr.trainable_parameters(model).items():
            p_.assign(saved[2][n])

    # the label of config.parallelism is computed on EVERY rank: _routes() may issue a broadcast when its cache misses, and a
    # collective only rank 0 enters (the others already wait in the closing barrier) would hang the job
    split_on = bool(world > 1 and getattr(model, "_latent_split", None) is not None
                    and model._latent_split(model._routes(1e-9)))
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        flops = kernel_flops(w, rows, batched=getattr(eng, "last_batched", False) or not w.get("separate"),
                             trmm_batch=getattr(eng, "last_trmm_batch", 1) if w.get("separate") else 1)
        byts = kernel_bytes(w, rows, esize, batched=getattr(eng, "last_batched", False))
        kern_ms = {k: v[1] for k, v in prof.items()}
        mfma = {k: kern_ms[k] for k in flops if k in kern_ms}
        dom = max(mfma, key=mfma.get) if mfma else None
        roofline = None
        if dom:
            achieved = flops[dom] / (mfma[dom] * 1e-3) / 1e12
            traffic = None
            tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
            if os.path.exists(tp):
                try:
                    traffic = json.load(open(tp)).get(args.workload, {}).get(dom)
                except Exception:
                    traffic = None
            roofline = {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_TFLOPS[w["dtype"]],
                        "unit": "TFLOP/s", "frac": round(achieved / PEAK_TFLOPS[w["dtype"]], 4), "traffic": traffic,
                        "algorithmic_flops_per_launch": flops[dom], "avg_launch_ms": round(mfma[dom], 4),
                        "min_launch_ms": round(prof[dom][2], 4), "median_launch_ms": round(prof[dom][3], 4),
                        "max_launch_ms": round(prof[dom][4], 4),
                        "launch_ms_in_order": [round(v, 3) for v in prof[dom][5]]}
        # the WHOLE step against the same peak: the algorithmic flops of its N-sized products on the route that ran
        # (moments + site sums, + one triangular product per whitening / projection launch) over the step's wall time --
        # everything that is not MFMA work (fill, M x M chain, host turn-around) counts as loss here
        step_flops = sum(flops[k] * prof[k][0] / args.steps for k in flops if k in prof)
        step_roofline = None if not step_flops else {
            "bound": "mfma", "achieved": round(step_flops / (ms_per_step * 1e-3) / 1e12, 3), "peak": PEAK_TFLOPS[w["dtype"]],
            "unit": "TFLOP/s", "frac": round(step_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_TFLOPS[w["dtype"]], 4),
            "algorithmic_flops_per_step": step_flops,
            "note": "sum over the step's MFMA launches of their algorithmic flops / ms_per_step (one rank's rows and time)"}
        kernels = {}
        for k, (n, ms, lo_ms, med_ms, hi_ms, _series) in sorted(prof.items()):
            e = {"launches_per_step": n / args.steps, "avg_ms": round(ms, 4), "min_ms": round(lo_ms, 4),
                 "median_ms": round(med_ms, 4), "max_ms": round(hi_ms, 4)}
            if k in flops:
                e["tflops"] = round(flops[k] / (ms * 1e-3) / 1e12, 2)
            if k in byts:
                e["gbs"] = round(byts[k] / (ms * 1e-3) / 1e9, 1)
                e["hbm_frac"] = round(e["gbs"] / HBM_PEAK_GBS, 4)
            kernels[k] = e
        out = {
            "metric": "natgrad E-steps/sec", "value": round(args.steps / elapsed, 4), "unit": "E-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": w["dtype"],
            "data": "synthetic",
            "config": {"workload": w["name"], "N": w["N"], "M": w["M"], "D": w["D"], "P": w["P"],
                       "likelihood": w["lik"], "rows_per_gpu": rows, "parallelism": (f"N-sharded x{world}, M x M work of the latents split over the ranks (all-gather of operands, "
                                       f"reduce-scatter of the sums, all-gather of the state)"
                                       if split_on else f"N-sharded x{world}, 1 all-reduce/step"),
                       "e_step": "cold (K_uu, K_uf, Choleskys, whitening rebuilt every step; the projection-route decision "
                                 "is cached per (theta, Z, jitter): see route_gate_ms)", "lr": 0.8,
                       "projection": getattr(model, "projection", None), "routes": routes,
                       "cond_Kuu_plus_jitter": conds},
            "route_gate_ms": None if route_gate_ms is None else round(route_gate_ms, 3),
            "other_routes": forced,
            "elbo_after_steps": elbo, "steps_before_elbo": steps_before_elbo,
            "warm": {"value": round(args.steps / warm_elapsed, 4), "unit": "E-steps/s",
                     "ms_per_step": round(warm_elapsed / args.steps * 1e3, 4),
                     "note": "cache_whitened=True: the factor of K_uu+jitter I, its inverse and the N x M operand (K_fu, or "
                             "the whitened B) reused across E-steps with unchanged hyperparameters; not the headline"},
            "hipgraph": graph_line,
            "em_loop": em_line,
            "skip_unused_variance": skip_line,
            "roofline": roofline,
            "step_roofline": step_roofline,
            "kernels": kernels,
            "kernel_ms_per_step": round(sum(v[0] * v[1] for v in prof.values()) / args.steps, 4),
            "kernel_timing": ("HIP events around every C-ABI launch, on its launch stream, " +
                              ("in an EAGER pass of the same steps right behind the timed region (the headline replays the step "
                               "from a captured hipGraph, which takes no events)" if replayed else "inside the timed region") +
                              "; the cyclic garbage collector is paused inside timed loops (as timeit does)"),
        }
        # The parity half of the metric and the CPU baseline ride on every line, also with N > 1 ranks: the oracle runs on
        # rank 0's host cores while the other ranks wait in the closing barrier.  (torch.distributed.run exports
        # OMP_NUM_THREADS=1 to its children unless told otherwise: the BLAS pool is widened to the box's cores here.)
        with blas_threads():
            sm = None
            if (args.model == "tsvgp" and world == 1 and not args.no_state_match and not args.no_elbo_match
                    and w["N"] * w["M"] * w["P"] >= 100_000_000):
                # one GPU, a full-size workload: ONE oracle E-step over all rows gives the ELBO match, the state match and the
                # measured CPU baseline together (with several ranks the others would wait minutes in the closing barrier:
                # the one-GPU line carries it)
                sm = state_match(model, w, X, Y, Z, Xd, Yd, args.state_budget)
            if sm is not None:
                out["elbo_match"], out["state_match"] = sm["elbo_match"], sm["state_match"]
            elif not args.no_elbo_match and args.model == "tsvgp":
                out["elbo_match"] = elbo_match(model, w, X, Y, Z, args.elbo_budget, elbo_all_ranks=elbo if world > 1 else None)
            elif not args.no_elbo_match:
                out["elbo_match"] = elbo_match_white(model, w, X, Y, Z)
            if not args.no_cpu_baseline:
                # the two-size fit (bounded sample) always runs; when the full-N step was measured it becomes the baseline and the
                # extrapolation stays beside it
                fit = cpu_baseline(w, args.cpu_budget if sm is None else min(args.cpu_budget, 6.0), model_kind=args.model)
                out["cpu_baseline"] = fit if sm is None else dict(sm["cpu_measured"], extrapolated=fit)
        out["elbo_vs_1gpu"] = elbo_vs_1gpu(args, w, elbo, steps_before_elbo)
        if collective is not None:
            # E-steps/s(N) = 1 / (N-pass + collective + chain): the N-sized kernels of the main stream from the per-kernel HIP
            # events (the K(X, Z) fill runs beside the chain on the side stream), the collectives from their own events inside
            # the timed region, the rest -- the replicated M x M chain and the host's turn-around -- by difference
            npass = sum(v[0] * v[1] for k, v in prof.items()
                        if k.split("(")[0] in ("tsvgp_moments", "tsvgp_site_accum", "tsvgp_trmm", "tsvgp_lik_map")) / args.steps
            coll_ms = sum(c["ms"]["mean"] * c["count_per_step"] for c in coll_times.values())
            ar = coll_times.get("all_reduce_sum")
            collective.update({
                "payload_bytes": None if ar is None else ar["payload_bytes"],
                "allreduce_ms": None if ar is None else {k: round(v, 4) for k, v in ar["ms"].items()},
                "per_step": {k: {"count_per_step": c["count_per_step"], "payload_bytes": c["payload_bytes"],
                                 "ms": {a: round(b, 4) for a, b in c["ms"].items()}, "host_ms_mean": round(c["host_ms_mean"], 4)}
                             for k, c in coll_times.items()},
                "npass_ms": round(npass, 4), "collective_ms": round(coll_ms, 4),
                "chain_ms": round(ms_per_step - npass - coll_ms, 4),
                "note": "rank 0's events: ms_per_step = npass_ms (moments + site sums + whitening launches) + collective_ms "
                        "(HIP events around every torch.distributed call of the timed steps) + chain_ms (the replicated M x M "
                        "chain, packing, host turn-around; by difference).  device_per_rank is all_gather_object of what each "
                        "rank's process group and device report"})
            out["collective"] = collective
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
