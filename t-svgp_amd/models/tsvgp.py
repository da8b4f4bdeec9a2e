"""t-SVGP model on MI355X (mirror of reference src/models/tsvgp.py).

Same constructor, properties and methods as the reference's ``t_SVGP`` so a driver written against it
(``natgrad_step(data, lr)``, ``elbo(data)``, ``predict_f``, ``new_predict_f``, ``lambda_1``, ``lambda_2_sqrt`` ...)
runs unchanged, but every N-sized operation goes through the HIP kernels behind ``include/tsvgp_hip.h`` and the
M x M site algebra runs on the GPU in fp64 through torch.  There is no CPU fallback.

Algebra of one E-step (P latents, shared kernel; K6 = Kuu + 1e-6 I, K9 = Kuu + jitter I).  Factorisations are taken
in upper form, W = U_W U_W^T and K9 = U9 U9^T (``util.rev_cholesky``), which makes every N-sized product triangular:

  reference (tsvgp.py:234-304)                       here (whitened route)
  -------------------------------------------------  ----------------------------------------------------------
  (m, chol S) = posterior_from_dense_site(K6, ...)   D = U_W^-1 L^T (upper), W = I + L^T K6 L       (util.py:168-175)
  mean = Kfu K6^-1 m                                 mean = B gamma,  gamma = U9^T beta,  beta = l1 - D^T D K6 l1
  var  = knn - |Lm^-1 k|^2 + |chol S^T K6^-1 k|^2    var  = knn - |T b|^2,  T = D U9 (upper)          [= knn - |D k|^2]
  A = Kfu K9^-1;  G1 = sum g1 a a^T                  B = Kfu U9^-T (HIP trmm);  acc2 = sum g1 b b^T (HIP syrk)
                                                     G1 = U9^-T acc2 U9^-1
  G0 = sum g0 a                                      G0 = U9^-T acc1
  lambda update + (-chol)                            identical (tsvgp.py:293-300)

Accumulating in whitened coordinates (B, not Kfu) keeps the M x M back-solves at cond(U9) = sqrt(cond(K9)).

Direct route (``projection="auto"`` picks it when cond(K9) <= DIRECT_MAX_COND): no N-sized whitening at all,
  mean = Kfu beta,  var = knn - |D k|^2 (D upper),  acc2 = sum g1 k k^T,  G1 = K9^-1 acc2 K9^-1,  G0 = K9^-1 acc1,
whose error grows faster than cond(K9) (measured: profiles/r02_route_gate_m1024.txt); it falls back to the whitened route
if the final factorisation fails.
"""
from __future__ import annotations

import abc
import functools
import os
import time
import warnings

import numpy as np
import torch

from .. import _backend as B
from .. import distributed as D_
from ..base import default_device, default_float, default_jitter, to_tensor
from ..estep import EStepStats
from ..inducing_variables import inducingpoint_wrapper
from ..kernels import SeparateIndependent, latent_kernels
from ..sites import DenseSites
from ..util import (
    bmv,
    cholesky_deferred,
    cond2_estimate,
    gradient_transformation_mean_var_to_expectation,
    info_sum,
    kl_from_dense_site,
    posterior_from_dense_site,
    rev_cholesky,
)


def _kmv(K: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """K v per latent: K [M, M] (shared kernel) or [P, M, M] (separate kernels), v [M, P] -> [M, P]."""
    return bmv(K, v)


def _ktmv(K: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """K^T v per latent (shapes as ``_kmv``)."""
    return bmv(K, v, transpose=True)


class base_SVGP(abc.ABC):
    """Mirror of the reference's ``base_SVGP`` (tsvgp.py:32-114): ELBO, prior KL and predict_f for a model that
    exposes q(u) through ``get_mean_chol_cov_inducing_posterior``."""

    def __init__(self, kernel, likelihood, inducing_variable, *, mean_function=None, num_latent_gps=1, num_data=None,
                 compute_dtype=None, device=None):
        if mean_function is not None:
            raise NotImplementedError("only the default Zero mean function is on the hot path")
        self.kernel = kernel
        self.likelihood = likelihood
        self.mean_function = None
        self.num_latent_gps = num_latent_gps
        self.num_data = num_data
        self.inducing_variable = inducingpoint_wrapper(inducing_variable)
        self.compute_dtype = compute_dtype or default_float()
        self.device = torch.device(device) if device is not None else default_device()
        self._engine = None
        self.poll_status = True  # how the step's status read waits for the GPU: see _read_flags
        self.data_parallel = None  # None = automatic: shard-reduce whenever torch.distributed has > 1 rank

    # -- engine ------------------------------------------------------------------------------------------------
    def _get_engine(self):
        """The HIP kernel launcher; creating it fails loudly when the extension or the GPU is missing."""
        if self._engine is None:
            from ..estep import EStepEngine

            self._engine = EStepEngine(self.compute_dtype, self.device)
        return self._engine

    def _reduce(self) -> bool:
        return D_.collectives_on() if self.data_parallel is None else bool(self.data_parallel)

    def _read_flags(self, flags: torch.Tensor) -> torch.Tensor:
        """The step's one device->host read.  On the GPU: an asynchronous copy into pinned memory and a POLLED event
        instead of a blocking ``.cpu()``: the wait never sleeps on an interrupt, so the host is back the moment the last
        kernel of the step retires.  The poll yields between queries (``time.sleep(0)``), so one host core is busy for the
        length of a step but other Python threads are not starved of the GIL; ``model.poll_status = False`` waits in a
        blocking stream synchronisation instead (an idle core per rank, the host back some tens of microseconds later)."""
        if not flags.is_cuda:
            return flags
        host = getattr(self, "_flags_host", None)
        if host is None or host.numel() != flags.numel():
            host = self._flags_host = torch.empty(flags.numel(), dtype=flags.dtype).pin_memory()
        host.copy_(flags, non_blocking=True)
        if not self.poll_status:
            torch.cuda.current_stream(flags.device).synchronize()
            return host.clone()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(flags.device))
        while not ev.query():
            time.sleep(0)  # gives up the GIL and the time slice: other Python threads (loaders, logging) keep running
        return host.clone()

    @abc.abstractmethod
    def get_mean_chol_cov_inducing_posterior(self):
        """Returns the mean and cholesky factor of the covariance matrix of q(u)"""
        raise NotImplementedError

    def maximum_log_likelihood_objective(self, data):
        return self.elbo(data)

    def training_loss(self, data):
        return -self.elbo(data)

    def training_loss_closure(self, data, *, compile=False):
        """``gpflow.models.ExternalDataTrainingLossMixin.training_loss_closure`` [ext] (reference
        experiments/uci_regression.py:114): a zero-argument callable returning the negative ELBO.  ``data`` is a
        (X, Y) tuple or an iterator of such tuples (one batch per call).  ``compile`` is accepted and ignored: there is
        no tracing compiler here."""
        if isinstance(data, (tuple, list)):
            return lambda: self.training_loss(tuple(data))
        it = iter(data)
        return lambda: self.training_loss(next(it))

    @property
    def trainable_parameters(self):
        """The parameters the reference's M-step trains (``model.trainable_variables``, experiments/uci_regression.py:160):
        kernel variance(s) and lengthscales, the likelihood's parameters and the inducing inputs; the sites are not
        trainable (src/sites.py:56-63)."""
        out, seen = [], set()
        kernels = self.kernel.kernels if hasattr(self.kernel, "kernels") else [self.kernel]
        for k in kernels:
            for par in (k.variance, k.lengthscales):
                if par.trainable and id(par) not in seen:
                    seen.add(id(par))
                    out.append(par)
        out += [par for par in vars(self.likelihood).values() if hasattr(par, "trainable") and par.trainable]
        if self.inducing_variable.Z.trainable:
            out.append(self.inducing_variable.Z)
        return tuple(out)

    trainable_variables = trainable_parameters  # GPflow's name for the unconstrained counterparts


class t_SVGP(base_SVGP):
    """Class for the t-SVGP model (reference tsvgp.py:117-304)."""

    def __init__(self, kernel, likelihood, inducing_variable, *, mean_function=None, num_latent_gps: int = 1,
                 lambda_1=None, lambda_2_sqrt=None, num_data=None, force=False, compute_dtype=None, device=None,
                 cache_whitened=False, projection="auto", use_graph="auto", skip_unused_variance=False,
                 overlap_fill=True, latent_split=None):
        super().__init__(kernel, likelihood, inducing_variable, mean_function=mean_function,
                         num_latent_gps=num_latent_gps, num_data=num_data, compute_dtype=compute_dtype, device=device)
        self.num_inducing = self.inducing_variable.num_inducing
        self._init_variational_parameters(self.num_inducing, lambda_1, lambda_2_sqrt)
        self.whiten = False
        self.force = force
        # Opt-in "warm" E-steps: keep chol(K_uu + jitter I), its inverse and the whitened B = K_fu L^-T between calls while
        # the kernel parameters, Z, the jitter and the data tensor are unchanged (the reference rebuilds them on every
        # call; with the cache off -- the default -- so does this class).
        self.cache_whitened = cache_whitened
        self._warm = None
        # How the projection A = K_fu K_uu^-1 of tsvgp.py:268-271 enters the site sums:
        #   "whitened": B = K_fu L^-T by an N-sized triangular product, sums over b b^T, two M x M back-solves
        #               (error ~ sqrt(cond K_uu) eps: safe for any K_uu the reference can factorise);
        #   "direct":   sums over k k^T on K_fu itself, then K_uu^-1 (.) K_uu^-1 by M x M Cholesky solves
        #               (error ~ cond(K_uu) eps; one N M^2 product fewer);
        #   "projected": a = K_uu^-1 k by a second N-sized triangular product, sums over a a^T as the reference does
        #               (G1 is a sum of outer products: stays definite at any cond(K_uu); one N M^2 product more);
        #   "auto":     by cond(K_uu + jitter I): direct, whitened or projected (see _routes); a failed final
        #               factorisation moves a latent one route down.
        if projection not in ("auto", "whitened", "direct", "projected"):
            raise ValueError("projection must be 'auto', 'whitened', 'direct' or 'projected'")
        self.projection = projection
        self._cond_cache = None
        # Replay natgrad_step from a captured hipGraph (see _graph_step): True, False or "auto" (by problem size, where
        # the step is launch bound; see _wants_graph)
        if use_graph not in (True, False, "auto"):
            raise ValueError('use_graph must be True, False or "auto"')
        self.use_graph = use_graph
        self._graphs = {}
        # Opt-in, Gaussian likelihood only: natgrad_step skips the predictive-variance product.  The reference computes
        # the variance because autodiff needs the forward pass (tsvgp.py:246-259), but d ve/d mean = (y - mean)/s2 and
        # d ve/d var = -1/(2 s2) do not depend on it, so the updated sites are the same numbers; what is lost is the
        # var > 0 check of the step (tsvgp.py:113), which elbo / predict_f still make.
        self.skip_unused_variance = skip_unused_variance
        # natgrad_step starts the K(X, Z) fill on a side stream before the M x M prelude (EStepEngine.start_fill)
        self.overlap_fill = overlap_fill
        # one kernel per latent on several ranks: split the latents' M x M algebra over the ranks (see _latent_split)
        self.latent_split = latent_split
        self.name = "t_svgp"  # tf.Module derives this from the class name (experiments/uci_regression.py:150)

    def _init_variational_parameters(self, num_inducing, lambda_1, lambda_2_sqrt, **kwargs):
        """Constructs the site parameters lambda_1, Lambda_2 of t(u) = exp(u^T l1 - 1/2 u^T L2 u) (tsvgp.py:159-185)."""
        lambda_1 = np.zeros((num_inducing, self.num_latent_gps)) if lambda_1 is None else lambda_1
        if lambda_2_sqrt is None:
            lambda_2_sqrt = np.array([-np.eye(num_inducing) * 1e-10 for _ in range(self.num_latent_gps)])
        else:
            lambda_2_sqrt = lambda_2_sqrt.value if hasattr(lambda_2_sqrt, "value") else lambda_2_sqrt
            assert lambda_2_sqrt.ndim == 3
            self.num_latent_gps = lambda_2_sqrt.shape[0]
        self.sites = DenseSites(to_tensor(lambda_1, device=self.device), to_tensor(lambda_2_sqrt, device=self.device))

    @property
    def lambda_1(self):
        """first natural parameter"""
        return self.sites.lambda_1

    @property
    def lambda_2_sqrt(self):
        """Cholesky factor of the second natural parameter"""
        return self.sites.lambda_2_sqrt

    @property
    def lambda_2(self):
        """second natural parameter"""
        L = self.lambda_2_sqrt.value
        return L @ L.transpose(-1, -2)

    # -- M x M prelude -----------------------------------------------------------------------------------------
    def _kmv(self, K: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        """K v per latent on the engine's own one-wave-per-row kernel (``tsvgp_gemv_f64``) where there is one: rocBLAS takes
        24-30 us for the 8 MB read of an M = 1024 matrix-vector product on the replicated chain."""
        eng = self._get_engine()
        if hasattr(eng, "gemv") and K.is_cuda and K.dtype == torch.float64 and v.dtype == torch.float64:
            return eng.gemv(K, v)
        return _kmv(K, v)

    def _Z(self) -> torch.Tensor:
        return self.inducing_variable.Z.value.to(self.device)

    def _eye(self, M: int) -> torch.Tensor:
        """Cached fp64 identity on the model device (read-only: callers add it, never write to it)."""
        e = getattr(self, "_eye_cache", None)
        if e is None or e.shape[0] != M or e.device != self.device:
            e = self._eye_cache = torch.eye(M, dtype=torch.float64, device=self.device)
        return e

    @staticmethod
    def _k6_of(Kzz: torch.Tensor) -> torch.Tensor:
        """K_uu + default_jitter() I (tsvgp.py:209-211) as a fresh tensor."""
        K6 = Kzz.clone()
        K6.diagonal(dim1=-2, dim2=-1).add_(default_jitter())
        return K6

    def _k6(self, Kzz: torch.Tensor) -> torch.Tensor:
        """``_k6_of`` in one launch on the GPU (``tsvgp_tri_copy_shift_f64``: a copy with the jitter on the diagonal)."""
        eng = self._get_engine()
        if Kzz.is_cuda and Kzz.dtype == torch.float64 and hasattr(eng, "tri_copy"):
            M = Kzz.shape[-1]
            return eng.tri_copy(Kzz.reshape(-1, M, M), M, 1.0, 3, diag_add=default_jitter()).reshape(Kzz.shape)
        return self._k6_of(Kzz)

    def _warm_key(self, X, jitter):
        """Cache key of everything B = K(X, Z) U9^-T depends on; None when caching is off or X is not a device tensor."""
        if not self.cache_whitened or not isinstance(X, torch.Tensor) or X.device != self.device:
            return None
        if isinstance(self.kernel, SeparateIndependent):
            return None  # one K(X, Z) buffer serves the latents in turn: nothing N-sized survives the call
        return (X.data_ptr(), tuple(X.shape), X._version, X.dtype, self._kernel_versions(),
                self.inducing_variable.Z.stamp(), float(jitter), self.compute_dtype)

    def _kernel_versions(self):
        # Parameter identity and the tensors' own edit counters ride along: a replaced Parameter restarts at version 0 and an
        # in-place edit of .value does not pass through assign()
        return tuple((id(k), k.variance.stamp(), k.lengthscales.stamp())
                     for k in latent_kernels(self.kernel, self.num_latent_gps))

    # Route gates on cond(K_uu + jitter I), per latent GP.  direct: K^-1 (sum g k k^T) K^-1 cancels two factors of K, so its
    # error grows faster than cond.  Measured against the CPU restatement of the reference at the benchmark's M = 1024, D = 8
    # (tools/route_gate.py; max rel err of lambda_1 / Lambda_2): round 2 over two steps (profiles/r02_route_gate_m1024.txt), round 3
    # over EIGHT steps -- what the reference's loop runs per M-step -- for the Gaussian AND the Bernoulli likelihood
    # (profiles/r03_route_gate_8steps.txt): the error settles by the second step and stays there: 7.5e-13 at cond 5e2, 2e-11 at
    # 4e3, 7e-11 at 1e4, 2e-10 at 2e4, 9.5e-10 / 6.4e-10 (Gaussian / Bernoulli) at 5.4e4, 1.2e-8 / 7e-9 at 2.5e5.  It reaches
    # 1e-9 -- a tenth of the stated 1e-8 -- near cond 6e4; the estimate the gate sees (util.cond2_estimate) is a LOWER bound whose
    # test accepts 10 % under, so the fp64 gate is 5e4 (round 1 had 1e3, set from M <= 64 fixtures; round 2 6e4 without the
    # slack).  tests/test_gpu_benchshape.py::test_direct_route_at_the_gate_holds_1e8_over_eight_steps pins it.  fp32: cond <= 30
    # keeps it <= ~1e-4.  whitened: the sums over b b^T carry an ABSOLUTE error ~eps |acc2| that U9^-T (.) U9^-1 amplifies
    # by |K9^-1| (measured 4e-10 at cond 1e7, 2e-8 at 1e9, and the final factorisation loses definiteness beyond):
    # cond <= 1e7.  projected: any cond.
    DIRECT_MAX_COND = {torch.float64: 5.0e4, torch.float32: 30.0}
    WHITENED_MAX_COND = {torch.float64: 1.0e7, torch.float32: float("inf")}
    _DEMOTE = {"direct": "whitened", "whitened": "projected"}

    def _routes(self, jitter) -> list:
        """The projection route of every latent GP (one decision for all latents under a shared kernel): "direct",
        "whitened" or "projected".  In "auto" mode the 2-norm condition number of K_uu + jitter I is estimated
        (``util.cond2_estimate``: one factorisation + power iterations, M x M, one host read) only when the kernel
        parameters, Z or the jitter changed."""
        P = self.num_latent_gps
        if self.projection != "auto":
            return [self.projection] * P
        key = (self._kernel_versions(), self.inducing_variable.Z.stamp(), float(jitter))
        if self._cond_cache is None or self._cond_cache[0] != key:
            eng = self._get_engine()
            Kzz = eng.kuu(self._Z(), self.kernel)  # [M, M] or [P, M, M]
            Kzz.diagonal(dim1=-2, dim2=-1).add_(jitter)
            cond = cond2_estimate(Kzz, getattr(eng, "cholesky", None)).reshape(-1)
            # one decision for all ranks -- only when this model's steps are collective at all (data_parallel=False, or a
            # rank-0-only side computation, must not issue a broadcast the other ranks never join)
            cond = (D_.broadcast_from_rank0(cond.contiguous()) if self._reduce() else cond).tolist()
            self._cond_cache = (key, cond if len(cond) == P else cond * P, {})
        dmax, wmax = self.DIRECT_MAX_COND[self.compute_dtype], self.WHITENED_MAX_COND[self.compute_dtype]
        routes = ["direct" if c <= dmax else "whitened" if c <= wmax else "projected" for c in self._cond_cache[1]]
        for p, r in self._cond_cache[2].items():  # latents demoted after a failed step keep their route
            order = ("direct", "whitened", "projected")
            routes[p] = order[max(order.index(routes[p]), order.index(r))]
        return routes

    def _use_direct(self, jitter) -> list:
        return [r == "direct" for r in self._routes(jitter)]

    def _site_operands(self, whiten_jitter=None, warm_key=None, routes=None, latents=None, fork=True, Kzz=None, K6=None,
                       after_w=None, beside_fill=False):
        """Everything the N-pass needs that depends only on (theta, Z, lambda): O(M^3), fp64, replicated.
        No host synchronisation happens here: Cholesky statuses are collected in ops["infos"] and checked once per
        call by ``_check_step`` (TF raises immediately; here the raise comes at the end of the same call).

        Both factorisations are taken in UPPER form, A = U U^T (``rev_cholesky``).  With W = U_W U_W^T the projection
        D = U_W^-1 L^T (util.py:168-175 up to an orthogonal factor: D^T D = L W^-1 L^T either way) is itself upper
        triangular, and so is T = D U9 for K_uu + jitter I = U9 U9^T: the predictive variance knn - |D k|^2 =
        knn - |T b|^2 is a triangular product without any further factorisation."""
        eng = self._get_engine()
        Z = self._Z()
        M = Z.shape[0]
        infos = []
        warm = self._warm if (warm_key is not None and self._warm is not None and self._warm[0] == warm_key) else None
        kernel = self.kernel
        l1 = self.lambda_1.value
        L = self.lambda_2_sqrt.value
        if latents is not None:
            # the latent GPs this rank owns (one kernel per latent, several ranks: ``_step_device_split``): everything below is
            # per latent, so the subset is a smaller batch of the same algebra.  ``routes`` is already the subset's.
            idx = torch.as_tensor(list(latents), device=l1.device)
            kernel = SeparateIndependent([self.kernel.kernels[p] for p in latents])
            l1, L = l1.index_select(1, idx), L.index_select(0, idx)
        if warm:
            Kzz = warm[1]["Kzz"]
        elif Kzz is None or latents is not None:
            Kzz = eng.kuu(Z, kernel)  # HIP fill kernel, no jitter (``Kzz``: the caller already has it, see _step_front)
        Id = self._eye(M)
        if warm and "K6" in warm[1]:
            K6 = warm[1]["K6"]  # read only from here on (40 us of copy + strided add per step otherwise)
        elif K6 is None or latents is not None:
            K6 = self._k6(Kzz)
        P_ = L.shape[0]
        potrf = getattr(eng, "cholesky", None)  # HIP blocked Cholesky (tsvgp_potrf_f64)
        robust = potrf is not None and routes is not None and any(r == "projected" for r in routes)
        if robust:
            # cond(K_uu + jitter I) beyond 1e7: K9 and the new Lambda_2 are barely definite in fp64 there, and the
            # factorisation that follows the reference's success / failure is the one with substitution panels
            # (TSVGP_POTRF_SUBST through EStepEngine.cholesky(robust=True))
            potrf = functools.partial(potrf, robust=True)
        # W = I + L^T K6 L (util.py:171-172, formed without chol(K6)).  The factorisations of W and K_uu + jitter I are
        # independent and both latency bound (one workgroup per diagonal block): do them in ONE batched call, whose
        # input batch is assembled in place (the addition of I rides on the GEMM; no concatenation).  Only one
        # triangle of W is read, so the rounding-level asymmetry of L^T (K6 L) needs no symmetrisation.
        # The factorisation also returns the inverse factors (tsvgp_potrf_inv_f64) and they are applied as GEMMs: a
        # rocBLAS trsm with an M x M right-hand side costs ~0.25 ms at M = 1024, a GEMM ~0.05 ms.
        # Uniformly "projected" (cond(K9) beyond 1e7): K9 is factored in the REFERENCE's order (plain lower Cholesky,
        # tsvgp.py:270) in its own call instead of joining the upper-form batch -- that close to singular, whether a
        # factorisation goes through depends on the elimination order, and the reference's is the one to follow
        lower9 = (whiten_jitter is not None and routes is not None and len(routes) > 0
                  and all(r == "projected" for r in routes))
        with_k9 = whiten_jitter is not None and not warm and not lower9
        n9 = (Kzz.shape[0] if Kzz.dim() == 3 else 1) if with_k9 else 0
        K6l = self._kmv(K6, l1)  # [M, P]: needed behind the factorisation (beta) -- issued in front of it, off the path to the moments
        # (the transposed-operand rocBLAS kernels take 60 us at M = 1024 where the plain one takes 40: L^T is copied out first,
        # for this product and for L L^T below)
        Lt = L.transpose(-1, -2).contiguous()
        solve = potrf is not None and hasattr(eng, "cholesky_solve_upper") and os.environ.get("TSVGP_POTRF_SOLVE", "1") != "0"
        if solve:
            # the identity of W and the jitter of K9 = K_uu + jitter I (tsvgp.py:270) ride on the pass that hands the matrices to
            # the factorisation (EStepEngine.solve_upper_put): no assembled batch, no 8 MB copy of K_uu, no strided additions --
            # three launches fewer on the chain.  What exists before W does -- K9 and both right-hand sides -- is handed over
            # first, on a chip the shard's fill has not reached yet (three passes of 7 us instead of 20-30 each beside it; the
            # step does not notice: handed over behind the fill's start it measures the same to 0.03 ms, the fill then runs that
            # much longer -- one more section in which the fill and the chain trade one for one).
            job = eng.solve_upper_begin(P_ + n9, M)
            eng.solve_upper_put_rhs(job, 0, L)
            if with_k9:
                eng.solve_upper_put(job, P_, Kzz if Kzz.dim() == 3 else Kzz[None], whiten_jitter)
                eng.solve_upper_put_rhs(job, P_, Id.expand(n9, M, M))
            eng.solve_upper_put(job, 0, torch.bmm(Lt, K6 @ L), 1.0)
            if after_w is not None:
                after_w()  # a small shard's K(X, Z) fill starts here, behind the two GEMMs it would otherwise starve (_step_front)
        else:
            batch = torch.empty((P_ + n9, M, M), dtype=torch.float64, device=Kzz.device)
            # (bmm + a strided add of the identity: baddbmm first copies its [P, M, M] addend into the output, 8 MB per latent)
            torch.bmm(Lt, K6 @ L, out=batch[:P_])
            batch[:P_].diagonal(dim1=-2, dim2=-1).add_(1.0)
            if after_w is not None:
                after_w()
            if with_k9:
                batch[P_:].copy_(Kzz if Kzz.dim() == 3 else Kzz[None])
                batch[P_:].diagonal(dim1=-2, dim2=-1).add_(whiten_jitter)  # K9 = K_uu + jitter I, tsvgp.py:270
        # Factor AND solve in one pass (EStepEngine.cholesky_solve_upper): D = U_W^-1 L^T comes out of the factorisation itself,
        # J L J riding along as panel rows -- and with the identity riding along K_uu + jitter I, its inverse factor -- instead of
        # from the inverse recursion (three levels of launch pairs), two triangle copies, a 1024^3 GEMM and a triu pass.
        # (Measured and NOT kept, profiles/r04_chain_ab.txt: K_uu + jitter I factored on the side stream, off the chain -- its 24
        # dependent launches then queue behind the moments kernel's workgroups: 4 ms per call at N = 1e6, the step 0.2 ms slower.)
        Dm = None
        if solve:
            both, info_w, sol = eng.solve_upper_run(job, robust=robust, beside_fill=bool(beside_fill))
            infos.append(info_w.reshape(-1).to(torch.int32))
            U_W, Dm, Uinv_W, inv_both = both[:P_], sol[:P_], None, sol
        else:
            both, inv_both = rev_cholesky(batch, infos, potrf, inverse=True)
            U_W, Uinv_W = both[:P_], inv_both[:P_]
        L9inv = None
        if with_k9:
            U9, Uinv9 = (both[P_:], inv_both[P_:]) if Kzz.dim() == 3 else (both[-1], inv_both[-1])
        elif lower9:
            K9 = Kzz.clone()
            K9.diagonal(dim1=-2, dim2=-1).add_(whiten_jitter)
            _, L9inv = cholesky_deferred(K9 if K9.dim() == 3 else K9[None], infos, potrf, inverse=True, overwrite=True)
            L9inv = L9inv if Kzz.dim() == 3 else L9inv[0]
            U9, Uinv9 = None, None
        elif whiten_jitter is not None:
            U9, Uinv9 = warm[1]["U9"], warm[1]["Uinv9"]
        else:
            U9, Uinv9 = None, None
        if Dm is None:
            Dm = (Uinv_W @ L.transpose(-1, -2)).triu()  # D = U_W^-1 L^T, [P, M, M], upper triangular
        if hasattr(eng, "site_beta") and Dm.is_cuda and os.environ.get("TSVGP_SITE_BETA", "1") != "0":
            beta = eng.site_beta(Dm, K6l, l1)  # K6^-1 m = l1 - D^T D K6 l1: two triangular matrix-vector launches
        else:
            beta = l1 - bmv(Dm, bmv(Dm, K6l), transpose=True)
        # (Lt rides along because the side stream reads it below: inside a capture nothing else keeps a block of the graph's pool
        # from being handed to the next allocation of the capturing stream while the forked branch still reads it -- the race
        # of profiles/r04_c4_fork_capture_race.txt, met again in round 5 through this very tensor)
        ops = dict(Z=Z, Kzz=Kzz, K6=K6, D=Dm, U_W=U_W, beta=beta, Id=Id, infos=infos, potrf=potrf, Lt=Lt,
                   routes=["whitened"] * P_, moment_mode=B.TRI_UPPER, whiten_mode=B.TRI_UPPER,
                   whiten_T=None, project_T=None)
        if whiten_jitter is None:
            return ops
        ops["U9"], ops["Uinv9"] = U9, Uinv9  # K_uu + jitter I = U9 U9^T, tsvgp.py:268-270
        ops["moments_on_kfu"], ops["project_mode"] = False, B.TRI_LOWER
        # What only the EPILOGUE needs of (theta, Z, the old sites): L L^T (tsvgp.py:293), predict_f(Z)'s mean K_uu beta
        # (:249-254) and, for the direct route, K9^-1 = U9^-T U9^-1.  Three M^3 GEMMs that nothing in the N-pass waits for:
        # they go to the side stream (behind the K(X, Z) fill) and run beside the moments kernel instead of in front of
        # it; ``_apply_site_update`` waits for ops["epi_event"].
        want_k9inv = routes is not None and "direct" in routes and Uinv9 is not None

        def epilogue_operands():
            ops["LLt"] = L @ Lt
            ops["meanZ"] = self._kmv(Kzz, beta)
            if want_k9inv:
                ops["K9inv"] = Uinv9.transpose(-1, -2) @ Uinv9

        side = getattr(eng, "_side", None)
        if not fork or os.environ.get("TSVGP_EPI_INLINE") == "1":
            # in line, in front of the moments kernel: a captured launch-bound step (see _step_front); as an experiment at
            # N = 1e6 it measured 33.83 / 33.90 ms against 33.92 / 33.96 on the side stream -- no difference worth a second path
            epilogue_operands()
        elif self.overlap_fill and side is not None and Kzz.is_cuda:
            capturing = torch.cuda.is_current_stream_capturing()  # the side stream then joins the capture (fork / join by events)
            main = torch.cuda.current_stream(self.device)
            ready = torch.cuda.Event()
            ready.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ready)
                epilogue_operands()
                done = torch.cuda.Event()
                done.record(side)
            if not capturing:
                for t in (L, Lt, Kzz, beta) + ((Uinv9,) if want_k9inv else ()):
                    t.record_stream(side)  # blocks of the main stream's pool read on the side stream
                for k in ("LLt", "meanZ", "K9inv"):
                    if k in ops:
                        ops[k].record_stream(main)  # and the other way round
            ops["epi_event"] = done
        else:
            epilogue_operands()
        if lower9:
            # a = K9^-1 k by two triangular products with the lower factor: b = L9^-1 k, a = L9^-T b; the moments act on
            # K(X, Z) with D and beta, so nothing on their side depends on the factor of K9
            ops["routes"] = list(routes)
            ops.update(gamma=beta, moment_Tm=Dm, whiten_T=L9inv, whiten_mode=B.TRI_LOWER,
                       project_T=L9inv.transpose(-1, -2).contiguous(), project_mode=B.TRI_UPPER, moments_on_kfu=True)
            return ops
        if warm_key is not None and not warm:
            self._warm = (warm_key, dict(Kzz=Kzz, K6=K6, U9=U9, Uinv9=Uinv9))
        routes = list(routes) if routes is not None else ["whitened"] * P_
        ops["routes"] = routes
        if all(r == "direct" for r in routes):
            # direct projection: the moments act on K_fu with D itself, the sums are mapped by K9^-1 (.) K9^-1
            # afterwards; no N-sized whitening
            ops["gamma"], ops["moment_Tm"] = beta, Dm
        else:
            gamma_w = _ktmv(U9, beta)  # mean = k^T beta = b^T U9^T beta with b = U9^-1 k
            T_w = (Dm @ U9).triu()  # var = knn - |D k|^2 = knn - |T b|^2
            Uinv9t = Uinv9.transpose(-1, -2).contiguous()  # projected route: a = U9^-T b  (lower triangular product)
            if len(set(routes)) == 1:
                ops["whiten_T"], ops["gamma"], ops["moment_Tm"] = Uinv9, gamma_w, T_w  # B = K_fu U9^-T
                ops["project_T"] = Uinv9t if routes[0] == "projected" else None
            else:
                # separate kernels, latents of different conditioning: each latent takes its own route
                per = lambda m, p: m[p] if m.dim() == 3 else m
                sel = torch.tensor([r == "direct" for r in routes], device=Dm.device)
                ops["gamma"] = torch.where(sel[None, :], beta, gamma_w)
                ops["moment_Tm"] = torch.where(sel[:, None, None], Dm, T_w)
                ops["whiten_T"] = [None if r == "direct" else per(Uinv9, p) for p, r in enumerate(routes)]
                ops["project_T"] = [per(Uinv9t, p) if r == "projected" else None for p, r in enumerate(routes)]
        return ops

    def _status_flags(self, ops, nonpos, extra_infos=()) -> torch.Tensor:
        """Device tensor [3]: failed prelude factorisations, count of non-positive variances, failed final one
        (one kernel, ``tsvgp_step_status_f64``)."""
        eng = self._get_engine()
        if hasattr(eng, "step_status") and nonpos.is_cuda:
            return eng.step_status(ops["infos"], list(extra_infos), nonpos)
        zero = torch.zeros(1, dtype=torch.float64, device=self.device)
        final = info_sum(extra_infos) if len(extra_infos) else zero
        return torch.cat([info_sum(ops["infos"]), nonpos.reshape(1).to(torch.float64), final])

    @staticmethod
    def _judge(flags, soft_final=False):
        """Host side of the status check.  Returns True when the step stands; "whiten" when soft_final is set and the
        final factorisation failed (the direct projection lost definiteness: the caller retries with the whitened
        route); raises FloatingPointError for what TensorFlow raises on (tsvgp.py:113 assert_positive; failed Cholesky)."""
        if float(flags[0]) != 0:  # factorisation of W or of K_uu + jitter I
            raise FloatingPointError("Cholesky decomposition was not successful (matrix not positive definite)")
        if not (float(flags[1]) == 0):  # a NaN count also lands here
            raise FloatingPointError(f"non-positive predictive variance at {float(flags[1]):.0f} point(s)")
        if float(flags[2]) != 0:  # chol(-2 lambda_2 + jitter I), tsvgp.py:300
            if soft_final:
                return "whiten"
            raise FloatingPointError("Cholesky decomposition was not successful (matrix not positive definite)")
        return True

    def _check_step(self, ops, nonpos, extra_infos=(), soft_final=False):
        """ONE device->host read per call: Cholesky statuses and the count of non-positive variances."""
        return self._judge(self._read_flags(self._status_flags(ops, nonpos, extra_infos)), soft_final)

    def get_mean_chol_cov_inducing_posterior(self):
        """Mean and Cholesky factor of q(u) = N(u; m, S) (tsvgp.py:202-212)."""
        eng = self._get_engine()
        Kzz = eng.kuu(self._Z(), self.kernel)
        K_uu = Kzz + default_jitter() * torch.eye(Kzz.shape[-1], dtype=Kzz.dtype, device=Kzz.device)
        return posterior_from_dense_site(K_uu, self.lambda_1.value, self.lambda_2_sqrt.value)

    def prior_kl(self):
        """KL[q(u) || p(u)] (tsvgp.py:65-70)."""
        ops = self._site_operands()
        kl = kl_from_dense_site(ops["K6"], self.lambda_1.value, ops["D"], ops["U_W"], ops["beta"])
        self._check_step(ops, torch.zeros(1, dtype=torch.float64, device=self.device))
        return kl

    # -- data plumbing -----------------------------------------------------------------------------------------
    def _as_device(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(self.device)
        return torch.as_tensor(np.asarray(a)).to(self.device)

    # -- predictions -------------------------------------------------------------------------------------------
    def predict_f(self, Xnew, full_cov=False, full_output_cov=False):
        """Posterior prediction at new input Xnew [N, D] (tsvgp.py:97-114): mean = k^T beta, var = knn - |D k|^2 on
        K(Xnew, Z) itself -- the reference's conditional only involves K_uu + 1e-6 I (tsvgp.py:209-211), and so does this
        form (no factor of K_uu + 1e-9 I, no N-sized whitening)."""
        if full_cov or full_output_cov:
            raise NotImplementedError("full covariances are not on the E-step hot path")
        Xnew = self._as_device(Xnew)
        ops = self._site_operands()
        st = self._get_engine().run(Xnew, None, ops["Z"], self.kernel, moment_Tm=ops["D"], moment_mode=ops["moment_mode"],
                                    gamma=ops["beta"], want_moments=True)
        self._check_step(ops, st.nonpos)
        return st.mean, st.var

    def new_predict_f(self, Xnew, full_cov=False, full_output_cov=False):
        """Same moments straight from the sites: var = knn - |D k|^2 (tsvgp.py:215-232, util.py:91-185)."""
        if full_cov or full_output_cov:
            raise NotImplementedError("full covariances are not on the E-step hot path")
        if isinstance(self.kernel, SeparateIndependent):
            # "todo : make broadcastable" (tsvgp.py:214): the reference form only covers one shared kernel
            raise NotImplementedError("new_predict_f is not broadcastable over separate kernels in the reference")
        ops = self._site_operands()
        st = self._get_engine().run(self._as_device(Xnew), None, ops["Z"], self.kernel, moment_Tm=ops["D"],
                                    moment_mode=ops["moment_mode"], gamma=ops["beta"], want_moments=True)
        self._check_step(ops, st.nonpos)
        return st.mean, st.var

    def moments_and_gradients(self, data):
        """The N-sized intermediates of one E-step at the current state, without updating it (tsvgp.py:246-263):
        mean, var = predict_f(X); g0 = d ve/d mean; g1 = min(d ve/d var, -1e-8).  All [N, P] fp64 on the model device.
        Not part of the reference's API: what parity checks (tests/, bench.py's ``elbo_match``) read."""
        X, Y = self._as_device(data[0]), self._as_device(data[1])
        ops = self._site_operands()
        st = self._get_engine().run(X, Y, ops["Z"], self.kernel, moment_Tm=ops["D"], moment_mode=ops["moment_mode"],
                                    gamma=ops["beta"], lik_id=self.likelihood.lik_id, lik_param=self.likelihood.lik_param,
                                    want_moments=True, want_grads=True)
        self._check_step(ops, st.nonpos)
        return st.mean, st.var, st.g0, st.g1

    def predict_y(self, Xnew):
        return self.likelihood.predict_mean_and_var(*self.predict_f(Xnew))

    def predict_log_density(self, data):
        X, Y = data
        Fmu, Fvar = self.predict_f(X)
        return self.likelihood.predict_log_density(Fmu, Fvar, self._as_device(Y).to(Fmu.dtype))

    # -- ELBO --------------------------------------------------------------------------------------------------
    def elbo(self, data):
        """Evidence lower bound  sum_n E_q[log p(y_n | f_n)] * scale - KL[q(u) || p(u)]  (tsvgp.py:79-95).
        With more than one rank ``data`` is this rank's row shard and the sum is all-reduced."""
        X, Y = data
        ops = self._site_operands()
        kl = kl_from_dense_site(ops["K6"], self.lambda_1.value, ops["D"], ops["U_W"], ops["beta"])
        st = self._get_engine().run(self._as_device(X), self._as_device(Y), ops["Z"], self.kernel,
                                    moment_Tm=ops["D"], moment_mode=ops["moment_mode"], gamma=ops["beta"],
                                    lik_id=self.likelihood.lik_id, lik_param=self.likelihood.lik_param)
        _, _, ve_sum, nonpos, rows, _ = D_.reduce_stats(st, self.num_latent_gps, self.num_inducing, False, self._reduce(),
                                                        self._get_engine())
        self._check_step(ops, nonpos)
        scale = (float(self.num_data) / rows) if self.num_data is not None else 1.0
        return ve_sum * scale - kl

    # -- M-step gradient -----------------------------------------------------------------------
    def elbo_and_grads(self, data):
        """ELBO and its gradient with respect to the kernel variance, the lengthscales, the inducing inputs Z and (Gaussian
        likelihood) the noise variance, with the sites held fixed -- what the reference's M-step differentiates
        (experiments/uci_regression.py:159-160: Adam on ``training_loss_closure``; TensorFlow autodiff there).

        With g0 = d ve/d mean and g1 = d ve/d var at the current parameters, the chain rule gives
            d ELBO = scale * sum_np (g0 d mean + g1 d var) - d KL,   mean = k^T beta,  var = kff - k^T Q k,
        which splits into an N-sized contraction with dK_fu (HIP: fill, moments, the site sums a1 = sum g0 k,
        A2 = sum g1 k k^T, U = K_fu Q, and ``tsvgp_kernel_grad``) and an M x M part in which a1, A2 are constants
        (torch autograd over K_uu(theta, Z), its factorisations and the KL).
        Returns (elbo, {"variance", "lengthscales", "Z", "likelihood_variance" (Gaussian only)}), gradients of the ELBO
        with respect to the constrained parameter values; with one kernel per latent (``SeparateIndependent``) the kernel
        entries are named "kernels.<p>.variance" / "kernels.<p>.lengthscales".  With more than one rank ``data`` is this
        rank's shard."""
        sep = isinstance(self.kernel, SeparateIndependent)
        X, Y = self._as_device(data[0]), self._as_device(data[1])
        eng, P, M = self._get_engine(), self.num_latent_gps, self.num_inducing
        Dn = X.shape[1]
        ops = self._site_operands()
        Dm, beta = ops["D"], ops["beta"]
        gaussian = self.likelihood.lik_id == B.LIK_GAUSSIAN
        kernels = list(self.kernel.kernels) if sep else [self.kernel]
        nk = len(kernels)
        Q = Dm.transpose(-1, -2) @ Dm  # [P, M, M]
        dvar = torch.zeros(nk, dtype=torch.float64, device=self.device)
        dls = torch.zeros((nk, Dn), dtype=torch.float64, device=self.device)
        dZ = torch.zeros((M, Dn), dtype=torch.float64, device=self.device)
        sum_g1 = torch.zeros(nk, dtype=torch.float64, device=self.device)
        res = torch.zeros((), dtype=torch.float64, device=self.device)
        # One N-pass per kernel: a shared kernel serves all P latents from one K(X, Z); separate kernels take their
        # latent alone, through the same buffers (as EStepEngine._run_separate).
        # The TRUE d ve / d var here: the crop of tsvgp.py:262-263 belongs to the site update, not to the ELBO (with the
        # 1e-3 jitter of the probit link log p is not log-concave in the far tails, so some g1 are positive)
        parts = []
        for ki, kern in enumerate(kernels):
            lat = [ki] if sep else list(range(P))
            sl = slice(lat[0], lat[-1] + 1)
            # One latent: the moments' own triangular product t_n = D k_n is KEPT (tsvgp_trmm) and Q k_n = D^T t_n is a second
            # triangular product of it -- 2 N M^2 flops for moments + U instead of the 3 N M^2 of the fused moments kernel plus a
            # dense GEMM with Q = D^T D (round 5; 16.8 + 28.1 ms -> 15.5 + ~5 + 16 ms at N = 1e6, M = 1024).
            tile_path = len(lat) == 1 and hasattr(eng, "trmm") and os.environ.get("TSVGP_MSTEP_TILE", "1") != "0"
            st = eng.run(X, Y[:, sl], ops["Z"], kern, moment_Tm=Dm[sl], moment_mode=ops["moment_mode"], gamma=beta[:, sl],
                         lik_id=self.likelihood.lik_id | B.LIK_NOCROP, lik_param=self.likelihood.lik_param, sites=True,
                         want_moments=gaussian, **({"keep_tile": True} if tile_path else {}))
            parts.append(st)
            Kfu, g0, g1 = eng._buf["Kfu"], eng._buf["g0"], eng._buf["g1"]  # [Np, Mp], [Np, len(lat)] (rows >= N are zero)
            Ubuf = eng._get("U", tuple(Kfu.shape), Kfu.dtype)
            for c, p_ in enumerate(lat):
                if tile_path:
                    # U[n, m] = sum_{i <= m} t[n, i] D[i, m]: the lower-form product with D^T
                    eng.trmm(st.tile, eng._pad_square(Dm[p_].transpose(-1, -2).contiguous(), Kfu.shape[1], "pad_Dt"), Ubuf, B.TRI_LOWER)
                    v, l, z = eng.kernel_grad(X, ops["Z"], kern, Ubuf, g0[:, c], g1[:, c], beta[:, p_])
                    dvar[ki] += v
                    dls[ki] += l
                    dZ += z
                    continue
                # U = K_fu Q_p: a plain dense GEMM, so it goes to the BLAS library (rocBLAS DGEMM holds 0.95 of the fp64 MFMA
                # peak at this shape, tools/dgemm_ref.py: 28.1 ms at N = 1e6, M = 1024 against 33.8 for tsvgp_trmm's dense mode)
                torch.mm(Kfu, eng._pad_square(0.5 * (Q[p_] + Q[p_].T), Kfu.shape[1], "pad_Q"), out=Ubuf)
                v, l, z = eng.kernel_grad(X, ops["Z"], kern, Ubuf, g0[:, c], g1[:, c], beta[:, p_])
                dvar[ki] += v
                dls[ki] += l
                dZ += z
            sum_g1[ki] = g1.sum(dtype=torch.float64)
            if gaussian:  # d ve / d s2 = -1/(2 s2) + ((y - m)^2 + v) / (2 s2^2), summed
                res = res + torch.sum((Y[:, sl].to(st.mean.dtype) - st.mean) ** 2 + st.var)
        if sep:
            st = EStepStats(n_rows=parts[0].n_rows, ve_sum=sum(s_.ve_sum for s_ in parts),
                            nonpos=sum(s_.nonpos for s_ in parts))
            st.acc2, st.acc1 = torch.cat([s_.acc2 for s_ in parts], dim=0), torch.cat([s_.acc1 for s_ in parts], dim=0)
        extra = torch.cat([dvar, dls.reshape(-1), dZ.reshape(-1), sum_g1, res.reshape(1)])
        acc2, acc1, ve_sum, nonpos, rows, tail = D_.reduce_stats(st, P, M, True, self._reduce(), eng, extra=extra)
        o = 0
        dvar, o = tail[o:o + nk], o + nk
        dls, o = tail[o:o + nk * Dn].reshape(nk, Dn), o + nk * Dn
        dZ, o = tail[o:o + M * Dn].reshape(M, Dn), o + M * Dn
        sum_g1, o = tail[o:o + nk], o + nk
        res = tail[o]
        self._check_step(ops, nonpos)
        scale = (float(self.num_data) / rows) if self.num_data is not None else 1.0

        # M x M part: a1, A2, sum g1 are constants here.  With K = K_uu + 1e-6 I, W = I + L^T K L, Q = L W^-1 L^T = D^T D and
        # beta = l1 - Q K l1 (util.py:168-179, tsvgp.py:65-70):   dQ = -Q dK Q,   d beta = -Q dK beta,   d log|W| = tr(Q dK),  so
        #   d KL            = 1/2 [ beta^T dK beta - 2 (Q K beta)^T dK beta + tr(Q K Q dK) ]
        #   d (beta^T a1 - tr(Q A2)) = -beta^T dK (Q a1) + tr(Q A2 Q dK)
        # i.e. one symmetric M x M matrix G = d surrogate / d K per latent from six GEMMs on the prelude's own Q, beta -- and autograd
        # only through the elementwise K(Z, Z; theta) (round 5; the factorisation and the triangular solve used to sit inside the
        # autograd graph: ~4.5 ms of rocSOLVER / rocBLAS launches per evaluation, forward and backward).  TSVGP_MSTEP_AUTOGRAD=1: that
        # form, kept as the cross-check of tests/test_gpu_mstep.py.
        var_t = [k.variance.value.detach().to(self.device).clone().requires_grad_(True) for k in kernels]
        ls_t = [k.lengthscales.value.detach().to(self.device).clone().requires_grad_(True) for k in kernels]
        Z_t = self._Z().detach().clone().requires_grad_(True)
        l1, L = self.lambda_1.value.detach(), self.lambda_2_sqrt.value.detach()
        Id = ops["Id"]
        if os.environ.get("TSVGP_MSTEP_AUTOGRAD", "0") == "1":
            with torch.enable_grad():
                K6 = torch.stack([k.K_torch(Z_t, v_, l_) for k, v_, l_ in zip(kernels, var_t, ls_t)]) + default_jitter() * Id
                K6 = K6.expand(P, M, M)  # one shared kernel: the same matrix for every latent
                W = Id + L.transpose(-1, -2) @ (K6 @ L)
                cW = torch.linalg.cholesky(0.5 * (W + W.transpose(-1, -2)))
                Dt = torch.linalg.solve_triangular(cW, L.transpose(-1, -2), upper=False)
                Qt = Dt.transpose(-1, -2) @ Dt
                K6l = torch.einsum("pmk,kp->mp", K6, l1)
                beta_t = l1 - torch.einsum("pmk,kp->mp", Qt, K6l)
                kl = 0.5 * (torch.einsum("mp,pmk,kp->", beta_t, K6, beta_t) - torch.sum(Qt * K6)
                            + 2.0 * torch.sum(torch.log(torch.diagonal(cW, dim1=-2, dim2=-1))))
                knn = sum(v_ * s_ for v_, s_ in zip(var_t, sum_g1))  # sum_np g1 * k(x, x), k(x, x) = variance
                surrogate = scale * (torch.sum(beta_t * acc1.transpose(-1, -2)) - torch.sum(Qt * acc2) + knn) - kl
                g_all = torch.autograd.grad(surrogate, var_t + ls_t + [Z_t])
            kl = kl.detach()
        else:
            K6c = ops["K6"].expand(P, M, M)
            outer = lambda a, b: torch.einsum("mp,kp->pmk", a, b)  # [M, P] x [M, P] -> [P, M, M]
            QK = Q @ K6c
            QKb = torch.einsum("pmk,kp->mp", QK, beta)
            Qa1 = torch.einsum("pmk,pk->mp", Q, acc1)
            G_kl = 0.5 * (outer(beta, beta) - outer(QKb, beta) - outer(beta, QKb) + QK @ Q)
            G_t = (Q @ acc2) @ Q - 0.5 * (outer(beta, Qa1) + outer(Qa1, beta))
            G = scale * G_t - G_kl
            G = 0.5 * (G + G.transpose(-1, -2))
            if not sep:
                G = G.sum(dim=0, keepdim=True)  # one kernel behind every latent
            kl = kl_from_dense_site(ops["K6"], l1, Dm, ops["U_W"], beta)
            with torch.enable_grad():
                Kt = torch.stack([k.K_torch(Z_t, v_, l_) for k, v_, l_ in zip(kernels, var_t, ls_t)])
                knn = sum(v_ * s_ for v_, s_ in zip(var_t, sum_g1))  # sum_np g1 * k(x, x), k(x, x) = variance
                g_all = torch.autograd.grad(torch.sum(G.detach() * Kt) + scale * knn, var_t + ls_t + [Z_t])
        g_var, g_ls, g_Z = g_all[:nk], g_all[nk:2 * nk], g_all[-1]
        grads = {"Z": g_Z + scale * dZ}
        for ki, k in enumerate(kernels):
            ls_shape = k.lengthscales.value.shape
            n_ls = scale * dls[ki]
            pre = f"kernels.{ki}." if sep else ""
            grads[pre + "variance"] = g_var[ki] + scale * dvar[ki]
            grads[pre + "lengthscales"] = g_ls[ki] + (n_ls.sum() if len(ls_shape) == 0 else n_ls.reshape(ls_shape))
        if gaussian:
            s2 = self.likelihood.lik_param
            grads["likelihood_variance"] = scale * (-0.5 * rows * P / s2 + 0.5 * res / (s2 * s2))
        elbo = ve_sum * scale - kl
        return elbo, grads

    # -- the hot path ------------------------------------------------------------------------------------------
    def natgrad_step(self, data, lr=0.1, jitter=1e-9):
        """One natural-gradient step on the site parameters (tsvgp.py:234-304):
            lambda <- (1 - lr) lambda + lr * scale * grad_mu E_q[log p(y | f)].
        ``data = (X [N, D], Y [N, P])``; with more than one rank, this rank's contiguous row shard.
        Updates the parameters in place and returns None.  The whole step is enqueued without host
        synchronisation; one device->host read of the status flags ends it."""
        X, Y = self._as_device(data[0]), self._as_device(data[1])
        routes = self._routes(jitter)
        if self._wants_graph(X) and self._graph_step(X, Y, lr, jitter, routes):
            return
        old_l1, old_L = self.lambda_1.value, self.lambda_2_sqrt.value
        while True:
            soft = any(r != "projected" for r in routes)
            try:
                flags = self._step_device(X, Y, lr, jitter, routes)
                verdict = self._judge(self._read_flags(flags), soft_final=soft)
            except FloatingPointError:
                self.lambda_1.assign(old_l1)  # the reference raises before its assigns: leave the state untouched
                self.sites.assign_lambda_2_sqrt(old_L)
                raise
            if verdict is True:
                return
            # The final factorisation failed on a cheaper route (its error exceeded the absolute jitter): put the state
            # back, move every latent one route down (direct -> whitened -> projected) and remember it until the
            # parameters change.  The projected route forms G1 as a sum of outer products like the reference, so a
            # failure there is a failure of the reference's own algorithm and raises.
            self.lambda_1.assign(old_l1)
            self.sites.assign_lambda_2_sqrt(old_L)
            routes = [self._DEMOTE.get(r, r) for r in routes]
            if self.projection == "auto" and self._cond_cache is not None:
                self._cond_cache[2].update({p: r for p, r in enumerate(routes)})

    def _step_front(self, X, Y, lr, jitter, routes):
        """M x M prelude and the N-pass of one E-step (everything in front of the all-reduce): returns (per-shard
        statistics, prelude operands).  No host synchronisation."""
        warm_key = self._warm_key(X, jitter)
        eng = self._get_engine()
        # the clock keeper bridges the M x M prelude to the N-pass (EStepEngine.keeper_begin); the epilogue has its own
        self._keep_clock = (self.device.type == "cuda" and getattr(eng, "clock_keeper", 0) != 0
                            and X.shape[0] * self.num_inducing >= self.KEEPER_MIN_NM)
        keeper = eng.keeper_begin() if self._keep_clock else None
        # K(X, Z) depends on neither lambda nor the M x M factors: its fill runs on a side stream beside the prelude.
        # (Starting it only behind the two GEMMs that assemble W -- they take 130-190 us each under the fill instead of 40 --
        # measured 0.1-0.2 ms SLOWER per step, and again 36.61 vs 36.46 ms after the fill and the factorisation were reworked:
        # the factorisation and the moments kernel behind a later fill lose more than the two GEMMs gain.)
        pre = ops = None
        # (inside a capture of a launch-bound size the fork / join costs a replay more than the overlap gains: in line there)
        fork = not (self.device.type == "cuda" and torch.cuda.is_current_stream_capturing()
                    and X.shape[0] * self.num_inducing < self.GRAPH_FORK_MIN_NM)
        Kzz = K6 = None
        if self.overlap_fill and fork and hasattr(eng, "start_fill"):
            # K(Z, Z) opens the M x M chain and is a launch of a few microseconds: it goes out BEFORE the N-sized fill, whose
            # workgroups otherwise take every CU first (the small fill then waited for slots: 54 us instead of ~8 at M = 1024)
            if not (warm_key is not None and self._warm is not None and self._warm[0] == warm_key):
                Kzz = eng.kuu(self._Z(), self.kernel)
                K6 = self._k6(Kzz)
            want = "Kfu" if all(r == "direct" for r in routes) else "B"
            if X.shape[0] * self.num_inducing <= self.FILL_INLINE_MAX_NM:
                # One rank's share of a large job, smaller still: the fill goes in line, right in front of the moments kernel
                # (``EStepEngine.run`` fills when it gets no ticket).  Beside the factorisation it gains nothing -- the two contend
                # one for one -- and as the last thing the chip does before the N-pass it hands the MFMA kernels a higher clock than
                # the latency-bound chain does (profiles/r05_clock_lab.txt, r05_fill_placement.txt).
                pass
            elif self._late_fill(X):
                # One rank's share of a large job: the fill is shorter than the factorisation chain, and its workgroups take every
                # CU slot from the two M^3 GEMMs that assemble W (126 + 129 us under the fill against 40 + 40 alone at 125 000 x
                # 1024, profiles/r05_v1_ns_mxm_timeline_rows125000.txt): it starts behind them and runs beside the factorisation.
                # (At N = 1e6 the fill outlasts the whole chain and starts first -- measured in round 3, docs/history_r01-r03.md.)
                box = {}
                ops = self._site_operands(whiten_jitter=jitter, warm_key=warm_key, routes=routes, fork=fork, Kzz=Kzz, K6=K6,
                                          after_w=lambda: box.update(pre=eng.start_fill(X, self._Z(), self.kernel, b_tag=warm_key,
                                                                                        want=want, routes=routes)),
                                          beside_fill=self._fill_outlasts_factorisation(X))
                pre = box.get("pre")
            else:
                pre = eng.start_fill(X, self._Z(), self.kernel, b_tag=warm_key, want=want, routes=routes)
        if ops is None:
            ops = self._site_operands(whiten_jitter=jitter, warm_key=warm_key, routes=routes, fork=fork, Kzz=Kzz, K6=K6,
                                      beside_fill=pre is not None)  # a long fill is already under way: see cholesky_solve_upper
        if keeper is not None:
            eng.keeper_end(keeper)
        st = eng.run(X, Y, ops["Z"], self.kernel, moment_Tm=ops["moment_Tm"], prefill=pre,
                     moment_mode=ops["moment_mode"], gamma=ops["gamma"],
                     lik_id=self.likelihood.lik_id, lik_param=self.likelihood.lik_param,
                     whiten_T=ops["whiten_T"], whiten_mode=ops["whiten_mode"], project_T=ops["project_T"],
                     moments_on_kfu=ops["moments_on_kfu"], project_mode=ops["project_mode"],
                     sites=True, b_tag=warm_key,
                     mean_only=self.skip_unused_variance and self.likelihood.lik_id == B.LIK_GAUSSIAN)
        return st, ops

    KEEPER_MIN_NM = int(os.environ.get("TSVGP_KEEPER_MIN_NM", "50000000"))  # N * M from which the M x M sections get a clock keeper
    FILL_INLINE_MAX_NM = int(os.environ.get("TSVGP_FILL_INLINE_MAX_NM", "0"))  # N * M up to which the fill runs in line in front of the moments
    LATE_FILL_MAX_NM = int(os.environ.get("TSVGP_LATE_FILL_MAX_NM", "300000000"))  # N * M up to which the fill starts behind W's GEMMs

    def _fill_outlasts_factorisation(self, X) -> bool:
        """A late fill of at least ~1.6 GB (0.4 ms and more) covers the whole factor-and-solve call of the prelude: round 4's block
        step, whose kernels share a CU with the fill's workgroups, is then the faster one (EStepEngine.cholesky_solve_upper,
        ``beside_fill``); measured, replayed steps on one box: 250 000 x 1024 fp64 9.90-9.98 -> 9.75-9.84 ms, 125 000 x 1024
        fp64 5.65 -> 5.63 (kept on the default step), 125 000 x 1024 fp32 3.59 -> 3.62 (worse)."""
        esize = 8 if self.compute_dtype == torch.float64 else 4
        return X.shape[0] * self.num_inducing * esize >= 1_600_000_000

    def _late_fill(self, X) -> bool:
        return X.shape[0] * self.num_inducing <= self.LATE_FILL_MAX_NM

    def _step_device(self, X, Y, lr, jitter, routes, inplace=False) -> torch.Tensor:
        """The whole E-step as device work, no host synchronisation: M x M prelude, N-pass, all-reduce, epilogue, state
        assignment.  Returns the status flags (device).  With ``inplace`` the state tensors are overwritten in place
        (what a captured graph needs) instead of being replaced."""
        if not inplace and self._latent_split(routes):
            return self._step_device_split(X, Y, lr, jitter, routes)
        st, ops = self._step_front(X, Y, lr, jitter, routes)
        return self._apply_site_update(st, ops, lr, jitter, inplace=inplace)

    # -- one kernel per latent over several ranks: the M x M work split over the latents (BASELINE configs[4]) ----
    def _latent_split(self, routes) -> bool:
        """With one kernel per latent (K_uu [P, M, M]) the replicated M x M prelude / epilogue is P factorisation pairs and a
        dozen GEMMs per latent -- ~20 ms of a 34 ms 8-way shard step at P = 8, M = 1024 -- so with several ranks every latent
        gets ONE owner (p mod world).  ``latent_split``: None = whenever that applies, False = never."""
        if self.latent_split is False or not isinstance(self.kernel, SeparateIndependent) or not self._reduce():
            return False
        if D_.world_size() < 2 and not D_.FORCE_COLLECTIVES:
            return False
        # "projected" latents factor K9 in the reference's elimination order inside a uniformly projected batch: not split
        return self.num_latent_gps >= 2 and all(r in ("direct", "whitened") for r in routes) and not self.cache_whitened

    def _step_device_split(self, X, Y, lr, jitter, routes) -> torch.Tensor:
        """One E-step with the latents' M x M algebra split over the ranks (reference src/models/tsvgp.py:249-254, 268-303 with
        K_uu [P, M, M]; docs/notebooks/heteroskedastic.py:62-76):
          1. prelude for the OWNED latents only (factorisations, D_p, beta_p, and U9_p^-1 where the latent is whitened);
          2. all-gather of what the N-pass needs of every latent: [D_p | gamma_p (| U9_p^-1)];
          3. the N-pass over this rank's rows for ALL latents (the batched launches);
          4. reduce-scatter of the packed accumulators BY LATENT (the owner receives the sums over all ranks' rows) + one
             tiny all-reduce of (sum ve, #var <= 0, rows);
          5. epilogue for the owned latents; all-gather of the new (lambda_1[:, p], lambda_2_sqrt[p]); the status words are
             max-reduced so that every rank raises or retries together.
        Returns the status flags (device); the state is assigned on every rank."""
        eng = self._get_engine()
        P, M = self.num_latent_gps, self.num_inducing
        G, r = max(D_.world_size(), 1), D_.rank()
        own = [p for p in range(P) if p % G == r]
        per = (P + G - 1) // G  # latents per rank, the last ranks padded with empty slots
        owner_order = [p for q in range(G) for p in range(q, P, G)]
        dev, f64 = self.device, torch.float64
        tri = M * (M + 1) // 2
        pre = None
        if self.overlap_fill and hasattr(eng, "start_fill"):
            pre = eng.start_fill(X, self._Z(), self.kernel, want="B", routes=routes)
        # 1. prelude of the owned latents (a rank without latents still takes part in the collectives)
        ops = self._site_operands(whiten_jitter=jitter, routes=[routes[p] for p in own], latents=own) if own else None
        # 2. all-gather of the N-pass operands, one fixed-size slot per latent: D | gamma | U9^-1 (zeros when direct)
        slot = 2 * M * M + M
        send = torch.zeros(per * slot, dtype=f64, device=dev)
        for i, p in enumerate(own):
            o = i * slot
            send[o:o + M * M] = ops["moment_Tm"][i].reshape(-1)
            send[o + M * M:o + M * M + M] = ops["gamma"][:, i]
            wt = eng._per_latent(ops["whiten_T"], i) if hasattr(eng, "_per_latent") else (
                None if ops["whiten_T"] is None else (ops["whiten_T"][i] if isinstance(ops["whiten_T"], (list, tuple)) else
                                                      (ops["whiten_T"][i] if ops["whiten_T"].dim() == 3 else ops["whiten_T"])))
            if routes[p] != "direct" and wt is not None:
                send[o + M * M + M:o + slot] = wt.reshape(-1)
        allops = D_.all_gather_flat(send).reshape(G, per, slot)
        Tm_full = torch.empty((P, M, M), dtype=f64, device=dev)
        gamma_full = torch.empty((M, P), dtype=f64, device=dev)
        whiten_full = [None] * P
        for p in range(P):
            blk = allops[p % G, p // G]
            Tm_full[p] = blk[:M * M].reshape(M, M)
            gamma_full[:, p] = blk[M * M:M * M + M]
            if routes[p] != "direct":
                whiten_full[p] = blk[M * M + M:].reshape(M, M)
        if all(wt is None for wt in whiten_full):
            whiten_full = None
        # 3. the N-pass: every latent over this rank's rows
        st = eng.run(X, Y, self._Z(), self.kernel, moment_Tm=Tm_full, prefill=pre, moment_mode=B.TRI_UPPER, gamma=gamma_full,
                     lik_id=self.likelihood.lik_id, lik_param=self.likelihood.lik_param, whiten_T=whiten_full,
                     whiten_mode=B.TRI_UPPER, project_T=None, sites=True,
                     mean_only=self.skip_unused_variance and self.likelihood.lik_id == B.LIK_GAUSSIAN)
        # 4. reduce-scatter by latent (slots in owner order) + the three scalars
        blk = tri + M
        packed = torch.zeros(G * per * blk, dtype=f64, device=dev)
        acc2, acc1 = st.acc2.to(f64), st.acc1.to(f64)
        i_, j_ = torch.tril_indices(M, M, device=dev)
        for q in range(G):
            for i, p in enumerate(range(q, P, G)):
                o = (q * per + i) * blk
                packed[o:o + tri] = acc2[p][i_, j_]
                packed[o + tri:o + blk] = acc1[p]
        mine = D_.reduce_scatter_sum(packed, G)
        scal = torch.stack([st.ve_sum.reshape(()).to(f64), st.nonpos.reshape(()).to(f64),
                            torch.full((), float(st.n_rows), dtype=f64, device=dev)])
        D_.all_reduce_sum(scal)
        nonpos, rows = scal[1], scal[2]
        # 5. epilogue of the owned latents, then the new state to every rank
        state = torch.zeros(per * blk, dtype=f64, device=dev)
        flags = torch.zeros(3, dtype=f64, device=dev)
        if own:
            a2 = torch.zeros((len(own), M, M), dtype=f64, device=dev)
            a1 = torch.empty((len(own), M), dtype=f64, device=dev)
            for i in range(len(own)):
                v = mine[i * blk:i * blk + tri]
                a2[i][i_, j_] = v
                a2[i][j_, i_] = v
                a1[i] = mine[i * blk + tri:(i + 1) * blk]
            flags, l1_new, L_new = self._apply_site_update(None, ops, lr, jitter, reduced=(a2, a1, nonpos, rows), latents=own)
            for i in range(len(own)):
                state[i * blk:i * blk + tri] = L_new[i][i_, j_]
                state[i * blk + tri:(i + 1) * blk] = l1_new[:, i]
        else:
            flags[1] = nonpos
        allstate = D_.all_gather_flat(state).reshape(G, per, blk)
        L_full = torch.zeros((P, M, M), dtype=f64, device=dev)
        l1_full = torch.empty((M, P), dtype=f64, device=dev)
        for p in range(P):
            sb = allstate[p % G, p // G]
            L_full[p][i_, j_] = sb[:tri]
            l1_full[:, p] = sb[tri:]
        self.lambda_1.assign_owned(l1_full)
        self.sites.assign_lambda_2_sqrt(L_full, lower_and_owned=True)
        return D_.all_reduce_max(flags.abs())

    # -- hipGraph replay of the step (launch-bound problem sizes) -------------------------------------------------
    GRAPH_AUTO_MAX_NM = int(os.environ.get("TSVGP_GRAPH_AUTO_MAX_NM", "200000000"))  # "auto": replay when N * M is at most this (tools/bench_graph_sizes.py, below)
    # A captured step forks the K(X, Z) fill and the epilogue operands onto the side stream from this N * M on; below it they
    # are captured in line: the cross-stream edges cost a replay 0.12-0.15 ms, more than the overlap gains at launch-bound
    # sizes -- replayed step, forked / in line (gpurun_out/r3m/c1_routes.txt, graph_fork.txt, graph_fork_sizes.txt):
    # N = 1000, M = 32: 0.46 / 0.31 ms; 5000 x 128: 0.44 / 0.38; 62 500 x 1024: 3.91 / 3.83; 125 000 x 1024: 5.96 / 5.97;
    # 250 000 x 1024: 9.94 / 10.02.
    GRAPH_FORK_MIN_NM = int(os.environ.get("TSVGP_GRAPH_FORK_MIN_NM", "100000000"))
    # Memory cost of use_graph: a captured step owns its own K(X, Z), partial-tile and operand buffers beside the eager path's
    # (8 N M bytes + ~0.5 GB in fp64: 1.0 GB at 125 000 x 1024, 1.6 GB at the "auto" limit).  Up to four captures are kept for
    # small problems (alternating minibatches, lr schedules); from this N * M on only the latest one (see _graph_step).
    GRAPH_SINGLE_MIN_NM = 10_000_000

    def _wants_graph(self, X) -> bool:
        """use_graph = True / False, or "auto" (the default): replay from a captured graph where that is faster.
        Enqueueing a step eagerly costs the host ~5.9 ms of Python and launch calls (about 120 dispatches), so below that much
        GPU time the step is host bound: eager -> replay at M = 1024, D = 8, fp64 (tools/bench_graph_sizes.py,
        profiles/r03_graph_vs_eager_by_shard_size.txt): 62 500 rows 5.94 -> 4.09 ms, 125 000 (one rank's share of an 8-way
        shard of the headline workload) 6.26 -> 6.15, 250 000 10.22 -> 10.25, 500 000 18.28 -> 18.27, 1e6 34.11 -> 34.00;
        small problems: N = 1000, M = 32: 981 -> 1765 E-steps/s, 5000 x 128: 1206 -> 2035.  Since round 3 a captured step
        forks the K(X, Z) fill and the epilogue's operands onto the side stream INSIDE the capture (event fork / join), so a
        replay no longer serialises what an eager step overlaps and never loses; "auto" still stops at N * M = 2e8 because a
        graph owns its work buffers (K(X, Z) among them: 1.6 GB there) beside the eager path's."""
        if self.use_graph == "auto":
            return X.shape[0] * self.num_inducing <= self.GRAPH_AUTO_MAX_NM
        return bool(self.use_graph)

    def _graph_step(self, X, Y, lr, jitter, routes) -> bool:
        """Runs the step by replaying a captured graph (torch.cuda.CUDAGraph = hipGraph on ROCm).  A step is ~130
        dispatches; at small N and M (BASELINE configs[0]) launching them costs more than running them.  The graph is
        keyed on everything that is baked into it at capture: the data buffers, kernel / likelihood / Z parameter
        versions (scalars travel as kernel arguments), lr, jitter and the projection routes.  The first occurrence of a
        key runs eagerly (library handles, buffers), the second captures, later ones replay.  Returns False when the
        caller should run eagerly (not capturable, first occurrence, or the replayed step failed its status check: the
        state has been restored and the eager path raises or retries exactly as without graphs)."""
        if self.device.type != "cuda":
            return False
        eng = self._get_engine()
        if self.cache_whitened or eng.profile is not None or isinstance(self.kernel, SeparateIndependent):
            return False
        # With several ranks the step is TWO graphs around the all-reduce of the packed accumulators: everything in
        # front of it (prelude, fill, moments, site sums, packing) and everything behind it (unpacking, epilogue, state
        # update, status words); the collective itself is issued between the two replays on the same stream.
        two = self._reduce()
        # scalars (likelihood variance, kernel variance) travel as kernel ARGUMENTS and Z is converted into a capture-owned
        # buffer, so every parameter is keyed on Parameter.stamp(): identity + assign counter + the tensor's own edit counter
        # (an in-place edit of .value -- likelihood.variance.value.mul_(2), Z.value.add_(..) -- never passes through assign)
        lik_v = tuple(p.stamp() for p in vars(self.likelihood).values() if hasattr(p, "stamp"))
        key = (X.data_ptr(), Y.data_ptr(), tuple(X.shape), tuple(Y.shape), X.dtype, Y.dtype, self._kernel_versions(),
               lik_v, self.inducing_variable.Z.stamp(), float(lr), float(jitter), tuple(routes), self.num_data)
        entry = self._graphs.get(key)
        if entry is None:
            if len(self._graphs) >= 64:  # ever-changing inputs (fresh minibatch tensors): keep the markers bounded
                self._graphs = {k: v for k, v in self._graphs.items() if isinstance(v, dict)}
            self._graphs[key] = "seen"
            return False
        l1p, Lp = self.lambda_1, self.sites._lambda_2_sqrt
        if entry == "seen":
            # each graph owns its work buffers (K(X, Z), the partial tiles, the operands: ~10 N M bytes): keep few, and above
            # GRAPH_SINGLE_MIN_NM only ONE -- every M-step changes the kernel versions and with them the key, so the previous
            # capture is dead weight (1.6-3.2 GB at the "auto" limit) the moment a new key is captured
            big = X.shape[0] * self.num_inducing > self.GRAPH_SINGLE_MIN_NM
            if len(self._graphs) > 4 or (big and any(isinstance(v, dict) for v in self._graphs.values())):
                self._graphs = {key: "seen"}
            sl1, sL = l1p.value.clone(), Lp.value.clone()  # static state tensors the graph reads and writes
            bl1, bL = torch.empty_like(sl1), torch.empty_like(sL)
            l1p._value, Lp._value = sl1, sL
            graph = torch.cuda.CUDAGraph()
            saved_buf, saved_tag = eng._buf, eng._b_tag
            eng._buf, eng._b_tag = {}, None  # the graph's own work buffers (kept alive by the entry)
            try:
                if not two:
                    with torch.cuda.graph(graph):
                        bl1.copy_(sl1)
                        bL.copy_(sL)
                        flags = self._step_device(X, Y, lr, jitter, routes, inplace=True)
                    entry = dict(graph=graph, flags=flags, state=(sl1, sL), backup=(bl1, bL), buf=eng._buf)
                else:
                    P_, M_ = self.num_latent_gps, self.num_inducing
                    with torch.cuda.graph(graph):
                        bl1.copy_(sl1)
                        bL.copy_(sL)
                        st, ops = self._step_front(X, Y, lr, jitter, routes)
                        if ops.get("epi_event") is not None:  # the side stream's branch ends inside THIS graph
                            torch.cuda.current_stream(self.device).wait_event(ops["epi_event"])
                            ops["epi_event"] = None
                        packed = D_.pack_stats(st, True, eng)
                    tail = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(tail, pool=graph.pool()):
                        acc2, acc1, _, nonpos, rows = D_.unpack_stats(packed, P_, M_, True, eng)
                        flags = self._apply_site_update(None, ops, lr, jitter, inplace=True,
                                                        reduced=(acc2, acc1, nonpos, rows))
                    entry = dict(graph=graph, tail=tail, packed=packed, ops=ops, flags=flags, state=(sl1, sL),
                                 backup=(bl1, bL), buf=eng._buf)
                self._graphs[key] = entry
            except RuntimeError as exc:
                # What torch / HIP raise when an operation is not permitted under stream capture: never try this key again
                # and say so once (anything else -- a TypeError, a KeyError -- is a bug and propagates).  With several ranks
                # the outcome needs no agreement: a rank that runs eagerly issues the same ONE all-reduce of the same
                # packed buffer per step as a rank that replays its two graphs around it.
                warnings.warn(f"natgrad_step: hipGraph capture failed, running eagerly for this configuration ({exc})",
                              RuntimeWarning, stacklevel=3)
                self._graphs[key] = "seen-uncapturable"
                entry = None
            finally:
                eng._buf, eng._b_tag = saved_buf, saved_tag
            if entry is None:
                return False
        elif not isinstance(entry, dict):
            return False
        sl1, sL = entry["state"]
        if l1p._value is not sl1:  # the state was replaced since (an eager step, a user assign): bring it in
            sl1.copy_(l1p.value)
            l1p._value = sl1
        if Lp._value is not sL:
            sL.copy_(Lp.value)
            Lp._value = sL
        entry["graph"].replay()
        if "tail" in entry:
            D_.all_reduce_sum(entry["packed"])
            entry["tail"].replay()
        try:
            ok = self._judge(self._read_flags(entry["flags"]), soft_final=any(r != "projected" for r in routes)) is True
        except FloatingPointError:
            ok = False
        if not ok:  # put the pre-step state back; the eager path then raises or falls back as it always does
            sl1.copy_(entry["backup"][0])
            sL.copy_(entry["backup"][1])
            return False
        l1p.version += 1
        Lp.version += 1
        return True

    def _map_sums(self, ops, acc2, acc1):
        """The accumulators of the N-pass in the coordinates their route took them in -> the reference's
        G1 = sum_n g1 a a^T [P, M, M] and G0 = sum_n g0 a [M, P] with a = K9^-1 k (tsvgp.py:278-281)."""
        Uinv9 = ops["Uinv9"]
        Uinv9t = Uinv9.transpose(-1, -2) if Uinv9 is not None else None  # None: projected route on the lower factor
        routes = ops["routes"]
        forms = {}
        if "direct" in routes:
            # direct projection: acc2 = sum g1 k k^T, acc1 = sum g0 k  ->  G1 = K9^-1 acc2 K9^-1, G0 = K9^-1 acc1
            # (K9^-1 = U9^-T U9^-1 from the prelude, applied as GEMMs; torch.cholesky_solve is not an option: it returned
            # wrong values for small batched right-hand sides on this ROCm build -- tools/check_cholesky_solve.py)
            K9inv = ops["K9inv"]
            forms["direct"] = (K9inv @ acc2 @ K9inv, self._kmv(K9inv, acc1.transpose(-1, -2).contiguous()))
        if "whitened" in routes:
            # G1 = U9^-T acc2 U9^-1,  G0 = U9^-T acc1   (tsvgp.py:279-280 in whitened coordinates)
            forms["whitened"] = (Uinv9t @ acc2 @ Uinv9, _kmv(Uinv9t, acc1.transpose(-1, -2)))
        if "projected" in routes:
            # the sums were taken over a = K9^-1 k itself: they ARE G1 and G0 (tsvgp.py:279-280), a sum of outer products
            forms["projected"] = (acc2, acc1.transpose(-1, -2))
        if len(forms) == 1:
            G1, G0 = next(iter(forms.values()))
        else:  # mixed routes (separate kernels): every latent keeps the form its sums were taken in
            G1 = torch.stack([forms[r][0][p] for p, r in enumerate(routes)])
            G0 = torch.stack([forms[r][1][:, p] for p, r in enumerate(routes)], dim=1)
        return G1, G0

    def site_sums(self, data, jitter=1e-9):
        """(G0 [M, P], G1 [P, M, M]) of tsvgp.py:279-280 at the CURRENT state -- the N-pass of one E-step on the route
        ``natgrad_step`` would take, mapped back to the reference's coordinates, without updating the state.  Not part of the
        reference's API: what full-size parity checks (tests/, bench.py's ``state_match``) compare with the CPU restatement's einsums."""
        X, Y = self._as_device(data[0]), self._as_device(data[1])
        routes = self._routes(jitter)
        st, ops = self._step_front(X, Y, 0.0, jitter, routes)
        acc2, acc1, _, nonpos, _, _ = D_.reduce_stats(st, self.num_latent_gps, self.num_inducing, True, self._reduce(),
                                                      self._get_engine())
        if ops.get("epi_event") is not None:
            torch.cuda.current_stream(self.device).wait_event(ops["epi_event"])
        G1, G0 = self._map_sums(ops, acc2, acc1)
        self._check_step(ops, nonpos)
        return G0, 0.5 * (G1 + G1.transpose(-1, -2))

    def _apply_site_update(self, st, ops, lr, jitter, inplace=False, reduced=None, latents=None):
        """All-reduce of the packed accumulators (RCCL) + the replicated M x M epilogue (tsvgp.py:278-303).
        Returns the status flags (device tensor, see ``_status_flags``).  ``reduced``: the already summed
        (acc2, acc1, nonpos, rows) when the caller did the collective itself (the two-graph replay)."""
        eng = self._get_engine()
        keeper = eng.keeper_begin() if getattr(self, "_keep_clock", False) else None  # to the next step's prelude
        try:
            return self._site_update_body(st, ops, lr, jitter, inplace, reduced, latents)
        finally:
            if keeper is not None:
                eng.keeper_end(keeper)

    def _site_update_body(self, st, ops, lr, jitter, inplace, reduced, latents):
        P, M = self.num_latent_gps, self.num_inducing
        eng = self._get_engine()
        if reduced is not None:
            acc2, acc1, nonpos, rows = reduced
        else:
            acc2, acc1, _, nonpos, rows, _ = D_.reduce_stats(st, P, M, True, self._reduce(), eng)
        if ops.get("epi_event") is not None:  # L L^T, K_uu beta, K9^-1 from the side stream (see _site_operands)
            torch.cuda.current_stream(self.device).wait_event(ops["epi_event"])

        G1, G0 = self._map_sums(ops, acc2, acc1)
        # tsvgp.py:286-300 in one pass (tsvgp_site_target_f64): with lambda_2 = -1/2 L L^T the matrix to factor is
        #   -2 [(1 - lr) lambda_2 + lr scale G1] + jitter I = (1 - lr) L L^T - 2 lr scale G1 + jitter I;
        # L L^T of the old factor comes from the prelude, `rows` (the global number of rows) is a device scalar: the
        # minibatch scale of :286-291 needs no synchronisation
        l1_old = self.lambda_1.value
        if latents is not None:
            l1_old = l1_old.index_select(1, torch.as_tensor(list(latents), device=l1_old.device))
        fused = hasattr(eng, "site_update") and G1.is_cuda and os.environ.get("TSVGP_SITE_UPDATE", "1") != "0"
        if fused:
            # symmetrisation, the matrix of the final factorisation, the chain rule of util.py:429-438 and the convex update of
            # lambda_1 (tsvgp.py:284-297) in ONE launch (tsvgp_site_update_f64; it was a kernel, a gemv and ~10 elementwise launches)
            target, lambda_1 = eng.site_update(G1, G0, ops["LLt"], ops["meanZ"], l1_old, lr, jitter, rows, self.num_data)
        elif hasattr(eng, "site_target") and G1.is_cuda:
            target, G1 = eng.site_target(G1, ops["LLt"], 1.0 - lr, -2.0 * lr, jitter, rows, self.num_data)
            scale = (float(self.num_data) / rows) if self.num_data is not None else 1.0
        else:
            G1 = 0.5 * (G1 + G1.transpose(-1, -2))
            scale = (float(self.num_data) / rows) if self.num_data is not None else 1.0
            target = (1.0 - lr) * ops["LLt"] + (-2.0 * lr * scale) * G1
            target.diagonal(dim1=-2, dim2=-1).add_(jitter)
        if not fused:
            grad_mu = gradient_transformation_mean_var_to_expectation(ops["meanZ"], [G0, G1])  # tsvgp.py:284
            lambda_1 = (1 - lr) * l1_old + lr * scale * grad_mu[0]  # tsvgp.py:296
        final_info = []
        # tsvgp.py:300; the leading minus rides on the factorisation's triangle copy, which also leaves exact zeros above
        lambda_2_sqrt = cholesky_deferred(target, final_info, ops["potrf"], overwrite=True, scale=-1.0)
        if latents is not None:  # the owner's part of a latent-split step: the caller gathers and assigns
            return self._status_flags(ops, nonpos, final_info), lambda_1, lambda_2_sqrt
        if inplace:
            self.lambda_1.value.copy_(lambda_1)
            self.lambda_2_sqrt.value.copy_(lambda_2_sqrt)
        else:
            self.lambda_1.assign_owned(lambda_1)  # tsvgp.py:302
            self.sites.assign_lambda_2_sqrt(lambda_2_sqrt, lower_and_owned=True)  # tsvgp.py:303
        # tsvgp.py:304 recomputes the posterior and discards it: dead work, not reproduced.
        return self._status_flags(ops, nonpos, final_info)
