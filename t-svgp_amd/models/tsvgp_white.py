"""t-SVGP with the site in K-whitened coordinates: mirror of the reference's ``t_SVGP_white``
(reference src/models/tsvgp_white.py:23-246), SURVEY section 8(f) #1.

q(u) = N(m, S) with  S^-1 = K^-1 + K^-1 Lambda_2 K^-1,  S^-1 m = K^-1 lambda_1;  the state is lambda_1 [M, 1] and the
FULL Lambda_2 [1, M, M] (no Cholesky factor, no final factorisation in the step).  The N-sized work is the same as
``t_SVGP``'s -- K(X, Z) fill, a whitening product, fused moments + likelihood gradients, weighted Gram accumulation --
through the same HIP kernels; only the M x M prelude / epilogue differ:

  reference (tsvgp_white.py / util.py:11-88)              here  (K6 = Kuu + 1e-6 I = U6 U6^T, E = Lambda_2 + 1e-9 I = U_E U_E^T)
  -----------------------------------------------------   ----------------------------------------------------------------
  R = Lambda_2 + K6 + 1e-9 I;  LR = chol R, LA = chol K6   b = U6^-1 k  (HIP trmm);  H = U6^-1 U_E (upper);  I + H^T H = C C^T
  var  = kff - |LA^-1 k|^2 + |LR^-1 k|^2                   var  = kff - |T b|^2,  T = C^-1 H^T  (LOWER triangular)
  mean = k^T R^-1 lambda_1                                 mean = b^T gamma,  gamma = v - T^T T v,  v = U6^-1 lambda_1
  A = Kfu K9^-1;  G1 = sum g1 a a^T, G0 = sum g0 a          acc2 = sum g1 b b^T, acc1 = sum g0 b  (HIP syrk);  S2 = U6 acc2 U6^T
  lambda_1 += .. Kuu (G0 - 2 G1 meanZ)                      Kuu G1 Kuu = Mj S2 Mj^T,  Mj = Kuu K9^-1 = I - jitter K9^-1
  Lambda_2 += .. -2 Kuu G1 Kuu                              (the product with Mj is benign: no cancellation with K^-1)

Direct route (``projection="auto"`` takes it when cond(K6) <= DIRECT_MAX_COND): no N-sized whitening.  With
Q = K6^-1 - R^-1 = (T U6^-1)^T (T U6^-1) formed and factorised in M x M (Q = Lq Lq^T):  var = kff - |Lq^T k|^2, an upper
triangular product on K(X, Z) itself;  mean = k^T (U6^-T gamma);  the sums run over k k^T, which is S2 directly.  One
N M^2 product fewer (56 -> 40 ms at N = 1e6, M = 1024); forming Q squares cond(K6), hence the gate, and a failed
factorisation of Q sends the call back to the whitened route.

Two-product form (round 3): when Lambda_2 + 1e-9 I has no Cholesky factor -- the class does not crop d ve / d var
(tsvgp_white.py:188-191), so Lambda_2 can lose definiteness -- the table's single-product variance does not exist, while the
reference, which only factors K6 and R, goes on.  The call is then repeated (and the model stays) on the reference's own form:
var = kff - |b|^2 + |T2 b|^2 with T2 = U_R^-1 U6 (upper triangular), mean = b^T U6^T R^-1 lambda_1; |T2 b|^2 is one triangular
product of the moments kernel, |b|^2 a row reduction of B, and the likelihood map runs on the assembled moments
(``tsvgp_lik_map_*``, ``EStepEngine.run_two_product``).

The whitened table rests on  |LA^-1 k|^2 - |LR^-1 k|^2 = b^T (I - (I + G)^-1) b  with  G = U6^-1 E U6^-T = H H^T  and
I - (I + H H^T)^-1 = H (I + H^T H)^-1 H^T.  The reference's util functions take element [0] of a latent-batched product
(util.py:87, :425), so the class is only defined for ONE latent GP; more raise NotImplementedError.
"""
from __future__ import annotations

import functools

import numpy as np
import torch

from .. import _backend as B
from .. import distributed as D_
from ..base import default_jitter, to_tensor
from ..kernels import SeparateIndependent
from ..sites import DenseSites
from ..util import cholesky_deferred, cond2_estimate, info_sum, rev_cholesky
from .tsvgp import base_SVGP


class _DirectRouteFailed(Exception):
    """The direct projection's own M x M factorisation failed: the caller repeats the call on the whitened route."""


class _SingleProductFailed(Exception):
    """Lambda_2 + 1e-9 I (or I + H^T H built on its factor) is not positive definite while K_uu's factorisations went through:
    the single-product variance does not exist; the caller repeats the call with the reference's own two-product form, which
    only needs K + Lambda_2 + 1e-9 I (src/util.py:73-86)."""


class t_SVGP_white(base_SVGP):
    """Class for the t-SVGP model with whitened parameterization (reference tsvgp_white.py:23-246)."""

    def __init__(self, kernel, likelihood, inducing_variable, *, mean_function=None, num_latent_gps: int = 1,
                 lambda_1=None, lambda_2=None, num_data=None, compute_dtype=None, device=None, projection="auto"):
        super().__init__(kernel, likelihood, inducing_variable, mean_function=mean_function,
                         num_latent_gps=num_latent_gps, num_data=num_data, compute_dtype=compute_dtype, device=device)
        if projection not in ("auto", "whitened", "direct"):
            raise ValueError("projection must be 'auto', 'whitened' or 'direct'")
        # "whitened": b = U6^-1 k by an N-sized triangular product, moments and sums on b (any K_uu the reference can
        # factorise).  "direct": no N-sized whitening -- the moments act on k with the triangular factor of
        # Q = K6^-1 - R^-1 = (T U6^-1)^T (T U6^-1), formed and factorised in M x M, and the sums are taken over k k^T, which
        # is what the update needs (Kuu G1 Kuu = Mj (sum g1 k k^T) Mj^T).  Forming Q squares cond(K6), hence "auto":
        # direct when cond(K6) <= DIRECT_MAX_COND, and back to whitened if the factorisation of Q fails.
        self.projection = projection
        self._cond_cache = None
        self._direct_failed = False
        self._two_product = False  # set once Lambda_2 + 1e-9 I failed to factor: the reference's two-product variance from then on
        if isinstance(kernel, SeparateIndependent):
            raise NotImplementedError("t_SVGP_white takes one shared kernel (util.py:52-56 asserts Kuu [M, M])")
        self.num_inducing = self.inducing_variable.num_inducing
        self._init_variational_parameters(self.num_inducing, lambda_1, lambda_2)
        if self.num_latent_gps != 1:
            raise NotImplementedError("the reference's whitened util functions take element [0] of the latent batch "
                                      "(util.py:87, :425): only num_latent_gps = 1 is defined")
        self.name = "t_svgp_white"

    def _init_variational_parameters(self, num_inducing, lambda_1, lambda_2):
        """lambda_1 = 0, Lambda_2 = 1e-10 I (tsvgp_white.py:62-90)."""
        lambda_1 = np.zeros((num_inducing, self.num_latent_gps)) if lambda_1 is None else lambda_1
        if lambda_2 is None:
            lambda_2 = np.array([np.eye(num_inducing) * 1e-10 for _ in range(self.num_latent_gps)])
        else:
            lambda_2 = lambda_2.value if hasattr(lambda_2, "value") else lambda_2
            assert lambda_2.ndim == 3
            self.num_latent_gps = lambda_2.shape[0]
        self.sites = DenseSites(to_tensor(lambda_1, device=self.device), lambda_2=to_tensor(lambda_2, device=self.device))

    @property
    def lambda_1(self):
        return self.sites.lambda_1

    @property
    def lambda_2(self):
        """The full second natural parameter, a Parameter (tsvgp_white.py:96-97)."""
        return self.sites._lambda_2

    # -- M x M prelude -----------------------------------------------------------------------------------------
    def _Z(self):
        return self.inducing_variable.Z.value.to(self.device)

    def _as_device(self, a):
        if isinstance(a, torch.Tensor):
            return a.to(self.device)
        return torch.as_tensor(np.asarray(a)).to(self.device)

    def _operands(self, jitter=None, *, kuu_jitter=None, lambda_1=None, lambda_2=None, direct=False, two_product=False):
        """Everything the N-pass needs (see the table in the module docstring); with ``jitter`` also K9^-1 for the
        site update.  No host synchronisation: factorisation statuses go to ops["infos"].
        ``kuu_jitter`` (default: gpflow's default_jitter, as predict_f) is the jitter of the K_uu the conditional is built
        on; ``lambda_1`` / ``lambda_2`` replace the state (predict_f_extra_data)."""
        eng = self._get_engine()
        Z = self._Z()
        M = Z.shape[0]
        infos = []
        potrf = getattr(eng, "cholesky", None)
        if potrf is not None and self._cond_k6() > self.ROBUST_MIN_COND:
            # Lambda_2 + 1e-9 I and K_uu + jitter I are barely definite in fp64 on an ill-conditioned K_uu: the blocked
            # kernel's inverse-based panels can fail there where a factorisation by substitution goes through
            potrf = functools.partial(potrf, robust=True)
        Kzz = eng.kuu(Z, self.kernel)
        Id = torch.eye(M, dtype=torch.float64, device=Kzz.device)
        K6 = Kzz + (default_jitter() if kuu_jitter is None else kuu_jitter) * Id
        l1 = self.lambda_1.value if lambda_1 is None else lambda_1
        L2 = self.lambda_2.value if lambda_2 is None else lambda_2
        E = 0.5 * (L2 + L2.transpose(-1, -2)) + 1e-9 * Id  # util.py:76 (jitter argument default)
        if two_product:
            # The reference's own form (util.py:73-86): R = Lambda_2 + K6 + 1e-9 I = U_R U_R^T and K6 = U6 U6^T are all that is
            # factored; var = kff - |U6^-1 k|^2 + |U_R^-1 k|^2 = kff - |b|^2 + |T2 b|^2 with T2 = U_R^-1 U6 (upper triangular),
            # mean = k^T R^-1 lambda_1 = b^T gamma, gamma = U6^T R^-1 lambda_1.  Two N-sized reductions instead of one
            # (EStepEngine.run_two_product), no factor of Lambda_2 + 1e-9 I anywhere.
            mats = [K6[None], (E + K6)[None] if E.dim() == 2 else E + K6] + ([(Kzz + jitter * Id)[None]] if jitter is not None else [])
            U, Uinv = rev_cholesky(torch.cat(mats, dim=0), infos, potrf, inverse=True)
            U6, Uinv6, UinvR = U[0], Uinv[0], Uinv[1]
            T2 = (UinvR @ U6).triu()
            gamma = U6.transpose(-1, -2) @ (UinvR.transpose(-1, -2) @ (UinvR @ l1))
            ops = dict(Z=Z, Kzz=Kzz, K6=K6, Id=Id, infos=infos, e_infos=[], U6=U6, Uinv6=Uinv6, moment_Tm=T2[None], gamma=gamma,
                       moment_mode=B.TRI_UPPER, whiten_T=Uinv6, direct=False, direct_info=None, two_product=True)
            if jitter is not None:
                ops["K9inv"] = Uinv[2].transpose(-1, -2) @ Uinv[2]
            return ops
        # ONE batched factorisation call, its statuses in two lists: K_uu's (a failure there is the reference's failure too) and
        # those that exist only for the single-product variance (Lambda_2 + 1e-9 I here, I + H^T H below): see _check
        mats = [K6[None], E] + ([(Kzz + jitter * Id)[None]] if jitter is not None else [])
        all_infos = []
        U, Uinv = rev_cholesky(torch.cat(mats, dim=0), all_infos, potrf, inverse=True)
        infos = [torch.cat([all_infos[0][:1], all_infos[0][2:]])]
        e_infos = [all_infos[0][1:2]]
        U6, Uinv6, U_E = U[0], Uinv[0], U[1]
        H = (Uinv6 @ U_E).triu()
        Wm = Id + H.transpose(-1, -2) @ H
        _, Cinv = cholesky_deferred(0.5 * (Wm + Wm.transpose(-1, -2))[None], e_infos, potrf, inverse=True, overwrite=True)
        Tm = (Cinv[0] @ H.transpose(-1, -2)).tril()
        v = Uinv6 @ l1
        gamma = v - Tm.transpose(-1, -2) @ (Tm @ v)
        ops = dict(Z=Z, Kzz=Kzz, K6=K6, Id=Id, infos=infos, e_infos=e_infos, U6=U6, Uinv6=Uinv6, moment_Tm=Tm[None], gamma=gamma,
                   moment_mode=B.TRI_LOWER, whiten_T=Uinv6, direct=False, direct_info=None, two_product=False)
        if direct:
            # var = kff - k^T Q k with Q = U6^-T T^T T U6^-1 = K6^-1 - R^-1; Q = Lq Lq^T gives |Lq^T k|^2, an UPPER
            # triangular product on K(X, Z) itself; mean = k^T (U6^-T gamma)
            G = Tm @ Uinv6
            Q = G.transpose(-1, -2) @ G
            dinfo = []
            Lq = cholesky_deferred(Q[None], dinfo, potrf, overwrite=True)
            ops.update(moment_Tm=Lq.transpose(-1, -2).contiguous(), moment_mode=B.TRI_UPPER, whiten_T=None, direct=True,
                       gamma=Uinv6.transpose(-1, -2) @ gamma, direct_info=dinfo)
        if jitter is not None:
            ops["K9inv"] = Uinv[2].transpose(-1, -2) @ Uinv[2]
        return ops

    def _run(self, X, Y, ops, lik_id, sites=False, want_moments=False):
        if ops.get("two_product"):
            return self._get_engine().run_two_product(X, Y, ops["Z"], self.kernel, whiten_T=ops["whiten_T"],
                                                      moment_Tm=ops["moment_Tm"], gamma=ops["gamma"], lik_id=lik_id,
                                                      lik_param=self.likelihood.lik_param, sites=sites, want_moments=want_moments)
        return self._get_engine().run(X, Y, ops["Z"], self.kernel, moment_Tm=ops["moment_Tm"], moment_mode=ops["moment_mode"],
                                      gamma=ops["gamma"], lik_id=lik_id, lik_param=self.likelihood.lik_param,
                                      whiten_T=ops["whiten_T"], whiten_mode=B.TRI_UPPER, sites=sites,
                                      want_moments=want_moments)

    def _check(self, ops, nonpos):
        zero = torch.zeros(1, dtype=torch.float64, device=self.device)
        parts = [info_sum(ops["infos"]), nonpos.reshape(1).to(torch.float64),
                 info_sum(ops["direct_info"]) if ops.get("direct_info") else zero,
                 info_sum(ops["e_infos"]) if ops.get("e_infos") else zero]
        flags = self._read_flags(torch.cat(parts))
        if float(flags[0]) != 0:  # K_uu + jitter I (or, two-product form, K + Lambda_2 + 1e-9 I): the reference fails here too
            raise FloatingPointError("Cholesky decomposition was not successful (matrix not positive definite)")
        if float(flags[3]) != 0:  # only the single-product variance needs these factors: the two-product form goes on
            raise _SingleProductFailed()
        if float(flags[2]) != 0:  # Q lost definiteness in M x M: not the reference's failure
            raise _DirectRouteFailed()
        if not (float(flags[1]) == 0):  # tsvgp_white.py:131
            if ops.get("direct") and self.projection == "auto":
                raise _DirectRouteFailed()  # let the whitened route decide whether the variance really is non-positive
            raise FloatingPointError(f"non-positive predictive variance at {float(flags[1]):.0f} point(s)")

    # -- projection route ------------------------------------------------------------------------------------------
    DIRECT_MAX_COND = {torch.float64: 1.0e3, torch.float32: 30.0}

    ROBUST_MIN_COND = 1.0e6  # beyond this the factorisations run by substitution (EStepEngine.cholesky(robust=True))

    def _cond_k6(self) -> float:
        """cond(K_uu + default_jitter I), estimated (``util.cond2_estimate``) once per change of the kernel parameters or
        Z, taken from rank 0."""
        k, Zp = self.kernel, self.inducing_variable.Z
        key = (id(k), k.variance.stamp(), k.lengthscales.stamp(), Zp.stamp())
        if self._cond_cache is None or self._cond_cache[0] != key:
            eng = self._get_engine()
            Kzz = eng.kuu(self._Z(), k)
            Kzz.diagonal(dim1=-2, dim2=-1).add_(default_jitter())
            cond = cond2_estimate(Kzz, getattr(eng, "cholesky", None)).reshape(1)
            self._cond_cache = (key, float(D_.broadcast_from_rank0(cond.contiguous())))
            self._direct_failed = False
        return self._cond_cache[1]

    def _use_direct(self) -> bool:
        """"auto": direct when cond(K_uu + default_jitter I) is small and the route has not failed since."""
        if self.projection != "auto":
            return self.projection == "direct"
        return self._cond_k6() <= self.DIRECT_MAX_COND[self.compute_dtype] and not self._direct_failed

    def _routed(self, fn):
        """fn(direct=..., two_product=...) on the chosen form.  A direct attempt that fails in its own M x M algebra is repeated
        whitened; a single-product attempt whose factor of Lambda_2 + 1e-9 I does not exist is repeated with the reference's
        two-product variance (and the model stays on that form: an indefinite Lambda_2 tends to stay indefinite)."""
        if not self._two_product and self._use_direct():
            try:
                return fn(direct=True, two_product=False)
            except _DirectRouteFailed:
                if self.projection != "auto":
                    raise FloatingPointError("the direct projection lost definiteness; use projection='whitened'")
                self._direct_failed = True
            except _SingleProductFailed:
                self._two_product = True
        if not self._two_product:
            try:
                return fn(direct=False, two_product=False)
            except _SingleProductFailed:
                self._two_product = True
        return fn(direct=False, two_product=True)

    # -- reference API -----------------------------------------------------------------------------------------
    def get_mean_chol_cov_inducing_posterior(self):
        """posterior_from_dense_site_white (util.py:394-426): m = K (K + Lambda_2 + 1e-9 I)^-1 lambda_1, chol(S),
        S = K R^-1 K."""
        ops = self._operands()
        K6 = ops["K6"]
        R = K6 + self.lambda_2.value[0] + 1e-9 * ops["Id"]
        LR = torch.linalg.cholesky(R)
        iLRK = torch.linalg.solve_triangular(LR, K6, upper=False)
        S_q = iLRK.transpose(-1, -2) @ iLRK
        Rl = torch.linalg.solve_triangular(LR.transpose(-1, -2), torch.linalg.solve_triangular(
            LR, self.lambda_1.value, upper=False), upper=True)  # R^-1 lambda_1 (two explicit triangular solves)
        m_q = K6 @ Rl
        return m_q, torch.linalg.cholesky(S_q)[None]

    def prior_kl(self):
        """kl_from_precision_sites_white(K6, lambda_1, L2=Lambda_2) (util.py:239-291; tsvgp_white.py:116-120)."""
        eng = self._get_engine()
        Kzz = eng.kuu(self._Z(), self.kernel)
        Id = torch.eye(Kzz.shape[0], dtype=torch.float64, device=Kzz.device)
        A = Kzz + default_jitter() * Id
        R = self.lambda_2.value[0] + A
        LR, LA = torch.linalg.cholesky(R), torch.linalg.cholesky(A)
        log_det = 2.0 * (torch.sum(torch.log(torch.diagonal(LR))) - torch.sum(torch.log(torch.diagonal(LA))))
        tmp = torch.linalg.solve_triangular(LR, LA, upper=False)
        Rl = torch.linalg.solve_triangular(LR.transpose(-1, -2), torch.linalg.solve_triangular(
            LR, self.lambda_1.value, upper=False), upper=True)
        return 0.5 * (log_det + torch.sum(tmp * tmp) - float(A.shape[0]) + torch.sum(torch.square(LA.transpose(-1, -2) @ Rl)))

    def predict_f(self, Xnew, full_cov=False, full_output_cov=False):
        """tsvgp_white.py:122-132."""
        if full_cov or full_output_cov:
            raise NotImplementedError("full covariances are not on the E-step hot path")
        Xd = self._as_device(Xnew)

        def go(direct, two_product):
            ops = self._operands(direct=direct, two_product=two_product)
            st = self._run(Xd, None, ops, B.LIK_NONE, want_moments=True)
            self._check(ops, st.nonpos)
            return st.mean, st.var

        return self._routed(go)

    def predict_y(self, Xnew):
        return self.likelihood.predict_mean_and_var(*self.predict_f(Xnew))

    def elbo(self, data):
        """tsvgp_white.py:162-177; with more than one rank ``data`` is this rank's row shard."""
        X, Y = self._as_device(data[0]), self._as_device(data[1])
        kl = self.prior_kl()

        def go(direct, two_product):
            ops = self._operands(direct=direct, two_product=two_product)
            st = self._run(X, Y, ops, self.likelihood.lik_id | B.LIK_NOCROP)
            _, _, ve_sum, nonpos, rows, _ = D_.reduce_stats(st, self.num_latent_gps, self.num_inducing, False,
                                                              self._reduce(), self._get_engine())
            self._check(ops, nonpos)
            scale = (float(self.num_data) / rows) if self.num_data is not None else 1.0
            return ve_sum * scale - kl

        return self._routed(go)

    def _site_sums_k(self, X, Y, jitter=1e-9, direct=False, two_product=False):
        """The N-pass of compute_data_natural_params (tsvgp_white.py:183-206) in terms of k_n = K(Z, x_n):
        returns (S2 = sum g1 k k^T [1, M, M], s1 = sum g0 k [M, 1], K9^-1 meanZ [M, 1], rows, nonpos, ops) with
        K9 = K_uu + jitter I and meanZ = predict_f(Z) (:186)."""
        ops = self._operands(jitter=jitter, direct=direct, two_product=two_product)
        # tsvgp_white.py:188-191: no crop of d ve / d var in this class
        st = self._run(X, Y, ops, self.likelihood.lik_id | B.LIK_NOCROP, sites=True)
        acc2, acc1, _, nonpos, rows, _ = D_.reduce_stats(st, self.num_latent_gps, self.num_inducing, True, self._reduce(),
                                                         self._get_engine())
        U6, Uinv6, K9inv, Id = ops["U6"], ops["Uinv6"], ops["K9inv"], ops["Id"]
        if ops["direct"]:  # the sums were taken over k itself
            S2, s1, gamma_k = acc2, acc1.transpose(-1, -2), ops["gamma"]
        else:
            S2 = U6 @ acc2 @ U6.transpose(-1, -2)  # sum g1 k k^T   [1, M, M]
            s1 = U6 @ acc1.transpose(-1, -2)  # sum g0 k     [M, 1]
            gamma_k = Uinv6.transpose(-1, -2) @ ops["gamma"]  # R^-1 lambda_1
        a_meanZ = (Id - jitter * K9inv) @ gamma_k  # K9^-1 meanZ, meanZ = Kuu R^-1 lambda_1 (predict_f at Z, :186)
        return S2, s1, a_meanZ, rows, nonpos, ops

    def _kuu_grad_mu(self, X, Y, jitter=1e-9, kuu_jitter=0.0, direct=False, two_product=False):
        """compute_data_natural_params (tsvgp_white.py:183-212) with K_uu + kuu_jitter I already applied, which is how both
        callers use it: returns (K grad_mu[0] [M, 1], K grad_mu[1] K [1, M, M], rows, nonpos, ops).
        With s1 = sum g0 k, S2 = sum g1 k k^T and K9 = K_uu + jitter I:  grad_mu[0] = K9^-1 (s1 - 2 S2 K9^-1 meanZ),
        grad_mu[1] = K9^-1 S2 K9^-1, so K grad_mu = (I - (jitter - kuu_jitter) K9^-1)(...): no product with an
        ill-conditioned inverse is ever formed."""
        S2, s1, a_meanZ, rows, nonpos, ops = self._site_sums_k(X, Y, jitter, direct, two_product)
        Mj = ops["Id"] - (jitter - kuu_jitter) * ops["K9inv"]  # (Kuu + kuu_jitter I) K9^-1
        KG1K = Mj @ S2 @ Mj.transpose(-1, -2)  # K G1 K
        Kg0 = Mj @ s1 - 2.0 * (Mj @ (S2[0] @ a_meanZ))  # K (G0 - 2 G1 meanZ), util.py:429-438
        return Kg0, KG1K, rows, nonpos, ops

    def compute_data_natural_params(self, data, jitter=1e-9, nat_params=None):
        """The data term's gradient with respect to the expectation parameters at the inducing points
        (tsvgp_white.py:181-209):  returns [grad_mu[0] [M, 1], grad_mu[1] [1, M, M]] with, for A = K(X, Z) K9^-1
        (K9 = K_uu + jitter I, :196-199),  G0 = A^T g0, G1 = A^T diag(g1) A (:203-206) and
        grad_mu = [G0 - 2 G1 meanZ, G1] (util.py:429-438).  Unlike natgrad_step and predict_f_extra_data, which only ever
        need K grad_mu, this forms the products with K9^-1 themselves, as the reference does: the result carries
        cond(K9) times the rounding of the sums.  ``nat_params`` is accepted and ignored, as in the reference.  The state is
        not touched.  With more than one rank ``data`` is this rank's row shard (the sums are all-reduced)."""
        X, Y = self._as_device(data[0]), self._as_device(data[1])

        def go(direct, two_product):
            S2, s1, a_meanZ, _, nonpos, ops = self._site_sums_k(X, Y, jitter, direct, two_product)
            self._check(ops, nonpos)  # predict_f(X) asserts positivity (:131)
            K9inv = ops["K9inv"]
            G1 = K9inv @ S2 @ K9inv
            return [K9inv @ (s1 - 2.0 * (S2[0] @ a_meanZ)), 0.5 * (G1 + G1.transpose(-1, -2))]

        return self._routed(go)

    def natgrad_step(self, dataset, lr=0.1, jitter=1e-9):
        """One natural-gradient step on (lambda_1, Lambda_2) (tsvgp_white.py:183-248); returns None."""
        X, Y = self._as_device(dataset[0]), self._as_device(dataset[1])

        def go(direct, two_product):
            Kg0, KG1K, rows, nonpos, ops = self._kuu_grad_mu(X, Y, jitter=jitter, kuu_jitter=0.0, direct=direct,
                                                             two_product=two_product)  # :231
            scale = (float(self.num_data) / rows) if self.num_data is not None else 1.0
            lambda_1 = (1.0 - lr) * self.lambda_1.value + lr * scale * Kg0  # :244
            lambda_2 = (1.0 - lr) * self.lambda_2.value - 2.0 * lr * scale * KG1K  # :241-248 (Lambda_2 = -2 lambda_2)
            old_l1, old_L2 = self.lambda_1.value, self.lambda_2.value
            self.lambda_1.assign(lambda_1)
            self.sites.assign_lambda_2(0.5 * (lambda_2 + lambda_2.transpose(-1, -2)))
            try:
                self._check(ops, nonpos)
            except (FloatingPointError, _DirectRouteFailed, _SingleProductFailed):
                self.lambda_1.assign(old_l1)  # the reference raises before its assigns: leave the state untouched
                self.sites.assign_lambda_2(old_L2)
                raise

        self._routed(go)

    def predict_f_extra_data(self, Xnew, extra_data, jitter=None):
        """Prediction at Xnew conditioned on ``extra_data`` as well (tsvgp_white.py:134-160): the sites receive the
        extra points' natural-gradient contribution (a full step, no learning rate, no minibatch scale) for this call
        only; the state is not touched.  ``jitter`` (default: gpflow's default_jitter) is the jitter of the K_uu the
        contribution is mapped with and the conditional is built on (:146); the projection inside
        compute_data_natural_params keeps its own 1e-9 (:183).  With more than one rank ``extra_data`` is this rank's
        shard of the extra points."""
        jitter = default_jitter() if jitter is None else float(jitter)
        Xe, Ye = self._as_device(extra_data[0]), self._as_device(extra_data[1])
        Xn = self._as_device(Xnew)

        def go(direct, two_product):  # (the whitened forms only, as before: direct is not offered on this path)
            Kg0, KG1K, _, nonpos, ops = self._kuu_grad_mu(Xe, Ye, jitter=1e-9, kuu_jitter=jitter, two_product=two_product)  # :141
            self._check(ops, nonpos)  # predict_f(X) inside compute_data_natural_params asserts positivity (:131)
            return self.lambda_1.value + Kg0, self.lambda_2.value - 2.0 * KG1K  # :148-149

        def routed(fn):
            if not self._two_product:
                try:
                    return fn(False, False)
                except _SingleProductFailed:
                    pass
            return fn(False, True)

        lambda_1c, lambda_2c = routed(go)
        lambda_2c = 0.5 * (lambda_2c + lambda_2c.transpose(-1, -2))

        def cond(direct, two_product):  # the conditioned sites may be indefinite even when the model's own are not
            ops_c = self._operands(kuu_jitter=jitter, lambda_1=lambda_1c, lambda_2=lambda_2c, two_product=two_product)
            st = self._run(Xn, None, ops_c, B.LIK_NONE, want_moments=True)
            self._check(ops_c, torch.zeros(1, dtype=torch.float64, device=self.device))  # no assert_positive on this path (:155-158)
            return st.mean, st.var

        return routed(cond)
