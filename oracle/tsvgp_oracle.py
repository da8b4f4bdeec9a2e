"""CPU fp64 oracle for the t-SVGP natural-gradient E-step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``t-svgp_amd/`` (the product path) may
import this module; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker / timed
baseline -- never as the thing shipped.

What it is
----------
An op-for-op NumPy/SciPy restatement of the reference's hot path
(``/root/reference/src/models/tsvgp.py:234-304`` and the functions it calls),
*including the reference's redundancies* (cross-covariance built twice, the
posterior factorisation run three times, the materialised ``[N,M,P]`` tile) so
that it doubles as the timed CPU baseline.  Each function cites the reference
``file:line`` it follows.

The reference executes on TensorFlow 2.5 / GPflow 2.2.1, neither of which is
under ``/root/reference`` nor installed here (plain missing modules, nothing was
refused).  GPflow semantics marked ``[ext]`` below are restated from GPflow
2.2.1's published algorithms:

* ``SquaredExponential.K``  : ``variance * exp(-0.5 * r2)`` with
  ``r2 = |x/l|^2 + |z/l|^2 - 2 (x/l).(z/l)`` (``square_distance``; no clamp).
* ``covariances.Kuu``       : ``K(Z) + jitter * I``;  ``Kuf`` : ``K(Z, X)``.
* ``conditionals.base_conditional`` (white=False, q_sqrt 3-D).
* ``kullback_leiblers.gauss_kl`` (dense K, q_sqrt 3-D).
* ``likelihoods.Gaussian.variational_expectations`` (closed form) and
  ``likelihoods.Bernoulli`` (probit link with 1e-3 jitter, 20-point
  Gauss-Hermite quadrature ``NDiagGHQuadrature``), differentiated the way
  ``tf.GradientTape`` differentiates the quadrature sum.

Pinning status
--------------
The reference's tests hold no golden vectors; every assertion is relational.
This oracle is pinned (``tests/test_oracle_pins.py``) by the relational tests
that have closed forms independent of GPflow:
``tests/models/test_tsvgp.py:106-165`` (ELBO == exact GP log marginal
likelihood, predictions == exact GP posterior, fixed point, minibatch scale)
and ``tests/test_utils.py:44-137`` (site conditionals, Woodbury, site KL), plus
an independent autodiff natural-gradient SVGP for the Bernoulli case
(the shape of ``tests/models/test_tsvgp.py:123-131``).  Because the Bernoulli
comparison model is our own restatement of GPflow's SVGP + NaturalGradient and
not GPflow itself, and the reference holds no golden vectors: **parity unpinned**
by the rule -- parity of the non-conjugate site update against the real
reference is **unpinned beyond those checks**.

Also restated here, with the same status: ``t_SVGP_white``
(``src/models/tsvgp_white.py``; pinned by ``tests/models/test_tsvgp_white.py:64-115``
and the SGPR equality and extra-data conditioning of ``tests/models/test_condit.py:69-104``, closed forms),
the multi-output layout ``SeparateIndependent`` +
``SharedIndependentInducingVariables`` (GPflow's
``separate_independent_conditional`` and batched ``gauss_kl`` [ext]; pinned by
equality with independent single-output models) and the Matern-3/2 / 5/2
kernels [ext].  The oracle holds NO gradient code: the M-step gradients of the
product are checked against central finite differences of this module's ELBO
(the reference's own gradient pin, ``tests/models/test_tsvgp.py:168-188``, needs
GPflow's SVGP and cannot be run: **gradient parity is unpinned beyond those
differences**).
"""

from __future__ import annotations

import numpy as np
from scipy.linalg import cholesky as _sp_cholesky
from scipy.linalg import solve_triangular as _sp_solve_triangular
from scipy.special import erf as _erf

DEFAULT_JITTER = 1e-6  # gpflow.config.default_jitter() [ext]
_LOG_2PI = np.log(2.0 * np.pi)


# --------------------------------------------------------------------------
# small dense helpers (TensorFlow op equivalents)
# --------------------------------------------------------------------------
def _chol(a):
    """tf.linalg.cholesky: lower factor, batched over leading dims, reads the
    lower triangle only.  Raises FloatingPointError when not positive definite
    (TF raises InvalidArgumentError 'Cholesky decomposition was not successful')."""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 2:
        try:
            return _sp_cholesky(a, lower=True, check_finite=False)
        except np.linalg.LinAlgError as e:  # pragma: no cover - error path
            raise FloatingPointError("Cholesky decomposition was not successful") from e
    return np.stack([_chol(x) for x in a])


def _trsm(L, b, lower=True, adjoint=False):
    """tf.linalg.triangular_solve(L, b, lower, adjoint); L [.., M, M], b [.., M, K]
    with NumPy-style broadcasting over the leading (batch) dims."""
    L = np.asarray(L, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if L.ndim == 2 and b.ndim == 2:
        return _sp_solve_triangular(L, b, lower=lower, trans="T" if adjoint else "N", check_finite=False)
    batch = np.broadcast_shapes(L.shape[:-2], b.shape[:-2])
    Lb = np.broadcast_to(L, batch + L.shape[-2:])
    bb = np.broadcast_to(b, batch + b.shape[-2:])
    out = np.empty(batch + b.shape[-2:], dtype=np.float64)
    for idx in np.ndindex(*batch):
        out[idx] = _sp_solve_triangular(Lb[idx], bb[idx], lower=lower, trans="T" if adjoint else "N", check_finite=False)
    return out


def _chol_solve(L, b):
    """tf.linalg.cholesky_solve(L, b) = (L L^T)^-1 b."""
    return _trsm(L, _trsm(L, b, lower=True), lower=True, adjoint=True)


def _T(a):
    return np.swapaxes(a, -1, -2)


def _einsum_nml_nl(A, g):
    """tf.einsum("nml,nl->ml", A, g) (src/models/tsvgp.py:279): per latent l the matrix-vector product A[:, :, l]^T g[:, l]."""
    # (a latent's slice of [N, M, P] has no unit stride for P > 1: copied once, or the product leaves BLAS)
    return np.stack([np.ascontiguousarray(A[:, :, l]).T @ g[:, l] for l in range(A.shape[2])], axis=1)


def _einsum_nml_nol_nl(A, g):
    """tf.einsum("nml,nol,nl->lmo", A, A, g) (src/models/tsvgp.py:280): per latent l the product A_l^T diag(g_l) A_l, as the ONE
    matrix product per latent that TensorFlow lowers this contraction to (a batched GEMM).  ``np.einsum`` evaluates a
    contraction that keeps a batch index (``l``) in its own single-threaded scalar loop instead of BLAS -- at N = 1e6,
    M = 1024 that loop alone was two thirds of this port's E-step and would have handicapped the CPU baseline by a factor the
    reference does not pay.  Same sums; the order inside a dot product is the BLAS library's."""
    out = np.empty((A.shape[2], A.shape[1], A.shape[1]), dtype=np.float64)
    for l in range(A.shape[2]):
        Al = np.ascontiguousarray(A[:, :, l])  # P > 1: a strided slice would take the product out of BLAS
        out[l] = (Al * g[:, l:l + 1]).T @ Al
    return out


# --------------------------------------------------------------------------
# GPflow-style objects [ext]
# --------------------------------------------------------------------------
class SquaredExponential:
    """gpflow.kernels.SquaredExponential [ext]: k(x,z) = variance * exp(-0.5 |(x-z)/l|^2)."""

    def __init__(self, variance=1.0, lengthscales=1.0):
        self.variance = float(variance)
        self.lengthscales = np.asarray(lengthscales, dtype=np.float64)

    def _scaled(self, X):
        return np.asarray(X, dtype=np.float64) / self.lengthscales

    def K(self, X, X2=None):
        Xs = self._scaled(X)
        if X2 is None:
            sq = np.sum(Xs * Xs, axis=-1)
            r2 = -2.0 * Xs @ Xs.T + sq[:, None] + sq[None, :]
        else:
            X2s = self._scaled(X2)
            r2 = -2.0 * Xs @ X2s.T + np.sum(Xs * Xs, -1)[:, None] + np.sum(X2s * X2s, -1)[None, :]
        return self.variance * np.exp(-0.5 * r2)

    def K_diag(self, X):
        return np.full((np.asarray(X).shape[0],), self.variance, dtype=np.float64)


class _Matern(SquaredExponential):
    """gpflow.kernels.IsotropicStationary [ext]: K = variance * K_r(r) with r = sqrt(max(r2, 1e-36)), r2 the scaled
    squared distance in the same expanded form as SquaredExponential."""

    def _profile(self, r):
        raise NotImplementedError

    def K(self, X, X2=None):
        Xs = self._scaled(X)
        X2s = Xs if X2 is None else self._scaled(X2)
        r2 = -2.0 * Xs @ X2s.T + np.sum(Xs * Xs, -1)[:, None] + np.sum(X2s * X2s, -1)[None, :]
        return self.variance * self._profile(np.sqrt(np.maximum(r2, 1e-36)))


class Matern32(_Matern):
    def _profile(self, r):
        return (1.0 + np.sqrt(3.0) * r) * np.exp(-np.sqrt(3.0) * r)


class Matern52(_Matern):
    """gpflow.kernels.Matern52 [ext] (experiments/uci_regression.py:42-44)."""

    def _profile(self, r):
        return (1.0 + np.sqrt(5.0) * r + 5.0 / 3.0 * np.square(r)) * np.exp(-np.sqrt(5.0) * r)


class InducingPoints:
    """gpflow.inducing_variables.InducingPoints [ext]."""

    def __init__(self, Z):
        self.Z = np.array(Z, dtype=np.float64)

    @property
    def num_inducing(self):
        return self.Z.shape[0]


class SeparateIndependent:
    """gpflow.kernels.SeparateIndependent [ext]: P independent latent GPs, one kernel each, no mixing
    (docs/notebooks/heteroskedastic.py:62-67)."""

    def __init__(self, kernels):
        self.kernels = list(kernels)

    @property
    def num_latent_gps(self):
        return len(self.kernels)

    def K_diag(self, X):
        return np.stack([k.K_diag(X) for k in self.kernels], axis=1)  # [N, P]


class SharedIndependentInducingVariables:
    """gpflow.inducing_variables.SharedIndependentInducingVariables [ext]: one set of inducing points shared by
    all latent GPs (docs/notebooks/heteroskedastic.py:72-74)."""

    def __init__(self, inducing_variable):
        self.inducing_variable = inducingpoint_wrapper(inducing_variable)
        self.inducing_variables = [self.inducing_variable]  # the attribute src/models/tsvgp.py:252 reads

    @property
    def num_inducing(self):
        return self.inducing_variable.num_inducing


def inducingpoint_wrapper(iv):
    """gpflow.models.util.inducingpoint_wrapper [ext]: raw [M,D] arrays are wrapped."""
    return iv if isinstance(iv, (InducingPoints, SharedIndependentInducingVariables)) else InducingPoints(iv)


def _is_multi(iv, kernel):
    multi_k, multi_iv = isinstance(kernel, SeparateIndependent), isinstance(iv, SharedIndependentInducingVariables)
    if multi_k != multi_iv:
        raise NotImplementedError("only (SharedIndependentInducingVariables, SeparateIndependent) is restated")
    return multi_k


def Kuu(iv, kernel, jitter=0.0):
    """gpflow.covariances.Kuu [ext]: K(Z) + jitter*I; [P, M, M] for (shared inducing points, separate kernels)."""
    if _is_multi(iv, kernel):
        Z = iv.inducing_variable.Z
        return np.stack([k.K(Z) for k in kernel.kernels]) + jitter * np.eye(Z.shape[0])
    K = kernel.K(iv.Z)
    return K + jitter * np.eye(K.shape[0])


def Kuf(iv, kernel, Xnew):
    """gpflow.covariances.Kuf [ext]: K(Z, Xnew) -> [M, N], or [P, M, N] for separate kernels."""
    if _is_multi(iv, kernel):
        return np.stack([k.K(iv.inducing_variable.Z, Xnew) for k in kernel.kernels])
    return kernel.K(iv.Z, Xnew)


class Gaussian:
    """gpflow.likelihoods.Gaussian [ext]."""

    def __init__(self, variance=1.0):
        self.variance = float(variance)

    def variational_expectations(self, Fmu, Fvar, Y):
        ve = -0.5 * _LOG_2PI - 0.5 * np.log(self.variance) - 0.5 * ((Y - Fmu) ** 2 + Fvar) / self.variance
        return np.sum(ve, axis=-1)

    def variational_expectations_grads(self, Fmu, Fvar, Y):
        """d ve / d(Fmu, Fvar) as tf.GradientTape returns them (tsvgp.py:256-259)."""
        g0 = (Y - Fmu) / self.variance
        g1 = np.full_like(Fvar, -0.5 / self.variance)
        return g0, g1

    def predict_mean_and_var(self, Fmu, Fvar):
        return Fmu, Fvar + self.variance

    def predict_log_density(self, Fmu, Fvar, Y):
        v = Fvar + self.variance
        return np.sum(-0.5 * (_LOG_2PI + np.log(v) + (Y - Fmu) ** 2 / v), axis=-1)


_GH_N = 20


def gh_points_and_weights(n_gh=_GH_N):
    """gpflow.quadrature.gauss_hermite.gh_points_and_weights [ext]."""
    z, dz = np.polynomial.hermite.hermgauss(n_gh)
    return z * np.sqrt(2.0), dz / np.sqrt(np.pi)


def inv_probit(x):
    """gpflow.likelihoods.utils.inv_probit [ext] (1e-3 jitter)."""
    jitter = 1e-3
    return 0.5 * (1.0 + _erf(x / np.sqrt(2.0))) * (1 - 2 * jitter) + jitter


class Bernoulli:
    """gpflow.likelihoods.Bernoulli with the default probit link [ext];
    variational expectations by 20-point Gauss-Hermite quadrature."""

    num_gauss_hermite_points = _GH_N

    def _logp(self, F, Y):
        p = inv_probit(F)
        return np.log(np.where(Y == 1, p, 1.0 - p))

    def _dlogp(self, F, Y):
        jitter = 1e-3
        p = inv_probit(F)
        dp = (1 - 2 * jitter) * np.exp(-0.5 * F * F) / np.sqrt(2.0 * np.pi)
        return np.where(Y == 1, dp / p, -dp / (1.0 - p))

    def variational_expectations(self, Fmu, Fvar, Y):
        z, w = gh_points_and_weights(self.num_gauss_hermite_points)
        F = Fmu[..., None] + np.sqrt(Fvar)[..., None] * z
        ve = np.sum(self._logp(F, np.asarray(Y)[..., None]) * w, axis=-1)
        return np.sum(ve, axis=-1)

    def variational_expectations_grads(self, Fmu, Fvar, Y):
        """Derivative OF THE QUADRATURE SUM (what GradientTape computes):
        d/dm = sum_i w_i l'(f_i);  d/dv = sum_i w_i l'(f_i) z_i / (2 sqrt(v))."""
        z, w = gh_points_and_weights(self.num_gauss_hermite_points)
        sd = np.sqrt(Fvar)
        F = Fmu[..., None] + sd[..., None] * z
        dl = self._dlogp(F, np.asarray(Y)[..., None])
        g0 = np.sum(dl * w, axis=-1)
        g1 = np.sum(dl * w * z, axis=-1) / (2.0 * sd)
        return g0, g1

    def predict_mean_and_var(self, Fmu, Fvar):
        """Bernoulli._predict_mean_and_var with the probit link [ext]: closed form p = inv_probit(m / sqrt(1 + v))."""
        p = inv_probit(Fmu / np.sqrt(1.0 + Fvar))
        return p, p - p * p

    def predict_log_density(self, Fmu, Fvar, Y):
        """Bernoulli._predict_log_density [ext]: logdensities.bernoulli(Y, p) with p the predictive mean, summed over the
        output columns (what experiments/uci_classification.py:139 averages into the NLPD)."""
        p = self.predict_mean_and_var(Fmu, Fvar)[0]
        return np.sum(np.log(np.where(np.asarray(Y) == 1, p, 1.0 - p)), axis=-1)


# --------------------------------------------------------------------------
# GPflow conditionals / KL [ext]
# --------------------------------------------------------------------------
def base_conditional(Kmn, Kmm, Knn, f, q_sqrt=None, white=False, _Lm=None):
    """gpflow.conditionals.util.base_conditional [ext] (full_cov=False).
    Kmn [M,N], Kmm [M,M], Knn [N], f [M,P], q_sqrt [P,M,M] -> mean [N,P], var [N,P].
    ``_Lm`` (not GPflow's): chol(Kmm) from a previous call with the same Kmm -- row-blocked callers factor once, as the
    reference's single call over all rows does."""
    Lm = _chol(Kmm) if _Lm is None else _Lm
    A = _trsm(Lm, Kmn, lower=True)  # [M,N]
    P = f.shape[-1]
    fvar = Knn - np.sum(A * A, axis=-2)  # [N]
    fvar = np.tile(fvar[None, :], [P, 1])  # [P,N]
    if not white:
        A = _trsm(Lm, A, lower=True, adjoint=True)
    fmean = A.T @ f  # [N,P]
    if q_sqrt is not None:
        L = np.tril(q_sqrt)  # band_part(q_sqrt, -1, 0)
        A_tiled = np.tile(A[None], [P, 1, 1])
        # [P,M,N]: one 2-D product per latent (np.matmul on a stack of transposed views leaves BLAS: 6x slower at M = 1024)
        LTA = np.stack([L[p].T @ A_tiled[p] for p in range(P)])
        fvar = fvar + np.sum(LTA * LTA, axis=-2)
    return fmean, fvar.T


def conditional(Xnew, iv, kernel, f, q_sqrt=None, white=False, _Lm=None):
    """gpflow.conditionals.conditional (InducingPoints, Kernel) [ext]: jitter = default_jitter().
    For (SharedIndependentInducingVariables, SeparateIndependent) GPflow dispatches to
    separate_independent_conditional [ext]: base_conditional per latent with its own Kmm / Kmn / Knn."""
    Kmm = Kuu(iv, kernel, jitter=DEFAULT_JITTER)
    Kmn = Kuf(iv, kernel, Xnew)
    Knn = kernel.K_diag(Xnew)
    if _is_multi(iv, kernel):
        outs = [base_conditional(Kmn[p], Kmm[p], Knn[:, p], f[:, p:p + 1],
                                 q_sqrt=None if q_sqrt is None else q_sqrt[p:p + 1], white=white,
                                 _Lm=None if _Lm is None else _Lm[p])
                for p in range(len(kernel.kernels))]
        return np.concatenate([o[0] for o in outs], axis=1), np.concatenate([o[1] for o in outs], axis=1)
    return base_conditional(Kmn, Kmm, Knn, f, q_sqrt=q_sqrt, white=white, _Lm=_Lm)


def gauss_kl(q_mu, q_sqrt, K):
    """gpflow.kullback_leiblers.gauss_kl [ext], dense K [M,M] or one K per latent [P,M,M]; q_sqrt [P,M,M]."""
    M, P = q_mu.shape
    if K.ndim == 3:  # is_batched: every latent against its own prior
        Lp = _chol(K)  # [P,M,M]
        alpha = _trsm(Lp, _T(q_mu)[..., None], lower=True)  # [P,M,1]
        Lq = np.tril(q_sqrt)
        LpiLq = _trsm(Lp, Lq, lower=True)
        twoKL = np.sum(alpha * alpha) - float(M * P) - np.sum(np.log(np.square(np.diagonal(Lq, axis1=-2, axis2=-1))))
        twoKL += np.sum(LpiLq * LpiLq) + np.sum(np.log(np.square(np.diagonal(Lp, axis1=-2, axis2=-1))))
        return 0.5 * twoKL
    Lp = _chol(K)
    alpha = _trsm(Lp, q_mu, lower=True)
    Lq = np.tril(q_sqrt)
    mahalanobis = np.sum(alpha * alpha)
    constant = -float(M * P)
    logdet_qcov = np.sum(np.log(np.square(np.diagonal(Lq, axis1=-2, axis2=-1))))
    LpiLq = _trsm(np.tile(Lp[None], [P, 1, 1]), Lq, lower=True)
    trace = np.sum(LpiLq * LpiLq)
    twoKL = mahalanobis + constant - logdet_qcov + trace
    twoKL += P * np.sum(np.log(np.square(np.diagonal(Lp))))
    return 0.5 * twoKL


def prior_kl(iv, kernel, q_mu, q_sqrt, whiten=False):
    """gpflow.kullback_leiblers.prior_kl [ext]."""
    if whiten:
        raise NotImplementedError
    return gauss_kl(q_mu, q_sqrt, Kuu(iv, kernel, jitter=DEFAULT_JITTER))


# --------------------------------------------------------------------------
# src/util.py restated
# --------------------------------------------------------------------------
def posterior_from_dense_site(K, lambda_1, lambda_2_sqrt):
    """src/util.py:349-391.  K [M,M], lambda_1 [M,P], lambda_2_sqrt [P,M,M]
    -> m [M,P], chol(S) [P,M,M]."""
    if K.shape[-1] != K.shape[-2] or lambda_2_sqrt.ndim != 3 or lambda_2_sqrt.shape[-1] != K.shape[-1] \
            or lambda_1.shape[-1] != lambda_2_sqrt.shape[0]:
        raise ValueError("posterior_from_dense_site() arguments: shape mismatch")  # util.py:368-372
    L = lambda_2_sqrt
    Id = np.eye(K.shape[-1])
    C = _chol(K)  # :377
    CtL = _T(C) @ L  # :380
    W = Id + _T(CtL) @ CtL  # :381
    chol_W = _chol(W)  # :382
    LtK = _T(L) @ K  # :385
    iwLtK = _trsm(chol_W, LtK, lower=True)  # :386
    S_q = K - _T(iwLtK) @ iwLtK  # :387
    chol_S_q = _chol(S_q)  # :388
    m_q = np.einsum("lmn,nl->ml", S_q, lambda_1)  # :389
    return m_q, chol_S_q


def conditional_from_precision_sites(Kuu_, Kff, Kuf_, l, L=None, L2=None):
    """src/util.py:91-185.  Kuu [M,M], Kff [N,1], Kuf [M,N], l [M,P], L [P,M,M]
    -> mean [N,P], cov [N,P]."""
    if L is None:
        L = _chol(L2)  # :163-164
    Id = np.eye(Kuu_.shape[-1])
    C = _chol(Kuu_)  # :168
    CtL = _T(C) @ L  # :171
    W = Id + _T(CtL) @ CtL  # :172
    chol_W = _chol(W)  # :173
    D = _trsm(chol_W, _T(L), lower=True)  # :175  [P,M,M]
    tmp = D @ Kuf_  # :176  [P,M,N]
    DKl = D @ (Kuu_ @ _T(l)[..., None])  # :180  [P,M,1]
    mean = Kuf_.T @ l - _T(np.sum(DKl * tmp, axis=-2))  # :178-182
    cov = Kff - _T(np.sum(np.square(tmp), axis=-2))  # :184
    return mean, cov


def gradient_transformation_mean_var_to_expectation(inputs, grads):
    """src/util.py:429-438."""
    return grads[0] - 2.0 * np.einsum("lmo,ol->ml", grads[1], inputs), grads[1]


def kl_from_precision_sites_white(A, l, L=None, L2=None):
    """src/util.py:239-291."""
    if L2 is None:
        L2 = L @ _T(L)
    m = L2.shape[-2]
    R = L2 + A
    LR = _chol(R)
    LA = _chol(A)
    log_det = np.sum(np.log(np.square(np.diagonal(LR, axis1=-2, axis2=-1)))) - np.sum(
        np.log(np.square(np.diagonal(LA)))
    )
    tmp = _trsm(LR, LA, lower=True)
    trace_plus_const = np.sum(np.square(tmp)) - float(m)
    mahalanobis = np.sum(np.square(LA.T @ _chol_solve(LR, l)))
    return 0.5 * (log_det + trace_plus_const + mahalanobis)


def mean_cov_from_precision_site(A, l, L):
    """tests/tools.py:4-40 (test-only Woodbury reference)."""
    R = L @ _T(L) + A
    LR = _chol(R)
    tmp = _trsm(LR, A, lower=True)
    cov = _T(tmp) @ tmp
    mean = (_T(tmp) @ _trsm(LR, l, lower=True))[0]
    return mean, cov


def project_diag_sites(Kuf_, lambda_1, lambda_2, Kuu_=None, cholesky=True):
    """src/util.py:188-236 (used only by the pin tests)."""
    num_latent = lambda_1.shape[-1]
    P = np.tile(Kuf_[None], [num_latent, 1, 1]) if Kuf_.ndim == 2 else Kuf_
    if Kuu_ is not None:
        Kuu_ = Kuu_[None] if Kuu_.ndim == 2 else Kuu_
        Luu = _chol(Kuu_)
        P = _chol_solve(Luu, P)
    l = np.einsum("lmn,nl->ml", P, lambda_1)
    L = np.einsum("lmn,lon,nl->lmo", P, P, lambda_2)
    if cholesky:
        L = _chol(L)
    return l, L


def conditional_from_precision_sites_white(Kuu_, Kff, Kuf_, l, L=None, L2=None, jitter=1e-9):
    """src/util.py:11-88."""
    if L2 is None:
        L2 = L @ _T(L)
    m = Kuu_.shape[-1]
    R = L2 + Kuu_ + np.eye(m) * jitter
    LR = _chol(R)
    LA = _chol(Kuu_)[None]
    tmp1 = _trsm(LR, Kuf_, lower=True)
    tmp2 = _trsm(LA, Kuf_, lower=True)
    cov = Kff - _T(np.sum(np.square(tmp2), axis=-2) - np.sum(np.square(tmp1), axis=-2))
    mean = (Kuf_.T @ _chol_solve(LR, l))[0]
    return mean, cov


def posterior_from_dense_site_white(K, lambda_1, lambda_2, jitter=1e-9):
    """src/util.py:394-426."""
    m = K.shape[-1]
    R = K + lambda_2
    LR = _chol(R + np.eye(m) * jitter)
    iLRK = _trsm(LR, K, lower=True)
    S_q = _T(iLRK) @ iLRK
    chol_S_q = _chol(S_q)
    m_q = (K @ _chol_solve(LR, lambda_1))[0]
    return m_q, chol_S_q


# --------------------------------------------------------------------------
# src/sites.py restated
# --------------------------------------------------------------------------
class DenseSites:
    """src/sites.py:43-80.  lambda_1 [M,P]; lambda_2_sqrt [P,M,M] under a
    triangular() transform (only the lower triangle is kept)."""

    def __init__(self, lambda_1, lambda_2_sqrt=None, lambda_2=None):
        assert (lambda_2_sqrt is not None) or (lambda_2 is not None)
        self.lambda_1 = np.array(lambda_1, dtype=np.float64)
        self.num_latent_gps = self.lambda_1.shape[0]  # sites.py:57 (sic)
        if lambda_2_sqrt is not None:
            self.factor = True
            self._lambda_2_sqrt = np.tril(np.array(lambda_2_sqrt, dtype=np.float64))
        else:
            self.factor = False
            self._lambda_2 = np.array(lambda_2, dtype=np.float64)

    @property
    def lambda_2(self):
        if self.factor:
            return self._lambda_2_sqrt @ _T(self._lambda_2_sqrt)
        return self._lambda_2

    @property
    def lambda_2_sqrt(self):
        if self.factor:
            return self._lambda_2_sqrt
        return _chol(self._lambda_2)


# --------------------------------------------------------------------------
# src/models/tsvgp.py restated
# --------------------------------------------------------------------------
class t_SVGP:
    """src/models/tsvgp.py:117-304 (+ base_SVGP :32-114), NumPy fp64."""

    def __init__(self, kernel, likelihood, inducing_variable, *, mean_function=None,
                 num_latent_gps=1, lambda_1=None, lambda_2_sqrt=None, num_data=None, force=False):
        if mean_function is not None:
            raise NotImplementedError("only the Zero mean function is restated")
        self.kernel = kernel
        self.likelihood = likelihood
        self.num_latent_gps = num_latent_gps
        self.num_data = num_data
        self.inducing_variable = inducingpoint_wrapper(inducing_variable)  # :150
        self.num_inducing = self.inducing_variable.num_inducing
        self._init_variational_parameters(self.num_inducing, lambda_1, lambda_2_sqrt)
        self.whiten = False
        self.force = force
        self.last = {}  # intermediates of the last natgrad_step, for parity tests

    def _init_variational_parameters(self, num_inducing, lambda_1, lambda_2_sqrt):
        """:159-185."""
        lambda_1 = np.zeros((num_inducing, self.num_latent_gps)) if lambda_1 is None else lambda_1
        if lambda_2_sqrt is None:
            lambda_2_sqrt = np.array([-np.eye(num_inducing) * 1e-10 for _ in range(self.num_latent_gps)])
        else:
            lambda_2_sqrt = np.asarray(lambda_2_sqrt)
            assert lambda_2_sqrt.ndim == 3
            self.num_latent_gps = lambda_2_sqrt.shape[0]
        self.sites = DenseSites(lambda_1, lambda_2_sqrt)

    @property
    def lambda_1(self):
        return self.sites.lambda_1

    @property
    def lambda_2_sqrt(self):
        return self.sites.lambda_2_sqrt

    @property
    def lambda_2(self):
        return self.lambda_2_sqrt @ _T(self.lambda_2_sqrt)  # :200

    def get_mean_chol_cov_inducing_posterior(self):
        """:202-212."""
        K_uu = Kuu(self.inducing_variable, self.kernel, jitter=DEFAULT_JITTER)
        return posterior_from_dense_site(K_uu, self.lambda_1, self.lambda_2_sqrt)

    def prior_kl(self):
        """:65-70."""
        q_mu, q_sqrt = self.get_mean_chol_cov_inducing_posterior()
        return prior_kl(self.inducing_variable, self.kernel, q_mu, q_sqrt, whiten=False)

    def predict_f(self, Xnew, full_cov=False, full_output_cov=False):
        """:97-114."""
        if full_cov or full_output_cov:
            raise NotImplementedError
        q_mu, q_sqrt = self.get_mean_chol_cov_inducing_posterior()
        mu, var = conditional(Xnew, self.inducing_variable, self.kernel, q_mu, q_sqrt=q_sqrt, white=False)
        if not np.all(var > 0):  # tf.debugging.assert_positive :113
            raise FloatingPointError("predict_f: non-positive predictive variance")
        return mu, var

    def new_predict_f(self, Xnew, full_cov=False, full_output_cov=False):
        """:215-232."""
        K_uu = Kuu(self.inducing_variable, self.kernel, jitter=DEFAULT_JITTER)
        K_uf = Kuf(self.inducing_variable, self.kernel, Xnew)
        if K_uu.ndim == 3:  # "todo : make broadcastable" (:214): the reference form is for one shared kernel
            raise NotImplementedError("new_predict_f is not broadcastable over separate kernels in the reference")
        K_ff = self.kernel.K_diag(Xnew)[..., None]
        mu, var = conditional_from_precision_sites(K_uu, K_ff, K_uf, self.lambda_1, L=self.lambda_2_sqrt)
        if not np.all(var > 0):  # :231
            raise FloatingPointError("new_predict_f: non-positive predictive variance")
        return mu, var

    def predict_y(self, Xnew):
        return self.likelihood.predict_mean_and_var(*self.predict_f(Xnew))

    def predict_log_density(self, data):
        X, Y = data
        return self.likelihood.predict_log_density(*self.predict_f(X), Y)

    def elbo(self, data):
        """:79-95."""
        X, Y = data
        X = np.asarray(X, dtype=np.float64)
        Y = np.asarray(Y, dtype=np.float64)
        kl = self.prior_kl()
        f_mean, f_var = self.predict_f(X)
        var_exp = self.likelihood.variational_expectations(f_mean, f_var, Y)
        scale = (float(self.num_data) / X.shape[0]) if self.num_data is not None else 1.0
        return np.sum(var_exp) * scale - kl

    def maximum_log_likelihood_objective(self, data):
        return self.elbo(data)

    def training_loss(self, data):
        return -self.elbo(data)

    def natgrad_step(self, data, lr=0.1, jitter=1e-9):
        """:234-304, op for op (including the redundant work)."""
        X, Y = data
        X = np.asarray(X, dtype=np.float64)
        Y = np.asarray(Y, dtype=np.float64)
        mean, var = self.predict_f(X)  # :246
        if isinstance(self.inducing_variable, SharedIndependentInducingVariables):  # :249-252
            meanZ, _ = self.predict_f(self.inducing_variable.inducing_variables[0].Z)
        else:
            meanZ, _ = self.predict_f(self.inducing_variable.Z)  # :254

        g0, g1 = self.likelihood.variational_expectations_grads(mean, var, Y)  # :256-259
        eps = 1e-8
        g1 = np.minimum(g1, -eps * np.ones_like(g1))  # :262-263

        Id = np.eye(self.num_inducing)  # :265
        K_uu = Kuu(self.inducing_variable, self.kernel)  # :268 (no jitter)
        K_uf = Kuf(self.inducing_variable, self.kernel, X)  # :269
        chol_Kuu = _chol(K_uu + Id * jitter)  # :270
        A = np.transpose(_chol_solve(chol_Kuu, K_uf))  # :271  [N,M], or [N,M,P] from [P,M,N] (tf.transpose reverses)

        if A.ndim == 2:
            A = np.tile(A[..., None], [1, 1, self.num_latent_gps])  # :276-277
        grads = [
            _einsum_nml_nl(A, g0),  # :279
            _einsum_nml_nol_nl(A, g1),  # :280
        ]
        grad_mu = gradient_transformation_mean_var_to_expectation(meanZ, grads)  # :284

        scale = (float(self.num_data) / X.shape[0]) if self.num_data is not None else 1.0  # :286-291

        lambda_2 = -0.5 * self.lambda_2  # :293
        lambda_1 = self.lambda_1
        lambda_1 = (1 - lr) * lambda_1 + lr * scale * grad_mu[0]  # :296
        lambda_2 = (1 - lr) * lambda_2 + lr * scale * grad_mu[1]  # :297

        lambda_2_sqrt = -_chol(-2.0 * lambda_2 + Id * jitter)  # :300
        self.sites.lambda_1 = lambda_1  # :302
        self.sites._lambda_2_sqrt = np.tril(lambda_2_sqrt)  # :303 (triangular() transform)
        self.get_mean_chol_cov_inducing_posterior()  # :304 (result discarded)
        self.last = dict(mean=mean, var=var, meanZ=meanZ, g0=g0, g1=g1, G0=grads[0], G1=grads[1])


# --------------------------------------------------------------------------
# src/models/tsvgp_white.py restated
# --------------------------------------------------------------------------
class t_SVGP_white:
    """src/models/tsvgp_white.py:23-246, NumPy fp64: the t-SVGP with the site in K-whitened coordinates,
    q(u) with precision K^-1 + K^-1 Lambda_2 K^-1 and S^-1 m = K^-1 lambda_1; the state is the FULL Lambda_2.
    The util functions it calls take element [0] of a latent-batched product (util.py:87, :425), so the class is
    only meaningful for num_latent_gps = 1; that is what is restated and pinned."""

    def __init__(self, kernel, likelihood, inducing_variable, *, mean_function=None, num_latent_gps=1,
                 lambda_1=None, lambda_2=None, num_data=None):
        if mean_function is not None:
            raise NotImplementedError("only the Zero mean function is restated")
        self.kernel = kernel
        self.likelihood = likelihood
        self.num_latent_gps = num_latent_gps
        self.num_data = num_data
        self.inducing_variable = inducingpoint_wrapper(inducing_variable)  # :56
        self.num_inducing = self.inducing_variable.num_inducing
        lambda_1 = np.zeros((self.num_inducing, self.num_latent_gps)) if lambda_1 is None else lambda_1  # :78
        if lambda_2 is None:
            lambda_2 = np.array([np.eye(self.num_inducing) * 1e-10 for _ in range(self.num_latent_gps)])  # :80-85
        else:
            lambda_2 = np.asarray(lambda_2)
            assert lambda_2.ndim == 3  # :87
            self.num_latent_gps = lambda_2.shape[0]
        self.sites = DenseSites(lambda_1=lambda_1, lambda_2=lambda_2)  # :90
        self.last = {}

    @property
    def lambda_1(self):
        return self.sites.lambda_1

    @property
    def lambda_2(self):
        return self.sites.lambda_2

    def get_mean_chol_cov_inducing_posterior(self):
        """:99-109."""
        K_uu = Kuu(self.inducing_variable, self.kernel, jitter=DEFAULT_JITTER)
        return posterior_from_dense_site_white(K_uu, self.lambda_1, self.lambda_2)

    def prior_kl(self):
        """:116-120."""
        K_uu = Kuu(self.inducing_variable, self.kernel, jitter=DEFAULT_JITTER)
        return kl_from_precision_sites_white(K_uu, self.lambda_1, L2=self.lambda_2)

    def predict_f(self, Xnew, full_cov=False, full_output_cov=False):
        """:122-132."""
        K_uu = Kuu(self.inducing_variable, self.kernel, jitter=DEFAULT_JITTER)
        K_uf = Kuf(self.inducing_variable, self.kernel, Xnew)
        K_ff = self.kernel.K_diag(Xnew)[..., None]
        mu, var = conditional_from_precision_sites_white(K_uu, K_ff, K_uf, self.lambda_1, L2=self.lambda_2)
        if not np.all(var > 0):  # :131
            raise FloatingPointError("predict_f: non-positive predictive variance")
        return mu, var

    def predict_f_extra_data(self, Xnew, extra_data, jitter=DEFAULT_JITTER):
        """:134-160.  Prediction at Xnew conditioned on ``extra_data`` as well: the sites get the natural-gradient
        contribution of the extra points (one full step, no learning rate) without the state being touched."""
        Xe, Ye = extra_data
        grad_mu = self.compute_data_natural_params((np.asarray(Xe, dtype=np.float64), np.asarray(Ye, dtype=np.float64)))  # :141
        lambda_1 = self.lambda_1  # :143
        lambda_2 = -0.5 * self.lambda_2  # :144
        K_uu = Kuu(self.inducing_variable, self.kernel, jitter=jitter)  # :146
        lambda_1c = lambda_1 + K_uu @ grad_mu[0]  # :148
        lambda_2c = -2 * (lambda_2 + K_uu @ grad_mu[1] @ K_uu)  # :149
        K_uf = Kuf(self.inducing_variable, self.kernel, Xnew)  # :152
        K_ff = self.kernel.K_diag(Xnew)[..., None]
        mu, var = conditional_from_precision_sites_white(K_uu, K_ff, K_uf, lambda_1c, L2=lambda_2c)  # :155-156
        return mu, var  # no assert_positive on this path

    def elbo(self, data):
        """:162-177."""
        X, Y = data
        X = np.asarray(X, dtype=np.float64)
        Y = np.asarray(Y, dtype=np.float64)
        kl = self.prior_kl()
        f_mean, f_var = self.predict_f(X)
        var_exp = self.likelihood.variational_expectations(f_mean, f_var, Y)
        scale = (float(self.num_data) / X.shape[0]) if self.num_data is not None else 1.0
        return np.sum(var_exp) * scale - kl

    def compute_data_natural_params(self, data, jitter=1e-9):
        """:183-212 (no cropping of the second gradient here, unlike tsvgp.py:262-263)."""
        X, Y = data
        mean, var = self.predict_f(X)  # :185
        meanZ, _ = self.predict_f(self.inducing_variable.Z)  # :186
        g0, g1 = self.likelihood.variational_expectations_grads(mean, var, Y)  # :188-191
        Id = np.eye(self.num_inducing)
        K_uu = Kuu(self.inducing_variable, self.kernel)  # :196
        K_uf = Kuf(self.inducing_variable, self.kernel, X)
        chol_Kuu = _chol(K_uu + Id * jitter)  # :198
        A = np.transpose(_chol_solve(chol_Kuu, K_uf))  # :199
        A = np.tile(A[..., None], [1, 1, self.num_latent_gps])  # :201
        grads = [_einsum_nml_nl(A, g0), _einsum_nml_nol_nl(A, g1)]  # :203-206
        self.last = dict(mean=mean, var=var, meanZ=meanZ, g0=g0, g1=g1, G0=grads[0], G1=grads[1])
        return gradient_transformation_mean_var_to_expectation(meanZ, grads)  # :209

    def natgrad_step(self, dataset, lr=0.1, jitter=1e-9):
        """:215-246."""
        X, Y = dataset
        X = np.asarray(X, dtype=np.float64)
        Y = np.asarray(Y, dtype=np.float64)
        grad_mu = self.compute_data_natural_params((X, Y))  # :230
        K_uu = Kuu(self.inducing_variable, self.kernel)  # :231 (no jitter)
        scale = (float(self.num_data) / X.shape[0]) if self.num_data is not None else 1.0  # :233-238
        lambda_1 = self.lambda_1
        lambda_2 = -0.5 * self.lambda_2  # :241
        lambda_1 = (1.0 - lr) * lambda_1 + lr * scale * K_uu @ grad_mu[0]  # :244
        lambda_2 = (1.0 - lr) * lambda_2 + lr * scale * K_uu @ grad_mu[1] @ K_uu  # :245
        self.sites.lambda_1 = lambda_1  # :247
        self.sites._lambda_2 = -2.0 * lambda_2  # :248


def sgpr_predict_f(kernel, X, Y, Z, noise_variance, Xnew):
    """Titsias' collapsed sparse GP regression posterior (what gpflow.models.SGPR.predict_f computes [ext]), in the
    textbook form: Sigma = (Kuu + Kuf Kfu / s2)^-1, mean = Ksu Sigma Kuf y / s2, var = kss - Ksu Kuu^-1 Kus + Ksu Sigma Kus.
    Independent pin for reference tests/models/test_condit.py:69-83."""
    Kuu_ = kernel.K(Z) + DEFAULT_JITTER * np.eye(Z.shape[0])
    Kuf_ = kernel.K(Z, X)
    Kus = kernel.K(Z, Xnew)
    Sigma_inv = Kuu_ + Kuf_ @ Kuf_.T / noise_variance
    mean = Kus.T @ np.linalg.solve(Sigma_inv, Kuf_ @ Y) / noise_variance
    var = kernel.K_diag(Xnew)[:, None] - np.sum(Kus * np.linalg.solve(Kuu_, Kus), 0)[:, None] \
        + np.sum(Kus * np.linalg.solve(Sigma_inv, Kus), 0)[:, None]
    return mean, var


# --------------------------------------------------------------------------
# closed forms used as independent pins (exact GP regression)
# --------------------------------------------------------------------------
def gpr_log_marginal_likelihood(kernel, X, Y, noise_variance):
    """gpflow.models.GPR.log_marginal_likelihood [ext]: log N(Y | 0, K + s2 I), summed over columns."""
    K = kernel.K(X) + noise_variance * np.eye(X.shape[0])
    L = _chol(K)
    alpha = _trsm(L, Y, lower=True)
    n, p = Y.shape
    return float(-0.5 * np.sum(alpha * alpha) - p * np.sum(np.log(np.diag(L))) - 0.5 * n * p * _LOG_2PI)


def gpr_predict_f(kernel, X, Y, noise_variance, Xnew):
    """gpflow.models.GPR.predict_f [ext] (full_cov=False)."""
    K = kernel.K(X) + noise_variance * np.eye(X.shape[0])
    Kmn = kernel.K(X, Xnew)
    L = _chol(K)
    A = _trsm(L, Kmn, lower=True)
    V = _trsm(L, Y, lower=True)
    mean = A.T @ V
    var = kernel.K_diag(Xnew) - np.sum(A * A, 0)
    return mean, np.tile(var[:, None], [1, Y.shape[1]])


# --------------------------------------------------------------------------
# row-blocked evaluation of the same quantities (for sizes where the [M, N] temporaries of one call would not fit)
# --------------------------------------------------------------------------
def predict_f_chunked(model, Xnew, chunk_rows=20000):
    """``base_SVGP.predict_f`` (src/models/tsvgp.py:97-114) over row blocks: (m, chol S) once (:102), then GPflow's
    ``conditional`` [ext] per block of ``chunk_rows`` rows (rows are independent: the same numbers as one call)."""
    Xnew = np.asarray(Xnew, dtype=np.float64)
    q_mu, q_sqrt = model.get_mean_chol_cov_inducing_posterior()
    mus, vrs = [], []
    for lo in range(0, Xnew.shape[0], chunk_rows):
        mu, var = conditional(Xnew[lo:lo + chunk_rows], model.inducing_variable, model.kernel, q_mu, q_sqrt=q_sqrt,
                              white=False)
        if not np.all(var > 0):  # :113
            raise FloatingPointError("predict_f: non-positive predictive variance")
        mus.append(mu)
        vrs.append(var)
    return np.concatenate(mus), np.concatenate(vrs)


def elbo_chunked(model, data, chunk_rows=20000, progress=None):
    """``base_SVGP.elbo`` (src/models/tsvgp.py:79-95) with the sum over the N rows taken block by block: the M x M part
    (posterior factorisation :102, prior KL :65-70) once, then ``conditional`` [ext] + ``variational_expectations``
    [ext] per block, the block sums added exactly (``math.fsum``).  What ``bench.py`` evaluates on the HIP model's
    state at N = 1e6, M = 1024 ("ELBO match" half of the metric); ~4 N M^2 flops."""
    import math

    X, Y = data
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    kl = model.prior_kl()
    q_mu, q_sqrt = model.get_mean_chol_cov_inducing_posterior()
    sums = []
    for lo in range(0, X.shape[0], chunk_rows):
        mu, var = conditional(X[lo:lo + chunk_rows], model.inducing_variable, model.kernel, q_mu, q_sqrt=q_sqrt,
                              white=False)
        if not np.all(var > 0):  # :113
            raise FloatingPointError("predict_f: non-positive predictive variance")
        sums.append(float(np.sum(model.likelihood.variational_expectations(mu, var, Y[lo:lo + chunk_rows]))))
        if progress is not None:
            progress(min(lo + chunk_rows, X.shape[0]), X.shape[0])
    scale = (float(model.num_data) / X.shape[0]) if model.num_data is not None else 1.0
    return math.fsum(sums) * scale - kl


class _CompensatedSum:
    """Running sum of equally shaped arrays with Neumaier's compensation (the array form of ``math.fsum``'s idea): the
    rounding error of every addition is kept in a second array and added back at the end, so the order and the number of
    row blocks do not show in the result beyond one rounding."""

    def __init__(self):
        self.s = None
        self.c = None

    def add(self, x):
        x = np.asarray(x, dtype=np.float64)
        if self.s is None:
            self.s, self.c = x.copy(), np.zeros_like(x)
            return
        t = self.s + x
        big = np.abs(self.s) >= np.abs(x)
        self.c += np.where(big, (self.s - t) + x, (x - t) + self.s)
        self.s = t

    def value(self):
        return self.s + self.c


def natgrad_step_chunked(model, data, lr=0.1, jitter=1e-9, chunk_rows=20000, progress=None):
    """``t_SVGP.natgrad_step`` (src/models/tsvgp.py:234-304) with the N rows taken block by block, so that the reference's op
    sequence can be run at N = 1e6, M = 1024 (one call over all rows would hold ~8 arrays of N M 8 bytes = 67 GB).
    Everything that does not depend on the rows is done ONCE, where the reference does it (posterior factorisation :246 -> :102,
    predict_f(Z) :249-254, K_uu and its factor :268-270, the update :284-304); per block of ``chunk_rows`` rows the reference's
    N-sized ops run unchanged and in its order: ``conditional`` [ext] (:246 -- K_uf, two triangular solves, the dense
    q_sqrt^T A product), the likelihood gradients and the crop (:256-263), K_uf AGAIN (:269), ``cholesky_solve`` (:270-271), the
    tile (:276-277) and the two einsums (:279-280), whose block results G0 [M,P], G1 [P,M,M] are added with compensated
    summation (``_CompensatedSum``).  Same numbers as ``natgrad_step`` up to the order of the row sums (checked in
    tests/test_oracle_pins.py); what tests/test_gpu_fullsize.py compares the HIP step with at full size, and -- timed -- the
    MEASURED CPU baseline of the metric (bench.py ``state_match``)."""
    import math

    X, Y = data
    X = np.asarray(X, dtype=np.float64)
    Y = np.asarray(Y, dtype=np.float64)
    iv, kernel = model.inducing_variable, model.kernel
    # :246 -> :102-103: (m, chol S) once; the factor of K_uu + 1e-6 I that ``conditional`` takes per call
    q_mu, q_sqrt = model.get_mean_chol_cov_inducing_posterior()
    Lm = _chol(Kuu(iv, kernel, jitter=DEFAULT_JITTER))
    if isinstance(iv, SharedIndependentInducingVariables):  # :249-252
        meanZ, _ = model.predict_f(iv.inducing_variables[0].Z)
    else:
        meanZ, _ = model.predict_f(iv.Z)  # :254
    Id = np.eye(model.num_inducing)  # :265
    K_uu = Kuu(iv, kernel)  # :268 (no jitter)
    chol_Kuu = _chol(K_uu + Id * jitter)  # :270
    G0, G1, A_abs = _CompensatedSum(), _CompensatedSum(), _CompensatedSum()
    N = X.shape[0]
    means, vars_, g0s, g1s, ve = [], [], [], [], []
    import time as _time

    extras = 0.0  # seconds spent on what is NOT part of the reference's step (marked below): a timed call subtracts them
    _t = _time.perf_counter()
    kl_before = model.prior_kl()  # :65-70 at the state the step starts from (for ``last["elbo_before"]``, below); not part of the step
    extras += _time.perf_counter() - _t
    for lo in range(0, N, chunk_rows):
        Xb, Yb = X[lo:lo + chunk_rows], Y[lo:lo + chunk_rows]
        mean, var = conditional(Xb, iv, kernel, q_mu, q_sqrt=q_sqrt, white=False, _Lm=Lm)  # :246 -> :103
        if not np.all(var > 0):  # :113
            raise FloatingPointError("predict_f: non-positive predictive variance")
        g0, g1 = model.likelihood.variational_expectations_grads(mean, var, Yb)  # :256-259
        g1 = np.minimum(g1, -1e-8 * np.ones_like(g1))  # :262-263
        K_uf = Kuf(iv, kernel, Xb)  # :269
        A = np.transpose(_chol_solve(chol_Kuu, K_uf))  # :271
        if A.ndim == 2:
            A = np.tile(A[..., None], [1, 1, model.num_latent_gps])  # :276-277
        G0.add(_einsum_nml_nl(A, g0))  # :279
        # (not part of the step) sum_n |a_n|: the sensitivity of G0 to its inputs -- an error e in g0 moves G0 by at most
        # e * sum_n |a_n|.  G0 = A^T (y - mean) / s2 is a small difference of N-sized terms once the sites fit the data, so
        # its RELATIVE error is unbounded while lambda_1, which it updates, is not affected; comparisons scale by this
        _t = _time.perf_counter()
        A_abs.add(np.sum(np.abs(A), axis=0))
        extras += _time.perf_counter() - _t
        G1.add(_einsum_nml_nol_nl(A, g1))  # :280
        _t = _time.perf_counter()
        means.append(mean), vars_.append(var), g0s.append(g0), g1s.append(g1)
        # not part of the step: the block's term of ``elbo`` (:88-95) at the state the step STARTS from rides along (O(n P)),
        # so one pass over the rows yields both halves of the metric's parity check
        ve.append(float(np.sum(model.likelihood.variational_expectations(mean, var, Yb))))
        extras += _time.perf_counter() - _t
        if progress is not None:
            progress(min(lo + chunk_rows, N), N)
    grads = [G0.value(), G1.value()]
    grad_mu = gradient_transformation_mean_var_to_expectation(meanZ, grads)  # :284
    scale = (float(model.num_data) / N) if model.num_data is not None else 1.0  # :286-291
    lambda_2 = -0.5 * model.lambda_2  # :293
    lambda_1 = (1 - lr) * model.lambda_1 + lr * scale * grad_mu[0]  # :296
    lambda_2 = (1 - lr) * lambda_2 + lr * scale * grad_mu[1]  # :297
    lambda_2_sqrt = -_chol(-2.0 * lambda_2 + Id * jitter)  # :300
    model.sites.lambda_1 = lambda_1  # :302
    model.sites._lambda_2_sqrt = np.tril(lambda_2_sqrt)  # :303
    model.get_mean_chol_cov_inducing_posterior()  # :304 (result discarded)
    _t = _time.perf_counter()
    model.last = dict(mean=np.concatenate(means), var=np.concatenate(vars_), meanZ=meanZ, g0=np.concatenate(g0s),
                      g1=np.concatenate(g1s), G0=grads[0], G1=grads[1],
                      elbo_before=math.fsum(ve) * scale - kl_before, A_abs_colsum=A_abs.value())
    model.last["extras_seconds"] = extras + (_time.perf_counter() - _t)
