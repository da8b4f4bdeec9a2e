"""Site-form linear algebra of the hot path (mirror of reference src/util.py), on torch tensors.

Everything here is M x M (replicated, fp64) work: O(M^3), never O(N).  The N-sized contractions that the
reference performs with dense TensorFlow ops are in the HIP kernels (csrc/tsvgp_kernels.hip); the functions
below prepare their small operands and consume their M-sized outputs.

Shapes follow the reference: K [M, M], lambda_1 [M, P], lambda_2_sqrt [P, M, M].
"""
from __future__ import annotations

import torch


def bmv(A: torch.Tensor, v: torch.Tensor, transpose: bool = False) -> torch.Tensor:
    """Per-latent matrix-vector products: out[:, p] = A_p v[:, p] (A_p^T v[:, p] with ``transpose``), A [M, M] (shared by
    all latents) or [P, M, M], v [M, P] -> [M, P].  Few latents go through gemv (``torch.mv``) for A v -- einsum / matmul hand
    a one-column right-hand side to a GEMM kernel that takes 30-60 us at M = 1024 where gemv takes ~8 -- and through a
    broadcast multiply + column sum for A^T v (rocBLAS's non-transposed gemv on a row-major matrix takes 50 us)."""
    P = v.shape[1]
    if A.dim() == 2 and P > 4:
        return (A.transpose(-1, -2) if transpose else A) @ v
    if P <= 4:
        cols = []
        for p in range(P):
            Ap = A if A.dim() == 2 else A[p]
            cols.append((Ap * v[:, p, None]).sum(dim=0) if transpose else torch.mv(Ap, v[:, p]))
        return torch.stack(cols, dim=1)
    return torch.einsum("pkm,kp->mp" if transpose else "pmk,kp->mp", A, v)


def cholesky(a: torch.Tensor) -> torch.Tensor:
    """tf.linalg.cholesky equivalent; FloatingPointError stands in for TF's
    'Cholesky decomposition was not successful' InvalidArgumentError."""
    L, info = torch.linalg.cholesky_ex(a, upper=False, check_errors=False)
    if bool((info != 0).any()):
        raise FloatingPointError("Cholesky decomposition was not successful (matrix not positive definite)")
    return L


def cholesky_deferred(a: torch.Tensor, infos: list, potrf=None, inverse: bool = False, overwrite: bool = False,
                      scale: float = 1.0):
    """Cholesky without a host synchronisation: the LAPACK-style ``info`` tensor is appended to ``infos`` and
    checked later in one device->host read (see ``t_SVGP._check_step``).  ``potrf`` is the engine's HIP
    factorisation (``EStepEngine.cholesky``); without it torch's is used.  With ``inverse`` returns (L, inv(L)).
    ``overwrite``: ``a`` is a temporary that the factorisation may destroy.  ``scale``: the factor is returned times
    this (not the inverse)."""
    Linv = None
    if potrf is not None:
        res = potrf(a, inverse=inverse, overwrite=overwrite, scale=scale)
        L, info = res[0], res[1]
        Linv = res[2] if inverse else None
    else:
        L, info = torch.linalg.cholesky_ex(a, upper=False, check_errors=False)
        if inverse:
            Id = torch.eye(a.shape[-1], dtype=a.dtype, device=a.device)
            Linv = torch.linalg.solve_triangular(L, Id, upper=False)
        if scale != 1.0:
            L = L * scale
    infos.append(info.reshape(-1).to(torch.int32))  # raw: reduced once per call by ``info_sum``
    return (L, Linv) if inverse else L


def info_sum(infos) -> torch.Tensor:
    """[1] fp64 tensor: sum of |info| over the factorisations collected by ``cholesky_deferred`` (0 = all succeeded)."""
    return torch.cat(list(infos)).abs().sum().to(torch.float64).reshape(1)


def rev_cholesky(a: torch.Tensor, infos: list, potrf=None, inverse: bool = False):
    """Upper-form Cholesky a = U U^T, U upper triangular: the lower factor of the index-reversed matrix, reversed back
    (J a J = C C^T  =>  a = (J C J)(J C J)^T, and U^-1 = J C^-1 J).  Status handling as in ``cholesky_deferred``.
    With ``inverse`` returns (U, inv(U)).  The engine's factorisation takes the reversal into its own copy passes
    (``EStepEngine.cholesky(upper_form=True)``: no separate flip kernels)."""
    if potrf is not None:
        res = potrf(a, inverse=inverse, upper_form=True)
        infos.append(res[1].reshape(-1).to(torch.int32))
        return (res[0], res[2]) if inverse else res[0]
    res = cholesky_deferred(torch.flip(a, (-2, -1)), infos, None, inverse, overwrite=True)  # the flipped copy is ours
    if inverse:
        return torch.flip(res[0], (-2, -1)), torch.flip(res[1], (-2, -1))
    return torch.flip(res, (-2, -1))


def _cond2_engine(A: torch.Tensor, eng, iters: int, v0: torch.Tensor) -> torch.Tensor:
    """``cond2_estimate`` of ONE matrix through the engine's own kernels: the upper-form factorisation with the identity
    riding along (``EStepEngine.cholesky_solve_upper``: A = U U^T and D = U^-1 in one pass, so A^-1 = D^T D) and row GEMVs
    (``EStepEngine.gemv``, ~5 us each at M = 1024 instead of a rocBLAS call plus a norm and a division per product).  The
    iterates are normalised every fourth step only: the growth per step is at most lambda_max (resp. 1 / lambda_min
    <= 1 / jitter), far inside fp64's range over four steps."""
    M = A.shape[-1]
    Id = torch.eye(M, dtype=torch.float64, device=A.device)
    _, info, D = eng.cholesky_solve_upper(A[None], Id[None])
    D = D[0]
    Dt = D.transpose(0, 1).contiguous()
    v, w = v0.clone(), v0.clone()
    for it in range(iters):
        v = eng.gemv(A, v)
        w = eng.gemv(Dt, eng.gemv(D, w))
        if it % 4 == 3 or it == iters - 1:
            v = v / torch.linalg.vector_norm(v, dim=0, keepdim=True)
            w = w / torch.linalg.vector_norm(w, dim=0, keepdim=True)
    lam_max = torch.sum(v * eng.gemv(A, v), dim=0).amax()
    Dw = eng.gemv(D, w)
    inv_min = torch.sum(Dw * Dw, dim=0).amax()
    cond = (lam_max * inv_min).reshape(1)
    bad = (info.reshape(-1)[:1] != 0) | ~torch.isfinite(cond)
    return torch.where(bad, torch.full_like(cond, float("inf")), cond)


def cond2_estimate(A: torch.Tensor, potrf=None, iters: int = 32) -> torch.Tensor:
    """Estimate of the 2-norm condition number of the symmetric positive definite A [..., M, M] (one value per matrix,
    device tensor, no host synchronisation): lambda_max by power iteration on A, 1 / lambda_min by power iteration on
    A^-1 = X^T X with X the inverse Cholesky factor (one factorisation; ``potrf`` as in ``cholesky_deferred``).  Both
    Rayleigh quotients approach their eigenvalue from inside the spectrum, so the estimate is a LOWER bound of cond_2,
    within a few percent after 32 steps on kernel matrices (their small eigenvalues decay geometrically or sit on the
    jitter); ``inf`` where the factorisation fails.  Replaces a symmetric eigendecomposition (22 ms at M = 1024 through
    rocSOLVER) by ~2 ms of GEMVs in the route gate of the models (``_cond2_engine`` when ``potrf`` is the engine's)."""
    A = A if A.dim() == 3 else A[None]
    M = A.shape[-1]
    g = torch.Generator(device="cpu").manual_seed(1234)
    v0 = torch.randn(M, 2, generator=g, dtype=torch.float64).to(A.device)
    eng = getattr(potrf, "__self__", None)  # ``potrf`` is the bound EStepEngine.cholesky: the engine's kernels do the rest
    if eng is not None and A.is_cuda and A.dtype == torch.float64 and hasattr(eng, "gemv"):
        return torch.cat([_cond2_engine(A[b].contiguous(), eng, iters, v0) for b in range(A.shape[0])])
    infos = []
    _, X = cholesky_deferred(A.clone(), infos, potrf, inverse=True, overwrite=True)
    v = v0.expand(A.shape[0], M, 2).clone()  # two start vectors per matrix: the larger quotient is kept
    w = v.clone()
    for _ in range(iters):
        v = A @ v
        v = v / torch.linalg.vector_norm(v, dim=-2, keepdim=True)
        w = X.transpose(-1, -2) @ (X @ w)
        w = w / torch.linalg.vector_norm(w, dim=-2, keepdim=True)
    lam_max = torch.sum(v * (A @ v), dim=-2).amax(dim=-1)
    Xw = X @ w
    inv_min = torch.sum(Xw * Xw, dim=-2).amax(dim=-1)
    cond = lam_max * inv_min
    bad = (infos[0].reshape(-1) != 0) | ~torch.isfinite(cond)
    return torch.where(bad, torch.full_like(cond, float("inf")), cond)


def _check_site_shapes(K, lambda_1, lambda_2_sqrt, who):
    if K.dim() < 2 or K.shape[-1] != K.shape[-2]:
        raise ValueError(f"{who}: K must be [..., M, M]")
    M = K.shape[-1]
    if lambda_2_sqrt.dim() != 3 or lambda_2_sqrt.shape[1] != M or lambda_2_sqrt.shape[2] != M:
        raise ValueError(f"{who}: lambda_2_sqrt must be [P, M, M]")
    if lambda_1.dim() != 2 or lambda_1.shape[0] != M or lambda_1.shape[1] != lambda_2_sqrt.shape[0]:
        raise ValueError(f"{who}: lambda_1 must be [M, P]")


def posterior_from_dense_site(K, lambda_1, lambda_2_sqrt):
    """Mean and Cholesky factor of q(u) = p(u) t(u) = N(u; m, S)   (reference src/util.py:349-391).

    S = (K^-1 + L L^T)^-1 = K - K L W^-1 L^T K,  W = I + L^T K L,  m = S lambda_1.
    Returns m [M, P], chol(S) [P, M, M].
    """
    _check_site_shapes(K, lambda_1, lambda_2_sqrt, "posterior_from_dense_site()")
    L = lambda_2_sqrt
    Id = torch.eye(K.shape[-1], dtype=K.dtype, device=K.device)
    C = cholesky(K)
    CtL = C.transpose(-1, -2) @ L
    W = Id + CtL.transpose(-1, -2) @ CtL
    chol_W = cholesky(W)
    LtK = L.transpose(-1, -2) @ K
    iwLtK = torch.linalg.solve_triangular(chol_W, LtK, upper=False)
    S_q = K - iwLtK.transpose(-1, -2) @ iwLtK
    chol_S_q = cholesky(S_q)
    m_q = torch.einsum("lmn,nl->ml", S_q, lambda_1)
    return m_q, chol_S_q


def conditional_from_precision_sites(Kuu, Kff, Kuf, l, L=None, L2=None):
    """Predictive moments straight from the sites (reference src/util.py:91-185), dense torch form for small N.

    Kuu [M, M], Kff [N, 1], Kuf [M, N], l [M, P], L [P, M, M] -> mean [N, P], cov [N, P].
    The model's ``new_predict_f`` runs the same algebra through the HIP moments kernel instead.
    """
    if L is None:
        L = cholesky(L2)
    _check_site_shapes(Kuu, l, L, "conditional_from_precision_sites()")
    if Kuf.dim() != 2 or Kuf.shape[0] != Kuu.shape[0] or Kff.shape != (Kuf.shape[1], 1):
        raise ValueError("conditional_from_precision_sites(): Kuf must be [M, N] and Kff [N, 1]")
    D = site_projection_D(Kuu, L)
    tmp = D @ Kuf  # [P, M, N]
    DKl = D @ (Kuu @ l.transpose(-1, -2)[..., None])  # [P, M, 1]
    mean = Kuf.transpose(-1, -2) @ l - torch.sum(DKl * tmp, dim=-2).transpose(-1, -2)
    cov = Kff - torch.sum(torch.square(tmp), dim=-2).transpose(-1, -2)
    return mean, cov


def _dense_site(L, L2, who):
    if L is None and L2 is None:
        raise ValueError(f"{who}: one of L (a factor of the precision site) and L2 (the site itself) is needed")
    return L @ L.transpose(-1, -2) if L2 is None else L2


def conditional_from_precision_sites_white(Kuu, Kff, Kuf, l, L=None, L2=None, jitter=1e-9):
    """Predictive moments for sites stored pre-multiplied by K_uu (reference src/util.py:11-88), dense torch form for small N.

    With R = L2 + Kuu + jitter I:  mean = Kuf^T R^-1 l,  cov = Kff - |chol(Kuu)^-1 Kuf|^2 + |chol(R)^-1 Kuf|^2 (column sums).
    Kuu [M, M], Kff [N, 1], Kuf [M, N], l [M, 1], L / L2 [1, M, M] -> mean [N, 1], cov [N, 1] (one latent, as the
    reference's ``[0]`` at :87).  ``t_SVGP_white.predict_f`` runs the same algebra through the HIP moments kernel instead.
    """
    L2 = _dense_site(L, L2, "conditional_from_precision_sites_white()")
    if Kuf.dim() != 2 or Kuf.shape[0] != Kuu.shape[-1] or Kff.shape != (Kuf.shape[1], 1):
        raise ValueError("conditional_from_precision_sites_white(): Kuf must be [M, N] and Kff [N, 1]")
    Id = torch.eye(Kuu.shape[-1], dtype=Kuu.dtype, device=Kuu.device)
    LR = cholesky(L2 + Kuu + jitter * Id)  # [1, M, M]
    LA = cholesky(Kuu)
    a = torch.linalg.solve_triangular(LA, Kuf, upper=False)  # [M, N]
    r = torch.linalg.solve_triangular(LR, Kuf.expand(LR.shape[0], -1, -1), upper=False)  # [1, M, N]
    cov = Kff - (torch.sum(a * a, dim=-2) - torch.sum(r * r, dim=-2)).transpose(-1, -2)
    mean = (Kuf.transpose(-1, -2) @ torch.cholesky_solve(l, LR))[0]
    return mean, cov


def kl_from_precision_sites_white(A, l, L=None, L2=None):
    """KL[q(u) || p(u)] for q(u) ~ N(u; A R^-1 l, A R^-1 A), R = L2 + A, p(u) = N(0, A)   (reference src/util.py:239-291):
        1/2 [ log|R| - log|A| + tr(R^-1 A) - M + |chol(A)^T R^-1 l|^2 ].   A [M, M], l [M, 1], L / L2 [1, M, M] -> scalar."""
    L2 = _dense_site(L, L2, "kl_from_precision_sites_white()")
    LR = cholesky(L2 + A)
    LA = cholesky(A)
    logdet = lambda C: 2.0 * torch.sum(torch.log(torch.diagonal(C, dim1=-2, dim2=-1)))
    t = torch.linalg.solve_triangular(LR, LA.expand(LR.shape[0], -1, -1), upper=False)
    maha = LA.transpose(-1, -2) @ torch.cholesky_solve(l, LR)
    return 0.5 * (logdet(LR) - logdet(LA) + torch.sum(t * t) - float(A.shape[-1]) + torch.sum(maha * maha))


# the reference keeps a second copy of the same function under this name (src/util.py:294-346)
kl_from_precision_sites = kl_from_precision_sites_white


def posterior_from_dense_site_white(K, lambda_1, lambda_2, jitter=1e-9):
    """Mean and Cholesky factor of q(u) for sites stored pre-multiplied by K (reference src/util.py:394-426):
    R = K + lambda_2 (+ jitter I in its factor),  S = K R^-1 K,  m = K R^-1 lambda_1.
    K [M, M], lambda_1 [M, 1], lambda_2 [1, M, M] -> m [M, 1], chol(S) [1, M, M]."""
    Id = torch.eye(K.shape[-1], dtype=K.dtype, device=K.device)
    LR = cholesky(K + lambda_2 + jitter * Id)
    t = torch.linalg.solve_triangular(LR, K.expand(LR.shape[0], -1, -1), upper=False)
    S_q = t.transpose(-1, -2) @ t
    m_q = (K @ torch.cholesky_solve(lambda_1, LR))[0]
    return m_q, cholesky(S_q)


_factor = cholesky  # project_diag_sites keeps the reference's keyword ``cholesky``, which hides the function inside it


def project_diag_sites(Kuf, lambda_1, lambda_2, Kuu=None, cholesky=True):
    """Per-datum (diagonal) sites projected onto the inducing points (reference src/util.py:188-236):
        l = P lambda_1,  L2 = P diag(lambda_2) P^T,  P = Kuf, or Kuu^-1 Kuf when ``Kuu`` is given;  returns (l, chol(L2)) or (l, L2).
    Kuf [M, N] or [P, M, N], lambda_1 / lambda_2 [N, P], Kuu [M, M] or [P, M, M] -> l [M, P], L [P, M, M].  Dense torch form:
    the reference uses it in its per-datum-site model (t_SVGP_sites, out of this package's scope) and its tests."""
    nl = lambda_1.shape[-1]
    Pm = Kuf[None].expand(nl, -1, -1) if Kuf.dim() == 2 else Kuf
    if Kuu is not None:
        Kb = Kuu[None] if Kuu.dim() == 2 else Kuu
        Pm = torch.cholesky_solve(Pm, _factor(Kb).expand(nl, -1, -1))
    l = torch.einsum("lmn,nl->ml", Pm, lambda_1)
    L2 = torch.einsum("lmn,lon,nl->lmo", Pm, Pm, lambda_2)
    return l, (_factor(L2) if cholesky else L2)


def site_projection_D(K, L, return_chol=False, infos=None, potrf=None):
    """D = chol(W)^-1 L^T with W = I + L^T K L  (reference src/util.py:168-175); [P, M, M].

    W is formed directly from K (no Cholesky of K needed): W = I + L^T (K L).
    With ``infos`` (a list) the factorisation does not synchronise; its status is appended for a deferred check.
    """
    Id = torch.eye(K.shape[-1], dtype=K.dtype, device=K.device)
    W = Id + L.transpose(-1, -2) @ (K @ L)
    W = 0.5 * (W + W.transpose(-1, -2))
    chol_W = cholesky(W) if infos is None else cholesky_deferred(W, infos, potrf)
    D = torch.linalg.solve_triangular(chol_W, L.transpose(-1, -2), upper=False)
    return (D, chol_W) if return_chol else D


def gradient_transformation_mean_var_to_expectation(inputs, grads):
    """Chain rule (mean, var) -> (mu_1, mu_2) (reference src/util.py:429-438)."""
    return grads[0] - 2.0 * bmv(grads[1], inputs), grads[1]


def kl_from_dense_site(K, lambda_1, D, chol_W, beta):
    """KL[q(u) || p(u)] for the dense site, per the identities
        tr(K^-1 S) = M - tr(D K D^T),  log|S| = log|K| - log|W|,  m^T K^-1 m = (K beta)^T beta,
    which make gpflow.kullback_leiblers.gauss_kl (reference tsvgp.py:65-70) computable without chol(S):
        KL = 1/2 sum_p [ m_p^T beta_p - tr(D_p K D_p^T) + log|W_p| ].
    """
    m = bmv(K, beta)  # [M, P]; K [P, M, M]: one prior per latent
    maha = torch.sum(m * beta)
    trace = torch.sum(D * (D @ K))
    logdetW = 2.0 * torch.sum(torch.log(torch.diagonal(chol_W, dim1=-2, dim2=-1)))
    return 0.5 * (maha - trace + logdetW)
