"""Host-side driver of one N-pass of the E-step on one GPU: owns the HBM work buffers and launches the HIP kernels.

Data layout in HBM (T = compute dtype, fp64 or fp32; Np/Mp = N/M rounded up to 128, padding zero-filled):

    X    [N  x D ]  inputs (row-major, read once per pass by the fill kernel)
    Y    [N  x P ]  targets
    Kfu  [Np x Mp]  cross-covariance K(X, Z), written once by ``tsvgp_se_fill``
    B    [Np x Mp]  whitened cross-covariance  B = Kfu U9^-T  (Kuu + jitter I = U9 U9^T), written by ``tsvgp_trmm``
                    (whitened / projected routes only; on the projected route a = U9^-T b then overwrites Kfu)
    g0,g1[Np x P ]  likelihood gradients (rows >= N are zero)
    work            partial 128x128 tiles of the weighted Gram, [P][nsplit][ntri][128*128] (+ first-order partials)

Kernel sequence of ``run(..., sites=True)`` by projection route (models/tsvgp.py):
    direct:     fill -> moments(UPPER, on Kfu) -> site_accum(Kfu)
    whitened:   fill -> trmm(UPPER) -> moments(UPPER, on B) -> site_accum(B)
    projected:  fill -> trmm(UPPER) -> moments(UPPER, on B) -> trmm(LOWER) -> site_accum(a)
``mean_only`` replaces the moments product by one HBM-bound sweep (Gaussian likelihood, TSVGP_LIK_MEANONLY).
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Optional

import torch

from . import _backend as B
from .distributed import world_size
from .kernels import SeparateIndependent

MAX_INPUT_DIM = 32  # the fill kernel pads D to a compile-time size (1, 2, 4, 8, 16, 32)
MAX_INPUT_DIM_GRAD = 16  # tsvgp_kernel_grad_* (M-step): 1, 2, 4, 8, 16; beyond: GEMM form (tsvgp_gram_to_gradw_*)


@dataclass
class EStepStats:
    """Per-shard outputs of one N-pass (all fp64 on the model device)."""

    n_rows: int
    ve_sum: torch.Tensor  # scalar: sum of variational expectations over the shard's rows
    nonpos: torch.Tensor  # scalar: number of rows with non-positive predictive variance
    acc2: Optional[torch.Tensor] = None  # [P, M, M] sum_n g1 b_n b_n^T (whitened coordinates)
    acc1: Optional[torch.Tensor] = None  # [P, M]    sum_n g0 b_n
    mean: Optional[torch.Tensor] = None  # [N, P]
    var: Optional[torch.Tensor] = None  # [N, P]
    g0: Optional[torch.Tensor] = None  # [N, P]
    g1: Optional[torch.Tensor] = None  # [N, P]
    tile: Optional[torch.Tensor] = None  # [Np, Mp] the stored triangular product t_n = Tm k_n (run(keep_tile=True))


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def site_sum_chunk_rows(f64: bool, P: int) -> int:
    """Rows per k-chunk of the site-sum kernel that runs for this type and latent count: syrk1_kernel (fp64) 16,
    syrk1f_kernel (fp32, P <= 8) 32, the round-2 syrk_kernel (P > 8, either type) 16.  A slice shorter than one chunk is an
    empty workgroup (harmless, wasted): the slice count is capped at Np / this."""
    return 32 if (not f64 and P <= 8) else 16


def site_sum_slices(Mp: int, P: int, Np, slots: int, f64: bool, oversubscribe=None) -> int:
    """The slice count of ``EStepEngine.choose_nsplit`` as a pure function of the shape and of the resident workgroup slots
    (256 for fp64, 512 for fp32 on MI355X): see there for the cost model and the measurements behind it."""
    nt = Mp // B.TILE
    n_off = nt * (nt - 1) // 2
    num = 20 if f64 else 22  # TSVGP_SYRK1_DIAG_NUM / TSVGP_SYRK1F_DIAG_NUM
    ns_diag = lambda ns: max(1, (num * ns + 31) // 32)
    if Np is None or oversubscribe is not None:
        # (no row count, or a forced oversubscription: the largest count that fits one round, times the factor)
        ns = 1
        while P * (n_off * (ns + 1) + nt * ns_diag(ns + 1)) <= slots:
            ns += 1
        return ns * (oversubscribe if oversubscribe is not None else 8)
    chunks = max(1, Np // site_sum_chunk_rows(f64, P))  # the kernels' chunks: 16 rows (fp64) / 32 rows (fp32, P <= 8)
    overhead = 14.0
    best, best_t = 1, None
    for ns in range(1, max(1, min(chunks // 32, 1024)) + 1):
        nd = ns_diag(ns)
        rounds = -(-(P * (n_off * ns + nt * nd)) // slots)
        per = max(-(-chunks // ns) if n_off else 0, -(-chunks // nd) * num / 32.0)
        t = rounds * (per + overhead) * (1.02 if rounds == 1 else 1.0)
        if best_t is None or t < best_t * (1.0 - 1e-9):
            best, best_t = ns, t
    return best


class EStepEngine:
    """Launches the C-ABI kernels (include/tsvgp_hip.h) on torch-allocated device memory."""

    def __init__(self, compute_dtype=torch.float64, device=None):
        self.lib = B.lib()  # raises HipExtensionError when the extension is not built
        if not torch.cuda.is_available():
            raise B.HipExtensionError("no ROCm device visible: the t-SVGP E-step has no CPU fallback")
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        if self.device.type != "cuda":
            raise B.HipExtensionError(f"EStepEngine needs a ROCm device, got {self.device}")
        self.dtype = compute_dtype
        self.sfx = B.suffix(compute_dtype)
        self._buf = {}
        self._slots = None
        self.nsplit_override = None
        self.syrk_oversubscribe = None  # None: by kernel (see choose_nsplit)
        self._b_tag = None  # identifies the contents of the cached whitened buffer B
        self._side = None  # side stream of start_fill
        self.last_batched = False  # the last pass over separate kernels ran as batched launches
        self.last_trmm_batch = 1  # latents per whitening launch of that pass
        self.profile = None  # set to a dict to record (start, stop) HIP events per kernel launch on the launch stream
        self.profile_only = None  # a set of kernel names: bracket only these launches
        # A/B switch (tools/dev_diag2.py): round 4's diagonal-block kernel instead of round 5's (TSVGP_POTRF_DIAG_V1)
        self.potrf_flags = B.POTRF_DIAG_V1 if os.environ.get("TSVGP_POTRF_DIAG_V1") == "1" else 0
        # Clock keeper (``keeper_begin`` / ``keeper_end``), an experiment that is OFF by default: it holds the clock in isolation
        # but costs the M x M chain as much as the N-pass gains (profiles/r05_clock_lab.txt).  TSVGP_CLOCK_KEEPER=-1: one
        # workgroup per CU, =N: N workgroups; TSVGP_KEEPER_MAX_US: the bound after which its waves leave on their own
        self.clock_keeper = int(os.environ.get("TSVGP_CLOCK_KEEPER", "0"))
        self.keeper_max_us = float(os.environ.get("TSVGP_KEEPER_MAX_US", "4000"))
        self._keeper_side = None
        self._keeper_flag = None

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def _fn(self, name, dtype=None):
        return getattr(self.lib, f"{name}_{B.suffix(dtype) if dtype is not None else self.sfx}")

    def _get(self, key, shape, dtype):
        t = self._buf.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=self.device)
            self._buf[key] = t
        return t

    def _launch(self, name, status_fn):
        """Runs one C-ABI launch; with profiling on, brackets it with events on the stream it is launched on (events
        come from a pool: creating two per launch costs the host ~0.1 ms per step, which shows at small shards)."""
        if self.profile is None or (self.profile_only is not None and name not in self.profile_only):
            B.check(status_fn(), name)
            return
        e0, e1 = self._event(), self._event()
        e0.record(torch.cuda.current_stream(self.device))
        B.check(status_fn(), name)
        e1.record(torch.cuda.current_stream(self.device))
        self.profile.setdefault(name, []).append((e0, e1))

    def _event(self):
        pool = self.__dict__.setdefault("_event_pool", [])
        return pool.pop() if pool else torch.cuda.Event(enable_timing=True)

    def reserve_events(self, n: int):
        """Pre-creates n timing events (bench.py: before the timed region)."""
        pool = self.__dict__.setdefault("_event_pool", [])
        while len(pool) < n:
            pool.append(torch.cuda.Event(enable_timing=True))

    def profile_summary(self):
        """{kernel: (launches, mean ms, min ms, median ms, max ms, [ms in launch order])} from the recorded events (synchronises); the events go
        back to the pool."""
        torch.cuda.synchronize(self.device)
        out = {}
        pool = self.__dict__.setdefault("_event_pool", [])
        for name, evs in (self.profile or {}).items():
            series = [a.elapsed_time(b) for a, b in evs]  # in launch order
            ms = sorted(series)
            n = len(ms)
            med = 0.0 if n == 0 else (ms[n // 2] if n % 2 else 0.5 * (ms[n // 2 - 1] + ms[n // 2]))
            out[name] = (n, sum(ms) / max(n, 1), ms[0] if n else 0.0, med, ms[-1] if n else 0.0, series)
            for a, b in evs:
                pool += [a, b]
        if self.profile is not None:
            self.profile.clear()
        return out

    def release(self):
        """Drop the cached work buffers."""
        self._buf.clear()
        self._b_tag = None

    def slots(self) -> int:
        if self._slots is None:
            with torch.cuda.device(self.device):
                s = int(self._fn("tsvgp_site_accum_slots")())
            if s <= 0:
                raise B.HipExtensionError("tsvgp_site_accum_slots failed")
            self._slots = s
        return self._slots

    def choose_nsplit(self, Mp: int, P: int, Np: int = None) -> int:
        """Number of N-slices per off-diagonal tile of the weighted Gram (diagonal tiles cost less per row and get
        correspondingly fewer, longer slices: the rule of syrk_ns_diag in the kernel source, mirrored in ``ns_diag`` below).
        All workgroups of the launch take the same time by construction, so the launch runs in ROUNDS of `slots` workgroups
        (one per CU for fp64, two for fp32) and costs  rounds * (chunks per slice + o):  o ~ 14 chunk-times per workgroup for
        its first fetch, the 128 x 128 partial tile it writes and the reduction's read of it (fitted at M = 512: 58 / 116 /
        232 / 464 / 651 slices take 4.20 / 4.27 / 4.43 / 4.64 / 4.85 ms, gpurun_out/r3m/kbench_m512.txt).  The slice count is
        the one that minimises this: whole rounds, and few of them.  At N = 1e6 (gpurun_out/r3m/nsplit_rounds.txt):
        M = 1024 fp64: 62 slices (2048 workgroups = 8 rounds exactly) 15.19 ms, 124: 15.32, 224 (28.9 rounds): 15.62,
        56 (7.2 rounds): 17.0; M = 1024 fp32: 61 slices 7.65 ms, 120: 7.99; M = 512 fp64: 120 slices 4.15 ms, 651: 4.72.
        A launch of exactly one round is charged 2 % more (nothing evens out the diagonal against the off-diagonal tiles:
        M = 512, 29 slices = one round 4.33 ms against 4.20 for two)."""
        return site_sum_slices(Mp, P, Np, self.slots(), self.dtype == torch.float64, self.syrk_oversubscribe)

    def _padded_gamma(self, gamma: torch.Tensor, Mp: int, P: int) -> torch.Tensor:
        """gamma [M, P] fp64 -> [Mp, P] in the compute dtype, rows >= M zero: a cached buffer whose padding is zeroed once (it was a
        fill and a copy per pass in front of the moments kernel)."""
        M = gamma.shape[0]
        if M == Mp and gamma.dtype == self.dtype and gamma.is_contiguous():
            return gamma
        out = self._buf.get("pad_gamma")
        if out is None or tuple(out.shape) != (Mp, P) or out.dtype != self.dtype:
            out = self._buf["pad_gamma"] = torch.zeros((Mp, P), dtype=self.dtype, device=self.device)
        out[:M].copy_(gamma)
        return out

    def _pad_square(self, A: torch.Tensor, Mp: int, key: str = None) -> torch.Tensor:
        """[.., M, M] fp64 -> zero-padded contiguous [.., Mp, Mp] in the compute dtype (a cached buffer per key)."""
        M = A.shape[-1]
        if M == Mp and A.dtype == self.dtype and A.is_contiguous():
            return A
        shape = tuple(A.shape[:-2]) + (Mp, Mp)
        out = self._buf.get(key) if key else None
        if out is None or tuple(out.shape) != shape:
            out = torch.zeros(shape, dtype=self.dtype, device=self.device)
            if key:
                self._buf[key] = out
        out[..., :M, :M] = A
        return out

    # ------------------------------------------------------------------ kernels
    def se_fill(self, X: torch.Tensor, Z: torch.Tensor, inv_ls: torch.Tensor, variance: float, out: torch.Tensor,
                kind: int = B.KERNEL_SE):
        """out[Np x Mp] <- K(X, Z) (padding zero) for the stationary kernel ``kind``.  X, Z, inv_ls, out share one dtype."""
        N, D = X.shape
        M = Z.shape[0]
        if D > MAX_INPUT_DIM:
            # beyond the fused kernel's compile-time sizes the scaled distance is a GEMM (GPflow's own square_distance
            # form): x~ z~^T from the BLAS library into the padded buffer, then tsvgp_gram_to_kernel_* in place
            Xs, Zs = X * inv_ls, torch.zeros((out.shape[1], D), dtype=X.dtype, device=X.device)
            Zs[:M] = Z * inv_ls
            torch.mm(Xs, Zs.t(), out=out[:N])
            xx, zz = (Xs * Xs).sum(dim=1).contiguous(), (Zs[:M] * Zs[:M]).sum(dim=1).contiguous()
            fn = self._fn("tsvgp_gram_to_kernel", X.dtype)
            with torch.cuda.device(self.device):
                self._launch("tsvgp_se_fill", lambda: fn(kind, out.data_ptr(), xx.data_ptr(), zz.data_ptr(), float(variance),
                                                         N, M, out.shape[1], self._stream()))
            return out
        fn = self._fn("tsvgp_kernel_fill", X.dtype)
        with torch.cuda.device(self.device):
            self._launch("tsvgp_se_fill" if N != M or X.data_ptr() != Z.data_ptr() else "tsvgp_se_fill(Kuu)",
                         lambda: fn(kind, X.data_ptr(), Z.data_ptr(), inv_ls.data_ptr(), float(variance), out.data_ptr(),
                                    N, M, D, out.shape[1], self._stream()))
        return out

    def kuu(self, Z: torch.Tensor, kernel) -> torch.Tensor:
        """K(Z, Z) in fp64, [M, M] (no jitter); [P, M, M] for separate per-latent kernels."""
        if isinstance(kernel, SeparateIndependent):
            return torch.stack([self.kuu(Z, k) for k in kernel.kernels])
        Z = Z.to(device=self.device, dtype=torch.float64).contiguous()
        M, D = Z.shape
        Mp = B.round_up(M)
        out = torch.empty((Mp, Mp), dtype=torch.float64, device=self.device)
        inv_ls = kernel.inv_lengthscales(D, torch.float64, self.device)
        self.se_fill(Z, Z, inv_ls, kernel.variance.item(), out, kernel.kind)
        return out[:M, :M].contiguous()

    def tri_copy(self, src: torch.Tensor, M: int, scale: float = 1.0, flip: int = 0, out: torch.Tensor = None,
                 diag_add: float = 0.0) -> torch.Tensor:
        """[nb, M, M] contiguous <- triangle / index reversal of the leading M x M block of src [nb, R, C]
        (``tsvgp_tri_copy_shift_f64``: flip 0 = lower triangle, 1 = reversed indices, upper triangle, 2 = reversed, everything,
        3 = as it is, everything); ``diag_add`` is added on the diagonal behind the scale.
        ``out``: a [nb, >= M, >= M] view (unit element stride) whose leading M x M blocks are written instead."""
        nb, R, C = src.shape
        if out is None:
            out = torch.empty((nb, M, M), dtype=torch.float64, device=self.device)
        assert src.stride(2) == 1 and out.stride(2) == 1 and out.shape[0] == nb
        with torch.cuda.device(self.device):
            B.check(self.lib.tsvgp_tri_copy_shift_f64(src.data_ptr(), src.stride(1), src.stride(0), out.data_ptr(), out.stride(1),
                                                      out.stride(0), M, nb, float(scale), float(diag_add), int(flip), self._stream()),
                    "tsvgp_tri_copy")
        return out

    def cholesky_solve_upper(self, A: torch.Tensor, Lrhs, robust: bool = False, beside_fill: bool = False):
        """Upper-form factorisation AND the triangular solve behind it in one pass (``tsvgp_potrf_solve_f64``):
            A = U U^T (U upper triangular, as ``cholesky(upper_form=True)``),   D = U^-1 Lrhs^T   (upper triangular)
        for A [.., M, M] and a lower triangular Lrhs [.., M, M] -- reference src/util.py:168-175 with W = I + L^T K L for A and the
        site factor L for Lrhs; with the identity for Lrhs, D = U^-1 is the inverse factor (what K_uu + jitter I needs, reference
        src/models/tsvgp.py:270-271: the same batch, the same launches).  The index-reversed right-hand side J L J rides through
        the factorisation of J A J as extra panel rows and comes out as (J L J) C^-T = (C^-1 J L^T J)^T; one transposing pass turns
        that into D.  Replaces the inverse recursion (three levels of launch pairs), two triangle copies, a 1024^3 GEMM and a triu
        pass on the critical path of the replicated M x M chain.  ``Lrhs`` may be a LIST of [n_i, M, M] tensors, one run of the
        batch each.  Returns (U, info, D); no host synchronisation.
        ``beside_fill``: the N-sized K(X, Z) fill is in flight on the side stream and will outlast this call.  Round 5's panel
        kernel holds a strip and two column blocks of operands in 250 registers per lane and cannot share a CU with the fill's
        workgroups (224 registers per lane left), round 4's block step (78-87 registers per kernel) can: under the fill of
        N = 1e6 rows the call takes 1.69 ms with round 4's step and 2.13 ms with round 5's (profiles/r05_potrf_under_fill_ab.txt),
        alone 0.49 against 0.45."""
        # ``A`` may also be a LIST of (matrices [n_i, M, M], shift) runs of the batch: run i is factored as matrices_i + shift_i I.
        # The shift rides on the pass that writes J A J where the factorisation reads it, so W = I + L^T K L and K_uu + jitter I
        # need no assembled batch, no copy and no additions on the diagonal in front of this call (round 5).
        if isinstance(A, (list, tuple)):
            runs = [(a.reshape(-1, a.shape[-1], a.shape[-1]), float(sh)) for a, sh in A]
            M = runs[0][0].shape[-1]
            nb = sum(r.shape[0] for r, _ in runs)
            batch_shape = (nb,)
        else:
            M = A.shape[-1]
            batch_shape = A.shape[:-2]
            nb = 1
            for d in batch_shape:
                nb *= int(d)
            runs = [(A.reshape(nb, M, M), 0.0)]
        job = self.solve_upper_begin(nb, M)
        a0 = 0
        for A3, shift in runs:
            self.solve_upper_put(job, a0, A3, shift)
            a0 += A3.shape[0]
        parts, b0 = (list(Lrhs) if isinstance(Lrhs, (list, tuple)) else [Lrhs]), 0
        for part in parts:
            b0 += self.solve_upper_put_rhs(job, b0, part)
        assert b0 == nb, "one right-hand side per matrix of the batch"
        U, info, Dm = self.solve_upper_run(job, robust=robust, beside_fill=beside_fill)
        out_shape = tuple(batch_shape) + (M, M)
        return U.reshape(out_shape), info, Dm.reshape(out_shape)

    # The same call in pieces, for a caller whose operands become available at different times (t_SVGP._site_operands: K_uu +
    # jitter I and both right-hand sides exist before the two GEMMs that make W, and a shard's K(X, Z) fill starts right behind W:
    # handed over early, their passes run on an idle chip -- 7 us each -- instead of beside the fill -- 20-30 us each).
    def solve_upper_begin(self, nb: int, M: int) -> dict:
        """The tall operand [nb, 2 Mp, Mp] of ``tsvgp_potrf_solve_f64``: J A J on top, the right-hand-side rows below."""
        Mp = B.round_up(M)
        tall = (torch.empty if Mp == M else torch.zeros)((nb, 2 * Mp, Mp), dtype=torch.float64, device=self.device)
        if Mp != M:
            tall[:, :Mp].diagonal(dim1=-2, dim2=-1)[:, M:] = 1.0  # chol([[A, 0], [0, I]]) = [[C, 0], [0, I]]
        return dict(tall=tall, nb=nb, M=M, Mp=Mp)

    def solve_upper_put(self, job: dict, b0: int, A3: torch.Tensor, shift: float = 0.0) -> int:
        """Matrices b0 .. of the batch <- A3 [n, M, M] + shift I (index-reversed where the factorisation reads them)."""
        M, Mp = job["M"], job["Mp"]
        A3 = A3.to(device=self.device, dtype=torch.float64).reshape(-1, M, M)
        if A3.stride(2) != 1:
            A3 = A3.contiguous()
        self.tri_copy(A3, M, 1.0, 2, out=job["tall"][b0:b0 + A3.shape[0], :Mp], diag_add=shift)  # J (A + shift I) J
        return A3.shape[0]

    def solve_upper_put_rhs(self, job: dict, b0: int, L3: torch.Tensor) -> int:
        """Right-hand sides b0 .. <- the lower triangular L3 [n, M, M] (J tril(L) J: upper triangular, the rows the solve carries along)."""
        M, Mp = job["M"], job["Mp"]
        L3 = L3.to(device=self.device, dtype=torch.float64).reshape(-1, M, M)
        if L3.stride(2) != 1:
            L3 = L3.contiguous()
        self.tri_copy(L3, M, 1.0, 1, out=job["tall"][b0:b0 + L3.shape[0], Mp:])
        return L3.shape[0]

    def solve_upper_run(self, job: dict, robust: bool = False, beside_fill: bool = False):
        """Factor and solve what ``solve_upper_put`` / ``solve_upper_put_rhs`` handed over: (U [nb, M, M], info, D [nb, M, M])."""
        tall, nb, M, Mp = job["tall"], job["nb"], job["M"], job["Mp"]
        info = torch.empty(nb, dtype=torch.int32, device=self.device)
        work = self._get("potrf_work", (nb, 128 * 128), torch.float64)
        flags = (B.POTRF_SUBST if robust else 0) | B.POTRF_RHS_UPPER | self.potrf_flags | (B.POTRF_DIAG_V1 if beside_fill else 0)
        with torch.cuda.device(self.device):
            self._launch("tsvgp_potrf", lambda: self.lib.tsvgp_potrf_solve_f64(
                tall.data_ptr(), Mp, Mp, nb, 2 * Mp * Mp, info.data_ptr(), work.data_ptr(), Mp, flags, self._stream()))
        U = self.tri_copy(tall[:, :Mp], M, 1.0, 1)
        Dm = torch.empty((nb, M, M), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            B.check(self.lib.tsvgp_flip_transpose_f64(tall[:, Mp:].data_ptr(), Mp, 2 * Mp * Mp, Dm.data_ptr(), M, M * M, M, nb,
                                                      self._stream()), "tsvgp_flip_transpose")
        return U, info, Dm

    def cholesky(self, A: torch.Tensor, inverse: bool = False, overwrite: bool = False, robust: bool = False,
                 scale: float = 1.0, upper_form: bool = False):
        """Batched lower Cholesky on the GPU through ``tsvgp_potrf_f64`` (no host synchronisation).
        A [.., M, M] fp64 (lower triangle referenced) -> (L [.., M, M] with zeros above the diagonal, info [batch] int32);
        with ``inverse`` also inv(L) (``tsvgp_potrf_inv_f64``), lower triangular with exact zeros above.
        ``overwrite``: A is a temporary of the caller and may be factored in place (no copy when M is a multiple of 128).
        ``robust``: solve the panels below each diagonal block by substitution (TSVGP_POTRF_SUBST) instead of multiplying
        with the INVERTED diagonal block, which costs a factor cond(L_kk) of accuracy in the trailing matrix -- irrelevant
        for the matrices of a well-conditioned K_uu (and ~0.15 ms faster), but a numerically barely definite matrix
        (cond ~ 1e14, lambda_min ~ 30 eps lambda_max) then fails where LAPACK-style factorisations go through.  The
        callers ask for it on the "projected" route, whose matrices are of that kind.
        ``scale``: the factor is returned times this (the leading minus of tsvgp.py:300 rides on the triangle copy).
        ``upper_form``: A = U U^T with U upper triangular -- the input is read with reversed indices, factored, and the
        factor (and inverse) written back reversed (``util.rev_cholesky`` without its four flip passes)."""
        A = A.to(device=self.device, dtype=torch.float64)
        flags = (B.POTRF_SUBST if robust else 0) | self.potrf_flags
        M = A.shape[-1]
        batch_shape = A.shape[:-2]
        Mp = B.round_up(M)
        nb = 1
        for d in batch_shape:
            nb *= int(d)
        if upper_form:
            W = self.tri_copy(A.reshape(nb, M, M), M, 1.0, 2)  # J A J (a fresh buffer of ours)
            if Mp != M:
                Wp = torch.zeros((nb, Mp, Mp), dtype=torch.float64, device=self.device)
                Wp[:, :M, :M] = W
                Wp.diagonal(dim1=-2, dim2=-1)[:, M:] = 1.0
                W = Wp
        elif Mp == M:  # no padding: one copy (or none) instead of a zero fill plus a copy
            W = A.reshape(nb, M, M)
            if not (overwrite and W.is_contiguous() and W.data_ptr() == A.data_ptr()):
                W = W.clone(memory_format=torch.contiguous_format)
        else:
            W = torch.zeros((nb, Mp, Mp), dtype=torch.float64, device=self.device)
            W[:, :M, :M] = A.reshape(nb, M, M)
            W.diagonal(dim1=-2, dim2=-1)[:, M:] = 1.0  # chol([[A, 0], [0, I]]) = [[L, 0], [0, I]]  (a fill: capturable)
        info = torch.empty(nb, dtype=torch.int32, device=self.device)
        work = self._get("potrf_work", (nb, 128 * 128), torch.float64)
        out_shape = tuple(batch_shape) + (M, M)
        with torch.cuda.device(self.device):
            if inverse:
                X = torch.empty((2, nb, Mp, Mp), dtype=torch.float64, device=self.device)  # inv(L) and its transpose
                T = self._get("potrf_T", (nb, Mp, Mp), torch.float64)
                self._launch("tsvgp_potrf", lambda: self.lib.tsvgp_potrf_inv_f64(
                    W.data_ptr(), Mp, Mp, nb, Mp * Mp, info.data_ptr(), work.data_ptr(), X[0].data_ptr(), X[1].data_ptr(),
                    T.data_ptr(), flags, self._stream()))
            else:
                self._launch("tsvgp_potrf", lambda: self.lib.tsvgp_potrf_f64(
                    W.data_ptr(), Mp, Mp, nb, Mp * Mp, info.data_ptr(), work.data_ptr(), flags, self._stream()))
        L = self.tri_copy(W, M, scale, 1 if upper_form else 0).reshape(out_shape)
        if inverse:
            if upper_form:
                Xi = self.tri_copy(X[0], M, 1.0, 1).reshape(out_shape)
            else:
                Xi = X[0, :, :M, :M].reshape(out_shape)
            return L, info, Xi
        return L, info

    # ------------------------------------------------------------------ fused M x M helpers of the epilogue
    def site_target(self, G1, LLt, c_ll, c_g, jitter, rows, num_data):
        """(target, G1s) of ``tsvgp_site_target_f64``: G1s = (G1 + G1^T) / 2, target = c_ll LLt + c_g s G1s + jitter I with
        s = num_data / rows (rows: device scalar) or 1 when num_data is None.  [P, M, M] fp64."""
        G1, LLt = G1.contiguous(), LLt.contiguous()
        P, M = G1.shape[0], G1.shape[-1]
        target, G1s = torch.empty_like(G1), torch.empty_like(G1)
        with torch.cuda.device(self.device):
            B.check(self.lib.tsvgp_site_target_f64(G1.data_ptr(), LLt.data_ptr(), target.data_ptr(), G1s.data_ptr(), M, P,
                                                   float(c_ll), float(c_g), float(jitter),
                                                   rows.data_ptr() if num_data is not None else None,
                                                   float(num_data) if num_data is not None else 0.0, self._stream()),
                    "tsvgp_site_target")
        return target, G1s

    def site_update(self, G1, G0, LLt, meanZ, l1_old, lr, jitter, rows, num_data):
        """(target [P, M, M], lambda_1_new [M, P]) of ``tsvgp_site_update_f64``: the symmetrised G1 goes straight into the matrix of
        the final factorisation (1 - lr) LLt - 2 lr s sym(G1) + jitter I and into the chain rule G0 - 2 sym(G1) meanZ, whose
        convex combination with the old lambda_1 comes out of the same call."""
        G1, LLt = G1.contiguous(), LLt.contiguous()
        G0, meanZ, l1_old = G0.contiguous(), meanZ.contiguous(), l1_old.contiguous()
        P, M = G1.shape[0], G1.shape[-1]
        target, l1_new = torch.empty_like(G1), torch.empty_like(l1_old)
        work = self._get("site_update_work", (P * ((M + 31) // 32) * M,), torch.float64)
        with torch.cuda.device(self.device):
            B.check(self.lib.tsvgp_site_update_f64(G1.data_ptr(), G0.data_ptr(), LLt.data_ptr(), meanZ.data_ptr(), l1_old.data_ptr(),
                                                   target.data_ptr(), l1_new.data_ptr(), work.data_ptr(), M, P, float(lr),
                                                   float(jitter), rows.data_ptr() if num_data is not None else None,
                                                   float(num_data) if num_data is not None else 0.0, self._stream()),
                    "tsvgp_site_update")
        return target, l1_new

    def site_beta(self, Dm, v, l1):
        """beta [M, P] = l1 - D^T (D v) per latent (``tsvgp_site_beta_f64``): D [P, M, M] upper triangular, v = K6 l1 and l1 [M, P]."""
        Dm, v, l1 = Dm.contiguous(), v.contiguous(), l1.contiguous()
        P, M = Dm.shape[0], Dm.shape[-1]
        beta = torch.empty_like(l1)
        work = self._get("site_beta_work", (P * M * (1 + (M + 63) // 64),), torch.float64)
        with torch.cuda.device(self.device):
            B.check(self.lib.tsvgp_site_beta_f64(Dm.data_ptr(), v.data_ptr(), l1.data_ptr(), work.data_ptr(), beta.data_ptr(), M, P,
                                                 self._stream()), "tsvgp_site_beta")
        return beta

    def gemv(self, A: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
        """A v per latent (``tsvgp_gemv_f64``): A [M, M] (one matrix for every latent) or [P, M, M] row-major, v [M, P] -> [M, P]."""
        A, v = A.contiguous(), v.contiguous()
        M, P = v.shape
        y = torch.empty_like(v)
        with torch.cuda.device(self.device):
            B.check(self.lib.tsvgp_gemv_f64(A.data_ptr(), 0 if A.dim() == 2 else M * M, v.data_ptr(), y.data_ptr(), M, P, self._stream()),
                    "tsvgp_gemv")
        return y

    def keeper_begin(self):
        """Starts the clock keeper (``tsvgp_keeper_run``) on its own side stream beside whatever the current stream runs next and
        returns a ticket for ``keeper_end``; None when it is switched off.  The M x M sections of a step keep a few workgroups
        busy at a time; the chip's clock sags over them and the N-sized kernels behind pay for the climb back (moments at
        125 000 x 1024: 2.22 ms behind 1.5 ms of near-idle, 1.95 ms back-to-back or behind the keeper -- profiles/r05_clock_lab.txt).
        Inside a stream capture the side stream joins the capture (fork here, join in ``keeper_end``)."""
        if self.clock_keeper == 0:
            return None
        dev = self.device
        main = torch.cuda.current_stream(dev)
        if self._keeper_side is None:
            self._keeper_side = torch.cuda.Stream(dev)
            self._keeper_flag = torch.zeros(16, dtype=torch.int32, device=dev)
        side, flag = self._keeper_side, self._keeper_flag
        with torch.cuda.device(dev):
            B.check(self.lib.tsvgp_keeper_signal(flag.data_ptr(), 0, main.cuda_stream), "tsvgp_keeper_signal")
            side.wait_stream(main)
            B.check(self.lib.tsvgp_keeper_run(flag.data_ptr(), self.keeper_max_us, max(self.clock_keeper, 0), side.cuda_stream),
                    "tsvgp_keeper_run")
            done = torch.cuda.Event()
            done.record(side)
        return done

    def keeper_end(self, ticket):
        """Raises the keeper's flag in the order of the current stream and makes that stream wait for the keeper's waves to leave
        (microseconds): call it in front of the N-sized launch the keeper was bridging to."""
        if ticket is None:
            return
        main = torch.cuda.current_stream(self.device)
        with torch.cuda.device(self.device):
            B.check(self.lib.tsvgp_keeper_signal(self._keeper_flag.data_ptr(), 1, main.cuda_stream), "tsvgp_keeper_signal")
        main.wait_event(ticket)

    def step_status(self, infos_a, infos_b, nonpos):
        """[3] fp64 device tensor (sum |info| of the prelude factorisations, nonpos, sum |info| of the final one)."""
        cat = lambda ts: None if len(ts) == 0 else (ts[0] if len(ts) == 1 else torch.cat(list(ts))).contiguous()
        a, b = cat(infos_a), cat(infos_b)
        nonpos = nonpos.reshape(1).to(torch.float64)
        flags = torch.empty(3, dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            B.check(self.lib.tsvgp_step_status_f64(_ptr(a), 0 if a is None else a.numel(), _ptr(b), 0 if b is None else b.numel(),
                                                   nonpos.data_ptr(), flags.data_ptr(), self._stream()), "tsvgp_step_status")
        return flags

    def sym_pack(self, acc2: torch.Tensor, out: torch.Tensor):
        """Lower triangles of acc2 [P, M, M] (any row / matrix stride) -> out [P * M (M + 1) / 2] (``tsvgp_sym_pack_f64``)."""
        P, M = acc2.shape[0], acc2.shape[-1]
        assert acc2.stride(2) == 1
        with torch.cuda.device(self.device):
            B.check(self.lib.tsvgp_sym_pack_f64(acc2.data_ptr(), acc2.stride(1), acc2.stride(0), M, P, out.data_ptr(),
                                                self._stream()), "tsvgp_sym_pack")
        return out

    def sym_unpack(self, packed: torch.Tensor, P: int, M: int) -> torch.Tensor:
        out = torch.empty((P, M, M), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            B.check(self.lib.tsvgp_sym_unpack_f64(packed.data_ptr(), out.data_ptr(), M, M * M, M, P, self._stream()),
                    "tsvgp_sym_unpack")
        return out

    def trmm(self, A: torch.Tensor, Tm: torch.Tensor, C: torch.Tensor, mode: int):
        Np, Mp = A.shape
        with torch.cuda.device(self.device):
            self._launch("tsvgp_trmm", lambda: self._fn("tsvgp_trmm")(A.data_ptr(), Tm.data_ptr(), C.data_ptr(), Np, Mp,
                                                                      mode, self._stream()))
        return C

    def kernel_grad(self, X, Z, kernel, U, g0, g1, beta):
        """N-sized part of d ELBO / d (variance, lengthscales, Z) for ONE latent GP (``tsvgp_kernel_grad_*``):
        sum_{n,m} V[n,m] dK[n,m]/d theta with V = g0 beta^T - 2 g1 * U.  X [N, D], Z [M, D], U [Np, Mp] (compute dtype);
        g0, g1: columns of the [Np, P] gradient buffers (any element stride), beta [M] fp64.
        Returns (d_variance scalar, d_lengthscales [D], d_Z [M, D]) in fp64."""
        T, dev = self.dtype, self.device
        X = X.to(device=dev, dtype=T).contiguous()
        Z = Z.to(device=dev, dtype=T).contiguous()
        N, D = X.shape
        M = Z.shape[0]
        if D > MAX_INPUT_DIM_GRAD:
            return self._kernel_grad_gemm(X, Z, kernel, U, g0, g1, beta)
        Mp = B.round_up(M)
        inv_ls = kernel.inv_lengthscales(D, T, dev)
        rows, Dp = int(self.lib.tsvgp_kernel_grad_rows()), int(self.lib.tsvgp_kernel_grad_dpad(D))
        nrb, ncb = (N + rows - 1) // rows, (Mp + 511) // 512
        zpart = torch.empty((nrb, Mp, Dp), dtype=torch.float64, device=dev)
        lpart = torch.empty((nrb, ncb, Dp), dtype=torch.float64, device=dev)
        vpart = torch.empty((nrb, ncb), dtype=torch.float64, device=dev)
        bt = beta.to(device=dev, dtype=T).contiguous()
        assert g0.dim() == 1 and g1.dim() == 1 and g0.stride(0) == g1.stride(0)
        with torch.cuda.device(dev):
            self._launch("tsvgp_kernel_grad", lambda: self._fn("tsvgp_kernel_grad")(
                kernel.kind, X.data_ptr(), Z.data_ptr(), inv_ls.data_ptr(), kernel.variance.item(), U.data_ptr(), U.shape[1],
                g0.data_ptr(), g1.data_ptr(), g0.stride(0), bt.data_ptr(), 1, N, M, D, zpart.data_ptr(), lpart.data_ptr(),
                vpart.data_ptr(), self._stream()))
        return vpart.sum(), lpart.sum(dim=(0, 1))[:D], zpart.sum(dim=0)[:M, :D]

    def _kernel_grad_gemm(self, X, Z, kernel, U, g0, g1, beta):
        """``kernel_grad`` for D beyond the fused kernel's sizes, in the GEMM form of the large-D fill (reference
        docs/notebooks/mnist.py:117-192, D = 784): G = x~ z~^T from the BLAS library, ``tsvgp_gram_to_gradw_*`` turns it into
        W = -2 variance V * k'(r2) in place and sums V k(r2); what is left is one [Mp, N] x [N, D] GEMM and the row / column
        sums of W.  X, Z already in the compute dtype on the device."""
        T, dev = self.dtype, self.device
        N, D = X.shape
        M = Z.shape[0]
        Np, Mp = B.round_up(N), B.round_up(M)
        inv_ls = kernel.inv_lengthscales(D, T, dev)
        Xs = X * inv_ls
        Zs = torch.zeros((Mp, D), dtype=T, device=dev)
        Zs[:M] = Z * inv_ls
        W = self._get("gradW", (Np, Mp), T)
        torch.mm(Xs, Zs.t(), out=W[:N])
        xx = (Xs * Xs).sum(dim=1).contiguous()
        zz = (Zs[:M] * Zs[:M]).sum(dim=1).contiguous()
        vpart = torch.empty(int(self.lib.tsvgp_gram_to_gradw_parts(N, M)), dtype=torch.float64, device=dev)
        bt = beta.to(device=dev, dtype=T).contiguous()
        assert g0.dim() == 1 and g1.dim() == 1 and g0.stride(0) == g1.stride(0)
        with torch.cuda.device(dev):
            self._launch("tsvgp_kernel_grad", lambda: self._fn("tsvgp_gram_to_gradw")(
                kernel.kind, W.data_ptr(), xx.data_ptr(), zz.data_ptr(), kernel.variance.item(), U.data_ptr(), U.shape[1],
                g0.data_ptr(), g1.data_ptr(), g0.stride(0), bt.data_ptr(), 1, N, M, Mp, vpart.data_ptr(), self._stream()))
        f64 = torch.float64
        colsum = W.sum(dim=0, dtype=f64)[:M]  # [M]
        rowsum = W[:N].sum(dim=1, dtype=f64)  # [N]
        WtX = torch.mm(W[:N, :M].t(), Xs).to(f64)  # [M, D]
        Zs64, Xs64, il = Zs[:M].to(f64), Xs.to(f64), inv_ls.to(f64)
        dZ = (WtX - Zs64 * colsum[:, None]) * il
        dl = (torch.einsum("nd,n->d", Xs64 * Xs64, rowsum) - 2.0 * (Zs64 * WtX).sum(dim=0)
              + torch.einsum("md,m->d", Zs64 * Zs64, colsum)) * il
        return vpart.sum(), dl, dZ

    def selftest_mfma(self, dtype=None):
        """Runs one MFMA and returns (a, b, c) for a host-side check of the fragment maps."""
        dtype = dtype or self.dtype
        g = torch.Generator(device="cpu").manual_seed(7)
        a = torch.randn(16, 4, generator=g, dtype=torch.float64).to(dtype).to(self.device)
        b = torch.randn(4, 16, generator=g, dtype=torch.float64).to(dtype).to(self.device)
        c = torch.zeros(16, 16, dtype=dtype, device=self.device)
        with torch.cuda.device(self.device):
            B.check(self._fn("tsvgp_selftest_mfma", dtype)(a.data_ptr(), b.data_ptr(), c.data_ptr(), self._stream()),
                    "tsvgp_selftest_mfma")
        return a, b, c

    # ------------------------------------------------------------------ one kernel per latent, batched over the latents
    batch_separate = True  # False: always one pass per latent through a single K(X, Z) buffer (the round-1 path)
    batch_mem_fraction = 0.8  # of the device memory still free: what the [P, Np, Mp] operand of a batched pass may take

    @staticmethod
    def _per_latent(t, p):
        """Entry p of a per-latent operand given as None, a list, a [P, M, M] tensor or one shared [M, M] tensor."""
        if t is None:
            return None
        if isinstance(t, (list, tuple)):
            return t[p]
        return t[p] if t.dim() == 3 else t

    def _batch_plan(self, N, M, kernel, P, whiten_T, whiten_mode, project_T, moments_on_kfu, D=0):
        """Can the pass over P separately-parameterised latents run as batched launches (tsvgp_*_batched_*)?  Needs one
        kernel family, at most MAX_BATCH latents, no "projected" latent (its second product is not in place), the
        whitening in upper form (in place) and room for the [P, Np, Mp] operand.  Returns the list of whitened latents
        or None for the per-latent path."""
        if not self.batch_separate or P < 2 or P > B.MAX_BATCH or moments_on_kfu:
            return None
        if len({k.kind for k in kernel.kernels}) != 1:
            return None
        if D > MAX_INPUT_DIM:  # the GEMM-based fill of large input dimensions is per latent
            return None
        if any(self._per_latent(project_T, p) is not None for p in range(P)):
            return None
        whitened = [p for p in range(P) if self._per_latent(whiten_T, p) is not None]
        if whitened and whiten_mode != B.TRI_UPPER:
            return None
        Np, Mp = B.round_up(N), B.round_up(M)
        need = P * Np * Mp * torch.empty((), dtype=self.dtype).element_size()
        have = self._buf.get("KfuP")
        if have is None or tuple(have.shape) != (P, Np, Mp) or have.dtype != self.dtype:
            free, _ = torch.cuda.mem_get_info(self.device)
            free += torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)
            if have is not None:
                free += have.numel() * have.element_size()
            if need > self.batch_mem_fraction * free:
                return None
        return whitened

    def _fill_batched(self, X, Z, kernel, KfuP):
        """K(X, Z) of every latent's kernel into KfuP [P, Np, Mp] with one launch per MAX_BATCH latents."""
        T, dev = self.dtype, self.device
        N, D = X.shape
        M = Z.shape[0]
        if D > MAX_INPUT_DIM:
            raise ValueError(f"input dimension D = {D} exceeds the HIP fill kernel's limit of {MAX_INPUT_DIM}")
        P = len(kernel.kernels)
        inv_ls = torch.stack([k.inv_lengthscales(D, T, dev) for k in kernel.kernels]).contiguous()  # [P, D]
        ctype = ctypes.c_double if T == torch.float64 else ctypes.c_float
        fn = self._fn("tsvgp_kernel_fill_batched", T)
        stride = KfuP.shape[1] * KfuP.shape[2]
        with torch.cuda.device(dev):
            for p0 in range(0, P, B.MAX_BATCH):
                n = min(B.MAX_BATCH, P - p0)
                var = (ctype * n)(*[k.variance.item() for k in kernel.kernels[p0:p0 + n]])
                self._launch("tsvgp_se_fill", lambda: fn(kernel.kernels[0].kind, X.data_ptr(), Z.data_ptr(),
                                                         inv_ls[p0].data_ptr(), var, KfuP[p0].data_ptr(), stride, N, M, D,
                                                         KfuP.shape[2], n, self._stream()))
        return KfuP

    def _run_batched(self, X, Y, Z, kernel, whitened, *, moment_Tm, moment_mode, gamma, lik_id, lik_param, whiten_T,
                     sites, want_moments, want_grads, mean_only, prefill=None) -> EStepStats:
        """One pass for P latents with one kernel each as batched launches: fill [P, Np, Mp] -> in-place whitening of the
        latents that ask for it -> moments -> site sums (4 launches; BASELINE configs[4])."""
        T, dev = self.dtype, self.device
        X = X.to(device=dev, dtype=T).contiguous()
        Z = Z.to(device=dev, dtype=T).contiguous()
        N, M, P = X.shape[0], Z.shape[0], moment_Tm.shape[0]
        Np, Mp = B.round_up(N), B.round_up(M)
        need_g = lik_id != B.LIK_NONE
        if need_g:
            Y = Y.to(device=dev, dtype=T).contiguous()
        self._b_tag = None
        self._buf.pop("Kfu", None)  # the single-latent operand buffers are not needed beside the batched one
        self._buf.pop("B", None)
        KfuP = self._get("KfuP", (P, Np, Mp), T)
        stride = Np * Mp
        if prefill is not None:
            if prefill.get("KfuP") is not KfuP:
                raise RuntimeError("prefill ticket does not belong to this pass")
            torch.cuda.current_stream(dev).wait_event(prefill["event"])
        else:
            if self._side is not None and not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream(dev).wait_stream(self._side)
            self._fill_batched(X, Z, kernel, KfuP)
        # whitening in place, one launch per run of consecutive whitened latents: B_p = K_p U9_p^-T (upper form)
        runs, start = [], None
        for p in range(P + 1):
            on = p < P and p in whitened
            if on and start is None:
                start = p
            if not on and start is not None:
                runs.append((start, p))
                start = None
        fn_trmm = self._fn("tsvgp_trmm_batched")
        self.last_trmm_batch = max([hi - lo for lo, hi in runs], default=1)
        for lo, hi in runs:
            Wt = torch.stack([self._per_latent(whiten_T, p) for p in range(lo, hi)])
            Wp = self._pad_square(Wt, Mp, f"pad_LinvP{hi - lo}")
            with torch.cuda.device(dev):
                self._launch("tsvgp_trmm", lambda: fn_trmm(KfuP[lo].data_ptr(), stride, Wp.data_ptr(), Mp * Mp,
                                                           KfuP[lo].data_ptr(), stride, Np, Mp, B.TRI_UPPER, hi - lo,
                                                           self._stream()))
        Tm = self._pad_square(moment_Tm, Mp, "pad_Tm")
        gam = self._padded_gamma(gamma, Mp, P)
        nblk = Np // B.TILE
        ve_partial = self._get("ve_partial", (nblk,), torch.float64)
        nonpos_partial = self._get("nonpos_partial", (nblk,), torch.int32)
        g0 = self._get("g0", (Np, P), T) if need_g else None
        g1 = self._get("g1", (Np, P), T) if need_g else None
        mean = torch.empty((N, P), dtype=T, device=dev) if want_moments else None
        var = torch.empty((N, P), dtype=T, device=dev) if (want_moments and not mean_only) else None
        lik_flags = (lik_id | B.LIK_MEANONLY) if mean_only else lik_id
        kdiag = (ctypes.c_double * P)(*[k.variance.item() for k in kernel.kernels])
        with torch.cuda.device(dev):
            self._launch("tsvgp_moments", lambda: self._fn("tsvgp_moments_batched")(
                KfuP.data_ptr(), stride, Tm.data_ptr(), gam.data_ptr(), _ptr(Y) if need_g else None, kdiag, lik_flags,
                float(lik_param), _ptr(mean), _ptr(var), _ptr(g0), _ptr(g1), ve_partial.data_ptr(),
                nonpos_partial.data_ptr(), N, Np, Mp, P, moment_mode, self._stream()))
        stats = EStepStats(n_rows=N, ve_sum=ve_partial.sum(), nonpos=nonpos_partial.sum().to(torch.float64))
        if want_moments:
            stats.mean, stats.var = mean.to(torch.float64), (None if var is None else var.to(torch.float64))
        if want_grads and need_g:
            stats.g0, stats.g1 = g0[:N].to(torch.float64), g1[:N].to(torch.float64)
        if sites:
            nsplit = self.nsplit_override or self.choose_nsplit(Mp, P, Np)
            nsplit = max(1, min(nsplit, Np // site_sum_chunk_rows(T == torch.float64, P)))
            nbytes = int(self._fn("tsvgp_site_accum_work_bytes")(Mp, P, nsplit))
            work = self._get("work", (nbytes,), torch.uint8)
            acc2 = torch.empty((P, Mp, Mp), dtype=torch.float64, device=dev)
            acc1 = torch.empty((P, Mp), dtype=torch.float64, device=dev)
            with torch.cuda.device(dev):
                self._launch("tsvgp_site_accum", lambda: self._fn("tsvgp_site_accum_batched")(
                    KfuP.data_ptr(), stride, g0.data_ptr(), g1.data_ptr(), acc2.data_ptr(), acc1.data_ptr(),
                    work.data_ptr(), Np, Mp, P, nsplit, self._stream()))
            stats.acc2 = acc2[:, :M, :M]
            stats.acc1 = acc1[:, :M]
        self.last_batched = True
        return stats

    def _run_separate(self, X, Y, Z, kernel, *, moment_Tm, gamma, whiten_T, project_T, want_grads, prefill=None,
                      **kw) -> EStepStats:
        """Separate per-latent kernels (K_uu [P, M, M]).  Batched launches over the latents when ``_batch_plan`` allows
        (one [P, Np, Mp] operand: 8.2 GB per latent at N = 1e6, M = 1024, fp64 -- 66 GB of the 288 GB at P = 8);
        otherwise one fill + moments + accumulation pass per latent through the single-latent kernels and ONE K(X, Z)
        buffer."""
        P = moment_Tm.shape[0]
        if len(kernel.kernels) != P:
            raise ValueError(f"{len(kernel.kernels)} kernels for {P} latent GPs")
        if Y is not None and (Y.dim() != 2 or Y.shape[1] != P):
            raise ValueError(f"Y must be [N, P] = [{X.shape[0]}, {P}], got {tuple(Y.shape)}")
        if X.shape[0] > 0:
            if prefill is not None and "KfuP" in prefill:  # the batched fill is already under way (start_fill)
                whitened = [p for p in range(P) if self._per_latent(whiten_T, p) is not None]
            else:
                whitened = self._batch_plan(X.shape[0], Z.shape[0], kernel, P, whiten_T,
                                            kw.get("whiten_mode", B.TRI_UPPER), project_T, kw.get("moments_on_kfu", False),
                                            D=X.shape[1])
            if whitened is not None:
                return self._run_batched(X, Y, Z, kernel, whitened, moment_Tm=moment_Tm, moment_mode=kw["moment_mode"],
                                         gamma=gamma, lik_id=kw.get("lik_id", B.LIK_NONE), lik_param=kw.get("lik_param", 0.0),
                                         whiten_T=whiten_T, sites=kw.get("sites", False),
                                         want_moments=kw.get("want_moments", False), want_grads=want_grads,
                                         mean_only=kw.get("mean_only", False), prefill=prefill)
        self.last_batched = False
        parts = []
        for p, kp in enumerate(kernel.kernels):
            if isinstance(whiten_T, (list, tuple)):  # per-latent routes: None = this latent works on K_fu directly
                wt = whiten_T[p]
            else:
                wt = None if whiten_T is None else (whiten_T[p] if whiten_T.dim() == 3 else whiten_T)
            pt = project_T[p] if isinstance(project_T, (list, tuple)) else (
                None if project_T is None else (project_T[p] if project_T.dim() == 3 else project_T))
            st = self.run(X, None if Y is None else Y[:, p:p + 1], Z, kp, moment_Tm=moment_Tm[p:p + 1],
                          gamma=gamma[:, p:p + 1], whiten_T=wt, project_T=pt, want_grads=want_grads, **kw)
            if want_grads and st.g0 is not None:
                st.g0, st.g1 = st.g0.clone(), st.g1.clone()  # views of a buffer the next latent overwrites
            parts.append(st)

        def cat(name, dim):
            vals = [getattr(s, name) for s in parts]
            return None if vals[0] is None else torch.cat(vals, dim=dim)

        out = EStepStats(n_rows=parts[0].n_rows, ve_sum=sum(s.ve_sum for s in parts), nonpos=sum(s.nonpos for s in parts))
        out.mean, out.var, out.g0, out.g1 = cat("mean", 1), cat("var", 1), cat("g0", 1), cat("g1", 1)
        out.acc2, out.acc1 = cat("acc2", 0), cat("acc1", 0)
        return out

    # ------------------------------------------------------------------ K(X, Z) fill beside the M x M prelude
    def start_fill(self, X, Z, kernel, b_tag=None, want="Kfu", routes=None):
        """Starts the K(X, Z) fill of the next ``run`` on a side stream and returns a ticket for ``run(prefill=...)``.
        The fill needs only X, Z and the kernel parameters, so it can run beside the latency-bound M x M prelude
        (factorisations of one workgroup each, small GEMMs) that the main stream executes between this call and
        ``run``: 3.26 -> 2.6 ms for the pair at N = 1e6, M = 1024 (tools/exp_overlap2.py).  Returns None when there is
        nothing to overlap: separate kernels on the per-latent path (one fill per latent), an operand ``run`` would reuse
        (warm E-steps), a stream capture in progress.  ``routes``: the projection route of every latent (separate kernels:
        the batched pass, whose fill this starts, needs all of them direct or whitened)."""
        if self.device.type != "cuda" or X.shape[0] == 0:
            return None
        # Under stream capture (hipGraph) the side stream JOINS the capture: it waits for an event recorded on the capturing
        # stream (a fork) and the consumer waits for its event (the join) inside the same capture, so a replayed step runs the
        # fill beside the prelude exactly as an eager one.  record_stream means nothing to a graph's private pool, so there the
        # ticket itself keeps every tensor the side stream reads alive until the join (`keep`, below).
        capturing = torch.cuda.is_current_stream_capturing()
        T, dev = self.dtype, self.device
        N, D = X.shape
        M = Z.shape[0]
        Np, Mp = B.round_up(N), B.round_up(M)
        if isinstance(kernel, SeparateIndependent):
            P = len(kernel.kernels)
            if routes is None or any(r == "projected" for r in routes):
                return None
            if self._batch_plan(N, M, kernel, P, None, B.TRI_UPPER, None, False, D=D) is None:
                return None
            main = torch.cuda.current_stream(dev)
            if self._side is None:
                self._side = torch.cuda.Stream(dev)
            side = self._side
            Xc = X.to(device=dev, dtype=T).contiguous()
            Zc = Z.to(device=dev, dtype=T).contiguous()
            self._b_tag = None
            self._buf.pop("Kfu", None)
            self._buf.pop("B", None)
            KfuP = self._get("KfuP", (P, Np, Mp), T)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._fill_batched(Xc, Zc, kernel, KfuP)
                done = torch.cuda.Event()
                done.record(side)
            if not capturing:
                for t in (Xc, Zc, KfuP):
                    t.record_stream(side)
            # `keep`: see the single-kernel ticket below
            return dict(event=done, KfuP=KfuP, keep=(Xc, Zc))
        if (b_tag is not None and self._b_tag == (want, b_tag) and self._buf.get(want) is not None
                and tuple(self._buf[want].shape) == (Np, Mp)):
            return None
        main = torch.cuda.current_stream(dev)
        if self._side is None:
            self._side = torch.cuda.Stream(dev)
        side = self._side
        Xc = X.to(device=dev, dtype=T).contiguous()
        Zc = Z.to(device=dev, dtype=T).contiguous()
        inv_ls = kernel.inv_lengthscales(D, T, dev)
        variance = kernel.variance.item()
        self._b_tag = None
        Kfu = self._get("Kfu", (Np, Mp), T)
        side.wait_stream(main)  # the buffer's last readers (the previous pass) and the conversions above
        with torch.cuda.stream(side):
            self.se_fill(Xc, Zc, inv_ls, variance, Kfu, kernel.kind)
            done = torch.cuda.Event()
            done.record(side)
        if not capturing:
            for t in (Xc, Zc, inv_ls, Kfu):  # blocks of the main stream's allocator pool that the side stream touches
                t.record_stream(side)
        # `keep`: the converted inputs must outlive the side stream's read of them.  Eagerly record_stream sees to that; under
        # capture nothing does -- a block freed DURING a capture goes straight back to the graph's pool, the next allocation on
        # the capturing stream (a prelude temporary) takes it, and in the replayed graph that kernel runs BESIDE the forked fill
        # still reading it.  Found in round 4 by the two-rank fp32 test of BASELINE configs[3] (fp32: Z [M, D] is converted into
        # such a temporary; fp64 passes its tensors through, which is why three rounds of fp64 tests never saw it): the state
        # after four replayed steps was off by 1e-5 ... 0.4, differently on every run.  The ticket holds them until ``run`` has
        # made the consuming stream wait for the fill.
        return dict(event=done, key=(X.data_ptr(), tuple(X.shape), Z.data_ptr(), tuple(Z.shape), id(kernel)), Kfu=Kfu,
                    keep=(Xc, Zc, inv_ls))

    # ------------------------------------------------------------------ one N-pass
    def run(self, X, Y, Z, kernel, *, moment_Tm, moment_mode, gamma, lik_id=B.LIK_NONE, lik_param=0.0,
            whiten_T=None, whiten_mode=B.TRI_UPPER, project_T=None, sites=False, want_moments=False, want_grads=False,
            b_tag=None, mean_only=False, prefill=None, moments_on_kfu=False, project_mode=B.TRI_LOWER,
            keep_tile=False) -> EStepStats:
        """One pass over the shard's rows.

        keep_tile (one latent, no whitening): the triangular product of the moments, t_n = Tm k_n, is STORED (``tsvgp_trmm``
        into the buffer "Tt", returned as ``stats.tile``) and the moments are assembled from it -- mean by a matrix-vector
        product, var = kdiag - |t_n|^2 by a row norm, the gradients by ``tsvgp_lik_map`` -- instead of being squared and summed
        inside the fused kernel.  For the M-step (t_SVGP.elbo_and_grads), which needs Q k_n = Tm^T t_n for every row: a second
        triangular product of the stored tile instead of a dense N M^2 GEMM with Q = Tm^T Tm.

        X [N, D], Y [N, P] (or None when lik_id == NONE), Z [M, D];
        whiten_T [M, M] fp64: the inverted triangular factor of Kuu + jitter I (B[n, i] = sum_j Kfu[n, j] whiten_T[i, j]
        over the triangle ``whiten_mode`` names), or None (moments then act on Kfu directly);
        project_T [M, M] fp64 (needs whiten_T): the "projected" route -- after the moments the site sums are taken over
        a = project_T-product of B (a[n, i] = sum_{j <= i} B[n, j] project_T[i, j], i.e. a = U9^-T b = K9^-1 k with
        project_T = U9^-T), written over the K(X, Z) buffer;
        moment_Tm [P, M, M] fp64 and gamma [M, P] fp64: operands of the fused moments kernel;
        sites=True also accumulates (acc2, acc1) = (sum g1 a a^T, sum g0 a) over the rows a of the same operand the
        moments used: the whitened B when whiten_T is given, Kfu itself otherwise (the "direct" projection).
        b_tag: a hashable description of (X, Z, kernel parameters, jitter).  When it equals the tag of the B buffer left
        by the previous call, the fill and the whitening are skipped and B is reused ("warm" E-step: consecutive
        E-steps with unchanged hyperparameters, as in the reference's E/M loop, experiments/uci_regression.py:152-153).
        moments_on_kfu (with whiten_T): the moments act on K(X, Z) itself (moment_Tm / gamma in k coordinates) and only the
        site sums use the whitened / projected operand, whose products then run AFTER the moments: the variant in
        which nothing on the moments side depends on the factor of K_uu + jitter I.  project_mode: triangle of project_T.
        prefill: the ticket of ``start_fill`` for the same (X, Z, kernel): the fill is already under way on the side
        stream; this call waits for it instead of filling.
        mean_only (likelihood NONE or GAUSSIAN): skip the variance product of the moments (TSVGP_LIK_MEANONLY) -- the
        Gaussian g0, g1 do not depend on it; ``var`` is then None and ``ve_sum`` NaN.
        """
        if isinstance(kernel, SeparateIndependent):
            return self._run_separate(X, Y, Z, kernel, moment_Tm=moment_Tm, moment_mode=moment_mode, gamma=gamma,
                                      lik_id=lik_id, lik_param=lik_param, whiten_T=whiten_T, whiten_mode=whiten_mode,
                                      project_T=project_T, sites=sites, want_moments=want_moments, want_grads=want_grads,
                                      mean_only=mean_only, moments_on_kfu=moments_on_kfu, project_mode=project_mode,
                                      prefill=prefill)
        if mean_only and (lik_id & 0xFF) not in (B.LIK_NONE, B.LIK_GAUSSIAN):
            raise ValueError("mean_only needs a likelihood whose gradients do not depend on the predictive variance")
        T, dev = self.dtype, self.device
        X = X.to(device=dev, dtype=T).contiguous()
        Z = Z.to(device=dev, dtype=T).contiguous()
        N, D = X.shape
        M = Z.shape[0]
        P = moment_Tm.shape[0]
        if N == 0:
            # One rank of several may hold no rows (N < world size): it contributes zeros to the all-reduce instead of
            # raising alone while the other ranks wait in the collective.  A single process with no data is an error.
            if world_size() == 1:
                raise ValueError("empty data: natgrad_step / elbo need at least one row")
            zero = torch.zeros((), dtype=torch.float64, device=dev)
            st = EStepStats(n_rows=0, ve_sum=zero, nonpos=zero.clone())
            if sites:
                st.acc2 = torch.zeros((P, M, M), dtype=torch.float64, device=dev)
                st.acc1 = torch.zeros((P, M), dtype=torch.float64, device=dev)
            if want_moments:
                st.mean = torch.zeros((0, P), dtype=torch.float64, device=dev)
                st.var = None if mean_only else torch.zeros((0, P), dtype=torch.float64, device=dev)
            return st
        if X.dim() != 2 or Z.dim() != 2 or Z.shape[1] != D:
            raise ValueError(f"X must be [N, D] and Z [M, D] with equal D, got {tuple(X.shape)} and {tuple(Z.shape)}")
        if lik_id != B.LIK_NONE:
            Y = Y.to(device=dev, dtype=T).contiguous()
            if Y.dim() != 2 or Y.shape[0] != N or Y.shape[1] != P:
                raise ValueError(f"Y must be [N, P] = [{N}, {P}], got {tuple(Y.shape)}")
        Np, Mp = B.round_up(N), B.round_up(M)
        inv_ls = kernel.inv_lengthscales(D, T, dev)
        variance = kernel.variance.item()

        # The N x M operand of the moments / site kernels: the whitened B = Kfu U^-T, or Kfu itself ("direct" route).
        # A tagged operand left by the previous call is reused when the tag matches (warm E-steps).
        want = "B" if whiten_T is not None else "Kfu"
        late_whiten = moments_on_kfu and whiten_T is not None
        if late_whiten:
            b_tag = None  # K(X, Z) and B are both consumed in this pass: nothing to carry over
        reuse = (b_tag is not None and self._b_tag == (want, b_tag) and self._buf.get(want) is not None
                 and tuple(self._buf[want].shape) == (Np, Mp))
        if reuse:
            A = self._buf[want]
        else:
            self._b_tag = None
            Kfu = self._get("Kfu", (Np, Mp), T)
            if prefill is not None:
                if prefill["Kfu"] is not Kfu:
                    raise RuntimeError("prefill ticket does not belong to this pass")
                torch.cuda.current_stream(dev).wait_event(prefill["event"])
            else:
                # a fill started for a pass that never ran must not land on top of this one (a capture begins synchronised)
                if self._side is not None and not torch.cuda.is_current_stream_capturing():
                    torch.cuda.current_stream(dev).wait_stream(self._side)
                self.se_fill(X, Z, inv_ls, variance, Kfu, kernel.kind)
            A = Kfu
            if whiten_T is not None and not late_whiten:
                Bw = self._get("B", (Np, Mp), T)
                self.trmm(Kfu, self._pad_square(whiten_T, Mp, "pad_Linv"), Bw, whiten_mode)
                A = Bw
            if b_tag is not None:
                self._b_tag = (want, b_tag)

        Tm = self._pad_square(moment_Tm, Mp, "pad_Tm")
        gam = self._padded_gamma(gamma, Mp, P)
        nblk = Np // B.TILE
        ve_partial = self._get("ve_partial", (nblk,), torch.float64)
        nonpos_partial = self._get("nonpos_partial", (nblk,), torch.int32)
        need_g = lik_id != B.LIK_NONE
        g0 = self._get("g0", (Np, P), T) if need_g else None
        g1 = self._get("g1", (Np, P), T) if need_g else None
        mean = torch.empty((N, P), dtype=T, device=dev) if want_moments else None
        var = torch.empty((N, P), dtype=T, device=dev) if (want_moments and not mean_only) else None
        lik_flags = (lik_id | B.LIK_MEANONLY) if mean_only else lik_id
        tile = None
        if keep_tile:
            if P != 1 or mean_only or whiten_T is not None or not need_g:
                raise ValueError("keep_tile: one latent, a likelihood, no whitening")
            tile = self._get("Tt", (Np, Mp), T)
            self.trmm(A, Tm[0], tile, moment_mode)
            mean = torch.mv(A[:N], gam[:, 0]).reshape(N, 1)  # gam: [Mp, 1], rows >= M zero
            var = (variance - torch.linalg.vector_norm(tile[:N], dim=1).square()).reshape(N, 1)
            with torch.cuda.device(dev):
                self._launch("tsvgp_lik_map", lambda: self._fn("tsvgp_lik_map")(
                    mean.data_ptr(), var.data_ptr(), Y.data_ptr(), lik_id, float(lik_param), g0.data_ptr(), g1.data_ptr(),
                    ve_partial.data_ptr(), nonpos_partial.data_ptr(), N, Np, 1, self._stream()))
        else:
            with torch.cuda.device(dev):
                self._launch("tsvgp_moments", lambda: self._fn("tsvgp_moments")(
                    A.data_ptr(), Tm.data_ptr(), gam.data_ptr(), _ptr(Y) if need_g else None, variance, lik_flags,
                    float(lik_param), _ptr(mean), _ptr(var), _ptr(g0), _ptr(g1), ve_partial.data_ptr(),
                    nonpos_partial.data_ptr(), N, Np, Mp, P, moment_mode, self._stream()))
        stats = EStepStats(n_rows=N, ve_sum=ve_partial.sum(), nonpos=nonpos_partial.sum().to(torch.float64))
        stats.tile = tile
        if want_moments:
            stats.mean, stats.var = mean.to(torch.float64), (None if var is None else var.to(torch.float64))
        if want_grads and need_g:
            stats.g0, stats.g1 = g0[:N].to(torch.float64), g1[:N].to(torch.float64)

        if late_whiten and sites:  # the site sums' operand, now that the moments have read K(X, Z)
            Bw = self._get("B", (Np, Mp), T)
            self.trmm(A, self._pad_square(whiten_T, Mp, "pad_Linv"), Bw, whiten_mode)
            A = Bw
        if sites and project_T is not None:
            if whiten_T is None:
                raise ValueError("project_T needs whiten_T")
            Aproj = self._get("Kfu", (Np, Mp), T)  # K(X, Z) itself is no longer needed once B exists
            self.trmm(A, self._pad_square(project_T, Mp, "pad_proj"), Aproj, project_mode)
            A = Aproj
        if sites:
            stats.acc2, stats.acc1 = self._site_sums(A, g0, g1, P, M)
        return stats

    def _site_sums(self, A, g0, g1, P, M):
        """(sum_n g1 a a^T [P, M, M], sum_n g0 a [P, M]) over the rows of one operand A [Np, Mp] shared by the P latents
        (``tsvgp_site_accum_*``); g0, g1 [Np, P] with zero padding rows."""
        dev = self.device
        Np, Mp = A.shape
        nsplit = self.nsplit_override or self.choose_nsplit(Mp, P, Np)
        nsplit = max(1, min(nsplit, Np // site_sum_chunk_rows(self.dtype == torch.float64, P)))
        nbytes = int(self._fn("tsvgp_site_accum_work_bytes")(Mp, P, nsplit))
        work = self._get("work", (nbytes,), torch.uint8)
        acc2 = torch.empty((P, Mp, Mp), dtype=torch.float64, device=dev)
        acc1 = torch.empty((P, Mp), dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            self._launch("tsvgp_site_accum", lambda: self._fn("tsvgp_site_accum")(
                A.data_ptr(), g0.data_ptr(), g1.data_ptr(), acc2.data_ptr(), acc1.data_ptr(), work.data_ptr(), Np,
                Mp, P, nsplit, self._stream()))
        return acc2[:, :M, :M], acc1[:, :M]

    def run_two_product(self, X, Y, Z, kernel, *, whiten_T, moment_Tm, gamma, lik_id=B.LIK_NONE, lik_param=0.0, sites=False,
                        want_moments=False) -> EStepStats:
        """The pass of ``t_SVGP_white`` with the reference's OWN two-product variance (src/util.py:76-86):
            var = kff - |LA^-1 k|^2 + |LR^-1 k|^2 = kff - |b|^2 + |T2 b|^2,   b = U6^-1 k (whiten_T = U6^-1, upper form),
        T2 = U_R^-1 U6 (``moment_Tm``, upper triangular), mean = b^T gamma.  Used when Lambda_2 + 1e-9 I is not positive definite
        (the single triangular product of ``run`` needs its factor; the reference only needs K + Lambda_2 + 1e-9 I).  Sequence:
        fill -> trmm (B) -> moments with NO likelihood (mean, kff - |T2 b|^2) -> |b|^2 by a row reduction of B -> the
        likelihood map on the assembled moments (``tsvgp_lik_map_*``) -> site sums over B.  One shared kernel, one latent."""
        T, dev = self.dtype, self.device
        st = self.run(X, None, Z, kernel, moment_Tm=moment_Tm, moment_mode=B.TRI_UPPER, gamma=gamma, lik_id=B.LIK_NONE,
                      whiten_T=whiten_T, whiten_mode=B.TRI_UPPER, want_moments=True)
        N, M = X.shape[0], Z.shape[0]
        Np, Mp = B.round_up(N), B.round_up(M)
        Bw = self._buf["B"]  # the whitened operand of the pass just run
        kff = kernel.variance.item()
        bn = torch.linalg.vector_norm(Bw[:N], dim=1).to(torch.float64)
        var = (2.0 * kff - bn * bn)[:, None] - st.var  # st.var = kff - |T2 b|^2
        stats = EStepStats(n_rows=N, ve_sum=torch.zeros((), dtype=torch.float64, device=dev), nonpos=(~(var > 0)).sum().to(torch.float64))
        if want_moments:
            stats.mean, stats.var = st.mean, var
        if lik_id != B.LIK_NONE:
            Yc = Y.to(device=dev, dtype=T).contiguous()
            mean_t, var_t = st.mean.to(T).contiguous(), var.to(T).contiguous()
            nblk = Np // B.TILE
            ve_partial = self._get("ve_partial", (nblk,), torch.float64)
            nonpos_partial = self._get("nonpos_partial", (nblk,), torch.int32)
            g0, g1 = self._get("g0", (Np, 1), T), self._get("g1", (Np, 1), T)
            with torch.cuda.device(dev):
                B.check(self._fn("tsvgp_lik_map")(mean_t.data_ptr(), var_t.data_ptr(), Yc.data_ptr(), lik_id, float(lik_param),
                                                   g0.data_ptr(), g1.data_ptr(), ve_partial.data_ptr(), nonpos_partial.data_ptr(),
                                                   N, Np, 1, self._stream()), "tsvgp_lik_map")
            stats.ve_sum = ve_partial.sum()
            stats.nonpos = nonpos_partial.sum().to(torch.float64)
            if sites:
                stats.acc2, stats.acc1 = self._site_sums(Bw, g0, g1, 1, M)
        return stats
