/*
 * tsvgp_hip.h -- C ABI of the MI355X (gfx950) t-SVGP natural-gradient E-step kernels.
 *
 * The reference (AaltoML/t-SVGP) is pure Python on TensorFlow/GPflow and has NO FFI;
 * this header is the seam introduced *under* its Python API.
 * Every entry point replaces a group of TensorFlow ops on the hot path
 * `t_SVGP.natgrad_step` (reference src/models/tsvgp.py:234-304); the citation on each
 * declaration names the reference lines it stands in for.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types.
 *   - All pointers are DEVICE pointers (caller-allocated, caller-owned) unless the parameter name ends in `_host`
 *     (small arrays of per-latent scalars, read during the call and passed on as kernel arguments); `stream` is a
 *     hipStream_t passed as void* (NULL = default stream).  No allocation and no
 *     synchronisation inside: safe to capture into a hipGraph.  The only process-wide
 *     state is one "dynamic-LDS opt-in done" flag per (kernel, device) for the two
 *     kernels that use more than 64 KB of LDS (set on first use, thread-safe; any
 *     device of the process may be current).
 *   - Input dimension policy (one for the fill and for the M-step gradient): the fused kernels pad D to a compile-time
 *     size -- D <= 32 for the fill (tsvgp_kernel_fill_*), D <= 16 for the gradient contraction (tsvgp_kernel_grad_*);
 *     larger D returns 1 from them.  Beyond that the scaled distance is a GEMM, r2 = |x~|^2 + |z~|^2 - 2 x~ z~^T,
 *     taken from the BLAS library by the host, and an elementwise kernel finishes it in place:
 *     tsvgp_gram_to_kernel_* (K(X, Z)) and tsvgp_gram_to_gradw_* (the gradient's weight matrix W); any D.
 *   - Suffix _f64 / _f32 selects the arithmetic type T of the N-sized arrays.
 *   - "Padded" dimensions: Np = N rounded up to 128, Mp = M rounded up to 128.  Work
 *     buffers (Kfu, B) are [Np x Mp] row-major with the padding ZERO-filled by the
 *     producing kernel, so the MFMA kernels need no bounds checks.
 *   - Return value: 0 ok; 1 invalid argument; 2 launch failure (hipGetLastError != 0).
 */
#ifndef TSVGP_HIP_H
#define TSVGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSVGP_OK 0
#define TSVGP_EINVAL 1
#define TSVGP_ELAUNCH 2

#define TSVGP_TILE 128 /* padding granule of N and M */
#define TSVGP_MAX_BATCH 32 /* latent GPs per launch of the *_batched entry points (their per-latent scalars travel as
                              kernel arguments); callers with more latents issue several calls */

/* likelihood selectors for tsvgp_moments_* */
#define TSVGP_LIK_NONE 0      /* moments only (predict_f) */
#define TSVGP_LIK_GAUSSIAN 1  /* gpflow.likelihoods.Gaussian: closed form */
#define TSVGP_LIK_BERNOULLI 2 /* gpflow.likelihoods.Bernoulli, probit + 1e-3 jitter, 20-pt Gauss-Hermite */
#define TSVGP_LIK_NOCROP 0x100 /* OR-ed into the selector: leave g1 = d ve/d var uncropped (reference
                                  src/models/tsvgp_white.py:188-191 has no crop; src/models/tsvgp.py:262-263 has) */
#define TSVGP_LIK_MEANONLY 0x200 /* OR-ed into the selector (NONE or GAUSSIAN only): skip the variance product.  Under a
                                    Gaussian likelihood g0 = (y - mean)/s2 and g1 = -1/(2 s2) do not depend on the
                                    predictive variance, so the natural-gradient step (src/models/tsvgp.py:246-263)
                                    needs the mean alone: one HBM-bound sweep of A instead of the MFMA product.
                                    Tm and mode are ignored, var must be NULL, Mp*P*sizeof(T) <= 128 KB (gamma sits in LDS), ve_partial is written as NaN (the
                                    variational expectation itself does need the variance), nonpos_partial counts rows with a
                                    non-finite mean or gradient. */

/* k-range selectors of the panel product C[n,i] = sum_j A[n,j] * Tm[i,j] */
#define TSVGP_TRI_LOWER 0 /* j <= i  (forward substitution with the inverted factor)   */
#define TSVGP_TRI_UPPER 1 /* j >= i  (product with a transposed lower factor)         */
#define TSVGP_TRI_DENSE 2 /* all j                                                    */

/* Library / build identification: returns a static string "tsvgp_hip gfx950 <version>". */
const char *tsvgp_version(void);
/* Number of this header's calling conventions: bumped whenever an entry point's argument list changes without a new symbol
 * name.  A binding checks it against the TSVGP_ABI_VERSION it was written for before the first call (t-svgp_amd/_backend.py
 * refuses a library whose number differs: a shifted argument would otherwise hand a kernel a garbage stream or pointer). */
#define TSVGP_ABI_VERSION 4
int tsvgp_abi_version(void);

/* Upper bound on the number of workgroup slots the site-accumulation kernel can keep resident
 * (CUs x resident workgroups per CU for that kernel); used by the host to choose `nsplit`. */
int tsvgp_site_accum_slots_f64(void);
int tsvgp_site_accum_slots_f32(void);

/* (1) Kernel-matrix fill: squared-exponential K(X, Z) -> K[n, m] = variance * exp(-0.5 * sum_d ((x_nd - z_md) * inv_ls_d)^2)
 *     Replaces gpflow.covariances.Kuf / Kuu (reference src/models/tsvgp.py:209-211, 222-225, 268-269).
 *     X [N x D] row-major, Z [M x D] row-major, inv_ls [D] (1/lengthscale per dimension),
 *     K [rows_alloc x ldk] row-major with rows_alloc >= round_up(N,128) and ldk >= round_up(M,128); entries with n >= N or
 *     m >= M (up to the padded extents) are written as 0.  D <= 32 (padded to 1, 2, 4, 8, 16 or 32 inside).
 *     fp64: exp is the device library's algorithm restated for arguments <= 0 (bit-identical values).  An output of 512 MB or
 *     more is written with non-temporal stores (it cannot stay cached for its reader; work on other streams keeps its L2). */
int tsvgp_se_fill_f64(const double *X, const double *Z, const double *inv_ls, double variance, double *K, int64_t N,
                      int M, int D, int64_t ldk, void *stream);
int tsvgp_se_fill_f32(const float *X, const float *Z, const float *inv_ls, float variance, float *K, int64_t N, int M,
                      int D, int64_t ldk, void *stream);

/* (1b) The same fill for the other stationary kernels of the reference's experiments (experiments/uci_regression.py:42-44
 *     uses gpflow.kernels.Matern52): K = variance * k(r), r^2 = sum_d ((x_nd - z_md) * inv_ls_d)^2 (r = sqrt(max(r^2, 1e-36))
 *     as GPflow), kind one of TSVGP_KERNEL_*.  TSVGP_KERNEL_SE is tsvgp_se_fill_*. */
#define TSVGP_KERNEL_SE 0       /* exp(-r^2 / 2) */
/* (1 is reserved: Matern-1/2 = exp(-r) is not offered -- it is not differentiable at r = 0, so GPflow's own expanded-form
 * r^2 leaves 1e-8 relative rounding noise on K(Z, Z) and no 1e-8 parity statement can be made for it) */
#define TSVGP_KERNEL_MATERN32 2 /* (1 + sqrt(3) r) exp(-sqrt(3) r) */
#define TSVGP_KERNEL_MATERN52 3 /* (1 + sqrt(5) r + 5 r^2 / 3) exp(-sqrt(5) r) */
int tsvgp_kernel_fill_f64(int kind, const double *X, const double *Z, const double *inv_ls, double variance, double *K,
                          int64_t N, int M, int D, int64_t ldk, void *stream);
int tsvgp_kernel_fill_f32(int kind, const float *X, const float *Z, const float *inv_ls, float variance, float *K,
                          int64_t N, int M, int D, int64_t ldk, void *stream);

/* (1c) The fill for P latent GPs with one kernel each on shared inputs and shared inducing points
 *     (gpflow SeparateIndependent + SharedIndependentInducingVariables, reference docs/notebooks/heteroskedastic.py:62-76;
 *     Kuf [P, M, N] at src/models/tsvgp.py:269 for BASELINE configs[4]) in ONE launch:
 *        K[p][n, m] = variance[p] * k(r_p),   r_p^2 = sum_d ((x_nd - z_md) * inv_ls[p*D + d])^2,   latent p at K + p*strideK.
 *     inv_ls [P x D] is a device array; variance_host [P] is a HOST array (the scalars travel as kernel arguments, exactly
 *     as `variance` does above).  1 <= P <= TSVGP_MAX_BATCH; strideK >= round_up(N,128) * ldk, even. */
int tsvgp_kernel_fill_batched_f64(int kind, const double *X, const double *Z, const double *inv_ls,
                                  const double *variance_host, double *K, int64_t strideK, int64_t N, int M, int D,
                                  int64_t ldk, int P, void *stream);
int tsvgp_kernel_fill_batched_f32(int kind, const float *X, const float *Z, const float *inv_ls,
                                  const float *variance_host, float *K, int64_t strideK, int64_t N, int M, int D,
                                  int64_t ldk, int P, void *stream);

/* (1d) Input dimensions beyond 32 (the reference's MNIST notebook has D = 784): the scaled squared distance is then a
 *     GEMM, r2[n,m] = xx[n] + zz[m] - 2 G[n,m] with G = (X/l)(Z/l)^T (GPflow's square_distance form [ext]), which the caller
 *     takes from the BLAS library into K [rows_pad x ldk]; this call turns it into K = variance * k(r2) in place and
 *     zero-fills the padding (rows >= N, columns >= M).  xx [N], zz [M]: squared norms of the scaled rows. */
int tsvgp_gram_to_kernel_f64(int kind, double *K, const double *xx, const double *zz, double variance, int64_t N, int M,
                             int64_t ldk, void *stream);
int tsvgp_gram_to_kernel_f32(int kind, float *K, const float *xx, const float *zz, float variance, int64_t N, int M,
                             int64_t ldk, void *stream);

/* (1c) M-step gradient for D beyond tsvgp_kernel_grad_*'s sizes, GEMM form (replaces TensorFlow autodiff through Kuf [ext],
 *     reference experiments/uci_regression.py:159-160, docs/notebooks/mnist.py:117-192 with D = 784).  G [N x ldk] holds the
 *     Gram block x~ z~^T on entry; on return W = -2 variance V * k'(r2) with V = g0 beta^T - 2 g1 * U (padding zero), and
 *     vpart [tsvgp_gram_to_gradw_parts(N, M)] one partial sum of V k(r2) per workgroup (d ELBO / d variance = their sum).
 *     g0, g1: element stride gstride; beta: element stride bstride; U [N x ldu].  The caller finishes in M x D:
 *     dZ = (W^T x~ - z~ * colsum W) / l,  d l_d = (sum_n x~_nd^2 rowsum_n - 2 sum_m z~_md (W^T x~)_md + sum_m z~_md^2 colsum_m) / l_d. */
int64_t tsvgp_gram_to_gradw_parts(int64_t N, int M);
int tsvgp_gram_to_gradw_f64(int kind, double *G, const double *xx, const double *zz, double variance, const double *U,
                            int64_t ldu, const double *g0, const double *g1, int gstride, const double *beta, int bstride,
                            int64_t N, int M, int64_t ldk, double *vpart, void *stream);
int tsvgp_gram_to_gradw_f32(int kind, float *G, const float *xx, const float *zz, float variance, const float *U, int64_t ldu,
                            const float *g0, const float *g1, int gstride, const float *beta, int bstride, int64_t N, int M,
                            int64_t ldk, double *vpart, void *stream);

/* (2) Blocked triangular solve with an N-sized right-hand side, in inverted-factor form:
 *        C[n, i] = sum_{j in range(i)} A[n, j] * Tm[i, j],   range = j<=i | j>=i | all j   (mode)
 *     With Tm = inv(chol(Kuu + jitter I)) and mode LOWER this is B = Kfu * L^-T, i.e. the forward substitution
 *     L^-1 Kuf of tf.linalg.cholesky_solve / triangular_solve (reference src/models/tsvgp.py:270-271 and GPflow's
 *     conditional behind :103), done as MFMA tile products.
 *     A, C [Np x Mp] row-major (lda/ldc = Mp), Tm [Mp x Mp] row-major, zero outside its triangle/valid block. */
int tsvgp_trmm_f64(const double *A, const double *Tm, double *C, int64_t Np, int Mp, int mode, void *stream);
int tsvgp_trmm_f32(const float *A, const float *Tm, float *C, int64_t Np, int Mp, int mode, void *stream);

/* (2b) The same solve batched over the latent dimension (the rank-3 A = cholesky_solve(chol(Kuu [P,M,M]),
 *     Kuf [P,M,N]) of reference src/models/tsvgp.py:270-277 for one kernel per latent): latent b < batch reads
 *     A + b*strideA (strideA = 0: one operand shared by all latents) and Tm + b*strideT, writes C + b*strideC, one launch.
 *     In place (C == A with strideC == strideA) is allowed for TSVGP_TRI_UPPER only: a workgroup owns its 128-row panel,
 *     takes the output column tiles in increasing order, and tile `it` reads only columns >= 128*it before it overwrites
 *     columns [128*it, 128*it + 128).  batch <= 65535; strideC >= Np*Mp when batch > 1. */
int tsvgp_trmm_batched_f64(const double *A, int64_t strideA, const double *Tm, int64_t strideT, double *C,
                           int64_t strideC, int64_t Np, int Mp, int mode, int batch, void *stream);
int tsvgp_trmm_batched_f32(const float *A, int64_t strideA, const float *Tm, int64_t strideT, float *C, int64_t strideC,
                           int64_t Np, int Mp, int mode, int batch, void *stream);

/* (3)+(4) Fused predictive moments and likelihood-gradient map.
 *     For every row n < N and latent p < P:
 *        q    = sum_i ( sum_{j in range(i)} A[n,j] * Tm[p][i,j] )^2
 *        mean = sum_j A[n,j] * gamma[j*P + p]
 *        var  = kdiag - q
 *     then (lik != NONE)  g0 = d ve/d mean,  g1 = min(d ve/d var, -1e-8),  ve_sum += ve.
 *     Replaces base_SVGP.predict_f / GPflow conditional (reference src/models/tsvgp.py:97-114, :246), the site-form
 *     predictive (src/util.py:175-184), likelihood.variational_expectations + GradientTape (:256-259) and the crop (:262-263).
 *     A [Np x Mp]; Tm [P x Mp x Mp]; gamma [Mp x P]; Y [N x P]; outputs mean,var [N x P] (may be NULL),
 *     g0,g1 [Np x P] (rows >= N written as 0; may be NULL when lik == NONE);
 *     ve_partial [Np/128] doubles (per-workgroup sums of ve; may be NULL when lik == NONE);
 *     nonpos_partial [Np/128] int32 (count of var <= 0, the tf.debugging.assert_positive of :113).
 *     lik_param: Gaussian noise variance (ignored otherwise). */
int tsvgp_moments_f64(const double *A, const double *Tm, const double *gamma, const double *Y, double kdiag, int lik,
                      double lik_param, double *mean, double *var, double *g0, double *g1, double *ve_partial,
                      int32_t *nonpos_partial, int64_t N, int64_t Np, int Mp, int P, int mode, void *stream);
int tsvgp_moments_f32(const float *A, const float *Tm, const float *gamma, const float *Y, double kdiag, int lik,
                      double lik_param, float *mean, float *var, float *g0, float *g1, double *ve_partial,
                      int32_t *nonpos_partial, int64_t N, int64_t Np, int Mp, int P, int mode, void *stream);

/* (4) The likelihood-gradient map on its own (reference src/models/tsvgp.py:256-263: GPflow variational_expectations [ext] +
 *     tf.GradientTape): mean, var, Y [N x P] -> g0 = d ve / d mean, g1 = d ve / d var [Np x P] (rows >= N zero; g1 cropped at
 *     -1e-8 unless TSVGP_LIK_NOCROP), ve_partial / nonpos_partial [Np / 128] as tsvgp_moments_*.  lik = TSVGP_LIK_GAUSSIAN
 *     (lik_param = noise variance) or TSVGP_LIK_BERNOULLI.  The moments kernels run this map in their epilogue; this entry
 *     point serves a caller that assembled the moments itself (t_SVGP_white's two-product variance). */
int tsvgp_lik_map_f64(const double *mean, const double *var, const double *Y, int lik, double lik_param, double *g0, double *g1,
                      double *ve_partial, int32_t *nonpos_partial, int64_t N, int64_t Np, int P, void *stream);
int tsvgp_lik_map_f32(const float *mean, const float *var, const float *Y, int lik, double lik_param, float *g0, float *g1,
                      double *ve_partial, int32_t *nonpos_partial, int64_t N, int64_t Np, int P, void *stream);

/* (3b) The moments for P latents with one kernel each: latent p has its own operand A + p*strideA ([Np x Mp] each; strideA = 0
 *     is the shared operand of tsvgp_moments_*) and its own prior variance kdiag_host[p] (HOST array of P doubles, passed
 *     as kernel arguments).  1 <= P <= TSVGP_MAX_BATCH.  Everything else as tsvgp_moments_*; one launch, each workgroup
 *     takes its 128-row panel through the P latents in turn. */
int tsvgp_moments_batched_f64(const double *A, int64_t strideA, const double *Tm, const double *gamma, const double *Y,
                              const double *kdiag_host, int lik, double lik_param, double *mean, double *var, double *g0,
                              double *g1, double *ve_partial, int32_t *nonpos_partial, int64_t N, int64_t Np, int Mp,
                              int P, int mode, void *stream);
int tsvgp_moments_batched_f32(const float *A, int64_t strideA, const float *Tm, const float *gamma, const float *Y,
                              const double *kdiag_host, int lik, double lik_param, float *mean, float *var, float *g0,
                              float *g1, double *ve_partial, int32_t *nonpos_partial, int64_t N, int64_t Np, int Mp,
                              int P, int mode, void *stream);

/* (5) Site accumulation (the two einsums of reference src/models/tsvgp.py:278-281 in whitened coordinates):
 *        acc2[p][i][j] = sum_n g1[n,p] * B[n,i] * B[n,j]        (full symmetric [P x Mp x Mp], fp64)
 *        acc1[p][i]    = sum_n g0[n,p] * B[n,i]                 ([P x Mp], fp64)
 *     B [Np x Mp]; g0,g1 contiguous [Np x P] with rows >= N equal to 0.  B, g0, g1 and work on 16-byte boundaries (operand
 *     rows and the 16 P weights of a chunk travel as 16-byte LDS-DMA pieces; TSVGP_EINVAL otherwise).
 *     The N range is cut into `nsplit` slices; partial tiles go to `work` and are summed in a fixed order (bitwise
 *     reproducible; no atomics).  work must hold tsvgp_site_accum_work_bytes_*(Mp, P, nsplit) bytes.  Any nsplit >= 1 is
 *     valid (a slice is a whole number of chunks -- 16 rows in fp64, 32 in fp32 with P <= 8 -- so a count above Np / chunk
 *     leaves workgroups with empty slices: correct, wasted); the launch runs in rounds of tsvgp_site_accum_slots_*() equal workgroups (per latent: n_off * nsplit off-diagonal
 *     + nt * ceil(20 nsplit / 32) diagonal ones, 22 / 32 in fp32), so a count that fills whole rounds is the fast one
 *     (N = 1e6, M = 1024, fp64: 62 slices = 2048 workgroups = 8 rounds of 256; the Python mirror's EStepEngine.choose_nsplit). */
int64_t tsvgp_site_accum_work_bytes_f64(int Mp, int P, int nsplit);
int64_t tsvgp_site_accum_work_bytes_f32(int Mp, int P, int nsplit);
int tsvgp_site_accum_f64(const double *B, const double *g0, const double *g1, double *acc2, double *acc1, void *work,
                         int64_t Np, int Mp, int P, int nsplit, void *stream);
int tsvgp_site_accum_f32(const float *B, const float *g0, const float *g1, double *acc2, double *acc1, void *work,
                         int64_t Np, int Mp, int P, int nsplit, void *stream);
/* (5b) The same sums with one operand per latent (the rank-3 A of reference src/models/tsvgp.py:271-281): latent p reads
 *     B + p*strideB ([Np x Mp] each; strideB = 0 is tsvgp_site_accum_*).  One launch over all P latents. */
int tsvgp_site_accum_batched_f64(const double *B, int64_t strideB, const double *g0, const double *g1, double *acc2,
                                 double *acc1, void *work, int64_t Np, int Mp, int P, int nsplit, void *stream);
int tsvgp_site_accum_batched_f32(const float *B, int64_t strideB, const float *g0, const float *g1, double *acc2,
                                 double *acc1, void *work, int64_t Np, int Mp, int P, int nsplit, void *stream);

/* (6) Batched lower Cholesky of the M x M site matrices, in place:  A[b] = L[b] L[b]^T, L written to the lower triangle
 *     of the 128x128 diagonal blocks and below (the strictly upper part of the diagonal blocks is zeroed; blocks above
 *     the diagonal are left untouched).  Replaces tf.linalg.cholesky of reference src/models/tsvgp.py:270,300 and
 *     src/util.py:377-388.  M must be a multiple of 128 (pad with an identity block), lda >= M, matrix b starts at
 *     A + b*stride.  info[b] = 0, or the 1-based index of the first non-positive pivot (LAPACK potrf convention).
 *     work: batch * 128 * 128 doubles. */
#define TSVGP_POTRF_SUBST 1 /* flags: solve the panels below each diagonal block by substitution against L_kk (16-wide
                               sub-blocks by forward substitution, the rest by MFMA updates, as LAPACK's blocked trsm)
                               instead of multiplying by inv(L_kk): ~0.15 ms slower at M = 1024, but as robust as a
                               LAPACK factorisation on numerically barely definite matrices (cond ~ 1e14), where the
                               inverse-based panels lose cond(L_kk) digits of the trailing matrix.  The callers set it
                               when cond(K_uu + jitter I) is beyond 1e7. */
#define TSVGP_POTRF_DIAG_V1 4 /* flags: round 4's block step -- the diagonal block's inverse assembled by the 2 x 2 recursion and
                                 the panel rows multiplied by it -- instead of round 5's (inverted 16 x 16 diagonal tiles only,
                                 panel rows by substitution on MFMA tile registers, tsvgp_chol.hip); for A/B measurements */
#define TSVGP_POTRF_DIAG_V2 8 /* flags: factor the diagonal blocks with the MFMA tile-dataflow kernel of tsvgp_chol.hip
                                 (experimental: measured slower than the row-per-lane kernel, profiles/r05_potrf_diag_lab.txt) */
#define TSVGP_POTRF_FUSE 16 /* flags: the diagonal block and the panel rows below it in ONE launch (every panel workgroup factors
                               the diagonal block for itself); experimental: measured slower than the two launches it replaces */
int tsvgp_potrf_f64(double *A, int M, int lda, int batch, int64_t stride, int32_t *info, double *work, int flags,
                    void *stream);

/* (6a) Factor AND solve in one pass: the buffer of matrix b holds, directly below its M rows, `rhs_rows` further rows B
 *     [rhs_rows x M] (same lda; stride >= (M + rhs_rows) * lda).  They ride through the factorisation as panel rows -- every
 *     block step solves them against the diagonal block and applies the trailing update to them with the same tile kernels --
 *     and come out as  B L^-T  (= (L^-1 B^T)^T: the triangular solve of reference src/util.py:173, D = chol(W)^-1 L^T, without
 *     the inverse factor of (6b), its recursion levels and the GEMM that applied it).  rhs_rows a multiple of 128.
 *     TSVGP_POTRF_RHS_UPPER: B is block upper triangular (row block i zero in the column blocks < i), as J L J of a lower
 *     triangular L is -- the zero blocks are skipped.  work, info, TSVGP_POTRF_SUBST as in (6). */
#define TSVGP_POTRF_RHS_UPPER 2
int tsvgp_potrf_solve_f64(double *A, int M, int lda, int batch, int64_t stride, int32_t *info, double *work, int rhs_rows,
                          int flags, void *stream);
/*     dst[b][i][j] = i <= j ? src[b][M-1-j][M-1-i] : 0  on the leading M x M block: transpose + index reversal + upper triangle
 *     in one pass -- what turns the solved rows (J L J) C^-T of (6a) into the upper triangular D = U_W^-1 L^T. */
int tsvgp_flip_transpose_f64(const double *src, int lds, int64_t sstride, double *dst, int ldd, int64_t dstride, int M, int batch,
                             void *stream);

/* (6b) The same factorisation plus the inverse factor: X[b] = inv(L[b]) (lower triangular, exact zeros above) and
 *     Xt[b] = X[b]^T, both [batch x M x M] row-major (leading dimension M).  The inverted diagonal blocks the panel
 *     solve needs anyway are combined by the 2x2 block recursion inv([[A,0],[C,B]]) = [[A^-1,0],[-B^-1 C A^-1, B^-1]]
 *     as MFMA tile products.  Replaces tf.linalg.triangular_solve / cholesky_solve with M x M right-hand sides
 *     (reference src/util.py:168-175, src/models/tsvgp.py:270-271): the callers apply X by GEMM.
 *     T: scratch, batch * M * M doubles.  Other arguments as tsvgp_potrf_f64. */
int tsvgp_potrf_inv_f64(double *A, int M, int lda, int batch, int64_t stride, int32_t *info, double *work, double *X,
                        double *Xt, double *T, int flags, void *stream);

/* (7) Triangle extraction for the factors above, fused with a scale and (optionally) the index reversal of the upper-form
 *     factorisation A = U U^T (U = J C J with C the lower factor of J A J, J the exchange matrix):
 *        dst[b][i][j] = keep ? scale * src[b][si][sj] : 0   on the leading M x M block,
 *        flip = 0: (si, sj) = (i, j), keep = i >= j;   flip = 1: (si, sj) = (M-1-i, M-1-j), keep = i <= j;
 *        flip = 2: (si, sj) = (M-1-i, M-1-j), keep everything (J A J, the input of the upper-form factorisation).
 *     Replaces tf.linalg.band_part / the triangular() transform of reference src/sites.py:63 and the leading minus of
 *     src/models/tsvgp.py:300 (scale = -1) in one pass.  src [batch x * x lds], dst [batch x * x ldd] (strides in elements). */
int tsvgp_tri_copy_f64(const double *src, int lds, int64_t sstride, double *dst, int ldd, int64_t dstride, int M, int batch,
                       double scale, int flip, void *stream);
/* (7') The same pass with a shift of the diagonal, dst[b][i][i] += diag_add (after the scale), and flip = 3: (si, sj) = (i, j), keep
 *     everything -- K_uu + jitter I (reference src/models/tsvgp.py:209-211, :270) and I + L^T K L (src/util.py:171-172) written where
 *     the factorisation reads them, instead of a copy, an addition on the diagonal and a triangle pass each. */
int tsvgp_tri_copy_shift_f64(const double *src, int lds, int64_t sstride, double *dst, int ldd, int64_t dstride, int M, int batch,
                             double scale, double diag_add, int flip, void *stream);

/* (7b) The matrix of the final factorisation of one E-step (reference src/models/tsvgp.py:286-300), in one pass:
 *        G1s    = (G1 + G1^T) / 2
 *        target = c_ll * LLt + c_g * s * G1s + jitter * I,     s = num_data / rows[0]  (1 when num_data <= 0)
 *     with LLt = L L^T of the old site factor, c_ll = 1 - lr, c_g = -2 lr:  -2 [(1-lr) lambda_2 + lr s G1] + jitter I.
 *     `rows` is a device scalar (the all-reduced row count of the step), so the minibatch scale of :286-291 needs no host
 *     read.  All matrices [P x M x M] contiguous. */
int tsvgp_site_target_f64(const double *G1, const double *LLt, double *target, double *G1s, int M, int P, double c_ll,
                          double c_g, double jitter, const double *rows, double num_data, void *stream);

/* (7b') The elementwise part of the site update in one call (reference src/util.py:429-438, src/models/tsvgp.py:284-300):
 *        Gs     = (G1 + G1^T) / 2
 *        target = (1 - lr) LLt - 2 lr s Gs + jitter I                      (what (7b) writes)
 *        l1_new = (1 - lr) l1_old + lr s (G0 - 2 Gs meanZ)                  (the chain rule of util.py:436 and the update of :296)
 *     G1, LLt, target [P x M x M]; G0, meanZ, l1_old, l1_new [M x P]; all contiguous; s and `rows` as in (7b);
 *     work: P * ceil(M / 32) * M doubles (per-tile partial row sums, added in a fixed order).  Replaces (7b), a matrix-vector
 *     product and about ten elementwise launches on the replicated critical path of a step (two launches). */
int tsvgp_site_update_f64(const double *G1, const double *G0, const double *LLt, const double *meanZ, const double *l1_old,
                          double *target, double *l1_new, double *work, int M, int P, double lr, double jitter, const double *rows,
                          double num_data, void *stream);

/* (7b'') beta = l1 - D^T (D v) per latent: D [P x M x M] upper triangular (only that triangle is read), v, l1, beta [M x P]
 *     contiguous; work: P * M * (1 + ceil(M / 64)) doubles.  With v = (K_uu + 1e-6 I) lambda_1 this is K^-1 m of reference
 *     src/util.py:176-179 -- the two triangular matrix-vector products between the factorisation and the moments kernel, three
 *     small launches, summation in a fixed order. */
int tsvgp_site_beta_f64(const double *D, const double *v, const double *l1, double *work, double *beta, int M, int P,
                        void *stream);

/* (7b''') y[:, p] = A_p v[:, p]: row-major A_p [M x M] at A + p * strideA (strideA = 0: one matrix for every latent), v and y
 *     [M x P] contiguous.  The matrix-vector products of the replicated chain -- (K_uu + 1e-6 I) lambda_1 and K_uu beta of
 *     reference src/util.py:176-179 / src/models/tsvgp.py:249-254, K9^-1 acc1 of :279 -- one wave per row. */
int tsvgp_gemv_f64(const double *A, int64_t strideA, const double *v, double *y, int M, int P, void *stream);

/* (7k) Clock keeper.  Not part of the reference's algorithm: the M x M section of reference src/util.py:168-185 and
 *     src/models/tsvgp.py:293-300 is latency-bound (a few workgroups at a time for ~1.5 ms at M = 1024), the chip's clock sags
 *     over it and the N-sized kernels behind it pay for the climb back.  tsvgp_keeper_run launches, on
 *     `stream` (a side stream), `workgroups` workgroups (0: one per CU) of register-only fp64 FMAs at the lowest wave priority
 *     that leave as soon as *flag != 0 or after max_us microseconds (0 < max_us <= 1e6), whichever comes first;
 *     tsvgp_keeper_signal stores `value` to *flag in stream order (0 in front of the launch, 1 on the stream of the chain when
 *     the chain is through).  flag: one int32 in device memory. */
int tsvgp_keeper_run(const int32_t *flag, double max_us, int workgroups, void *stream);
int tsvgp_keeper_signal(int32_t *flag, int value, void *stream);

/* (7c) Status word of one step: flags[0] = sum |info_a| (prelude factorisations), flags[1] = nonpos[0] (count of
 *     non-positive predictive variances, the assert_positive of :113; NULL = 0), flags[2] = sum |info_b| (the final
 *     factorisation, :300).  One device->host read of these three doubles ends a step. */
int tsvgp_step_status_f64(const int32_t *info_a, int na, const int32_t *info_b, int nb, const double *nonpos, double *flags,
                          void *stream);

/* (7d) Lower triangle of P symmetric M x M accumulators <-> packed [P x M (M + 1) / 2] (row by row): the payload of the
 *     all-reduce of the dual accumulators (half of P M^2).  Unpack writes both triangles. */
int tsvgp_sym_pack_f64(const double *A, int lda, int64_t stride, int M, int P, double *packed, void *stream);
int tsvgp_sym_unpack_f64(const double *packed, double *A, int lda, int64_t stride, int M, int P, void *stream);

/* (8) Kernel-parameter gradient contraction for the M-step (d ELBO / d theta with the sites fixed: reference
 *     experiments/uci_regression.py:159-160, pinned by tests/models/test_tsvgp.py:168-188; TensorFlow autodiff there).
 *     For one latent GP, with V[n,m] = g0[n] beta[m] - 2 g1[n] U[n,m] (U = K_fu Q from tsvgp_trmm, Q = D^T D):
 *        sum_{n,m} V[n,m] * dK[n,m]/d variance,   /d lengthscale_d,   and per m  sum_n V[n,m] * dK[n,m]/d Z[m,d].
 *     X [N x D], Z [M x D], U [N x ldu]; g0, g1 with element stride gstride (so a column of an [N x P] array can be
 *     passed), beta with stride bstride.  Outputs are per-block partial sums to be added by the caller:
 *        zpart [nrb x Mp x Dp], lpart [nrb x ncb x Dp], vpart [nrb x ncb],   nrb = ceil(N / tsvgp_kernel_grad_rows()),
 *        ncb = ceil(Mp / 512), Mp = M rounded up to 128, Dp = tsvgp_kernel_grad_dpad(D)  (D <= 16). */
int tsvgp_kernel_grad_rows(void);
int tsvgp_kernel_grad_dpad(int D);
int tsvgp_kernel_grad_f64(int kind, const double *X, const double *Z, const double *inv_ls, double variance,
                          const double *U, int64_t ldu, const double *g0, const double *g1, int gstride,
                          const double *beta, int bstride, int64_t N, int M, int D, double *zpart, double *lpart,
                          double *vpart, void *stream);
int tsvgp_kernel_grad_f32(int kind, const float *X, const float *Z, const float *inv_ls, float variance, const float *U,
                          int64_t ldu, const float *g0, const float *g1, int gstride, const float *beta, int bstride,
                          int64_t N, int M, int D, double *zpart, double *lpart, double *vpart, void *stream);

/* Device self-test of the MFMA fragment maps used above (writes a 16x16 product C = A*B, k = 4, for host checking).
 * a [16 x 4], b [4 x 16], c [16 x 16] row-major. */
int tsvgp_selftest_mfma_f64(const double *a, const double *b, double *c, void *stream);
int tsvgp_selftest_mfma_f32(const float *a, const float *b, float *c, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* TSVGP_HIP_H */
