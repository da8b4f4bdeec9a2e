// tsvgp_kernels.hip -- hand-written CDNA4 (gfx950 / MI355X) kernels for the t-SVGP natural-gradient E-step.
//
// Hot path replaced: t_SVGP.natgrad_step, reference src/models/tsvgp.py:234-304 (see include/tsvgp_hip.h for the
// per-entry-point citations).  Written for gfx950 only: 64-wide wavefronts, v_mfma_f64_16x16x4_f64 /
// v_mfma_f32_16x16x4_f32 matrix instructions, 160 KiB LDS per CU, 256 CUs in 8 XCDs.
//
// Kernel inventory
//   se_fill_kernel      K(X,Z) tile fill (SE, Matern-3/2, Matern-5/2).  Z tile and X rows staged in LDS, 16-byte
//                       coalesced stores.
//   panel_kernel<STORE> C = A * Tm^T restricted to a triangular k-range (inverted-factor triangular solve).  MFMA bound.
//   panel_kernel<MOMENTS>  same product, but the tile is squared and row-summed in registers (never stored); the mean
//                       GEMV rides on the first column tile (FUSE) or runs as a pre-pass; the likelihood-gradient map
//                       runs in the epilogue.
//   mean_lik_kernel     the moments without the variance product (TSVGP_LIK_MEANONLY): mean GEMV + Gaussian gradient
//                       map in one sweep of the operand.  HBM bound.
//   syrk_kernel         weighted Gram  sum_n g1[n] a_n a_n^T  over an N-slice per workgroup (lower tiles only) + the
//                       first-order sum on diagonal tiles.  MFMA bound.  Partial tiles -> syrk_reduce_kernel (fixed order).
//   potrf_diag_kernel, chol_tile_kernel, trtri_level_kernel   blocked Cholesky of the M x M site matrices and the
//                       inverse factor (latency bound; sub-blocked diagonal block, wave-tile products without LDS).
//   kgrad_kernel        kernel-parameter gradient contraction of the M-step.  HBM bound.
//
// Tiling shared by the N-sized MFMA kernels: 128x128 output tile per 256-thread workgroup; wave w owns row blocks
// {w, 7 - w} x all eight 16-column blocks (acc[2][8]); k-chunks of 16 (32 floats in the panel kernels) staged
// global->registers->LDS with two LDS buffers and one barrier per chunk, two chunks per loop body; 2 workgroups per CU (<= 80 KB LDS, <= 256 VGPRs each) so one workgroup's barrier
// and staging hide under the other's MFMAs.  LDS images are padded so that every fragment read and every staging write
// is bank-conflict free: [row][k] images use a row stride of 17 doubles / 34 floats, [k][row] images a k stride of 144.

#include <hip/hip_runtime.h>

#include <atomic>
#include <type_traits>
#include <stdint.h>
#include <stdlib.h>

#include "tsvgp_hip.h"
#include "tsvgp_chol.h"

namespace {

constexpr int TILE = TSVGP_TILE;  // 128
constexpr int KC = 16;            // k-chunk of the site-accumulation kernel ([k][row] images)
#ifndef TSVGP_XTILE
#define TSVGP_XTILE 2  // panel kernels: request the next column tile's first chunk before the current tile's epilogue (1);
                       // also the first tile's first chunk before gamma is staged (2)
#endif
#ifndef TSVGP_CHOL_PRIO
#define TSVGP_CHOL_PRIO 3  // wave priority of the latency-bound factorisation kernels (s_setprio, 0..3)
#endif
#ifndef TSVGP_PANEL_KC_F32
#define TSVGP_PANEL_KC_F32 32
#endif
constexpr int LDS_KS = 144;       // [k][row] image: k stride (elements)
constexpr int NTHREADS = 256;
constexpr int MODE_STORE = 0;
constexpr int MODE_MOMENTS = 1;

typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------------------------
// MFMA wrappers.  A operand: lane l holds A[i = l&15][k = l>>4]; B operand: lane l holds B[k = l>>4][j = l&15].
// C/D: column = l&15 for both types; row = (l>>4) + 4*r for f64, (l>>4)*4 + r for f32 (r = register index 0..3).
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
struct Mfma;
template <>
struct Mfma<double> {
    typedef v4d acc_t;
    typedef v2d pair_t;
    static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct Mfma<float> {
    typedef v4f acc_t;
    typedef v2f pair_t;
    static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};

// k-chunk of the panel kernels ([row][k] images): 16 doubles or 32 floats -- the same 128 bytes per row, the same 64
// bytes staged per thread and operand, and the same MFMA time between two barriers for both types (an fp32 MFMA takes
// half the cycles of an fp64 one; with 16-float chunks the barrier and the staging latency weigh twice as much).
// Row stride in elements: 17 doubles / 34 floats (18 for 16-float chunks) make the fragment reads (banked modulo 32
// dwords per group of 32 lanes) and the 8-byte staging writes conflict free.
template <typename T>
struct PanelK {
    static constexpr int KC = sizeof(T) == 8 ? 16 : TSVGP_PANEL_KC_F32;
    static constexpr int H = KC / 2;  // elements one thread stages per operand and chunk
    static constexpr int RS = KC + (sizeof(T) == 8 ? 1 : 2);
#ifdef TSVGP_PANEL_DEPTH_F32
    static constexpr int DEPTH = sizeof(T) == 8 ? 1 : TSVGP_PANEL_DEPTH_F32;
#else
    static constexpr int DEPTH = 1;  // chunks of global prefetch held in registers (1 or 2)
#endif
};

// H consecutive elements of one row: global -> registers (16-byte loads) and registers -> LDS (8-byte stores: the rows
// of the double image are only 8-byte aligned).
template <typename T, int H>
__device__ __forceinline__ void load_run(T (&r)[H], const T* __restrict__ p) {
    if constexpr (sizeof(T) == 8) {
#pragma unroll
        for (int q = 0; q < H / 2; ++q) {
            v2d v = *reinterpret_cast<const v2d*>(p + 2 * q);
            r[2 * q] = v[0];
            r[2 * q + 1] = v[1];
        }
    } else {
#pragma unroll
        for (int q = 0; q < H / 4; ++q) {
            v4f v = *reinterpret_cast<const v4f*>(p + 4 * q);
            r[4 * q] = v[0];
            r[4 * q + 1] = v[1];
            r[4 * q + 2] = v[2];
            r[4 * q + 3] = v[3];
        }
    }
}
template <typename T, int H>
__device__ __forceinline__ void store_run(T* p, const T (&r)[H]) {
    if constexpr (sizeof(T) == 8) {
#pragma unroll
        for (int q = 0; q < H; ++q) p[q] = r[q];
    } else {
#pragma unroll
        for (int q = 0; q < H / 2; ++q) {
            v2f v;
            v[0] = r[2 * q];
            v[1] = r[2 * q + 1];
            *reinterpret_cast<v2f*>(p + 2 * q) = v;
        }
    }
}

// Wave w (0..3) of a workgroup owns the 16-row blocks {w, 7 - w} and ALL eight 16-column blocks of the 128x128 tile
// (32 x 128 per wave, acc[2][8]).  Every wave therefore sees the same column structure -- the k-tile on the diagonal
// of a triangular product skips the same (chunk, column block) pairs in all waves, with compile-time masks -- and the
// row pairing {w, 7 - w} gives every wave 9 of its 16 accumulators on a diagonal tile of the symmetric product.
__device__ __forceinline__ int row_block(int w, int slot) { return slot == 0 ? w : 7 - w; }

// One k-chunk of MFMAs on [row][k] images: acc[s][n] += A(row block s) * B(column block n)^T.
// MLO / MHI (compile time): bit n set = column block n takes part in the first / second half of the chunk's k-steps
// (a 32-wide chunk spans two 16-wide k-blocks of a triangular operand; for 16-wide chunks both masks are equal).
template <typename T, int MLO, int MHI>
__device__ __forceinline__ void mma_chunk_rowk(typename Mfma<T>::acc_t (&acc)[2][8], const T* __restrict__ As,
                                               const T* __restrict__ Bs, int w, int lane) {
    constexpr int RS = PanelK<T>::RS, PKC = PanelK<T>::KC, NKS = PKC / 4;
    const int lr = lane & 15, lk = lane >> 4;
    const T* ap0 = As + (w * 16 + lr) * RS + lk;
    const T* ap1 = As + ((7 - w) * 16 + lr) * RS + lk;
    const T* bp = Bs + lr * RS + lk;
#ifdef TSVGP_ROWK_PIPE
    // Two fragment register sets of ONE k-step each (the same 2 x (2 + 8) values the unpipelined form holds for two k-steps):
    // the reads of k-step ks + 1 are issued in front of the MFMAs of k-step ks, so that within a chunk only the first
    // k-step's read latency is exposed to a wave that has the matrix pipe to itself (the partner workgroup of the CU in a
    // prologue, an epilogue or a latency-bound diagonal chunk).
    T a[2][2], b[2][8];
    auto rd = [&](const int set, const int ks, const int NMASK) __attribute__((always_inline)) {
        a[set][0] = ap0[ks * 4];
        a[set][1] = ap1[ks * 4];
#pragma unroll
        for (int n = 0; n < 8; ++n)
            if (NMASK & (1 << n)) b[set][n] = bp[n * 16 * RS + ks * 4];
    };
    rd(0, 0, MLO);
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const int NMASK = (ks < NKS / 2) ? MLO : MHI;
        if (ks + 1 < NKS) rd((ks + 1) & 1, ks + 1, (ks + 1 < NKS / 2) ? MLO : MHI);
#pragma unroll
        for (int n = 0; n < 8; ++n)
            if (NMASK & (1 << n)) {
                acc[0][n] = Mfma<T>::run(a[ks & 1][0], b[ks & 1][n], acc[0][n]);
                acc[1][n] = Mfma<T>::run(a[ks & 1][1], b[ks & 1][n], acc[1][n]);
            }
    }
#else
#pragma unroll
    for (int ks = 0; ks < PKC / 4; ++ks) {
        const int NMASK = (ks < PKC / 8) ? MLO : MHI;
        T a[2], b[8];
        a[0] = ap0[ks * 4];
        a[1] = ap1[ks * 4];
#pragma unroll
        for (int n = 0; n < 8; ++n)
            if (NMASK & (1 << n)) b[n] = bp[n * 16 * RS + ks * 4];
#pragma unroll
        for (int n = 0; n < 8; ++n)
            if (NMASK & (1 << n)) {
                acc[0][n] = Mfma<T>::run(a[0], b[n], acc[0][n]);
                acc[1][n] = Mfma<T>::run(a[1], b[n], acc[1][n]);
            }
    }
#endif
}

// One k-chunk (16) of MFMAs on [k][row] images.  DIAG: only accumulators with column block <= row block
// (W = the wave index, compile time, selects the row blocks {W, 7 - W}).
template <typename T, bool DIAG, int W>
__device__ __forceinline__ void mma_chunk_krow(typename Mfma<T>::acc_t (&acc)[2][8], const T* __restrict__ As,
                                               const T* __restrict__ Bs, int w, int lane) {
    const int lr = lane & 15, lk = lane >> 4;
    const T* ap0 = As + lk * LDS_KS + w * 16 + lr;
    const T* ap1 = As + lk * LDS_KS + (7 - w) * 16 + lr;
    const T* bp = Bs + lk * LDS_KS + lr;
#pragma unroll
    for (int ks = 0; ks < KC / 4; ++ks) {
        T a[2], b[8];
        a[0] = ap0[ks * 4 * LDS_KS];
        a[1] = ap1[ks * 4 * LDS_KS];
#pragma unroll
        for (int n = 0; n < 8; ++n)
            if (!DIAG || n <= 7 - W || n <= W) b[n] = bp[ks * 4 * LDS_KS + n * 16];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            if (!DIAG || n <= W) acc[0][n] = Mfma<T>::run(a[0], b[n], acc[0][n]);
            if (!DIAG || n <= 7 - W) acc[1][n] = Mfma<T>::run(a[1], b[n], acc[1][n]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Likelihood maps (fp64; round 5: the Bernoulli quadrature of fp32 N-arrays in fp32, bern_sums_f below).  Restates gpflow.likelihoods.{Gaussian,Bernoulli}.variational_expectations
// and its tf.GradientTape derivative (reference src/models/tsvgp.py:256-263).
// ---------------------------------------------------------------------------------------------------------------
__device__ const double GH_X[10] = {  // positive nodes of 20-pt Gauss-Hermite, times sqrt(2)
    3.46964157081355917e-01, 1.04294534880275114e+00, 1.74524732081412703e+00, 2.45866361117236787e+00,
    3.18901481655339003e+00, 3.94396735065731630e+00, 4.73458133404605519e+00, 5.57873880589320148e+00,
    6.51059015701365507e+00, 7.61904854167975909e+00};
__device__ const double GH_W[10] = {  // matching weights / sqrt(pi)
    2.60793063449554885e-01, 1.61739333983999978e-01, 6.15063720639768968e-02, 1.39978374471010220e-02,
    1.83010313108049002e-03, 1.28826279961929280e-04, 4.40212109023085101e-06, 6.12749025998292797e-08,
    2.48206236231517553e-10, 1.25780067243792340e-13};

__device__ __forceinline__ void bern_point(double f, bool y1, double& lp, double& dl) {
    const double jit = 1e-3;
    const double p = 0.5 * (1.0 + erf(f * 0.70710678118654752440)) * (1.0 - 2.0 * jit) + jit;
    const double dp = (1.0 - 2.0 * jit) * 0.39894228040143267794 * exp(-0.5 * f * f);
    if (y1) {
        lp = log(p);
        dl = dp / p;
    } else {
        const double q = 1.0 - p;
        lp = log(q);
        dl = -dp / q;
    }
}

// Gauss-Hermite partial sums over the node pairs [i0, i1) of the Bernoulli map (two threads share a row in the panel kernel)
__device__ __forceinline__ void bern_sums(double m, double sd, bool y1, int i0, int i1, double& a0, double& a1, double& av) {
    a0 = a1 = av = 0.0;
#pragma unroll 1
    for (int i = i0; i < i1; ++i) {
        const double z = GH_X[i], w = GH_W[i];
        double lp, dl;
        bern_point(m + sd * z, y1, lp, dl);
        av += w * lp;
        a0 += w * dl;
        a1 += w * dl * z;
        bern_point(m - sd * z, y1, lp, dl);
        av += w * lp;
        a0 += w * dl;
        a1 -= w * dl * z;
    }
}

// The same sums in fp32, for the fp32 N-arrays (round 5): mean and variance arrive with 1e-6 of relative error from the fp32 MFMA
// sums, and the fp64 quadrature -- 20 x (erf, exp, log, a division) per row on four waves per CU -- was 0.30 ms of the 8.0 ms fp32
// moments kernel at N = 1e6 (profiles/r05_lik_split_lab.txt).  The class probability is formed from erfc of the signed argument,
// never as 1 - p: p saturates at 1 - 1e-3, where 1 - p in fp32 would keep four digits.
__device__ const float GH_XF[10] = {3.46964157e-01f, 1.04294535e+00f, 1.74524732e+00f, 2.45866361e+00f, 3.18901482e+00f,
                                    3.94396735e+00f, 4.73458133e+00f, 5.57873881e+00f, 6.51059016e+00f, 7.61904854e+00f};
__device__ const float GH_WF[10] = {2.60793063e-01f, 1.61739334e-01f, 6.15063721e-02f, 1.39978374e-02f, 1.83010313e-03f,
                                    1.28826280e-04f, 4.40212109e-06f, 6.12749026e-08f, 2.48206236e-10f, 1.25780067e-13f};
__device__ __forceinline__ void bern_point_f(float f, bool y1, float& lp, float& dl) {
    const float jit = 1e-3f;
    const float pp = 0.5f * erfcf((y1 ? -f : f) * 0.70710678f) * (1.0f - 2.0f * jit) + jit;  // p(y | f), >= 1e-3
    const float dp = (1.0f - 2.0f * jit) * 0.39894228f * expf(-0.5f * f * f);
    lp = logf(pp);
    dl = (y1 ? dp : -dp) / pp;
}
__device__ __forceinline__ void bern_sums_f(float m, float sd, bool y1, int i0, int i1, float& a0, float& a1, float& av) {
    a0 = a1 = av = 0.0f;
#pragma unroll 1
    for (int i = i0; i < i1; ++i) {
        const float z = GH_XF[i], w = GH_WF[i];
        float lp, dl;
        bern_point_f(m + sd * z, y1, lp, dl);
        av += w * lp;
        a0 += w * dl;
        a1 += w * dl * z;
        bern_point_f(m - sd * z, y1, lp, dl);
        av += w * lp;
        a0 += w * dl;
        a1 -= w * dl * z;
    }
}
// The quadrature in the arithmetic of the N-arrays' type T (TSVGP_BERN_F64=1 at build time: fp64 for both, as up to round 5)
#ifndef TSVGP_BERN_F64
#define TSVGP_BERN_F64 0
#endif
template <typename T>
__device__ __forceinline__ void bern_sums_t(double m, double sd, bool y1, int i0, int i1, double& a0, double& a1, double& av) {
    if constexpr (sizeof(T) == 4 && !TSVGP_BERN_F64) {
        float b0, b1, bv;
        bern_sums_f((float)m, (float)sd, y1, i0, i1, b0, b1, bv);
        a0 = (double)b0;
        a1 = (double)b1;
        av = (double)bv;
    } else {
        bern_sums(m, sd, y1, i0, i1, a0, a1, av);
    }
}

__device__ __forceinline__ void lik_eval(int lik_flags, double s2, double m, double v, double y, double& g0, double& g1,
                                         double& ve) {
    const int lik = lik_flags & 0xFF;
    if (lik == TSVGP_LIK_GAUSSIAN) {
        const double r = y - m;
        g0 = r / s2;
        g1 = -0.5 / s2;
        ve = -0.5 * 1.83787706640934548356 - 0.5 * log(s2) - 0.5 * (r * r + v) / s2;
    } else {  // Bernoulli, probit, 20-pt Gauss-Hermite; derivative OF the quadrature sum
        const double sd = sqrt(v);
        double a0, a1, av;
        bern_sums(m, sd, y == 1.0, 0, 10, a0, a1, av);
        g0 = a0;
        g1 = a1 / (2.0 * sd);
        ve = av;
    }
    if (!(lik_flags & TSVGP_LIK_NOCROP)) g1 = fmin(g1, -1e-8);  // reference tsvgp.py:262-263 (tsvgp_white.py does not crop)
}

// ---------------------------------------------------------------------------------------------------------------
// se_fill_kernel: K[n, m] = variance * exp(-0.5 * sum_d ((x_nd - z_md) * inv_ls_d)^2), zero in the padding.
// grid = (row blocks of FILL_ROWS, column tiles of FILL_COLS); each thread owns two adjacent columns.
// ---------------------------------------------------------------------------------------------------------------
constexpr int FILL_ROWS = 64;
#ifndef TSVGP_FILL_ROWS_DEFAULT
#define TSVGP_FILL_ROWS_DEFAULT 64
#endif
constexpr int FILL_ROWS_DEFAULT = TSVGP_FILL_ROWS_DEFAULT;
constexpr int FILL_COLS = 512;

// exp(a) for a <= 0 (every kernel profile below).  fp64: the device library's algorithm restated -- k = rint(a log2 e),
// r = a - k ln2 (two-term), degree-11 polynomial, ldexp; same constants, same operation order, bit-identical values
// -- minus what a <= 0 does not need and with the coefficients as FMA addends: 22 VALU instructions instead of the 33
// the library call compiles to (9 of them register copies in front of v_fmac, 5 for the overflow / underflow selects).
// The K(X, Z) fill is VALU-issue bound (108 fp64-rate instructions per row pair, 1.66 ms of arithmetic against 1.35 ms
// of stores at N = 1e6, M = 1024), so this is 24 % of its arithmetic.  Below -1080 the result is 0 as in the library
// (ldexp underflows); NaN stays NaN.
template <typename T>
__device__ __forceinline__ T exp_nonpos(T a) {
    return exp(a);
}
// fp32: the hardware exponential, v_exp_f32(a log2 e) -- two instructions where the library's range-reduced expf takes thirteen
// (a quarter-rate v_exp_f32 among them either way): with the packed distance loop the exponentials were half of the fp32 fill's
// issue time.  Relative error ~|a| 2^-24 (the product a log2 e is rounded once), i.e. <= 1e-5 for kernel values down to 1e-30 and
// far inside the fp32 path's stated tolerance against the fp64 reference values (atol 1e-4 + rtol 1e-3); results below the
// normal range flush to zero.
template <>
__device__ __forceinline__ float exp_nonpos<float>(float a) {
    return __expf(a);
}
template <>
__device__ __forceinline__ double exp_nonpos<double>(double a) {
    constexpr auto C = [](unsigned long long bits) { return __builtin_bit_cast(double, bits); };
    a = (a < -1080.0) ? -1080.0 : a;
    const double k = __builtin_rint(a * C(0x3ff71547652b82feULL));
    double r = fma(C(0xbfe62e42fefa39efULL), k, a);
    r = fma(C(0xbc7abc9e3b39803fULL), k, r);
    // q <- r q + c with c in a scalar register pair (VOP3 takes one scalar operand): left to itself the compiler keeps the
    // coefficients in VGPRs and copies each one in front of a destructive v_fmac
    auto horner = [](double r_, double q_, double c) {
        double d;
        asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(r_), "v"(q_), "s"(c));
        return d;
    };
    double q = horner(r, C(0x3e5ade156a5dcb37ULL), C(0x3e928af3fca7ab0cULL));
    q = horner(r, q, C(0x3ec71dee623fde64ULL));
    q = horner(r, q, C(0x3efa01997c89e6b0ULL));
    q = horner(r, q, C(0x3f2a01a014761f6eULL));
    q = horner(r, q, C(0x3f56c16c1852b7b0ULL));
    q = horner(r, q, C(0x3f81111111122322ULL));
    q = horner(r, q, C(0x3fa55555555502a1ULL));
    q = horner(r, q, C(0x3fc5555555555511ULL));
    q = horner(r, q, C(0x3fe000000000000bULL));
    q = fma(r, q, 1.0);
    q = fma(r, q, 1.0);
    return ldexp(q, (int)k);
}

// stationary kernel profile k(s) as a function of the scaled squared distance s = sum_d ((x_d - z_d) / l_d)^2
// (GPflow [ext]: Matern kernels take r = sqrt(max(s, 1e-36)))
template <int KIND, typename T>
__device__ __forceinline__ T kernel_profile(T s) {
    if constexpr (KIND == TSVGP_KERNEL_SE) {
        return exp_nonpos(T(-0.5) * s);
    } else {
        const T r = sqrt(s > T(1e-36) ? s : T(1e-36));
        if constexpr (KIND == TSVGP_KERNEL_MATERN32) {
            const T a = T(1.7320508075688772935) * r;
            return (T(1) + a) * exp_nonpos(-a);
        } else {
            const T a = T(2.2360679774997896964) * r;
            return (T(1) + a + T(5.0 / 3.0) * r * r) * exp_nonpos(-a);
        }
    }
}

// DT = D padded to a compile-time size (padded dimensions are zeros on both sides).  The thread's two Z columns live in
// registers and only the X rows go through LDS (FILL_ROWS * DT elements, wave-uniform 16-byte reads), the distance loop
// is fully unrolled and the two columns' dependent chains are interleaved.  At N = 1e6, M = 1024, D = 8 (fp64) the
// kernel is bound by VALU issue, not by its stores: 81 fp64-rate instructions per row pair (32 distance, 2 x 23
// exp / scale, 3 loop) = 1.26 ms with the store disabled (-DTSVGP_EXP_NOSTORE), the same 8.2 GB written by a
// store-only kernel of this access pattern 1.35 ms (6.1 TB/s, tools/store_pattern.hip), together 1.86 ms -- was
// 2.23 ms with the library exp (108 instructions, 1.66 ms of arithmetic), per-row 64-bit VALU index compares and
// exec-masked column branches.  The occupancy cap costs nothing (3 / 4 / 8 waves per SIMD within 2 %), the order in
// which row blocks and column tiles are dealt to workgroups neither (2.18 vs 2.18 ms), non-temporal stores 3 %.
#ifndef TSVGP_FILL_MAXWAVES
#define TSVGP_FILL_MAXWAVES 3
#endif
// Latent batch (grid.z): latent p has its own lengthscales inv_ls[p * D ..], variance var.v[p] and output K + p * strideK;
// X and Z are shared (SharedIndependentInducingVariables + SeparateIndependent, reference docs/notebooks/heteroskedastic.py:62-76).
template <typename T>
struct FillBatch {
    T v[TSVGP_MAX_BATCH];
};

// CPT = columns per thread: 2, or 4 in fp32 -- a thread's output of one row is then 16 bytes in either type.  With 8-byte
// stores the fp32 fill reached 2.3 TB/s where the fp64 fill (16-byte stores) reaches 4.5: 8-byte accesses run at 0.54-0.70 of
// the 16-byte rate (MI355X guide, stores of each flavour), and the four columns give the distance loop four independent chains.
template <typename T, int CPT>
struct FillVec;
template <>
struct FillVec<double, 2> {
    typedef v2d type;
};
template <>
struct FillVec<float, 2> {
    typedef v2f type;
};
template <>
struct FillVec<float, 4> {
    typedef v4f type;
};

// A streaming store that also writes THROUGH the L2 (sc0 sc1 nt): it leaves no dirty line behind.  Experiment of round 5
// (TSVGP_FILL_STORE=wt): every kernel boundary of the M x M chain that runs beside the fill writes the L2s back -- the XCDs' L2s are
// not coherent with each other -- and with plain or nt stores that write-back finds the fill's dirty lines.
template <typename V>
__device__ __forceinline__ void store_write_through(V* p, V v) {
    if constexpr (sizeof(V) == 16) {
        asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    } else {
        static_assert(sizeof(V) == 8, "8- or 16-byte vectors");
        asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
    }
}

template <typename T, int KIND, int DT, int CPT = 2>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(1, TSVGP_FILL_MAXWAVES))) void se_fill_kernel(
    const T* __restrict__ X, const T* __restrict__ Z, const T* __restrict__ inv_ls, FillBatch<T> var, T* __restrict__ K,
    int64_t strideK, int64_t N, int M, int D, int64_t ldk, int64_t rows_pad, int cols_pad, int stream_out, int rows_blk) {
    // rows_blk <= FILL_ROWS: rows per block.  FILL_ROWS for the N-sized operand; a SMALL matrix -- K(Z, Z), which opens the
    // replicated M x M chain of every step -- takes 8, i.e. eight times as many workgroups (M = 1024: 256 instead of 32:
    // the launch was 50-60 us of a chain whose every microsecond is on the critical path of an 8-way shard)
    __shared__ __attribute__((aligned(16))) T Xs[FILL_ROWS][DT];  // pre-scaled by inv_ls
#ifdef TSVGP_FILL_EXPANDED
    __shared__ T Xn[FILL_ROWS];  // |x~|^2 of the staged rows
#endif
    typedef typename FillVec<T, CPT>::type vec_t;

    __builtin_amdgcn_s_setprio(1);  // in front of the clock keeper's waves (priority 0), behind the factorisation's (3)
    const T variance = var.v[blockIdx.z];
    inv_ls += (size_t)blockIdx.z * D;
    K += (size_t)blockIdx.z * strideK;
    const int t = threadIdx.x;
    const int ctile = blockIdx.y;
    const int64_t rb_first = blockIdx.x, rb_step = gridDim.x;
    const int m = ctile * (CPT * NTHREADS) + CPT * t;  // this thread's columns
    const bool active = (m < cols_pad);
    T varc[CPT], z[CPT][DT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const bool vc = m + c < M;
        varc[c] = vc ? variance : T(0);
#pragma unroll
        for (int d = 0; d < DT; ++d) z[c][d] = (vc && d < D) ? Z[(int64_t)(m + c) * D + d] * inv_ls[d] : T(0);
    }
#ifdef TSVGP_FILL_EXPANDED
    // Round 4 experiment (-DTSVGP_FILL_EXPANDED, measured and NOT the default: profiles/r04_fill_expanded_ab.txt): the scaled squared
    // distance in the expanded form r2 = |x~|^2 + |z~|^2 - 2 x~.z~ -- GPflow's square_distance [ext] -- instead of
    // sum_d (x~_d - z~_d)^2: one FMA per dimension and column instead of a subtraction and an FMA (D = 8: 18 instead of 32 of the
    // ~81 instructions per row pair; D = 16: 34 instead of 64).  fp64, D = 8: 1.93 -> 1.87 ms alone, nothing in the step; fp32,
    // D = 16: 1.76 -> 1.68 ms, the step 18.83 -> 18.62 ms -- the kernel is not purely VALU-issue bound -- and in fp32 the expanded
    // form costs accuracy exactly where the E-step has its inducing points (Z = X[:M]: r2 = 0 becomes +-1e-6).  The difference
    // form stays.
    T zz[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        zz[c] = T(0);
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            zz[c] += z[c][d] * z[c][d];
            z[c][d] *= T(-2);
        }
    }
#endif
    // Row blocks are dealt round-robin to the workgroups of a column tile.  The default grid has one workgroup per
    // row block; a smaller grid (tsvgp_kernel_fill's cap) leaves CU slots free for work on another stream.
    const int64_t nrb = (rows_pad + rows_blk - 1) / rows_blk;
    for (int64_t rb = rb_first; rb < nrb; rb += rb_step) {
        const int64_t n0 = rb * rows_blk;
        for (int idx = t; idx < rows_blk * DT; idx += NTHREADS) {
            const int rr = idx / DT, d = idx - rr * DT;
            const int64_t n = n0 + rr;
            Xs[rr][d] = (n < N && d < D) ? X[n * D + d] * inv_ls[d] : T(0);
        }
        __syncthreads();
#ifdef TSVGP_FILL_EXPANDED
        if (t < rows_blk) {
            T acc = T(0);
#pragma unroll
            for (int d = 0; d < DT; ++d) acc += Xs[t][d] * Xs[t][d];
            Xn[t] = acc;
        }
        __syncthreads();
#endif
        if (active) {
            // row counts of this block as wave-uniform ints: valid rows get kernel values, the padding rows up to
            // rows_pad zeros.  Invalid columns of the last group come out as exact zeros through their variance factor.
            const int64_t left = N - n0, left_pad = rows_pad - n0;
            const int nvalid = left >= rows_blk ? rows_blk : (left > 0 ? (int)left : 0);
            const int nrows = left_pad >= rows_blk ? rows_blk : (int)left_pad;
            T* const Kp = K + n0 * ldk + m;
            for (int rr = 0; rr < nvalid; ++rr) {
                T sv[CPT];
#ifdef TSVGP_FILL_EXPANDED
                const T xn = Xn[rr];
#pragma unroll
                for (int c = 0; c < CPT; ++c) sv[c] = xn + zz[c];
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    const T x = Xs[rr][d];
#pragma unroll
                    for (int c = 0; c < CPT; ++c) sv[c] = fma(x, z[c][d], sv[c]);
                }
#else
                if constexpr (sizeof(T) == 4) {
                    // fp32: two columns per PACKED instruction (v_pk_add_f32 / v_pk_fma_f32).  A plain fp32 vector instruction
                    // costs a wave64 four cycles on this chip -- the 157 TFLOP/s vector peak is the packed rate -- and the
                    // difference form is two dependent operations per (dimension, column): measured ~5 cycles per instruction
                    // and 2.3-2.55 TB/s at D = 16 before (tools/fill_alone_f32.py)
                    v2f sp[CPT / 2];
#pragma unroll
                    for (int c = 0; c < CPT / 2; ++c) sp[c] = v2f{0.0f, 0.0f};
#pragma unroll
                    for (int d = 0; d < DT; ++d) {
                        const float x = Xs[rr][d];
                        const v2f xx = {x, x};
#pragma unroll
                        for (int c = 0; c < CPT / 2; ++c) {
                            const v2f dd = xx - v2f{z[2 * c][d], z[2 * c + 1][d]};
                            sp[c] = __builtin_elementwise_fma(dd, dd, sp[c]);
                        }
                    }
#pragma unroll
                    for (int c = 0; c < CPT / 2; ++c) {
                        sv[2 * c] = sp[c][0];
                        sv[2 * c + 1] = sp[c][1];
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < CPT; ++c) sv[c] = T(0);
#pragma unroll
                    for (int d = 0; d < DT; ++d) {
                        const T x = Xs[rr][d];
#pragma unroll
                        for (int c = 0; c < CPT; ++c) {
                            const T dd = x - z[c][d];
                            sv[c] += dd * dd;
                        }
                    }
                }
#endif
                vec_t out;
#pragma unroll
                for (int c = 0; c < CPT; ++c) {
#ifdef TSVGP_EXP_NOEXP  // ablation switch (tools/exp_fill.py): the store-bound floor of the kernel
                    out[c] = varc[c] * (T(1) - T(0.5) * sv[c]);
#else
                    out[c] = varc[c] * kernel_profile<KIND>(sv[c]);
#endif
                }
#ifdef TSVGP_EXP_NOSTORE  // ablation switch: the arithmetic alone (the store never executes, the compiler cannot know)
                if (out[0] == T(-1)) *reinterpret_cast<vec_t*>(Kp + (int64_t)rr * ldk) = out;
#else
                if (stream_out == 2)
                    store_write_through(reinterpret_cast<vec_t*>(Kp + (int64_t)rr * ldk), out);
                else if (stream_out)
                    __builtin_nontemporal_store(out, reinterpret_cast<vec_t*>(Kp + (int64_t)rr * ldk));
                else
                    *reinterpret_cast<vec_t*>(Kp + (int64_t)rr * ldk) = out;
#endif
            }
            vec_t zero;
#pragma unroll
            for (int c = 0; c < CPT; ++c) zero[c] = T(0);
            for (int rr = nvalid; rr < nrows; ++rr) *reinterpret_cast<vec_t*>(Kp + (int64_t)rr * ldk) = zero;
        }
        if (rb + rb_step < nrb) __syncthreads();  // Xs is rewritten by the next row block
    }
}

// ---------------------------------------------------------------------------------------------------------------
// kgrad_kernel: the N-sized part of d ELBO / d (variance, lengthscales, Z) for one latent GP (M-step, reference
// experiments/uci_regression.py:159-160).  With s = sum_d s_d^2, s_d = (x_nd - z_md) / l_d and K = variance * f(s):
//   V[n, m]  = g0[n] * beta[m] - 2 * g1[n] * U[n, m]                       (U = K_fu Q, a tsvgp_trmm product)
//   dvar    += V * f(s)
//   dls[d]  += V * variance * f'(s) * (-2 s_d^2 / l_d)
//   dZ[m,d] += V * variance * f'(s) * (-2 s_d   / l_d)
// f'(s) in closed form: SE -f/2;  Matern-3/2 -(3/2) e^-a, a = sqrt(3 s);  Matern-5/2 -(5/6)(1 + a) e^-a, a = sqrt(5 s).
// HBM bound (one read of U).  Grid (row blocks of KG_ROWS, column tiles of FILL_COLS); thread = two columns, its dZ sums
// stay in registers over the block's rows; per-block partials are written out and summed by the caller in a fixed order
// (bitwise reproducible, no atomics).  DT = D padded to a compile-time size (padded dimensions contribute zeros).
// ---------------------------------------------------------------------------------------------------------------
constexpr int KG_ROWS = 1024;
constexpr int KG_CHUNK = 64;

template <int KIND, typename T>
__device__ __forceinline__ void kernel_profile_grad(T s, T& f, T& df) {
    if constexpr (KIND == TSVGP_KERNEL_SE) {
        f = exp(T(-0.5) * s);
        df = T(-0.5) * f;
    } else {
        const T r = sqrt(s > T(1e-36) ? s : T(1e-36));
        if constexpr (KIND == TSVGP_KERNEL_MATERN32) {
            const T a = T(1.7320508075688772935) * r, e = exp(-a);
            f = (T(1) + a) * e;
            df = T(-1.5) * e;
        } else {
            const T a = T(2.2360679774997896964) * r, e = exp(-a);
            f = (T(1) + a + T(5.0 / 3.0) * r * r) * e;
            df = T(-5.0 / 6.0) * (T(1) + a) * e;
        }
    }
}

template <typename T, int KIND, int DT>
__global__ __launch_bounds__(NTHREADS) void kgrad_kernel(const T* __restrict__ X, const T* __restrict__ Z,
                                                         const T* __restrict__ inv_ls, T variance,
                                                         const T* __restrict__ U, int64_t ldu, const T* __restrict__ g0,
                                                         const T* __restrict__ g1, int gstride,
                                                         const T* __restrict__ beta, int bstride, int64_t N, int M, int D,
                                                         double* __restrict__ zpart, double* __restrict__ lpart,
                                                         double* __restrict__ vpart, int Mp) {
    __shared__ T Xs[KG_CHUNK][DT];
    __shared__ T Gs[KG_CHUNK][2];
    __shared__ double red[NTHREADS / 64][DT + 1];
    typedef typename Mfma<T>::pair_t pair_t;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int64_t n0 = (int64_t)blockIdx.x * KG_ROWS;
    const int m0 = blockIdx.y * FILL_COLS + 2 * t;
    const bool c0 = m0 < M, c1 = m0 + 1 < M;
    T z[2][DT], il[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d) {
        il[d] = d < D ? inv_ls[d] : T(0);
        z[0][d] = (c0 && d < D) ? Z[(int64_t)m0 * D + d] * il[d] : T(0);
        z[1][d] = (c1 && d < D) ? Z[(int64_t)(m0 + 1) * D + d] * il[d] : T(0);
    }
    const T b0 = c0 ? beta[(int64_t)m0 * bstride] : T(0), b1 = c1 ? beta[(int64_t)(m0 + 1) * bstride] : T(0);
    double za[2][DT], la[DT], va = 0.0;
#pragma unroll
    for (int d = 0; d < DT; ++d) za[0][d] = za[1][d] = la[d] = 0.0;
    const int64_t nend = (n0 + KG_ROWS < N) ? n0 + KG_ROWS : N;
    for (int64_t nc = n0; nc < nend; nc += KG_CHUNK) {
        __syncthreads();
        for (int idx = t; idx < KG_CHUNK * DT; idx += NTHREADS) {
            const int rr = idx / DT, d = idx - rr * DT;
            const int64_t n = nc + rr;
            Xs[rr][d] = (n < nend && d < D) ? X[n * D + d] * inv_ls[d] : T(0);
        }
        if (t < KG_CHUNK) {
            const int64_t n = nc + t;
            Gs[t][0] = n < nend ? g0[n * gstride] : T(0);
            Gs[t][1] = n < nend ? g1[n * gstride] : T(0);
        }
        __syncthreads();
        const int nr = (int)((nend - nc < KG_CHUNK) ? nend - nc : KG_CHUNK);
        if (c0) {
#pragma unroll 2
            for (int rr = 0; rr < nr; ++rr) {
                const pair_t u = *reinterpret_cast<const pair_t*>(U + (nc + rr) * ldu + m0);
                const T gg0 = Gs[rr][0], gg1 = Gs[rr][1];
                T sd[2][DT], s0 = T(0), s1 = T(0);
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    const T x = Xs[rr][d];
                    sd[0][d] = x - z[0][d];
                    sd[1][d] = x - z[1][d];
                    s0 += sd[0][d] * sd[0][d];
                    s1 += sd[1][d] * sd[1][d];
                }
                T f0, df0, f1, df1;
                kernel_profile_grad<KIND>(s0, f0, df0);
                kernel_profile_grad<KIND>(s1, f1, df1);
                const T v0 = gg0 * b0 - T(2) * gg1 * u[0];
                const T v1 = c1 ? gg0 * b1 - T(2) * gg1 * u[1] : T(0);
                va += (double)(v0 * f0) + (double)(v1 * f1);
                const T w0 = T(-2) * variance * v0 * df0, w1 = T(-2) * variance * v1 * df1;
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    const T p0 = w0 * sd[0][d], p1 = w1 * sd[1][d];
                    za[0][d] += (double)p0;  // times 1 / l_d below
                    za[1][d] += (double)p1;
                    la[d] += (double)(p0 * sd[0][d]) + (double)(p1 * sd[1][d]);
                }
            }
        }
    }
    // dZ partials: this thread's two columns
    const int64_t zb = ((int64_t)blockIdx.x * Mp) * DT;
#pragma unroll
    for (int d = 0; d < DT; ++d) {
        if (c0) zpart[zb + (int64_t)m0 * DT + d] = za[0][d] * (double)il[d];
        if (c1) zpart[zb + (int64_t)(m0 + 1) * DT + d] = za[1][d] * (double)il[d];
    }
    // lengthscale / variance partials: sum over the workgroup's threads
#pragma unroll
    for (int d = 0; d <= DT; ++d) {
        double v = d < DT ? la[d < DT ? d : 0] * (double)il[d < DT ? d : 0] : va;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if (lane == 0) red[w][d] = v;
    }
    __syncthreads();
    if (t <= DT) {
        double v = 0.0;
        for (int u = 0; u < NTHREADS / 64; ++u) v += red[u][t];
        const int64_t blk = (int64_t)blockIdx.x * gridDim.y + blockIdx.y;
        if (t < DT)
            lpart[blk * DT + t] = v;
        else
            vpart[blk] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// panel_kernel: one workgroup per 128-row panel of A; loops over latents p, output column tiles `it`, k-chunks.
//   C[n, i] = sum_{j in range(i)} A[n, j] * Tm[p][i, j]
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
struct PanelArgs {
    const T* A;      // [Np x Mp]
    const T* Tm;     // [P x Mp x Mp]
    T* C;            // STORE: [Np x Mp]
    const T* gamma;  // MOMENTS: [Mp x P]
    const T* Y;      // [N x P]
    T* mean;         // [N x P] or null
    T* var;          // [N x P] or null
    T* g0;           // [Np x P] or null
    T* g1;           // [Np x P] or null
    double* ve_partial;
    int32_t* nonpos_partial;
    double lik_param;
    int64_t N, Np;
    int64_t strideA;  // elements between the operands of consecutive latents (0: one operand shared by all latents)
    int64_t strideC, strideT;  // STORE, latent batch on grid.y
    int Mp, P, mode, lik;
    int kdiag_uniform;              // MOMENTS: kdiag[0] holds for every latent (one shared kernel; any P)
    double kdiag[TSVGP_MAX_BATCH];  // MOMENTS: k(x, x) = kernel variance of latent p (one kernel per latent: they differ)
};

// FUSE (MOMENTS, triangular modes): the mean GEMV rides on the column tile whose k-range covers every chunk of the
// row panel (the first one for UPPER, the last one for LOWER): each thread multiplies the eight A values it stages by gamma (kept in dynamic LDS, Mp elements) before
// storing them, which removes the separate sweep of the panel from HBM (1.0 of 18.3 ms at N = 1e6, M = 1024).
template <typename T, int MODE, int TRI, bool FUSE = false>
__global__ __launch_bounds__(NTHREADS, 2) void panel_kernel(PanelArgs<T> a) {
    static_assert(!FUSE || (MODE == MODE_MOMENTS && TRI != TSVGP_TRI_DENSE), "FUSE rides on the tile with the full k sweep");
    extern __shared__ __attribute__((aligned(16))) unsigned char panel_dyn_smem[];
    T* const gsm = reinterpret_cast<T*>(panel_dyn_smem);  // FUSE: gamma_p, Mp elements
    constexpr int RS = PanelK<T>::RS, KC = PanelK<T>::KC, H = PanelK<T>::H;  // KC shadows the site kernel's constant
    constexpr int CPT = TILE / KC;  // chunks per 128-wide k-tile
    __shared__ __attribute__((aligned(16))) T lds[2][2][TILE * RS];
    __shared__ double rowq[TILE];
    __shared__ double red[NTHREADS / 64];
    __shared__ int redi[NTHREADS / 64];
    typedef typename Mfma<T>::acc_t acc_t;

    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
#ifdef TSVGP_DIAG_PANEL  // diagnostic build (tools/diag_panel.py): when and where every workgroup ran
    const unsigned long long diag_t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long diag_c0 = __builtin_amdgcn_s_memtime();
    // shader cycles of this wave by phase: [0] full k-chunks (64 MFMAs per wave and barrier), [1] the masked chunks of the
    // diagonal k-tile, [2] a column tile's prologue (first fetch, staging, barrier), [3] its epilogue (square-sum / store)
    unsigned long long diag_ph[4] = {0, 0, 0, 0};
    unsigned long long diag_pre = 0, diag_post0 = 0;  // cycles before the first column tile / stamp at the end of the last one
#define TSVGP_PHASE(i_, t_) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); diag_ph[i_] += now_ - (t_); (t_) = now_; }
#else
#define TSVGP_PHASE(i_, t_)
#endif
#ifdef TSVGP_EXP_SLOTPRIO  // experiment: the second resident workgroup of a CU (second dispatch half-round) at wave priority 1
    if ((blockIdx.x >> 8) & 1) __builtin_amdgcn_s_setprio(TSVGP_EXP_SLOTPRIO);
#endif
#ifdef TSVGP_EXP_STAGGER  // experiment: delay the second resident workgroup of a CU by a fraction of a chunk
    if ((blockIdx.x >> 8) & 1) {
        for (int i = 0; i < TSVGP_EXP_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    const int srow = t >> 1, skh = t & 1;  // staging role: row of the tile, which half of the k-chunk
    const int64_t n0 = (int64_t)blockIdx.x * TILE;
    const int Mp = a.Mp;
    const int ntile = Mp / TILE, nchunk = Mp / KC;

    T* const lds_wr = &lds[0][0][srow * RS + skh * H];
    constexpr int BUF_STRIDE = 2 * TILE * RS, OP_STRIDE = TILE * RS;
    double ve_acc = 0.0;
    int nonpos = 0;
    // STORE: one latent per grid.y slice (operands A, Tm and the output C each advance by their latent stride);
    // MOMENTS: the workgroup takes its row panel through all P latents in turn (their outputs interleave in [n * P + p])
    const int pb = (MODE == MODE_STORE) ? (int)blockIdx.y : 0;

    for (int p = 0; p < a.P; ++p) {
        const T* Tp = (MODE == MODE_STORE) ? a.Tm + (size_t)pb * a.strideT : a.Tm + (size_t)p * Mp * Mp;
        // operand addresses as a wave-uniform base (scalar registers) + one 32-bit per-thread offset shared by A and Tm (a row
        // panel and a row tile of Tm are both < 4 GB): no 64-bit per-thread pointers, no 64-bit VALU adds per chunk
        const T* Abase = a.A + (size_t)(MODE == MODE_STORE ? pb : p) * a.strideA + n0 * (int64_t)Mp;
        const unsigned roff = (unsigned)((srow * Mp + skh * H) * sizeof(T));  // in BYTES: stays 32-bit after scaling
#define TSVGP_AT(base_, c_) reinterpret_cast<const T*>(reinterpret_cast<const char*>(base_) + (roff + (unsigned)((c_) * KC * sizeof(T))))
#define TSVGP_AROW(c_) TSVGP_AT(Abase, c_)
        // Row sums of squares: after every column tile the 8 per-register partials (2 row blocks x 4 registers) are
        // summed over the 16 lanes that share (lane>>4) and lane lr keeps the one with index lr & 7.
        double rs_mine = 0.0;
        T mpart = T(0);
        // Registers of the global prefetch (DEPTH chunks in flight, see below).  They live outside the tile body so that
        // the FIRST chunk of the next column tile can be requested before this tile's epilogue (TSVGP_XTILE): its load
        // latency then hides behind the square-sum / store of the finished tile instead of opening the next one.
        constexpr int DEPTH = PanelK<T>::DEPTH;
        T ra[DEPTH][H], rb[DEPTH][H];
        bool pre = false;  // ra[0] / rb[0] already hold the first chunk of the tile about to start
        if constexpr (FUSE) {
#if TSVGP_XTILE >= 2
            // the first tile (it = 0 for both triangles) starts at chunk 0: its loads fly while gamma is staged
            load_run<T, H>(ra[0], TSVGP_AROW(0));
            load_run<T, H>(rb[0], TSVGP_AT(Tp, 0));
            pre = true;
#endif
            for (int j = t; j < Mp; j += NTHREADS) gsm[j] = a.gamma[(size_t)j * a.P + p];
            __syncthreads();
        } else if constexpr (MODE == MODE_MOMENTS) {
            // Mean GEMV phase: mean[n] = sum_j A[n, j] * gamma[j, p].  A memory/VALU-only sweep of this workgroup's row
            // panel with no accumulators live (it runs beside the partner workgroup's MFMAs on the same CU); gamma_p
            // is staged in LDS (reusing the staging buffers) and read as a two-address broadcast.
            T* gs = &lds[0][0][0];
            for (int j = t; j < Mp; j += NTHREADS) gs[j] = a.gamma[(size_t)j * a.P + p];
            __syncthreads();
            const T* gk = gs + skh * H;
#ifndef TSVGP_EXP_NOGEMV  // ablation switch (tools/exp_moments.py)
#pragma unroll 4
            for (int c = 0; c < nchunk; ++c) {
                T ra[H];
                load_run<T, H>(ra, TSVGP_AROW(c));
#pragma unroll
                for (int q = 0; q < H; ++q) mpart += ra[q] * gk[c * KC + q];
            }
#endif
            __syncthreads();
        }

        auto tile_body = [&](const int it, auto first_tag, const int it_next) {
            constexpr bool FIRST = decltype(first_tag)::value;  // FUSE: the tile that also accumulates the mean
            const T* Tbase = Tp + (size_t)it * TILE * Mp;
#define TSVGP_TROW(c_) TSVGP_AT(Tbase, c_)

            acc_t acc[2][8];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int n = 0; n < 8; ++n) acc[s][n] = acc_t{0, 0, 0, 0};
#ifdef TSVGP_DIAG_PANEL
            unsigned long long ph_t = __builtin_amdgcn_s_memtime();
#endif

            // DEPTH chunks of global prefetch are held in registers (one set of H + H values per chunk in flight).
            // fp64 has room for one set only (128 accumulator registers); fp32 could take two (-DTSVGP_PANEL_DEPTH_F32=2),
            // but that measured slower (8.68 vs 8.40 ms at N = 1e6, M = 1024), so both types run with one.  A set is
            // chosen by the parity of the chunk index, so every loop below steps by two chunks (all chunk ranges are
            // even); the pairing itself is worth 3 % (fp64) to 6 % (fp32) over a one-chunk loop body.
            int buf = 0;
            // one pipeline step on chunk c_: prefetch chunk c_ + DEPTH to registers, MFMAs on chunk c_ (LDS buffer
            // `buf`), then stage chunk c_ + 1 (fetched one step earlier when DEPTH == 2) into the other buffer;
            // one barrier per chunk.  PAR = parity of c_ (compile time).
#define TSVGP_FETCH(SET, cnext)                           \
    {                                                     \
        load_run<T, H>(ra[SET], TSVGP_AROW(cnext));       \
        load_run<T, H>(rb[SET], TSVGP_TROW(cnext));       \
    }
#define TSVGP_STAGE(SET, cnext, b_)                                                        \
    {                                                                                      \
        if constexpr (FIRST) {                                                             \
            const T* gq = gsm + (cnext) * KC + skh * H;                                    \
            _Pragma("unroll") for (int q = 0; q < H; ++q) mpart += ra[SET][q] * gq[q];     \
        }                                                                                  \
        store_run<T, H>(lds_wr + (b_) * BUF_STRIDE, ra[SET]);                              \
        store_run<T, H>(lds_wr + (b_) * BUF_STRIDE + OP_STRIDE, rb[SET]);                  \
    }
#ifdef TSVGP_EXP_NOLOAD
#define TSVGP_EXP_HASNEXT(x) false
#else
#define TSVGP_EXP_HASNEXT(x) (x)
#endif
#define TSVGP_STEP2(MLO, MHI, PAR, c_)                                                          \
    {                                                                                           \
        const bool has_f = TSVGP_EXP_HASNEXT((c_) + DEPTH < c_end);                             \
        const bool has_s = TSVGP_EXP_HASNEXT((c_) + 1 < c_end);                                 \
        if (has_f) TSVGP_FETCH(DEPTH == 2 ? (PAR) : 0, (c_) + DEPTH)                            \
        mma_chunk_rowk<T, MLO, MHI>(acc, &lds[buf][0][0], &lds[buf][1][0], w, lane);            \
        if (has_s) TSVGP_STAGE(DEPTH == 2 ? ((PAR) ^ 1) : 0, (c_) + 1, buf ^ 1)                 \
        __syncthreads();                                                                        \
        buf ^= 1;                                                                               \
    }
#define TSVGP_STEP(NMASK, PAR, c_) TSVGP_STEP2(NMASK, NMASK, PAR, c_)
            const int cd = it * CPT;  // first chunk of the diagonal k-tile (even)
            const int c_first = (TRI == TSVGP_TRI_UPPER) ? cd : 0;
            const int c_end = (TRI == TSVGP_TRI_LOWER) ? cd + CPT : nchunk;
            if (!pre) TSVGP_FETCH(0, c_first)
            TSVGP_STAGE(0, c_first, 0)
            if constexpr (DEPTH == 2) {
                if (TSVGP_EXP_HASNEXT(c_first + 1 < c_end)) TSVGP_FETCH(1, c_first + 1)
            }
            __syncthreads();
            TSVGP_PHASE(2, ph_t)

            if constexpr (TRI == TSVGP_TRI_DENSE) {
                for (int c = 0; c < nchunk; c += 2) {
                    TSVGP_STEP(0xFF, 0, c)
                    TSVGP_STEP(0xFF, 1, c + 1)
                }
            } else if constexpr (TRI == TSVGP_TRI_LOWER) {
                // full k-tiles 0..it-1, then the diagonal k-tile: chunk cl only meets column blocks cb >= cl
                for (int c = 0; c < cd; c += 2) {
                    TSVGP_STEP(0xFF, 0, c)
                    TSVGP_STEP(0xFF, 1, c + 1)
                }
                TSVGP_PHASE(0, ph_t)
                if constexpr (KC == 16) {
                    TSVGP_STEP(0xFF, 0, cd)
                    TSVGP_STEP(0xFE, 1, cd + 1)
                    TSVGP_STEP(0xFC, 0, cd + 2)
                    TSVGP_STEP(0xF8, 1, cd + 3)
                    TSVGP_STEP(0xF0, 0, cd + 4)
                    TSVGP_STEP(0xE0, 1, cd + 5)
                    TSVGP_STEP(0xC0, 0, cd + 6)
                    TSVGP_STEP(0x80, 1, cd + 7)
                } else {  // 32-wide chunks: two 16-wide k-blocks per chunk
                    TSVGP_STEP2(0xFF, 0xFE, 0, cd)
                    TSVGP_STEP2(0xFC, 0xF8, 1, cd + 1)
                    TSVGP_STEP2(0xF0, 0xE0, 0, cd + 2)
                    TSVGP_STEP2(0xC0, 0x80, 1, cd + 3)
                }
                TSVGP_PHASE(1, ph_t)
            } else {
                // the diagonal k-tile first: chunk cl only meets column blocks cb <= cl; then full k-tiles it+1..
                if constexpr (KC == 16) {
                    TSVGP_STEP(0x01, 0, cd)
                    TSVGP_STEP(0x03, 1, cd + 1)
                    TSVGP_STEP(0x07, 0, cd + 2)
                    TSVGP_STEP(0x0F, 1, cd + 3)
                    TSVGP_STEP(0x1F, 0, cd + 4)
                    TSVGP_STEP(0x3F, 1, cd + 5)
                    TSVGP_STEP(0x7F, 0, cd + 6)
                    TSVGP_STEP(0xFF, 1, cd + 7)
                } else {
                    TSVGP_STEP2(0x01, 0x03, 0, cd)
                    TSVGP_STEP2(0x07, 0x0F, 1, cd + 1)
                    TSVGP_STEP2(0x1F, 0x3F, 0, cd + 2)
                    TSVGP_STEP2(0x7F, 0xFF, 1, cd + 3)
                }
                TSVGP_PHASE(1, ph_t)
                for (int c = cd + CPT; c < nchunk; c += 2) {
                    TSVGP_STEP(0xFF, 0, c)
                    TSVGP_STEP(0xFF, 1, c + 1)
                }
                TSVGP_PHASE(0, ph_t)
            }
#undef TSVGP_STEP
#undef TSVGP_STEP2
#undef TSVGP_STAGE
#undef TSVGP_FETCH
#undef TSVGP_TROW
            pre = false;
#if TSVGP_XTILE
            if (DEPTH == 1 && it_next >= 0) {  // request the next tile's first chunk; the epilogue below covers its latency
                const int cn = (TRI == TSVGP_TRI_UPPER) ? it_next * CPT : 0;
                load_run<T, H>(ra[0], TSVGP_AROW(cn));
                load_run<T, H>(rb[0], TSVGP_AT(Tp + (size_t)it_next * TILE * Mp, cn));
                pre = true;
            }
#endif

            if constexpr (MODE == MODE_STORE) {
                T* Cb = a.C + (size_t)pb * a.strideC + n0 * (int64_t)Mp + it * TILE + (lane & 15);
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        T* Cr = Cb + (int64_t)(row_block(w, s) * 16 + Mfma<T>::row(lane, r)) * Mp;
#pragma unroll
                        for (int n = 0; n < 8; ++n) Cr[n * 16] = acc[s][n][r];
                    }
            } else {
                double keep = 0.0;
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        double q = 0.0;
#pragma unroll
                        for (int n = 0; n < 8; ++n) {
                            const double v = (double)acc[s][n][r];
                            q += v * v;
                        }
                        q += __shfl_xor(q, 1);
                        q += __shfl_xor(q, 2);
                        q += __shfl_xor(q, 4);
                        q += __shfl_xor(q, 8);
                        keep = ((lane & 7) == s * 4 + r) ? q : keep;
                    }
                rs_mine += keep;
            }
            TSVGP_PHASE(3, ph_t)
        };  // tile_body
        const auto next_of = [ntile](int it) { return it + 1 < ntile ? it + 1 : -1; };
#ifdef TSVGP_DIAG_PANEL
        if (p == 0) diag_pre = __builtin_amdgcn_s_memtime() - diag_c0;
#endif
        if constexpr (FUSE && TRI == TSVGP_TRI_UPPER) {  // upper triangle: the FIRST column tile sweeps every k-chunk
            tile_body(0, std::true_type{}, next_of(0));
            for (int it = 1; it < ntile; ++it) tile_body(it, std::false_type{}, next_of(it));
        } else if constexpr (FUSE) {  // lower triangle: the LAST one does
            for (int it = 0; it + 1 < ntile; ++it) tile_body(it, std::false_type{}, next_of(it));
            tile_body(ntile - 1, std::true_type{}, -1);
        } else {
            for (int it = 0; it < ntile; ++it) tile_body(it, std::false_type{}, next_of(it));
        }

#ifdef TSVGP_DIAG_PANEL
        diag_post0 = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (MODE == MODE_MOMENTS) {
            // lane (lr < 8, lane>>4) holds the complete sum of row  row_block(w, lr>>2)*16 + rowmap(lane, lr&3)
            if ((lane & 15) < 8) {
                const int lr = lane & 15;
                rowq[row_block(w, lr >> 2) * 16 + Mfma<T>::row(lane, lr & 3)] = rs_mine;
            }
            mpart += __shfl_xor(mpart, 1);
            __syncthreads();
            {
                const int64_t n = n0 + srow;
                const bool live = n < a.N;
                const double q = rowq[srow];
                const double mu = (double)mpart;
                const double v = a.kdiag[a.kdiag_uniform ? 0 : p] - q;
                double g0 = 0.0, g1 = 0.0, ve = 0.0;
                if ((a.lik & 0xFF) == TSVGP_LIK_BERNOULLI) {
                    // the quadrature is the long pole of this epilogue (20 erf / exp / log in fp64 per row): the two
                    // threads that staged a row take five node pairs each and add up
                    double a0, a1, av;
                    const double sd = sqrt(live ? v : 1.0);
                    bern_sums_t<T>(live ? mu : 0.0, sd, live && (double)a.Y[n * a.P + p] == 1.0, skh * 5, skh * 5 + 5, a0, a1, av);
                    a0 += __shfl_xor(a0, 1);
                    a1 += __shfl_xor(a1, 1);
                    av += __shfl_xor(av, 1);
                    g0 = a0;
                    g1 = a1 / (2.0 * sd);
                    if (!(a.lik & TSVGP_LIK_NOCROP)) g1 = fmin(g1, -1e-8);  // reference tsvgp.py:262-263
                    ve = av;
                } else if (a.lik != TSVGP_LIK_NONE && live) {
                    lik_eval(a.lik, a.lik_param, mu, v, (double)a.Y[n * a.P + p], g0, g1, ve);
                }
                if (skh == 0) {
                    if (live) {
                        if (!(v > 0.0)) nonpos += 1;
#ifndef TSVGP_DIAG_PANEL
                        if (a.mean) a.mean[n * a.P + p] = (T)mu;
#endif
                        if (a.var) a.var[n * a.P + p] = (T)v;
                        ve_acc += ve;
                    }
                    if (a.lik != TSVGP_LIK_NONE) {
                        a.g0[n * a.P + p] = (T)(live ? g0 : 0.0);  // rows >= N: zeros (the padding contract of site_accum)
                        a.g1[n * a.P + p] = (T)(live ? g1 : 0.0);
                    }
                }
            }
            __syncthreads();  // rowq reused by the next latent
        }
    }  // p

    if constexpr (MODE == MODE_MOMENTS) {
        double s = ve_acc;
        int c = nonpos;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s += __shfl_xor(s, o);
            c += __shfl_xor(c, o);
        }
        if (lane == 0) {
            red[w] = s;
            redi[w] = c;
        }
        __syncthreads();
        if (t == 0) {
            if (a.ve_partial) a.ve_partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
            if (a.nonpos_partial) a.nonpos_partial[blockIdx.x] = redi[0] + redi[1] + redi[2] + redi[3];
        }
    }
#ifdef TSVGP_DIAG_PANEL
    // Stamps go to a buffer of their own, passed in place of `mean` (which NO kernel of this build writes: every store
    // through `mean` is compiled out under TSVGP_DIAG_PANEL).  The buffer describes itself: word 0 holds the number of
    // 8-word slots that follow the 8-word header, and a workgroup whose index is not below it writes nothing.  (Round 2's
    // stamp write was unchecked, and an experiment's finishing kernel stored its N means through the same pointer: a
    // memory access fault, profiles/r02_moments_split_panel_experiment.txt.)
    if (t == 0 && a.mean) {
        unsigned long long* const hdr = reinterpret_cast<unsigned long long*>(a.mean);
        const unsigned long long slots = hdr[0];
        const unsigned long long slot = (unsigned long long)blockIdx.x + (unsigned long long)blockIdx.y * gridDim.x;
        if (slot < slots) {
            unsigned long long* dbg = hdr + 8 + slot * 8;
            dbg[0] = diag_t0;
            dbg[1] = __builtin_amdgcn_s_memrealtime();
            dbg[2] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) |          // HW_REG_HW_ID
                     ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) << 32);  // HW_REG_XCC_ID
            dbg[3] = __builtin_amdgcn_s_memtime() - diag_c0;  // shader cycles of this workgroup
            for (int i = 0; i < 4; ++i) dbg[4 + i] = diag_ph[i];  // wave 0's cycles by phase
            hdr[8 + (slots + slot) * 8 + 0] = diag_pre;  // second bank of slots: before the first tile, after the last
            hdr[8 + (slots + slot) * 8 + 1] = __builtin_amdgcn_s_memtime() - diag_post0;
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// panel1_kernel (round 3): the upper-form products of panel_kernel -- the moments (MODE_MOMENTS, mean fused) and the stored
// product of tsvgp_trmm (MODE_STORE) -- as ONE 256-thread workgroup per CU with a hand-laid instruction stream, for both
// types.  fp64 moments: 0.90 of the MFMA peak where panel_kernel holds 0.83 on the same box
// (profiles/r03_moments_lab_notes.txt; tools/panel_lab.hip is the bench it was developed in); alone at N = 1e6, M = 1024
// (tools/kbench.py, old -> new): fp64 moments 17.4 -> 15.2 ms, fp64 trmm 18.3 -> 16.2, fp32 moments 8.28 -> 8.04, fp32 trmm
// 8.6 -> 8.0.  TRI selects the triangle: the upper form (tile it: chunks from its diagonal k-tile on) or the lower form (tile it:
// the full k-tiles in front of the diagonal one, then that one, whose chunk j meets the column blocks from j on; projected route,
// t_SVGP_white's whitened single-product variance): fp64 lower trmm 17.3 -> 16.2 ms, lower moments 17.0 -> 16.0.  The dense
// products (M-step) stay on panel_kernel.
//   * one wave per SIMD (__launch_bounds__(256, 1)): 512 registers per wave, so two fragment register sets, the accumulators
//     and everything else live without a single scratch access (panel_kernel is pinned at 256 by its partner workgroup);
//   * the chunk stream of a row panel runs through all column tiles without draining: every MFMA is followed by at most one
//     other instruction -- a fragment read of the NEXT k-step, an LDS-DMA issue, a piece of the mean -- and
//     __builtin_amdgcn_sched_barrier(0) pins that order, so a wave keeps the matrix pipe busy on its own:
//         k-step 0: MFMAs on set X | reads of k-step 1 -> set Y     (+ the mean's LDS reads in column tile 0)
//         k-step 1: MFMAs on set Y | reads of k-step 2 -> set X     (+ the mean's FMAs)
//         k-step 2: MFMAs on set X | reads of k-step 3 -> set Y, in the first slots
//         s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier     (next chunk landed for every wave; this chunk's buffer is free)
//         k-step 3: MFMAs on set Y | 8 LDS-DMA issues for chunk c + 2 into this chunk's buffer, reads of (c + 1, 0) -> X
//   * operands go global -> LDS by LDS-DMA (global_load_lds_dwordx4 from inline asm: no staging registers, no ds_write,
//     and no compiler-inserted vmcnt waits in front of fragment reads of other buffers).  LDS images are
//     [128 rows][8 units of 16 B], unpadded (a DMA instruction writes 1 KiB lane-linear); the unit index is XOR-ed with
//     f(row) = (row & 7) ^ ((row >> 3) & 1) on the per-lane global source address and on the reads: conflict free for
//     ds_read_b64 over each half wave;
//   * the T fragments of a set are read first and the two A fragments -- operands of the step's LAST MFMAs -- last, and a set's
//     registers stay occupied to the end of its step.  Round 3 adopted this order believing that a ds_read landing in an A/B
//     operand register of a v_mfma_f64_16x16x4_f64 issued just before it had corrupted results.  Round 4 measured it
//     (tools/hazard_probe.hip; this kernel built with -DTSVGP_HAZARD_AFIRST, i.e. the suspect order, 100-190 such sites at
//     distance one per instantiation: every parity and repeatability test passes -- profiles/r04_hazard_probe.txt): there is
//     no such window, the operands are read when the MFMA issues.  The order stays because it costs nothing.  What the stream
//     really depends on -- the wait state between the write of M0 and the LDS-DMA, five wait states between v_readfirstlane and
//     a memory instruction using that SGPR as its base, no scratch access among the MFMAs -- nothing pads inside asm volatile,
//     so tests/test_isa_lint.py checks it on the compiler's assembly (tools/isa_hazards.py).
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void_t;
constexpr int P1_OPS = TILE * 16;    // 8-byte fragment units per operand image of a chunk (16 KB)
constexpr int P1_BUFS = 2 * P1_OPS;  // fragment units per chunk buffer (A image + T image, 32 KB)
constexpr size_t P1_GAMMA_LDS_MAX = 64 * 1024;  // dynamic LDS for gamma_p beside the 64 KB ring: Mp <= 8192 (fp32: 16384)
constexpr size_t P1W_GAMMA_LDS_MAX = 32 * 1024;  // two tiles per pass: beside the 96 KB ring (Mp <= 4096 in fp64)
// Upper form, one tile per pass: the chunks of a tile's DIAGONAL k-tile in the order 8, 1, 7, 2, 6, 3, 5, 4 column blocks (fp32: 8, 2,
// 6, 4) instead of 1, 2, ... 8 (round 4).  A chunk of c + 2 is requested while chunk c runs and must have landed when chunk c + 1
// ends: behind two SHORT chunks (1 and 2 column blocks: 8 + 16 MFMAs per wave, ~0.7 us) it has not -- the ascending order stalled
// on chunks 2, 3, 4 of every diagonal k-tile, ~2.5 us per column tile (phase accounting of round 3: the diagonal chunks at 0.81 of
// their MFMA time) -- while a long and a short chunk together last as long as a full one.  All of a tile's accumulators are then
// first touched by its first chunk (srcC = 0 folds into those MFMAs).  Measured (profiles/r04_moments_epilogue_ablation.txt,
// moments alone at N = 1e6): fp64 M = 512 4.20 -> 4.10 ms (-2.5 %), fp64 M = 1024 unchanged (15.18 ms), fp32 (four diagonal chunks of
// 2, 4, 6, 8 column blocks) 7.71 -> 7.81 ms SLOWER -- so fp64 only.  -DTSVGP_DIAG_ORDER=0: the ascending order (A/B builds).
#ifndef TSVGP_DIAG_ORDER
#define TSVGP_DIAG_ORDER 1
#endif
#ifndef TSVGP_MOMENTS_WIDE  // (-DTSVGP_MOMENTS_WIDE=0: A/B builds keep one column tile per pass)
#define TSVGP_MOMENTS_WIDE 0
#endif
#ifndef TSVGP_MOMENTS_OLD_F32  // (-DTSVGP_MOMENTS_OLD_F32=1: A/B builds keep round 2's panel_kernel for fp32)
#define TSVGP_MOMENTS_OLD_F32 0
#endif

// Sum over the 16 lanes of a DPP row (lanes 16 g .. 16 g + 15), left in every lane of the row: four rotate-and-add steps through
// v_mov_b32_dpp row_ror (two per double), no LDS crossbar and no s_waitcnt -- the xor butterfly of __shfl_xor compiles to
// ds_bpermute_b32 pairs, each waited for: 64 of them and 32 waits per column-tile epilogue of panel1_kernel (round 4: that
// epilogue is 1.8 % of the fp64 moments kernel, 3.7 % of the fp32 one -- profiles/r04_moments_epilogue_ablation.txt).  All lanes of
// a row end with bit-identical sums (every step adds the same two partial sums in either order).
template <int CTRL>
__device__ __forceinline__ double dpp_row_mov(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);  // (every lane of a rotation has a source:
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);  //  no "old" value to initialise)
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double q) {
    q += dpp_row_mov<0x128>(q);  // row_ror:8
    q += dpp_row_mov<0x124>(q);  // row_ror:4
    q += dpp_row_mov<0x122>(q);  // row_ror:2
    q += dpp_row_mov<0x121>(q);  // row_ror:1
    return q;
}

template <int I, int N, class F>
__device__ __forceinline__ void cfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        cfor<I + 1, N>(f);
    }
}
constexpr int popc8(int m) { return m ? (m & 1) + popc8(m >> 1) : 0; }
// The same stream serves both types: a k-chunk is 128 bytes per row (16 doubles / 32 floats), a lane's fragment read is 8
// bytes -- ONE fp64 operand, or the operands of TWO consecutive v_mfma_f32_16x16x4_f32 (half the cycles each: the same MFMA
// time per read) -- so "k-step" below is 4 columns in fp64 and 8 in fp32, and the masks, buffers and DMA pieces do not change.
template <typename T>
struct P1Types;
template <>
struct P1Types<double> {
    typedef double frag_t;
    typedef v2d unit_t;  // 16-byte unit
};
template <>
struct P1Types<float> {
    typedef v2f frag_t;
    typedef v4f unit_t;
};
template <typename F, int NT = 8>
struct Frag1 {
    F v[2 + NT];  // v[0], v[1]: A fragments of the wave's two row blocks; v[2 + n]: T fragment of column block n
};
#define TSVGP_AI __attribute__((always_inline))
#define TSVGP_IC(x) std::integral_constant<int, (x)>{}
#define TSVGP_BC(x) std::integral_constant<bool, (x)>{}
#define TSVGP_SB() __builtin_amdgcn_sched_barrier(0)

// W2: column tiles per pass (1, or 2 for the upper-form moments at an even number of tiles): with two, the A fragments of a
// k-step serve 32 MFMAs instead of 16, a barrier and a set of DMA issues come once per 128 MFMAs, and the row panel is swept
// 2.5 instead of 4.5 times (M = 1024); the accumulators are then all 256 AGPRs of the wave (fp64).
#ifdef TSVGP_DIAG_PANEL1  // diagnostic build (tools/diag_panel1.py): when and where every workgroup of panel1_kernel ran, and the
// shader cycles of its prologue, tile epilogues and tail.  8 words per workgroup behind a one-word slot count; null: no stamps.
__device__ unsigned long long* g_panel1_stamps = nullptr;
#endif
template <typename T, int MODE = MODE_MOMENTS, int TRI = TSVGP_TRI_UPPER, int W2 = 1>
__global__ __launch_bounds__(NTHREADS, 1) void panel1_kernel(PanelArgs<T> a) {
    static_assert(W2 == 1 || (W2 == 2 && MODE == MODE_MOMENTS && TRI == TSVGP_TRI_UPPER), "two tiles per pass: upper-form moments");
#ifdef TSVGP_DIAG_PANEL1
    const unsigned long long dg_t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long dg_c0 = __builtin_amdgcn_s_memtime();
    unsigned long long dg_pro = 0, dg_epi = 0, dg_loop_end = 0, dg_e0 = 0;
#endif
    constexpr int NB = 8 * W2;                // column blocks of a pass
    constexpr int BUFS = (1 + W2) * P1_OPS;   // fragment units per chunk buffer: the A image + W2 T images
    constexpr int NDMA = 4 + 4 * W2;          // DMA pieces per wave and chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char panel_dyn_smem[];
    typedef typename P1Types<T>::frag_t frag_t;
    typedef typename P1Types<T>::unit_t unit_t;
    typedef Frag1<frag_t, NB> Frag;
    T* const gsm = reinterpret_cast<T*>(panel_dyn_smem);  // gamma_p, Mp elements
    __shared__ __attribute__((aligned(1024))) frag_t lds[2 * BUFS];
    __shared__ double rowq[TILE];
    __shared__ double rowm[TILE];
    __shared__ double red[NTHREADS / 64];
    __shared__ int redi[NTHREADS / 64];
    constexpr int KC = 128 / (int)sizeof(T), UE = 16 / (int)sizeof(T);  // elements per chunk row / per 16-byte unit
    constexpr int CPT = TILE / KC;   // chunks of the diagonal k-tile: 8 (one more column block each) or 4 (two more each)
    constexpr int BPC = 8 / CPT;     // column blocks a diagonal chunk adds
    typedef typename Mfma<T>::acc_t acc_t;

    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    const int Mp = a.Mp;
    const int ntile = Mp / TILE, nchunk = Mp / KC;
    const int64_t n0 = (int64_t)blockIdx.x * TILE;
    const int srow = t >> 1, skh = t & 1;  // epilogue role: row of the panel, half of the Bernoulli quadrature

    // DMA role: wave w moves the 1-KiB pieces 4 q + w (q = 0..3) of each operand image = rows 32 q + 8 w .. + 7; lane L lands
    // at unit L & 7 of row (L >> 3) of the piece and fetches the global unit (L & 7) ^ f(row), f(row) = (L >> 3) ^ (w & 1)
    const int drow = lane >> 3;
    const int dlog = (lane & 7) ^ drow ^ (w & 1);
    const unsigned dvoff = (unsigned)(drow * Mp * sizeof(T) + 16 * dlog);
    const size_t grp = (size_t)32 * Mp * sizeof(T);
    const unsigned lds_base = (unsigned)(uintptr_t)(lds_void_t*)lds + (unsigned)(w * 1024);

    // fragment reads: row r of a 16-row block, 8-byte piece 4 ks + lk of the chunk row -> unit (2 ks + (lk >> 1)) ^ f(r), half
    // lk & 1 (fp64: element k = 4 ks + lk; fp32: elements 8 ks + 2 lk, + 1 -- any assignment of k to lanes that A and T share)
    const int lr = lane & 15, lk = lane >> 4;
    const int fr = (lr & 7) ^ (lr >> 3);
    int offa0[4], offa1[4], offb[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int o = lr * 16 + (((2 * ks + (lk >> 1)) ^ fr) << 1) + (lk & 1);
        offb[ks] = o + P1_OPS;
        offa0[ks] = o + w * 256;
        offa1[ks] = o + (7 - w) * 256;
    }
    // Two tiles per pass: the ring is 96 KB, beyond the 64 KB a ds_read's immediate offset reaches from one base.  Left to fold
    // the buffer offset into every read, the compiler materialises one address register per (k-step, fragment) of the second
    // buffer and spills; the second buffer therefore gets its own twelve bases, opaque to constant folding.
    int offa0_1[4], offa1_1[4], offb_1[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        offa0_1[ks] = offa0[ks] + BUFS;
        offa1_1[ks] = offa1[ks] + BUFS;
        offb_1[ks] = offb[ks] + BUFS;
        if constexpr (W2 == 2) {
            asm volatile("" : "+v"(offa0_1[ks]), "+v"(offa1_1[ks]), "+v"(offb_1[ks]));
        }
    }
    // mean (column tile 0): the thread reads the 16-byte unit t & 7 of rows 32 q + (t >> 3) of the landed A image; behind
    // that physical unit sits the logical unit glog (columns 2 glog, 2 glog + 1 of the chunk)
    const int glog = (t & 7) ^ ((t >> 3) & 7) ^ (w & 1);

    double ve_acc = 0.0;
    int nonpos = 0;

    // STORE (tsvgp_trmm): one latent per grid.y slice, the operands and the output advance by their latent strides, a.P = 1;
    // MOMENTS: the workgroup takes its row panel through all P latents in turn
    const int pb = (MODE == MODE_STORE) ? (int)blockIdx.y : 0;
    for (int p = 0; p < a.P; ++p) {
        const T* Tp = (MODE == MODE_STORE) ? a.Tm + (size_t)pb * a.strideT : a.Tm + (size_t)p * Mp * Mp;
        const char* Ab = reinterpret_cast<const char*>(a.A + (size_t)(MODE == MODE_STORE ? pb : p) * a.strideA +
                                                       (n0 + 8 * w) * (int64_t)Mp);
        if constexpr (MODE == MODE_MOMENTS) {
            __syncthreads();  // the previous latent's readers of gsm / rowq / rowm are done
            for (int j = t; j < Mp; j += NTHREADS) gsm[j] = a.gamma[(size_t)j * a.P + p];
        }

        struct Cursor {
            int it, c;
        } cf{0, 0};  // fetch position in the chunk stream of this panel: column tile, k-chunk (tile it: chunks 8 it .. nchunk - 1)
        // DMA addresses: one per-lane byte offset (VGPR) + wave-uniform 64-bit bases.  The bases are made scalar HERE, at the
        // start of a chunk (readfirstlane of both halves; the casts matter: readfirstlane returns int, and an int low half would
        // be sign-extended into the high one), long before the LDS-DMA instructions of k-step 3 read them: a scalar register
        // written by a vector instruction needs five wait states in front of a memory instruction that reads it, and nothing
        // inserts them inside an asm statement.
        auto uni64 = [](const void* ptr) TSVGP_AI {
            const uint64_t v = (uint64_t)(uintptr_t)ptr;
            return ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
                   (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
        };
        const uint64_t ab_u = uni64(Ab);
        const unsigned lds_u = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_base);
        unsigned dma_vo = 0;
        uint64_t tb_u = 0;
        constexpr bool REORDER = TSVGP_DIAG_ORDER && W2 == 1 && TRI == TSVGP_TRI_UPPER && sizeof(T) == 8;
        // REORDER: cu.c counts POSITIONS in the tile's stream; position q < CPT of the diagonal k-tile is its chunk
        // (q even ? CPT - 1 - q / 2 : q / 2), i.e. 7, 0, 6, 1, 5, 2, 4, 3
        auto kchunk = [&](const Cursor cu) TSVGP_AI {
            if constexpr (!REORDER) return cu.c;
            const int q = cu.c - cu.it * CPT;
            return q < CPT ? cu.it * CPT + ((q & 1) ? (q >> 1) : CPT - 1 - (q >> 1)) : cu.c;
        };
        auto dma_setup = [&](const Cursor cu) TSVGP_AI {
            dma_vo = dvoff + (unsigned)(kchunk(cu) * KC * sizeof(T));
            tb_u = uni64(Tp + ((size_t)cu.it * TILE + 8 * w) * Mp);
        };
        // piece I of a chunk: I < 8: I even -> A piece I / 2, I odd -> piece I / 2 of the first T image; I >= 8: piece I - 8 of the second
        auto dma_piece = [&](auto i_tag, const int buf) TSVGP_AI {
            constexpr int I = decltype(i_tag)::value, q = I < 8 ? (I >> 1) : I - 8;
            constexpr int IMG = I < 8 ? (I & 1) : 2;  // 0: A, 1: T of the pass's first tile, 2: T of its second tile
            const unsigned la = lds_u + (unsigned)((buf * BUFS + q * 512 + IMG * P1_OPS) * sizeof(frag_t));
            const uint64_t g = (IMG == 0 ? ab_u : tb_u + (IMG == 2 ? 4 * grp : 0)) + q * grp;
            const unsigned vo_ = dma_vo;  // (an asm operand alone does not capture a variable in a generic lambda)
            // one wait state between the write of M0 and the LDS-DMA that reads it.  (Fetching only the T pieces a diagonal chunk
            // reads -- the others issued with EXEC = 0 -- was 1.7 % SLOWER at M = 1024 and 512: profiles/r03_moments_lab_notes.txt)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(la), "v"(vo_), "s"(g) : "memory");
        };
        auto advance = [&](Cursor& cu) TSVGP_AI {  // saturates at the last chunk (a harmless re-fetch into a dead buffer)
            // upper form: tile it takes the chunks CPT it .. nchunk - 1; lower form: 0 .. CPT (it + 1) - 1 (its diagonal k-tile last)
            const int c_end = (TRI == TSVGP_TRI_UPPER) ? nchunk : (cu.it + 1) * CPT;
            if (cu.c + 1 == c_end) {
                if (cu.it + W2 < ntile) {
                    cu.it += W2;
                    cu.c = (TRI == TSVGP_TRI_UPPER) ? cu.it * CPT : 0;
                }
            } else {
                ++cu.c;
            }
        };

        acc_t acc[2][NB];
        Frag fx, fy;
        double rs_mine = 0.0;
        T mpart[4] = {0, 0, 0, 0};
        unit_t gx[4], gg;  // the mean's operands in flight (column tile 0)

        auto rd1 = [&](Frag& f, auto e_tag, auto ks_tag, auto buf_tag) TSVGP_AI {
            constexpr int E = decltype(e_tag)::value, KS = decltype(ks_tag)::value, BI = decltype(buf_tag)::value;
            if constexpr (W2 == 2 && BI == 1) {
                if constexpr (E == 0) f.v[0] = lds[offa0_1[KS]];
                else if constexpr (E == 1) f.v[1] = lds[offa1_1[KS]];
                else f.v[E] = lds[offb_1[KS] + ((E - 2) >> 3) * P1_OPS + ((E - 2) & 7) * 256];
            } else {
                constexpr int boff = BI * BUFS;
                if constexpr (E == 0) f.v[0] = lds[offa0[KS] + boff];
                else if constexpr (E == 1) f.v[1] = lds[offa1[KS] + boff];
                else f.v[E] = lds[offb[KS] + boff + ((E - 2) >> 3) * P1_OPS + ((E - 2) & 7) * 256];
            }
        };
        // slot S of a k-step's reads -> element of the set: T fragments first, the two A fragments last (see the header)
        // (a chunk of mask popcount MM meets the column blocks N0 .. N0 + MM - 1: N0 = 0 in the upper form, 8 - MM in the lower)
        auto rds = [&](Frag& f, auto slot_tag, auto m_tag, auto ks_tag, auto boff) TSVGP_AI {
            constexpr int S = decltype(slot_tag)::value, MM = decltype(m_tag)::value;
            constexpr int N0 = (TRI == TSVGP_TRI_UPPER) ? 0 : 8 - MM;
#ifdef TSVGP_HAZARD_AFIRST  // diagnostic build (tools/isa_hazards.py, profiles/r04_hazard_*): the order round 3 saw corrupt results with
            if constexpr (S < 2) rd1(f, TSVGP_IC(S), ks_tag, boff);
            else rd1(f, TSVGP_IC(2 + N0 + S - 2), ks_tag, boff);
#else
            if constexpr (S < MM) rd1(f, TSVGP_IC(2 + N0 + S), ks_tag, boff);
            else rd1(f, TSVGP_IC(S - MM), ks_tag, boff);
#endif
        };
        auto keep_set = [&](const Frag& f, auto m_tag) TSVGP_AI {  // the set's registers stay occupied up to this point
#ifdef TSVGP_HAZARD_AFIRST
            return;
#endif
            constexpr int MM = decltype(m_tag)::value, N0 = (TRI == TSVGP_TRI_UPPER) ? 0 : 8 - MM;
            cfor<0, 2 + MM>([&](auto e) TSVGP_AI {
                constexpr int E = decltype(e)::value;
                const frag_t x = f.v[E < 2 ? E : E + N0];
                asm volatile("" ::"v"(x));
            });
        };
        auto mf = [&](const Frag& f, auto i_tag, auto m_tag) TSVGP_AI {  // MFMA slot I of a k-step: column block N0 + I / 2, row block I % 2
            constexpr int I = decltype(i_tag)::value, sblk = I & 1;
            constexpr int n = ((TRI == TSVGP_TRI_UPPER) ? 0 : 8 - decltype(m_tag)::value) + (I >> 1);
            if constexpr (sizeof(T) == 8) {
                acc[sblk][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.v[sblk], f.v[2 + n], acc[sblk][n], 0, 0, 0);
            } else {  // the two halves of the 8-byte fragments: two MFMAs of half the cycles
                acc[sblk][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.v[sblk][0], f.v[2 + n][0], acc[sblk][n], 0, 0, 0);
                acc[sblk][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.v[sblk][1], f.v[2 + n][1], acc[sblk][n], 0, 0, 0);
            }
        };
        // One chunk of the stream.  M: its column-block mask (upper form: blocks 0 .. m - 1); MN: the mask of the NEXT chunk of
        // the stream (0: none); BUF: its LDS buffer (parity of the chunk index); GC: it belongs to column tile 0 (its A image
        // feeds the mean); c_this: its k-chunk index.  On entry set X holds the fragments of its k-step 0.
        auto chunk = [&](auto m_tag, auto mn_tag, auto buf_tag, auto gc_tag, const int c_this) TSVGP_AI {
            constexpr int M = decltype(m_tag)::value, MN = decltype(mn_tag)::value, BUF = decltype(buf_tag)::value;
            constexpr bool GC = decltype(gc_tag)::value;
            constexpr int m = popc8(M & 0xFF) + popc8(M >> 8), mn = popc8(MN & 0xFF) + popc8(MN >> 8), NM = 2 * m, NR = 2 + m, NRN = 2 + mn;
            constexpr int B0 = BUF * BUFS, B1 = (BUF ^ 1) * BUFS;
            if constexpr (MN != 0) {  // the DMA addresses of chunk c + 2: scalar work in front of the first MFMAs
                dma_setup(cf);
                advance(cf);
                TSVGP_SB();
            }
            constexpr int S0 = NR + (GC ? 5 : 0);
            cfor<0, (NM > S0 ? NM : S0)>([&](auto i) TSVGP_AI {
                constexpr int I = decltype(i)::value;
                if constexpr (I < NM) mf(fx, i, TSVGP_IC(m));
                if constexpr (I < NR) rds(fy, i, TSVGP_IC(m), TSVGP_IC(1), TSVGP_IC(BUF));
                else if constexpr (GC && I == NR) gg = *reinterpret_cast<const unit_t*>(gsm + c_this * KC + UE * glog);
                else if constexpr (GC && I > NR && I < NR + 5) gx[I - NR - 1] = *reinterpret_cast<const unit_t*>(lds + B0 + (I - NR - 1) * 512 + t * 2);
                TSVGP_SB();
            });
            keep_set(fx, TSVGP_IC(m));
            constexpr int S1 = NR + (GC ? 4 : 0);
            cfor<0, (NM > S1 ? NM : S1)>([&](auto i) TSVGP_AI {
                constexpr int I = decltype(i)::value;
                if constexpr (I < NM) mf(fy, i, TSVGP_IC(m));
                if constexpr (I < NR) rds(fx, i, TSVGP_IC(m), TSVGP_IC(2), TSVGP_IC(BUF));
                else if constexpr (GC && I < NR + 4) {
                    T dot = gx[I - NR][0] * gg[0];
#pragma unroll
                    for (int u = 1; u < UE; ++u) dot += gx[I - NR][u] * gg[u];
                    mpart[I - NR] += dot;
                }
                TSVGP_SB();
            });
            keep_set(fy, TSVGP_IC(m));
            cfor<0, (NM > NR ? NM : NR)>([&](auto i) TSVGP_AI {
                constexpr int I = decltype(i)::value;
                if constexpr (I < NM) mf(fx, i, TSVGP_IC(m));
                if constexpr (I < NR) rds(fy, i, TSVGP_IC(m), TSVGP_IC(3), TSVGP_IC(BUF));
                TSVGP_SB();
            });
            keep_set(fx, TSVGP_IC(m));
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            TSVGP_SB();
            if constexpr (MN != 0) {
                constexpr int S3 = NDMA + NRN;
                cfor<0, (NM > S3 ? NM : S3)>([&](auto i) TSVGP_AI {
                    constexpr int I = decltype(i)::value;
                    if constexpr (I < NM) mf(fy, i, TSVGP_IC(m));
                    if constexpr (I < NDMA) dma_piece(i, BUF);
                    else if constexpr (I < S3) rds(fx, TSVGP_IC(I - NDMA), TSVGP_IC(mn), TSVGP_IC(0), TSVGP_IC(BUF ^ 1));
                    TSVGP_SB();
                });
            } else {
                cfor<0, NM>([&](auto i) TSVGP_AI { mf(fy, i, TSVGP_IC(m)); });
            }
            keep_set(fy, TSVGP_IC(m));
        };

        // prologue: chunks (0, 0) and (0, 1) on their way into the two buffers, the first fragments read
        dma_setup(cf);
        cfor<0, NDMA>([&](auto i) TSVGP_AI { dma_piece(i, 0); });
        advance(cf);
        dma_setup(cf);
        cfor<0, NDMA>([&](auto i) TSVGP_AI { dma_piece(i, 1); });
        advance(cf);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();  // both chunks and gamma are in LDS for every wave
        // a0, a1, b0 (fp32: and b1) of chunk (0, 0); lower form: tile 0 starts with its diagonal k-tile, all eight column blocks
        cfor<0, ((TRI == TSVGP_TRI_UPPER && !REORDER) ? 2 + BPC : 10)>([&](auto i) TSVGP_AI { rd1(fx, i, TSVGP_IC(0), TSVGP_IC(0)); });
#ifdef TSVGP_DIAG_PANEL1
        if (p == 0) dg_pro = __builtin_amdgcn_s_memtime() - dg_c0;
#endif

        for (int it = 0; it < ntile; it += W2) {
            const int cd = it * CPT;
            const bool last_tile = it + W2 == ntile;
            // an accumulator is zeroed in front of the diagonal chunk that first touches its column block
#define TSVGP_ZACC(n_) { acc[0][n_] = acc_t{0, 0, 0, 0}; acc[1][n_] = acc_t{0, 0, 0, 0}; }
            auto diag = [&](auto gc) TSVGP_AI {
                if constexpr (REORDER) {
                    cfor<0, 8>([&](auto n) TSVGP_AI {
                        acc[0][decltype(n)::value] = acc_t{0, 0, 0, 0};
                        acc[1][decltype(n)::value] = acc_t{0, 0, 0, 0};
                    });
                    // position q: chunk j(q) of the diagonal k-tile, column blocks 0 .. BPC (j + 1) - 1
                    cfor<0, CPT - 1>([&](auto q_tag) TSVGP_AI {
                        constexpr int Q = decltype(q_tag)::value;
                        constexpr int J = (Q & 1) ? (Q >> 1) : CPT - 1 - (Q >> 1);
                        constexpr int JN = ((Q + 1) & 1) ? ((Q + 1) >> 1) : CPT - 1 - ((Q + 1) >> 1);
                        chunk(TSVGP_IC((1 << (BPC * (J + 1))) - 1), TSVGP_IC((1 << (BPC * (JN + 1))) - 1), TSVGP_IC(Q & 1), gc, cd + J);
                    });
                    constexpr int JL = ((CPT - 1) & 1) ? ((CPT - 1) >> 1) : CPT - 1 - ((CPT - 1) >> 1);  // the last position's chunk
                    if (!last_tile) chunk(TSVGP_IC((1 << (BPC * (JL + 1))) - 1), TSVGP_IC(0xFF), TSVGP_IC(1), gc, cd + JL);
                    else chunk(TSVGP_IC((1 << (BPC * (JL + 1))) - 1), TSVGP_IC(0), TSVGP_IC(1), gc, cd + JL);  // the end of the stream
                    return;
                }
                if constexpr (CPT == 8) {
                    TSVGP_ZACC(0) chunk(TSVGP_IC(0x01), TSVGP_IC(0x03), TSVGP_IC(0), gc, cd);
                    TSVGP_ZACC(1) chunk(TSVGP_IC(0x03), TSVGP_IC(0x07), TSVGP_IC(1), gc, cd + 1);
                    TSVGP_ZACC(2) chunk(TSVGP_IC(0x07), TSVGP_IC(0x0F), TSVGP_IC(0), gc, cd + 2);
                    TSVGP_ZACC(3) chunk(TSVGP_IC(0x0F), TSVGP_IC(0x1F), TSVGP_IC(1), gc, cd + 3);
                    TSVGP_ZACC(4) chunk(TSVGP_IC(0x1F), TSVGP_IC(0x3F), TSVGP_IC(0), gc, cd + 4);
                    TSVGP_ZACC(5) chunk(TSVGP_IC(0x3F), TSVGP_IC(0x7F), TSVGP_IC(1), gc, cd + 5);
                    TSVGP_ZACC(6) chunk(TSVGP_IC(0x7F), TSVGP_IC(0xFF), TSVGP_IC(0), gc, cd + 6);
                    TSVGP_ZACC(7)
                } else {  // 32-column chunks: two more column blocks each (the second one's first k-steps meet zeros of T)
                    TSVGP_ZACC(0) TSVGP_ZACC(1) chunk(TSVGP_IC(0x03), TSVGP_IC(0x0F), TSVGP_IC(0), gc, cd);
                    TSVGP_ZACC(2) TSVGP_ZACC(3) chunk(TSVGP_IC(0x0F), TSVGP_IC(0x3F), TSVGP_IC(1), gc, cd + 1);
                    TSVGP_ZACC(4) TSVGP_ZACC(5) chunk(TSVGP_IC(0x3F), TSVGP_IC(0xFF), TSVGP_IC(0), gc, cd + 2);
                    TSVGP_ZACC(6) TSVGP_ZACC(7)
                }
                if (!last_tile) chunk(TSVGP_IC(0xFF), TSVGP_IC(0xFF), TSVGP_IC(1), gc, cd + CPT - 1);
                else chunk(TSVGP_IC(0xFF), TSVGP_IC(0), TSVGP_IC(1), gc, cd + CPT - 1);  // the end of the stream
            };
            auto full = [&](auto gc) TSVGP_AI {  // the full k-tiles behind the diagonal one; the next column tile's first chunk follows
                const int c_last = nchunk - 1;
                for (int c = cd + CPT; c < c_last - 1; c += 2) {
                    chunk(TSVGP_IC(0xFF), TSVGP_IC(0xFF), TSVGP_IC(0), gc, c);
                    chunk(TSVGP_IC(0xFF), TSVGP_IC(0xFF), TSVGP_IC(1), gc, c + 1);
                }
                chunk(TSVGP_IC(0xFF), TSVGP_IC(0xFF), TSVGP_IC(0), gc, c_last - 1);
                chunk(TSVGP_IC(0xFF), TSVGP_IC(REORDER ? 0xFF : (1 << BPC) - 1), TSVGP_IC(1), gc, c_last);
            };
#undef TSVGP_ZACC
            // lower form: the full k-tiles 0 .. it - 1 first, then the diagonal one, whose chunk j meets the column blocks from
            // BPC j on; all accumulators start the tile at zero; the mean rides on the LAST tile (its k-range is the whole panel)
            auto lower_tile = [&](auto gc) TSVGP_AI {
                cfor<0, 8>([&](auto n) TSVGP_AI {
                    acc[0][decltype(n)::value] = acc_t{0, 0, 0, 0};
                    acc[1][decltype(n)::value] = acc_t{0, 0, 0, 0};
                });
                for (int c = 0; c < cd; c += 2) {
                    chunk(TSVGP_IC(0xFF), TSVGP_IC(0xFF), TSVGP_IC(0), gc, c);
                    chunk(TSVGP_IC(0xFF), TSVGP_IC(0xFF), TSVGP_IC(1), gc, c + 1);
                }
                if constexpr (CPT == 8) {
                    chunk(TSVGP_IC(0xFF), TSVGP_IC(0xFE), TSVGP_IC(0), gc, cd);
                    chunk(TSVGP_IC(0xFE), TSVGP_IC(0xFC), TSVGP_IC(1), gc, cd + 1);
                    chunk(TSVGP_IC(0xFC), TSVGP_IC(0xF8), TSVGP_IC(0), gc, cd + 2);
                    chunk(TSVGP_IC(0xF8), TSVGP_IC(0xF0), TSVGP_IC(1), gc, cd + 3);
                    chunk(TSVGP_IC(0xF0), TSVGP_IC(0xE0), TSVGP_IC(0), gc, cd + 4);
                    chunk(TSVGP_IC(0xE0), TSVGP_IC(0xC0), TSVGP_IC(1), gc, cd + 5);
                    chunk(TSVGP_IC(0xC0), TSVGP_IC(0x80), TSVGP_IC(0), gc, cd + 6);
                    if (!last_tile) chunk(TSVGP_IC(0x80), TSVGP_IC(0xFF), TSVGP_IC(1), gc, cd + 7);
                    else chunk(TSVGP_IC(0x80), TSVGP_IC(0), TSVGP_IC(1), gc, cd + 7);
                } else {
                    chunk(TSVGP_IC(0xFF), TSVGP_IC(0xFC), TSVGP_IC(0), gc, cd);
                    chunk(TSVGP_IC(0xFC), TSVGP_IC(0xF0), TSVGP_IC(1), gc, cd + 1);
                    chunk(TSVGP_IC(0xF0), TSVGP_IC(0xC0), TSVGP_IC(0), gc, cd + 2);
                    if (!last_tile) chunk(TSVGP_IC(0xC0), TSVGP_IC(0xFF), TSVGP_IC(1), gc, cd + 3);
                    else chunk(TSVGP_IC(0xC0), TSVGP_IC(0), TSVGP_IC(1), gc, cd + 3);
                }
            };
            // two tiles per pass (upper form): the diagonal k-tile of the first tile (its column blocks one by one), then the
            // diagonal k-tile of the second one under the full first tile, then the full k-tiles behind both
            auto wide_pass = [&](auto gc) TSVGP_AI {
                cfor<0, CPT>([&](auto j) TSVGP_AI {  // region A: chunk j adds the column blocks BPC j .. of the first tile
                    constexpr int J = decltype(j)::value;
                    constexpr int M_ = (1 << (BPC * (J + 1))) - 1;
                    constexpr int MN_ = J + 1 < CPT ? (1 << (BPC * (J + 2))) - 1 : (0xFF | (((1 << BPC) - 1) << 8));
                    cfor<0, BPC>([&](auto q) TSVGP_AI {
                        acc[0][BPC * J + decltype(q)::value] = acc_t{0, 0, 0, 0};
                        acc[1][BPC * J + decltype(q)::value] = acc_t{0, 0, 0, 0};
                    });
                    chunk(TSVGP_IC(M_), TSVGP_IC(MN_), TSVGP_IC(J & 1), gc, cd + J);
                });
                cfor<0, CPT>([&](auto j) TSVGP_AI {  // region B: the first tile in full, the second one block by block
                    constexpr int J = decltype(j)::value;
                    constexpr int M_ = 0xFF | (((1 << (BPC * (J + 1))) - 1) << 8);
                    constexpr int MN_ = J + 1 < CPT ? (0xFF | (((1 << (BPC * (J + 2))) - 1) << 8)) : 0xFFFF;
                    cfor<0, BPC>([&](auto q) TSVGP_AI {
                        acc[0][8 + BPC * J + decltype(q)::value] = acc_t{0, 0, 0, 0};
                        acc[1][8 + BPC * J + decltype(q)::value] = acc_t{0, 0, 0, 0};
                    });
                    // (no separate variant for the end of the stream: the last chunk of the last pass fetches the saturated cursor
                    // into a dead buffer and pre-reads fragments nobody uses -- one straight line instead of a branch with 256
                    // live accumulators on both sides of its join)
                    chunk(TSVGP_IC(M_), TSVGP_IC(MN_), TSVGP_IC(J & 1), gc, cd + CPT + J);
                });
                if (!last_tile) {
                    const int c_last = nchunk - 1;
                    for (int c = cd + 2 * CPT; c < c_last - 1; c += 2) {
                        chunk(TSVGP_IC(0xFFFF), TSVGP_IC(0xFFFF), TSVGP_IC(0), gc, c);
                        chunk(TSVGP_IC(0xFFFF), TSVGP_IC(0xFFFF), TSVGP_IC(1), gc, c + 1);
                    }
                    chunk(TSVGP_IC(0xFFFF), TSVGP_IC(0xFFFF), TSVGP_IC(0), gc, c_last - 1);
                    chunk(TSVGP_IC(0xFFFF), TSVGP_IC((1 << BPC) - 1), TSVGP_IC(1), gc, c_last);
                }
            };
            if constexpr (W2 == 2) {
                if (it == 0) wide_pass(TSVGP_BC(true));
                else wide_pass(TSVGP_BC(false));
            } else if constexpr (TRI == TSVGP_TRI_LOWER) {
                if (MODE == MODE_MOMENTS && last_tile) {
                    if constexpr (MODE == MODE_MOMENTS) lower_tile(TSVGP_BC(true));
                } else {
                    lower_tile(TSVGP_BC(false));
                }
            } else if (MODE == MODE_MOMENTS && it == 0) {
                if constexpr (MODE == MODE_MOMENTS) {
                    diag(TSVGP_BC(true));
                    if (!last_tile) full(TSVGP_BC(true));
                }
            } else {
                diag(TSVGP_BC(false));
                if (!last_tile) full(TSVGP_BC(false));
            }
            if constexpr (MODE == MODE_STORE) {
                // the column tile goes out as panel_kernel writes it (the next tile's first chunks are already on their way; its
                // first barrier waits for these stores with them -- they share vmcnt)
                T* Cb = a.C + (size_t)pb * a.strideC + n0 * (int64_t)Mp + it * TILE + (lane & 15);
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        T* Cr = Cb + (int64_t)(row_block(w, s) * 16 + Mfma<T>::row(lane, r)) * Mp;
#pragma unroll
                        for (int n = 0; n < 8; ++n) Cr[n * 16] = acc[s][n][r];
                    }
                continue;
            }
#ifdef TSVGP_EXP_NOEPI  // ablation (profiles/r04_moments_epilogue_ablation.txt): the column tile's square-sum epilogue left out --
            // what a second accumulator set could at most hide.  The accumulators stay live (the MFMAs are not dead code); results wrong.
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int n = 0; n < NB; ++n) asm volatile("" ::"a"(acc[s][n]));
            rs_mine += (double)acc[0][0][0];
            continue;
#endif
            // the column tile is complete: squares of its entries, summed per row (as panel_kernel)
#ifdef TSVGP_DIAG_PANEL1
            dg_e0 = __builtin_amdgcn_s_memtime();
#endif
            double keep = 0.0;
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double q = 0.0;
                    if constexpr (sizeof(T) == 4) {
                        // fp32 N-arrays: the tile's entries ARE fp32 accumulators (relative error ~1e-6 from the fp32 MFMAs); the
                        // squares of a row's NB entries are summed in fp32 (1e-7) and converted once -- it was a conversion and an
                        // fp64 FMA per entry: 64 v_cvt_f64_f32 per lane and tile in an epilogue that is 3.7 % of this kernel
                        float q32 = 0.0f;
#pragma unroll
                        for (int n = 0; n < NB; ++n) q32 = fmaf(acc[s][n][r], acc[s][n][r], q32);
                        q = (double)q32;
                    } else {
#pragma unroll
                        for (int n = 0; n < NB; ++n) {
                            const double v = (double)acc[s][n][r];
                            q += v * v;
                        }
                    }
#ifdef TSVGP_EPI_SHFL  // (A/B builds: round 3's butterfly through the LDS crossbar)
                    q += __shfl_xor(q, 1);
                    q += __shfl_xor(q, 2);
                    q += __shfl_xor(q, 4);
                    q += __shfl_xor(q, 8);
#else
                    q = row16_sum(q);
#endif
                    keep = ((lane & 7) == s * 4 + r) ? q : keep;
                }
            rs_mine += keep;
#ifdef TSVGP_DIAG_PANEL1
            asm volatile("" : "+v"(rs_mine));
            dg_epi += __builtin_amdgcn_s_memtime() - dg_e0;
#endif
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the saturated re-fetches of the last two chunks have landed
#ifdef TSVGP_DIAG_PANEL1
        dg_loop_end = __builtin_amdgcn_s_memtime();
#endif
        if constexpr (MODE == MODE_STORE) return;

        // row sums and means to LDS, then the likelihood map of panel_kernel's epilogue (two threads per row)
        if ((lane & 15) < 8) {
            const int l8 = lane & 15;
            rowq[row_block(w, l8 >> 2) * 16 + Mfma<T>::row(lane, l8 & 3)] = rs_mine;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mpart[q] += __shfl_xor(mpart[q], 1);
            mpart[q] += __shfl_xor(mpart[q], 2);
            mpart[q] += __shfl_xor(mpart[q], 4);
        }
        if ((t & 7) == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) rowm[q * 32 + (t >> 3)] = (double)mpart[q];
        }
        __syncthreads();
        {
            const int64_t n = n0 + srow;
            const bool live = n < a.N;
            const double q = rowq[srow];
            const double mu = rowm[srow];
            const double v = a.kdiag[a.kdiag_uniform ? 0 : p] - q;
            double g0 = 0.0, g1 = 0.0, ve = 0.0;
            if ((a.lik & 0xFF) == TSVGP_LIK_BERNOULLI) {
                double a0, a1, av;
                const double sd = sqrt(live ? v : 1.0);
                bern_sums_t<T>(live ? mu : 0.0, sd, live && (double)a.Y[n * a.P + p] == 1.0, skh * 5, skh * 5 + 5, a0, a1, av);
                a0 += __shfl_xor(a0, 1);
                a1 += __shfl_xor(a1, 1);
                av += __shfl_xor(av, 1);
                g0 = a0;
                g1 = a1 / (2.0 * sd);
                if (!(a.lik & TSVGP_LIK_NOCROP)) g1 = fmin(g1, -1e-8);  // reference tsvgp.py:262-263
                ve = av;
            } else if (a.lik != TSVGP_LIK_NONE && live) {
                lik_eval(a.lik, a.lik_param, mu, v, (double)a.Y[n * a.P + p], g0, g1, ve);
            }
            if (skh == 0) {
                if (live) {
                    if (!(v > 0.0)) nonpos += 1;
                    if (a.mean) a.mean[n * a.P + p] = (T)mu;
                    if (a.var) a.var[n * a.P + p] = (T)v;
                    ve_acc += ve;
                }
                if (a.lik != TSVGP_LIK_NONE) {
                    a.g0[n * a.P + p] = (T)(live ? g0 : 0.0);  // rows >= N: zeros (the padding contract of site_accum)
                    a.g1[n * a.P + p] = (T)(live ? g1 : 0.0);
                }
            }
        }
    }  // p

    double s = ve_acc;
    int c = nonpos;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        c += __shfl_xor(c, o);
    }
    if (lane == 0) {
        red[w] = s;
        redi[w] = c;
    }
    __syncthreads();
    if (t == 0) {
        if (a.ve_partial) a.ve_partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
        if (a.nonpos_partial) a.nonpos_partial[blockIdx.x] = redi[0] + redi[1] + redi[2] + redi[3];
    }
#ifdef TSVGP_DIAG_PANEL1
    if (t == 0 && g_panel1_stamps && blockIdx.y == 0 && (unsigned long long)blockIdx.x < g_panel1_stamps[0]) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* d = g_panel1_stamps + 8 + (size_t)blockIdx.x * 8;
        const unsigned long long c_end = __builtin_amdgcn_s_memtime();
        d[0] = dg_t0;
        d[1] = __builtin_amdgcn_s_memrealtime();
        d[2] = (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) |          // HW_REG_HW_ID
               ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11)) << 32);  // HW_REG_XCC_ID
        d[3] = c_end - dg_c0;        // shader cycles of the workgroup (wave 0)
        d[4] = dg_pro;               // entry -> first fragments of chunk (0, 0) in registers
        d[5] = dg_epi;               // the column tiles' square-sum epilogues
        d[6] = c_end - dg_loop_end;  // end of the chunk stream -> here: row sums to LDS, likelihood map, stores, reductions
        d[7] = dg_loop_end - dg_c0;
    }
#endif
}

#ifdef TSVGP_DIAG_PANEL1
extern "C" int tsvgp_diag_panel1_stamps(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_panel1_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif

// ---------------------------------------------------------------------------------------------------------------
// lik_map_kernel: the likelihood-gradient map on its own: (mean, var, Y) -> g0 = d ve / d mean,
// g1 = d ve / d var (cropped at -1e-8 unless TSVGP_LIK_NOCROP), per-128-row sums of ve and counts of non-positive variances --
// what the moments kernels do in their epilogue, for a caller that assembled the moments itself (t_SVGP_white's two-product
// variance).  Reference src/models/tsvgp.py:256-263.  One workgroup per 128 rows, two threads per row (the Bernoulli
// quadrature split as in panel_kernel).  O(N P): microseconds.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NTHREADS) void lik_map_kernel(const T* __restrict__ mean, const T* __restrict__ var,
                                                           const T* __restrict__ Y, int lik, double lik_param,
                                                           T* __restrict__ g0o, T* __restrict__ g1o, double* __restrict__ ve_partial,
                                                           int32_t* __restrict__ nonpos_partial, int64_t N, int P) {
    __shared__ double red[NTHREADS / 64];
    __shared__ int redi[NTHREADS / 64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int srow = t >> 1, skh = t & 1;
    const int64_t n = (int64_t)blockIdx.x * TILE + srow;
    const bool live = n < N;
    double ve_acc = 0.0;
    int nonpos = 0;
    for (int p = 0; p < P; ++p) {
        const double mu = live ? (double)mean[n * P + p] : 0.0;
        const double v = live ? (double)var[n * P + p] : 1.0;
        double g0 = 0.0, g1 = 0.0, ve = 0.0;
        if ((lik & 0xFF) == TSVGP_LIK_BERNOULLI) {
            double a0, a1, av;
            const double sd = sqrt(v);
            bern_sums_t<T>(mu, sd, live && (double)Y[n * P + p] == 1.0, skh * 5, skh * 5 + 5, a0, a1, av);
            a0 += __shfl_xor(a0, 1);
            a1 += __shfl_xor(a1, 1);
            av += __shfl_xor(av, 1);
            g0 = a0;
            g1 = a1 / (2.0 * sd);
            if (!(lik & TSVGP_LIK_NOCROP)) g1 = fmin(g1, -1e-8);
            ve = av;
        } else if (live) {
            lik_eval(lik, lik_param, mu, v, (double)Y[n * P + p], g0, g1, ve);
        }
        if (skh == 0) {
            if (live) {
                if (!(v > 0.0)) nonpos += 1;
                ve_acc += ve;
            }
            g0o[n * P + p] = (T)(live ? g0 : 0.0);  // rows >= N of the [Np x P] outputs: zeros
            g1o[n * P + p] = (T)(live ? g1 : 0.0);
        }
    }
    double s = ve_acc;
    int c = nonpos;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o);
        c += __shfl_xor(c, o);
    }
    if (lane == 0) {
        red[w] = s;
        redi[w] = c;
    }
    __syncthreads();
    if (t == 0) {
        ve_partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
        nonpos_partial[blockIdx.x] = redi[0] + redi[1] + redi[2] + redi[3];
    }
}

// ---------------------------------------------------------------------------------------------------------------
// mean_lik_kernel (TSVGP_LIK_MEANONLY): mean[n, p] = sum_j A[n, j] * gamma[j, p] and, for the Gaussian likelihood,
// g0 = (y - mean) / s2, g1 = -1 / (2 s2) -- neither depends on the predictive variance.  HBM bound: one sweep of A.
// One workgroup per 128-row panel (same grid as panel_kernel, so the per-workgroup partial buffers keep their
// meaning); a wave owns 32 rows and reads two of them at a time with 16-byte loads, gamma [P][Mp] in dynamic LDS.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(NTHREADS) void mean_lik_kernel(PanelArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char mean_dyn_smem[];
    T* const gs = reinterpret_cast<T*>(mean_dyn_smem);
    constexpr int V = 16 / sizeof(T);
    typedef T vec_t __attribute__((ext_vector_type(V)));
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int Mp = a.Mp, P = a.P;
    __shared__ int bad;  // rows whose mean or gradient is not finite: reported through nonpos_partial
    if (t == 0) bad = 0;
    for (int i = t; i < Mp * P; i += NTHREADS) {
        const int p = i / Mp, j = i - p * Mp;
        gs[i] = a.gamma[(size_t)j * P + p];
    }
    __syncthreads();
    const int64_t nw = (int64_t)blockIdx.x * TILE + w * (TILE / 4);
    for (int r = 0; r < TILE / 4; r += 2) {
        for (int p = 0; p < P; ++p) {  // shared operand (strideA = 0): P > 1 re-reads the two rows from cache
            const T* r0 = a.A + (size_t)p * a.strideA + (nw + r) * (int64_t)Mp;
            const T* r1 = r0 + Mp;
            const T* gp = gs + p * Mp;
            T s0 = T(0), s1 = T(0);
#pragma unroll 4
            for (int j = lane * V; j < Mp; j += 64 * V) {
                const vec_t x0 = *reinterpret_cast<const vec_t*>(r0 + j);
                const vec_t x1 = *reinterpret_cast<const vec_t*>(r1 + j);
                const vec_t g = *reinterpret_cast<const vec_t*>(gp + j);
#pragma unroll
                for (int q = 0; q < V; ++q) {
                    s0 += x0[q] * g[q];
                    s1 += x1[q] * g[q];
                }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                s0 += __shfl_xor(s0, o);
                s1 += __shfl_xor(s1, o);
            }
            if (lane < 2) {
                const int64_t n = nw + r + lane;
                const double mu = (double)(lane ? s1 : s0);
                double g0 = 0.0, g1 = 0.0, ve = 0.0;
                if (n < a.N) {
#ifndef TSVGP_DIAG_PANEL  // in that build `mean` is the stamp buffer of panel_kernel: nothing else may store through it
                    if (a.mean) a.mean[n * P + p] = (T)mu;
#endif
                    if (a.lik != TSVGP_LIK_NONE) lik_eval(a.lik, a.lik_param, mu, 0.0, (double)a.Y[n * P + p], g0, g1, ve);
                    if (!(fabs(mu) <= 1.79769313486231570815e308) || !(fabs(g0) <= 1.79769313486231570815e308)) atomicAdd(&bad, 1);
                }
                if (a.lik != TSVGP_LIK_NONE) {
                    a.g0[n * P + p] = (T)g0;  // rows >= N: zeros (the padding contract of site_accum)
                    a.g1[n * P + p] = (T)g1;
                }
            }
        }
    }
    __syncthreads();
    if (t == 0) {
        if (a.ve_partial) a.ve_partial[blockIdx.x] = __builtin_nan("");
        if (a.nonpos_partial) a.nonpos_partial[blockIdx.x] = bad;
    }
}

// ---------------------------------------------------------------------------------------------------------------
template <typename T>
struct SyrkArgs {
    const T* B;   // [Np x Mp]
    const T* g0;  // [Np x P]
    const T* g1;  // [Np x P]
    T* part2;     // [P][slabs_per_p][128*128], slab(tri, s) = (tri - it) * ns_off + it * ns_diag + s
    T* part1;     // [P][ns_diag][Mp]
    int64_t Np;
    int64_t strideB;  // elements between the operands of consecutive latents (0: shared)
    int Mp, P, nt, ns_off, ns_diag;
    int64_t chunks_off, chunks_diag;  // N-slice lengths in units of KC rows
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // bijective remap: workgroups that share blockIdx % 8 (one XCD under round-robin dispatch) get a contiguous range
    const int q = nwg / 8, r = nwg % 8, xcd = bid % 8;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bid / 8;
}

// Diagonal tiles run 9 of 16 MFMAs per k-step but the same loads, LDS traffic and barrier per chunk: measured with
// in-kernel stamps (tools/diag_syrk.py) a diagonal-tile chunk costs 0.71 of an off-diagonal one, not 9/16.
#ifndef TSVGP_SYRK_DIAG_NUM
#define TSVGP_SYRK_DIAG_NUM 23
#define TSVGP_SYRK_DIAG_DEN 32
#endif
#ifndef TSVGP_SYRK1_DIAG_NUM  // syrk1_kernel (fp64, round 3): 19/32 .. 23/32 measured within 1 % of each other; 20/32 best
#define TSVGP_SYRK1_DIAG_NUM 20
#endif
#ifndef TSVGP_SYRK1F_DIAG_NUM  // syrk1f_kernel (fp32, round 3)
#define TSVGP_SYRK1F_DIAG_NUM 22
#endif
#ifndef TSVGP_SYRK_OLD_F32  // (-DTSVGP_SYRK_OLD_F32=1: A/B builds keep round 2's syrk_kernel for fp32)
#define TSVGP_SYRK_OLD_F32 0
#endif
// (esize: sizeof of the N-sized arrays' type -- syrk1_kernel / syrk1f_kernel run unless the build keeps the old one)
__host__ __device__ inline int syrk_ns_diag(int ns_off, int esize) {
#ifndef TSVGP_SYRK_OLD
    const int num = esize == 8 ? TSVGP_SYRK1_DIAG_NUM : TSVGP_SYRK_OLD_F32 ? TSVGP_SYRK_DIAG_NUM : TSVGP_SYRK1F_DIAG_NUM;
#else
    const int num = TSVGP_SYRK_DIAG_NUM;
#endif
    const int d = (num * ns_off + TSVGP_SYRK_DIAG_DEN - 1) / TSVGP_SYRK_DIAG_DEN;
    return d < 1 ? 1 : d;
}

// The chunk loop of one workgroup.  DIAG / W are compile time so that the MFMA sequence is straight-line code.
template <typename T, bool DIAG, int W>
__device__ __forceinline__ void syrk_body(const SyrkArgs<T>& a, T (*lds)[2][KC * LDS_KS], T (*g0s)[KC], int p, int it,
                                          int jt, int sidx, int slab, int per_p, int w) {
    typedef typename Mfma<T>::acc_t acc_t;
    typedef typename Mfma<T>::pair_t pair_t;
    const int t = threadIdx.x, lane = t & 63;
    const int Mp = a.Mp, P = a.P;
    const int64_t total_chunks = a.Np / KC;
    const int64_t per = DIAG ? a.chunks_diag : a.chunks_off;
    int64_t c_lo = (int64_t)sidx * per;
    int64_t c_hi = c_lo + per;
    if (c_lo > total_chunks) c_lo = total_chunks;
    if (c_hi > total_chunks) c_hi = total_chunks;

    // staging role: k-row of the chunk and four 16-byte (fp64) / 8-byte (fp32) pieces spread over the 128 columns
    const int krow = t >> 4, cseg = t & 15;
    const T* Bi = a.B + (size_t)p * a.strideB + (int64_t)krow * Mp + it * TILE + cseg * 2;
    const T* Bj = a.B + (size_t)p * a.strideB + (int64_t)krow * Mp + jt * TILE + cseg * 2;

    acc_t acc[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[s][n] = acc_t{0, 0, 0, 0};
    T acc1 = T(0);

    pair_t ri[4], rj[4];
    T w1 = T(0), w0 = T(0);
    auto load_chunk = [&](int64_t c) {
        const int64_t n = c * KC + krow;
        const T* bi = Bi + c * KC * (int64_t)Mp;
#pragma unroll
        for (int q = 0; q < 4; ++q) ri[q] = *reinterpret_cast<const pair_t*>(bi + q * 32);
        if constexpr (!DIAG) {
            const T* bj = Bj + c * KC * (int64_t)Mp;
#pragma unroll
            for (int q = 0; q < 4; ++q) rj[q] = *reinterpret_cast<const pair_t*>(bj + q * 32);
        }
        w1 = a.g1[n * P + p];
        if (DIAG && cseg == 0) w0 = a.g0[n * P + p];
    };
    auto store_chunk = [&](int buf) {
        T* Ai = &lds[buf][0][krow * LDS_KS + cseg * 2];
        T* Aj = &lds[buf][1][krow * LDS_KS + cseg * 2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            pair_t sc;
            sc[0] = ri[q][0] * w1;
            sc[1] = ri[q][1] * w1;
            *reinterpret_cast<pair_t*>(Ai + q * 32) = sc;
            *reinterpret_cast<pair_t*>(Aj + q * 32) = DIAG ? ri[q] : rj[q];
        }
        if (DIAG && cseg == 0) g0s[buf][krow] = w0;
    };

#if defined(TSVGP_DIAG_CLOCK)
    unsigned long long seg[4] = {0, 0, 0, 0};
#define STAMP(i)
    const unsigned long long tstart = __builtin_amdgcn_s_memtime();
    const unsigned long long rstart = __builtin_amdgcn_s_memrealtime();
#elif defined(TSVGP_DIAG_STAMPS)
    unsigned long long seg[4] = {0, 0, 0, 0}, tprev;
#define STAMP(i)                                                                             \
    {                                                                                        \
        unsigned long long tn;                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tn)::"memory");           \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        seg[i] += tn - tprev;                                                                \
        tprev = tn;                                                                          \
    }
    const unsigned long long tstart = __builtin_amdgcn_s_memtime();
    const unsigned long long rstart = __builtin_amdgcn_s_memrealtime();
#else
#define STAMP(i)
#endif
    if (c_lo < c_hi) {
        load_chunk(c_lo);
        store_chunk(0);
    }
    __syncthreads();
#ifdef TSVGP_DIAG_STAMPS
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tprev)::"memory");
#endif
    // one pipeline step on chunk c in LDS buffer BUF (compile time): prefetch chunk c + 1 to registers, MFMAs, stage the
    // prefetched chunk into the other buffer, one barrier.  The loop body holds two steps so that the buffer index is a
    // constant in each (as in panel_kernel, where the pairing measured 3-6 % faster than a one-step body).
    auto step = [&](const int64_t c, auto buf_tag) {
        constexpr int BUF = decltype(buf_tag)::value;
        const bool has_next = (c + 1 < c_hi);
        if (has_next) load_chunk(c + 1);
        STAMP(0)
        mma_chunk_krow<T, DIAG, W>(acc, lds[BUF][0], lds[BUF][1], w, lane);
        STAMP(1)
        if (DIAG && t < TILE) {
            const T* bcol = &lds[BUF][1][t];
#pragma unroll
            for (int k = 0; k < KC; ++k) acc1 += g0s[BUF][k] * bcol[k * LDS_KS];
        }
        if (has_next) store_chunk(BUF ^ 1);
        STAMP(2)
        __syncthreads();
        STAMP(3)
    };
    {
        int64_t c = c_lo;
        for (; c + 1 < c_hi; c += 2) {
            step(c, std::integral_constant<int, 0>{});
            step(c + 1, std::integral_constant<int, 1>{});
        }
        if (c < c_hi) step(c, std::integral_constant<int, 0>{});
    }
#if defined(TSVGP_DIAG_STAMPS) || defined(TSVGP_DIAG_CLOCK)
    if (lane == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(a.part1 + (size_t)a.P * a.ns_diag * a.Mp) + ((size_t)blockIdx.x * 4 + w) * 8;
        dbg[0] = seg[0]; dbg[1] = seg[1]; dbg[2] = seg[2]; dbg[3] = seg[3];
        dbg[4] = __builtin_amdgcn_s_memtime() - tstart;
        dbg[5] = __builtin_amdgcn_s_memrealtime() - rstart;
        dbg[6] = (unsigned long long)(c_hi - c_lo);
        dbg[7] = DIAG;
#if defined(TSVGP_DIAG_CLOCK)
        dbg[0] = rstart;                             // absolute start (100 MHz ticks)
        dbg[1] = __builtin_amdgcn_s_memrealtime();   // absolute end
        dbg[2] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));   // HW_REG_HW_ID
        dbg[3] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
#endif
    }
#endif
#undef STAMP

    T* out = a.part2 + ((size_t)p * per_p + slab) * (TILE * TILE) + (lane & 15);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            T* orow = out + (row_block(w, s) * 16 + Mfma<T>::row(lane, r)) * TILE;
#pragma unroll
            for (int n = 0; n < 8; ++n)
                if (!DIAG || n <= (s == 0 ? W : 7 - W)) orow[n * 16] = acc[s][n][r];
        }
    if (DIAG && t < TILE) a.part1[((size_t)p * a.ns_diag + sidx) * Mp + it * TILE + t] = acc1;
}

template <typename T>
__global__ __launch_bounds__(NTHREADS, 2) void syrk_kernel(SyrkArgs<T> a) {
    __shared__ __attribute__((aligned(16))) T lds[2][2][KC * LDS_KS];
    __shared__ T g0s[2][KC];

    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nt = a.nt, n_off = nt * (nt - 1) / 2;
    const int per_p = n_off * a.ns_off + nt * a.ns_diag;
    int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lin / per_p;
    lin -= p * per_p;
    int it, jt, sidx;
    if (lin < n_off * a.ns_off) {
        sidx = lin / n_off;
        const int idx = lin - sidx * n_off;  // idx-th pair (it, jt) with jt < it
        it = 1;
        while (it * (it + 1) / 2 <= idx) ++it;
        jt = idx - it * (it - 1) / 2;
    } else {
        lin -= n_off * a.ns_off;
        sidx = lin / nt;
        it = jt = lin - sidx * nt;
    }
    const int tri = it * (it + 1) / 2 + jt;
    const int slab = (tri - it) * a.ns_off + it * a.ns_diag + sidx;

    // top-level uniform dispatch: no live state crosses the branches, each body is straight-line MFMA code
    if (it != jt) {
        syrk_body<T, false, 0>(a, lds, g0s, p, it, jt, sidx, slab, per_p, w);
    } else if (w == 0) {
        syrk_body<T, true, 0>(a, lds, g0s, p, it, jt, sidx, slab, per_p, w);
    } else if (w == 1) {
        syrk_body<T, true, 1>(a, lds, g0s, p, it, jt, sidx, slab, per_p, w);
    } else if (w == 2) {
        syrk_body<T, true, 2>(a, lds, g0s, p, it, jt, sidx, slab, per_p, w);
    } else {
        syrk_body<T, true, 3>(a, lds, g0s, p, it, jt, sidx, slab, per_p, w);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// syrk1_kernel (round 3, fp64): syrk_kernel's work -- one lower 128 x 128 tile of  sum_n g1[n] b_n b_n^T  over one N-slice, and
// the first-order sum on diagonal tiles -- with the recipe of panel1_kernel: ONE workgroup per CU (512 registers per wave, no
// scratch), the instruction stream of a chunk laid out by hand and pinned with sched_barrier(0), operands global -> LDS by
// LDS-DMA.  A chunk is 16 rows n of the operand; its two LDS images are [16 k-rows][144 doubles] (the k stride of syrk_kernel:
// conflict free for the fragment reads), and ONE DMA instruction moves one k-row (128 columns = 1 KiB lane-linear).  The
// weights no longer ride on the staging (there is none): a lane multiplies the two A fragments of a k-step by
// g1[n = 16 c + 4 ks + lk], which it loads itself two chunks ahead (four 8-byte loads per chunk).  First-order sum (diagonal
// tiles): lane (w, lr, lk) owns columns 32 w + 2 lr, + 1 and the k-rows of its MFMA role; reduced over lk at the end.
//     k-step 0: MFMAs on set X | reads of k-step 1 -> set Y   (tail: Y's A fragments times their weight)
//     k-step 1: MFMAs on set Y | reads of k-step 2 -> set X
//     k-step 2: MFMAs on set X | reads of k-step 3 -> set Y
//     s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier
//     k-step 3: MFMAs on set Y | LDS-DMA of chunk c + 2 into this chunk's buffer, its weights, reads of (c + 1, 0) -> set X
// ---------------------------------------------------------------------------------------------------------------
constexpr int S1_IMG = KC * LDS_KS;   // doubles per image (16 k-rows of 144)
constexpr int S1_BUF = 2 * S1_IMG;    // doubles per chunk buffer (image I + image J)

template <bool DIAG, int W>
__device__ __forceinline__ void syrk1_body(const SyrkArgs<double>& a, double* lds, double* wl, int p, int it, int jt, int sidx, int slab,
                                           int per_p, int w) {
    typedef v4d acc_t;
    const int t = threadIdx.x, lane = t & 63;
    const int Mp = a.Mp, P = a.P;
    const int64_t total_chunks = a.Np / KC;
    const int64_t per = DIAG ? a.chunks_diag : a.chunks_off;
    int64_t c_lo = (int64_t)sidx * per;
    int64_t c_hi = c_lo + per;
    if (c_lo > total_chunks) c_lo = total_chunks;
    if (c_hi > total_chunks) c_hi = total_chunks;
    const int nch = (int)(c_hi - c_lo);

    // column blocks the wave's MFMAs touch: all eight, or on a diagonal tile n <= W for row block W and n <= 7 - W for row
    // block 7 - W (9 of the 16 pairs); MB = T-side fragments needed per k-step
    constexpr int MB = DIAG ? 8 - W : 8;
    constexpr int NM = DIAG ? 9 : 16;
    constexpr int NR = 2 + MB;

    const int lr = lane & 15, lk = lane >> 4;
    const int offa0 = lk * LDS_KS + w * 16 + lr;
    const int offa1 = lk * LDS_KS + (7 - w) * 16 + lr;
    const int offb = lk * LDS_KS + lr + (DIAG ? 0 : S1_IMG);  // diagonal tile: both sides come from the one image
    const int off1 = lk * LDS_KS + w * 32 + lr * 2;            // first-order sum: the lane's column pair

    auto uni64 = [](const void* ptr) TSVGP_AI {
        const uint64_t v = (uint64_t)(uintptr_t)ptr;
        return ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
               (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    };
    // DMA: piece i (0..7): image i & 1 (0: columns of tile it, 1: of tile jt), k-row 4 (i >> 1) + w; a lane moves 16 bytes
    const unsigned lds_u = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)(lds_void_t*)lds + (unsigned)(w * LDS_KS * 8)));
    const unsigned dvo = (unsigned)(lane * 16);
    const double* Bp = a.B + (size_t)p * a.strideB + (int64_t)w * Mp;
    const uint64_t bi_u = uni64(Bp + it * TILE), bj_u = uni64(Bp + jt * TILE);
    const uint64_t row4 = (uint64_t)4 * Mp * sizeof(double), chunkb = (uint64_t)KC * Mp * sizeof(double);
    uint64_t ci_u = 0, cj_u = 0;  // bases of the chunk being fetched
    auto dma_setup = [&](const int64_t c) TSVGP_AI {
        ci_u = bi_u + (uint64_t)c * chunkb;
        cj_u = bj_u + (uint64_t)c * chunkb;
    };
    auto dma_piece = [&](auto i_tag, const int buf) TSVGP_AI {
        constexpr int I = decltype(i_tag)::value, rq = I >> 1;
        const unsigned la = lds_u + (unsigned)((buf * S1_BUF + (I & 1) * S1_IMG + rq * 4 * LDS_KS) * sizeof(double));
        const uint64_t g = ((I & 1) ? cj_u : ci_u) + rq * row4;
        const unsigned vo_ = dvo;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(la), "v"(vo_), "s"(g) : "memory");
    };
    constexpr int NDMA = DIAG ? 4 : 8;  // diagonal tile: image I only (pieces 0, 2, 4, 6)

    // Weights of a chunk: the 16 P doubles g1[16 c .. 16 c + 15][0 .. P) (P <= 8: at most 1 KiB) travel by ONE more LDS-DMA
    // instruction, issued by wave 0 with the chunk's operand pieces (wave 1: g0, diagonal tiles), lanes beyond the 16 P doubles
    // switched off in EXEC so that nothing is read behind the end of the array; behind the barrier every lane takes the weights
    // of its four k-rows from LDS.  (Loading them with ordinary global loads made the compiler wait -- s_waitcnt vmcnt(small) in
    // front of their first use -- for the LDS-DMA pieces in flight, which it cannot see: the whole memory latency, every chunk.)
    const unsigned wl_u = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(lds_void_t*)wl);
    const uint64_t wmask = (P >= 8) ? ~0ull : ((1ull << (8 * P)) - 1);
    const uint64_t g1_u = uni64(a.g1), g0_u = uni64(a.g0);
    const uint64_t wchunk = (uint64_t)KC * P * sizeof(double);
    auto dma_weights = [&](const int64_t c, const int buf) TSVGP_AI {  // wl: [buffer][g1 | g0][128 doubles]
        if (w == 0 || (DIAG && w == 1)) {
            const unsigned la = wl_u + (unsigned)((buf * 256 + (w == 1 ? 128 : 0)) * sizeof(double));
            const uint64_t g = (w == 1 ? g0_u : g1_u) + (uint64_t)c * wchunk;
            const unsigned vo_ = dvo;
            uint64_t saved;
            asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %3, %4\n\ts_mov_b64 exec, %0"
                         : "=&s"(saved)
                         : "s"(wmask), "s"(la), "v"(vo_), "s"(g)
                         : "memory");
        }
    };
    double w1r[4], w0r[4];  // the lane's weights of the chunk being computed: k-rows 4 ks + lk
    const int woff = lk * P + p;
    auto read_weight = [&](auto i_tag, const int buf) TSVGP_AI {  // I < 4: g1 of k-step I; I >= 4: g0 of k-step I - 4
        constexpr int I = decltype(i_tag)::value;
        if constexpr (I < 4) w1r[I] = wl[buf * 256 + woff + I * 4 * P];
        else w0r[I - 4] = wl[buf * 256 + 128 + woff + (I - 4) * 4 * P];
    };
    constexpr int NW = DIAG ? 8 : 4;

    acc_t acc[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[s][n] = acc_t{0, 0, 0, 0};
    double acc1v[2] = {0.0, 0.0};
    Frag1<double> fx, fy;

    auto rd1 = [&](Frag1<double>& f, auto e_tag, auto ks_tag, const int boff) TSVGP_AI {
        constexpr int E = decltype(e_tag)::value, KS_ = decltype(ks_tag)::value;
        if constexpr (E == 0) f.v[0] = lds[offa0 + boff + KS_ * 4 * LDS_KS];
        else if constexpr (E == 1) f.v[1] = lds[offa1 + boff + KS_ * 4 * LDS_KS];
        else f.v[E] = lds[offb + boff + KS_ * 4 * LDS_KS + (E - 2) * 16];
    };
    auto rds = [&](Frag1<double>& f, auto slot_tag, auto ks_tag, const int boff) TSVGP_AI {  // T-side fragments first, A-side last
        constexpr int S = decltype(slot_tag)::value;
        if constexpr (S < MB) rd1(f, TSVGP_IC(2 + S), ks_tag, boff);
        else rd1(f, TSVGP_IC(S - MB), ks_tag, boff);
    };
    auto keep_set = [&](const Frag1<double>& f) TSVGP_AI {
        cfor<0, NR>([&](auto e) TSVGP_AI {
            const double x = f.v[decltype(e)::value];
            asm volatile("" ::"v"(x));
        });
    };
    // MFMA number I of a k-step.  Off-diagonal: column block I / 2, row block I % 2.  Diagonal tile, wave W: the pairs
    // (row block s, column block n) with n <= W (s = 0) or n <= 7 - W (s = 1), ordered by n
    auto mf = [&](const Frag1<double>& f, auto i_tag) TSVGP_AI {
        constexpr int I = decltype(i_tag)::value;
        if constexpr (!DIAG) {
            constexpr int n = I >> 1, sblk = I & 1;
            acc[sblk][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.v[sblk], f.v[2 + n], acc[sblk][n], 0, 0, 0);
        } else {
            // n <= W: both row blocks (pairs 2 n, 2 n + 1); W < n <= 7 - W: row block 1 only
            constexpr int n = (I < 2 * (W + 1)) ? (I >> 1) : (I - (W + 1));
            constexpr int sblk = (I < 2 * (W + 1)) ? (I & 1) : 1;
            acc[sblk][n] = __builtin_amdgcn_mfma_f64_16x16x4f64(f.v[sblk], f.v[2 + n], acc[sblk][n], 0, 0, 0);
        }
    };
    v2d f1[4];  // first-order sum: the lane's column pair in its four k-rows
    // one chunk of the slice; on entry set X holds the (weighted) fragments of its k-step 0
    auto chunk = [&](auto buf_tag, auto next_tag, const int64_t c_fetch) TSVGP_AI {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr bool NEXT = decltype(next_tag)::value;  // a chunk follows in this slice
        constexpr int B0 = BUF * S1_BUF, B1 = (BUF ^ 1) * S1_BUF;
        if constexpr (NEXT) {
            dma_setup(c_fetch);
            TSVGP_SB();
        }
        // k-steps 0..2: MFMAs on one set, the next k-step's reads into the other, then its A fragments times their weight
        cfor<0, 3>([&](auto ks) TSVGP_AI {
            constexpr int KS_ = decltype(ks)::value;
            Frag1<double>& cur = (KS_ & 1) ? fy : fx;
            Frag1<double>& nxt = (KS_ & 1) ? fx : fy;
            constexpr int S = NR + ((DIAG && KS_ == 0) ? 4 : 0);
            cfor<0, (NM > S ? NM : S)>([&](auto i) TSVGP_AI {
                constexpr int I = decltype(i)::value;
                if constexpr (I < NM) mf(cur, i);
                if constexpr (I < NR) rds(nxt, i, TSVGP_IC(KS_ + 1), B0);
                else if constexpr (DIAG && KS_ == 0 && I < NR + 4)
                    f1[I - NR] = *reinterpret_cast<const v2d*>(lds + off1 + B0 + (I - NR) * 4 * LDS_KS);
                TSVGP_SB();
            });
            nxt.v[0] *= w1r[KS_ + 1];
            nxt.v[1] *= w1r[KS_ + 1];
            if constexpr (DIAG && KS_ == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc1v[0] += w0r[q] * f1[q][0];
                    acc1v[1] += w0r[q] * f1[q][1];
                }
            }
            TSVGP_SB();
            keep_set(cur);
        });
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        TSVGP_SB();
        if constexpr (NEXT) {
            // slots: the operand pieces of chunk c + 2, then the weights of chunk c + 1 (landed with it; this chunk's are all
            // consumed), then the fragments of (c + 1, k-step 0)
            constexpr int S3 = NDMA + NW + NR;
            cfor<0, (NM > S3 ? NM : S3)>([&](auto i) TSVGP_AI {
                constexpr int I = decltype(i)::value;
                if constexpr (I < NM) mf(fy, i);
                if constexpr (I < NDMA) dma_piece(TSVGP_IC(DIAG ? 2 * I : I), BUF);
                else if constexpr (I < NDMA + NW) read_weight(TSVGP_IC(I - NDMA), BUF ^ 1);
                else if constexpr (I < S3) rds(fx, TSVGP_IC(I - NDMA - NW), TSVGP_IC(0), B1);
                TSVGP_SB();
            });
            dma_weights(c_fetch, BUF);
            fx.v[0] *= w1r[0];
            fx.v[1] *= w1r[0];
            TSVGP_SB();
        } else {
            cfor<0, NM>([&](auto i) TSVGP_AI { mf(fy, i); });
        }
        keep_set(fy);
    };

    if (nch > 0) {
        // prologue: the first two chunks and their weights on their way, then the first weights and fragments
        dma_setup(c_lo);
        cfor<0, NDMA>([&](auto i) TSVGP_AI { dma_piece(TSVGP_IC(DIAG ? 2 * decltype(i)::value : decltype(i)::value), 0); });
        dma_weights(c_lo, 0);
        const int64_t c1 = (nch > 1) ? c_lo + 1 : c_lo;  // a slice of one chunk fetches it twice (the second copy is never read)
        dma_setup(c1);
        cfor<0, NDMA>([&](auto i) TSVGP_AI { dma_piece(TSVGP_IC(DIAG ? 2 * decltype(i)::value : decltype(i)::value), 1); });
        dma_weights(c1, 1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        cfor<0, NW>([&](auto i) TSVGP_AI { read_weight(i, 0); });
        cfor<0, NR>([&](auto i) TSVGP_AI { rds(fx, i, TSVGP_IC(0), 0); });
        fx.v[0] *= w1r[0];
        fx.v[1] *= w1r[0];
        // chunk i of the slice sits in buffer i & 1 and fetches chunk i + 2 (clamped to the last one: a harmless re-fetch
        // into a dead buffer) while a chunk follows it
        int i = 0;
        const int64_t c_last = c_hi - 1;
        auto fetch_of = [&](int idx) TSVGP_AI { const int64_t c = c_lo + idx + 2; return c < c_last ? c : c_last; };
        if (nch >= 3) {
            // The first pair of chunks stands in front of the loop: the accumulators then enter the loop as MFMA results.
            // (With the zero-initialised values as the loop's incoming ones the compiler kept all 128 accumulator registers in
            // VGPRs across the back edge and copied them into and out of the AGPRs the MFMAs run on in EVERY iteration:
            // 256 v_accvgpr moves per 128 MFMAs, 19.7 instead of 16.8 ms for the launch.)
            chunk(TSVGP_IC(0), TSVGP_BC(true), fetch_of(0));
            chunk(TSVGP_IC(1), TSVGP_BC(true), fetch_of(1));
            for (i = 2; i + 2 < nch; i += 2) {
                chunk(TSVGP_IC(0), TSVGP_BC(true), fetch_of(i));
                chunk(TSVGP_IC(1), TSVGP_BC(true), fetch_of(i + 1));
            }
        }
        if (nch - i == 2) {
            chunk(TSVGP_IC(0), TSVGP_BC(true), fetch_of(i));
            chunk(TSVGP_IC(1), TSVGP_BC(false), 0);
        } else {
            chunk(TSVGP_IC(0), TSVGP_BC(false), 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clamped re-fetches have landed before the LDS is given back
    }

    double* out = a.part2 + ((size_t)p * per_p + slab) * (TILE * TILE) + (lane & 15);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double* orow = out + (row_block(w, s) * 16 + Mfma<double>::row(lane, r)) * TILE;
#pragma unroll
            for (int n = 0; n < 8; ++n)
                if (!DIAG || n <= (s == 0 ? W : 7 - W)) orow[n * 16] = acc[s][n][r];
        }
    if constexpr (DIAG) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            acc1v[q] += __shfl_xor(acc1v[q], 16);
            acc1v[q] += __shfl_xor(acc1v[q], 32);
        }
        if (lk == 0) {
            double* o1 = a.part1 + ((size_t)p * a.ns_diag + sidx) * Mp + it * TILE + w * 32 + lr * 2;
            o1[0] = acc1v[0];
            o1[1] = acc1v[1];
        }
    }
}

__global__ __launch_bounds__(NTHREADS, 1) void syrk1_kernel(SyrkArgs<double> a) {
    __shared__ __attribute__((aligned(1024))) double lds[2 * S1_BUF];
    __shared__ __attribute__((aligned(1024))) double wl[2 * 256];  // per chunk buffer: 128 doubles of g1, 128 of g0

    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nt = a.nt, n_off = nt * (nt - 1) / 2;
    const int per_p = n_off * a.ns_off + nt * a.ns_diag;
    int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lin / per_p;
    lin -= p * per_p;
    int it, jt, sidx;
    if (lin < n_off * a.ns_off) {
        sidx = lin / n_off;
        const int idx = lin - sidx * n_off;  // idx-th pair (it, jt) with jt < it
        it = 1;
        while (it * (it + 1) / 2 <= idx) ++it;
        jt = idx - it * (it - 1) / 2;
    } else {
        lin -= n_off * a.ns_off;
        sidx = lin / nt;
        it = jt = lin - sidx * nt;
    }
    const int tri = it * (it + 1) / 2 + jt;
    const int slab = (tri - it) * a.ns_off + it * a.ns_diag + sidx;
    if (it != jt) {
        syrk1_body<false, 0>(a, lds, wl, p, it, jt, sidx, slab, per_p, w);
    } else if (w == 0) {
        syrk1_body<true, 0>(a, lds, wl, p, it, jt, sidx, slab, per_p, w);
    } else if (w == 1) {
        syrk1_body<true, 1>(a, lds, wl, p, it, jt, sidx, slab, per_p, w);
    } else if (w == 2) {
        syrk1_body<true, 2>(a, lds, wl, p, it, jt, sidx, slab, per_p, w);
    } else {
        syrk1_body<true, 3>(a, lds, wl, p, it, jt, sidx, slab, per_p, w);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// syrk1f_kernel (round 3, fp32): syrk1_kernel's recipe for the fp32 N-arrays.  A chunk is 32 rows n of the operand (the same
// 16 KB per image and the same eight 1-KiB LDS-DMA pieces per wave as fp64: one piece = TWO k-rows of 128 floats), a k-step is
// one v_mfma_f32_16x16x4_f32 per (row block, column block) -- half the cycles of the fp64 one, so a chunk of eight k-steps takes
// the MFMA time of an fp64 chunk of four.  The images are unpadded [32 k-rows][128 floats] (a DMA piece lands lane-linear);
// what the padding did in fp64 a swizzle does here: the 16-byte unit u of k-row k sits at unit u ^ 4 (k & 1) (applied to the
// per-lane global source address of the DMA and to the reads), i.e. the 16-column block n of an odd k-row sits where block
// n ^ 1 would: the two k-rows a half wave reads in one ds_read_b32 fall into different halves of the 32 banks.
// The weights of a chunk (32 P floats <= 1 KiB) travel as one more DMA piece, as in syrk1_kernel.
//     chunk start: weights of k-steps 4..7 (and g0's) from LDS
//     k-steps 0..6: MFMAs on one set | reads of the next k-step -> the other set (tail: its A fragments times their weight)
//                   diagonal tiles: the first-order operands in k-steps 0..3, their FMAs in k-steps 4, 5
//     s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier
//     k-step 7: MFMAs | LDS-DMA of chunk c + 2 into this chunk's buffer, weights of k-steps 0..3 and reads of (c + 1, 0)
// ---------------------------------------------------------------------------------------------------------------
constexpr int S1F_KC = 32;              // rows n per chunk
constexpr int S1F_IMG = S1F_KC * TILE;  // floats per image (16 KB)
constexpr int S1F_BUF = 2 * S1F_IMG;    // floats per chunk buffer (image I + image J)
constexpr int S1F_WL = 512;             // floats of weights per chunk buffer: 256 of g1, 256 of g0

template <bool DIAG, int W>
__device__ __forceinline__ void syrk1f_body(const SyrkArgs<float>& a, float* lds, float* wl, int p, int it, int jt, int sidx,
                                            int slab, int per_p, int w) {
    typedef v4f acc_t;
    typedef Frag1<float> Frag;
    const int t = threadIdx.x, lane = t & 63;
    const int Mp = a.Mp, P = a.P;
    const int64_t total_chunks = a.Np / S1F_KC;
    const int64_t per = DIAG ? a.chunks_diag : a.chunks_off;  // (the launcher counts them in chunks of 32 rows for this kernel)
    int64_t c_lo = (int64_t)sidx * per;
    int64_t c_hi = c_lo + per;
    if (c_lo > total_chunks) c_lo = total_chunks;
    if (c_hi > total_chunks) c_hi = total_chunks;
    const int nch = (int)(c_hi - c_lo);

    constexpr int MB = DIAG ? 8 - W : 8;  // as syrk1_body
    constexpr int NM = DIAG ? 9 : 16;
    constexpr int NR = 2 + MB;
    constexpr int NKS = S1F_KC / 4;  // k-steps per chunk

    // fragment of k-row k = 4 ks + lk, column 16 n + lr: float k * 128 + 16 (n ^ (k & 1)) + lr, and k & 1 = lk & 1
    const int lr = lane & 15, lk = lane >> 4, odd = lk & 1;
    const int offa0 = lk * TILE + 16 * (w ^ odd) + lr;
    const int offa1 = lk * TILE + 16 * ((7 - w) ^ odd) + lr;
    const int offb_e = lk * TILE + lr + 16 * odd + (DIAG ? 0 : S1F_IMG);  // even column blocks: n ^ odd = n + odd
    const int offb_o = lk * TILE + lr - 16 * odd + (DIAG ? 0 : S1F_IMG);  // odd ones:          n ^ odd = n - odd
    // first-order sum: the lane's column pair 32 w + 2 lr, + 1 = floats 2 (lr & 1), + 1 of unit 8 w + (lr >> 1)
    const int off1 = lk * TILE + 4 * ((8 * w + (lr >> 1)) ^ (4 * odd)) + 2 * (lr & 1);

    auto uni64 = [](const void* ptr) TSVGP_AI {
        const uint64_t v = (uint64_t)(uintptr_t)ptr;
        return ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
               (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    };
    // DMA: piece i (0..7): image i & 1, k-row pair 4 (i >> 1) + w; lane L lands at unit L & 31 of k-row 2 pair + (L >> 5) and
    // fetches the unit (L & 31) ^ 4 (L >> 5) of that row
    const unsigned lds_u = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)(lds_void_t*)lds + (unsigned)(w * 1024)));
    const unsigned dvo = (unsigned)((lane >> 5) * Mp * (int)sizeof(float) + (((lane & 31) ^ ((lane >> 5) << 2)) << 4));
    const unsigned dvw = (unsigned)(lane * 16);  // the weights piece is linear
    const float* Bp = a.B + (size_t)p * a.strideB + (int64_t)(2 * w) * Mp;
    const uint64_t bi_u = uni64(Bp + it * TILE), bj_u = uni64(Bp + jt * TILE);
    const uint64_t row8 = (uint64_t)8 * Mp * sizeof(float), chunkb = (uint64_t)S1F_KC * Mp * sizeof(float);
    uint64_t ci_u = 0, cj_u = 0;
    auto dma_setup = [&](const int64_t c) TSVGP_AI {
        ci_u = bi_u + (uint64_t)c * chunkb;
        cj_u = bj_u + (uint64_t)c * chunkb;
    };
    auto dma_piece = [&](auto i_tag, const int buf) TSVGP_AI {
        constexpr int I = decltype(i_tag)::value, rq = I >> 1;
        const unsigned la = lds_u + (unsigned)((buf * S1F_BUF + (I & 1) * S1F_IMG + rq * 1024) * sizeof(float));
        const uint64_t g = ((I & 1) ? cj_u : ci_u) + rq * row8;
        const unsigned vo_ = dvo;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(la), "v"(vo_), "s"(g) : "memory");
    };
    constexpr int NDMA = DIAG ? 4 : 8;

    const unsigned wl_u = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(uintptr_t)(lds_void_t*)wl);
    const uint64_t wmask = (P >= 8) ? ~0ull : ((1ull << (8 * P)) - 1);  // 32 P floats = 8 P lanes of 16 bytes
    const uint64_t g1_u = uni64(a.g1), g0_u = uni64(a.g0);
    const uint64_t wchunk = (uint64_t)S1F_KC * P * sizeof(float);
    auto dma_weights = [&](const int64_t c, const int buf) TSVGP_AI {  // wl: [buffer][g1 | g0][256 floats]
        if (w == 0 || (DIAG && w == 1)) {
            const unsigned la = wl_u + (unsigned)((buf * S1F_WL + (w == 1 ? 256 : 0)) * sizeof(float));
            const uint64_t g = (w == 1 ? g0_u : g1_u) + (uint64_t)c * wchunk;
            const unsigned vo_ = dvw;
            uint64_t saved;
            asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %3, %4\n\ts_mov_b64 exec, %0"
                         : "=&s"(saved)
                         : "s"(wmask), "s"(la), "v"(vo_), "s"(g)
                         : "memory");
        }
    };
    float w1r[NKS], w0r[NKS];  // the lane's weights of the chunk being computed: k-rows 4 ks + lk
    const int woff = lk * P + p;
    auto read_weight = [&](auto i_tag, const int buf) TSVGP_AI {  // I < 8: g1 of k-step I; I >= 8: g0 of k-step I - 8
        constexpr int I = decltype(i_tag)::value;
        if constexpr (I < NKS) w1r[I] = wl[buf * S1F_WL + woff + I * 4 * P];
        else w0r[I - NKS] = wl[buf * S1F_WL + 256 + woff + (I - NKS) * 4 * P];
    };
    // weight slot J of a half (HALF 0: k-steps 0..3, read in front of the chunk; HALF 1: k-steps 4..7, read at its start)
    constexpr int NWH = DIAG ? 8 : 4;
    auto read_weight_half = [&](auto j_tag, auto half_tag, const int buf) TSVGP_AI {
        constexpr int J = decltype(j_tag)::value, H = decltype(half_tag)::value;
        if constexpr (J < 4) read_weight(TSVGP_IC(4 * H + J), buf);
        else read_weight(TSVGP_IC(NKS + 4 * H + J - 4), buf);
    };

    acc_t acc[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int n = 0; n < 8; ++n) acc[s][n] = acc_t{0, 0, 0, 0};
    float acc1v[2] = {0.f, 0.f};
    Frag fx, fy;

    auto rd1 = [&](Frag& f, auto e_tag, auto ks_tag, const int boff) TSVGP_AI {
        constexpr int E = decltype(e_tag)::value, KS_ = decltype(ks_tag)::value;
        if constexpr (E == 0) f.v[0] = lds[offa0 + boff + KS_ * 4 * TILE];
        else if constexpr (E == 1) f.v[1] = lds[offa1 + boff + KS_ * 4 * TILE];
        else f.v[E] = lds[(((E - 2) & 1) ? offb_o : offb_e) + boff + KS_ * 4 * TILE + (E - 2) * 16];
    };
    auto rds = [&](Frag& f, auto slot_tag, auto ks_tag, const int boff) TSVGP_AI {  // T-side fragments first, A-side last
        constexpr int S = decltype(slot_tag)::value;
        if constexpr (S < MB) rd1(f, TSVGP_IC(2 + S), ks_tag, boff);
        else rd1(f, TSVGP_IC(S - MB), ks_tag, boff);
    };
    auto keep_set = [&](const Frag& f) TSVGP_AI {
        cfor<0, NR>([&](auto e) TSVGP_AI {
            const float x = f.v[decltype(e)::value];
            asm volatile("" ::"v"(x));
        });
    };
    auto mf = [&](const Frag& f, auto i_tag) TSVGP_AI {  // as syrk1_body
        constexpr int I = decltype(i_tag)::value;
        if constexpr (!DIAG) {
            constexpr int n = I >> 1, sblk = I & 1;
            acc[sblk][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.v[sblk], f.v[2 + n], acc[sblk][n], 0, 0, 0);
        } else {
            constexpr int n = (I < 2 * (W + 1)) ? (I >> 1) : (I - (W + 1));
            constexpr int sblk = (I < 2 * (W + 1)) ? (I & 1) : 1;
            acc[sblk][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f.v[sblk], f.v[2 + n], acc[sblk][n], 0, 0, 0);
        }
    };
    v2f f1[NKS];  // first-order sum: the lane's column pair in its eight k-rows
    auto chunk = [&](auto buf_tag, auto next_tag, const int64_t c_fetch) TSVGP_AI {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr bool NEXT = decltype(next_tag)::value;
        constexpr int B0 = BUF * S1F_BUF, B1 = (BUF ^ 1) * S1F_BUF;
        if constexpr (NEXT) {
            dma_setup(c_fetch);
            TSVGP_SB();
        }
        cfor<0, NKS - 1>([&](auto ks) TSVGP_AI {
            constexpr int KS_ = decltype(ks)::value;
            Frag& cur = (KS_ & 1) ? fy : fx;
            Frag& nxt = (KS_ & 1) ? fx : fy;
            // extra slots behind the fragment reads: k-step 0 the second half of the weights; diagonal tiles, k-steps 0..3: two
            // first-order operands each
            constexpr int XW = KS_ == 0 ? NWH : 0;
            constexpr int XF = (DIAG && KS_ < 4) ? 2 : 0;
            constexpr int S = NR + XW + XF;
            cfor<0, (NM > S ? NM : S)>([&](auto i) TSVGP_AI {
                constexpr int I = decltype(i)::value;
                if constexpr (I < NM) mf(cur, i);
                if constexpr (I < NR) rds(nxt, i, TSVGP_IC(KS_ + 1), B0);
                else if constexpr (I < NR + XW) read_weight_half(TSVGP_IC(I - NR), TSVGP_IC(1), BUF);
                else if constexpr (I < S)
                    f1[2 * KS_ + I - NR - XW] = *reinterpret_cast<const v2f*>(lds + off1 + B0 + (2 * KS_ + I - NR - XW) * 4 * TILE);
                TSVGP_SB();
            });
            nxt.v[0] *= w1r[KS_ + 1];
            nxt.v[1] *= w1r[KS_ + 1];
            if constexpr (DIAG && (KS_ == 4 || KS_ == 5)) {
#pragma unroll
                for (int q = 4 * (KS_ - 4); q < 4 * (KS_ - 4) + 4; ++q) {
                    acc1v[0] += w0r[q] * f1[q][0];
                    acc1v[1] += w0r[q] * f1[q][1];
                }
            }
            TSVGP_SB();
            keep_set(cur);
        });
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        TSVGP_SB();
        // (NKS - 1 is odd: the last k-step's fragments sit in set Y)
        if constexpr (NEXT) {
            constexpr int S3 = NDMA + NWH + NR;
            cfor<0, (NM > S3 ? NM : S3)>([&](auto i) TSVGP_AI {
                constexpr int I = decltype(i)::value;
                if constexpr (I < NM) mf(fy, i);
                if constexpr (I < NDMA) dma_piece(TSVGP_IC(DIAG ? 2 * I : I), BUF);
                else if constexpr (I < NDMA + NWH) read_weight_half(TSVGP_IC(I - NDMA), TSVGP_IC(0), BUF ^ 1);
                else if constexpr (I < S3) rds(fx, TSVGP_IC(I - NDMA - NWH), TSVGP_IC(0), B1);
                TSVGP_SB();
            });
            dma_weights(c_fetch, BUF);
            fx.v[0] *= w1r[0];
            fx.v[1] *= w1r[0];
            TSVGP_SB();
        } else {
            cfor<0, NM>([&](auto i) TSVGP_AI { mf(fy, i); });
        }
        keep_set(fy);
    };

    if (nch > 0) {
        dma_setup(c_lo);
        cfor<0, NDMA>([&](auto i) TSVGP_AI { dma_piece(TSVGP_IC(DIAG ? 2 * decltype(i)::value : decltype(i)::value), 0); });
        dma_weights(c_lo, 0);
        const int64_t c1 = (nch > 1) ? c_lo + 1 : c_lo;  // a slice of one chunk fetches it twice (the second copy is never read)
        dma_setup(c1);
        cfor<0, NDMA>([&](auto i) TSVGP_AI { dma_piece(TSVGP_IC(DIAG ? 2 * decltype(i)::value : decltype(i)::value), 1); });
        dma_weights(c1, 1);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        cfor<0, NWH>([&](auto i) TSVGP_AI { read_weight_half(i, TSVGP_IC(0), 0); });
        cfor<0, NR>([&](auto i) TSVGP_AI { rds(fx, i, TSVGP_IC(0), 0); });
        fx.v[0] *= w1r[0];
        fx.v[1] *= w1r[0];
        int i = 0;
        const int64_t c_last = c_hi - 1;
        auto fetch_of = [&](int idx) TSVGP_AI { const int64_t c = c_lo + idx + 2; return c < c_last ? c : c_last; };
        if (nch >= 3) {  // the first pair in front of the loop: see syrk1_body
            chunk(TSVGP_IC(0), TSVGP_BC(true), fetch_of(0));
            chunk(TSVGP_IC(1), TSVGP_BC(true), fetch_of(1));
            for (i = 2; i + 2 < nch; i += 2) {
                chunk(TSVGP_IC(0), TSVGP_BC(true), fetch_of(i));
                chunk(TSVGP_IC(1), TSVGP_BC(true), fetch_of(i + 1));
            }
        }
        if (nch - i == 2) {
            chunk(TSVGP_IC(0), TSVGP_BC(true), fetch_of(i));
            chunk(TSVGP_IC(1), TSVGP_BC(false), 0);
        } else {
            chunk(TSVGP_IC(0), TSVGP_BC(false), 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    float* out = a.part2 + ((size_t)p * per_p + slab) * (TILE * TILE) + (lane & 15);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* orow = out + (row_block(w, s) * 16 + Mfma<float>::row(lane, r)) * TILE;
#pragma unroll
            for (int n = 0; n < 8; ++n)
                if (!DIAG || n <= (s == 0 ? W : 7 - W)) orow[n * 16] = acc[s][n][r];
        }
    if constexpr (DIAG) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            acc1v[q] += __shfl_xor(acc1v[q], 16);
            acc1v[q] += __shfl_xor(acc1v[q], 32);
        }
        if (lk == 0) {
            float* o1 = a.part1 + ((size_t)p * a.ns_diag + sidx) * Mp + it * TILE + w * 32 + lr * 2;
            o1[0] = acc1v[0];
            o1[1] = acc1v[1];
        }
    }
}

#ifndef TSVGP_S1F_WAVES  // waves per SIMD the register allocation may assume at most (1: one workgroup per CU)
#define TSVGP_S1F_WAVES 2
#endif
__global__ __launch_bounds__(NTHREADS, 1) __attribute__((amdgpu_waves_per_eu(1, TSVGP_S1F_WAVES))) void syrk1f_kernel(SyrkArgs<float> a) {
    __shared__ __attribute__((aligned(1024))) float lds[2 * S1F_BUF];
    __shared__ __attribute__((aligned(1024))) float wl[2 * S1F_WL];
#ifdef TSVGP_S1F_LDS_PAD  // experiment: more than half of a CU's LDS, i.e. one workgroup per CU
    __shared__ float s1f_pad[TSVGP_S1F_LDS_PAD];
    {
        float* keep_pad = s1f_pad + threadIdx.x;
        asm volatile("" ::"v"(keep_pad));
    }
#endif

    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nt = a.nt, n_off = nt * (nt - 1) / 2;
    const int per_p = n_off * a.ns_off + nt * a.ns_diag;
    int lin = xcd_remap(blockIdx.x, gridDim.x);
    const int p = lin / per_p;
    lin -= p * per_p;
    int it, jt, sidx;
    if (lin < n_off * a.ns_off) {
        sidx = lin / n_off;
        const int idx = lin - sidx * n_off;  // idx-th pair (it, jt) with jt < it
        it = 1;
        while (it * (it + 1) / 2 <= idx) ++it;
        jt = idx - it * (it - 1) / 2;
    } else {
        lin -= n_off * a.ns_off;
        sidx = lin / nt;
        it = jt = lin - sidx * nt;
    }
    const int tri = it * (it + 1) / 2 + jt;
    const int slab = (tri - it) * a.ns_off + it * a.ns_diag + sidx;
    if (it != jt) {
        syrk1f_body<false, 0>(a, lds, wl, p, it, jt, sidx, slab, per_p, w);
    } else if (w == 0) {
        syrk1f_body<true, 0>(a, lds, wl, p, it, jt, sidx, slab, per_p, w);
    } else if (w == 1) {
        syrk1f_body<true, 1>(a, lds, wl, p, it, jt, sidx, slab, per_p, w);
    } else if (w == 2) {
        syrk1f_body<true, 2>(a, lds, wl, p, it, jt, sidx, slab, per_p, w);
    } else {
        syrk1f_body<true, 3>(a, lds, wl, p, it, jt, sidx, slab, per_p, w);
    }
}

// Sums the partial tiles over the splits in a fixed order and writes the full symmetric matrix (fp64).
// One block per 16-row strip of a 128 x 128 tile: the sums go out row by row (1 KB runs) and, through an LDS image of the
// strip, mirrored as 128 rows of 16 (128-byte runs).  (Up to round 5 a block took two rows and every thread wrote its mirrored
// element on its own: a column of 8-byte stores 8 KB apart, 67 us for M = 1024 where the 41 MB it moves are ~10 us.)
constexpr int SR_ROWS = 16;
template <typename T>
__global__ __launch_bounds__(NTHREADS) void syrk_reduce_kernel(const T* __restrict__ part2,
                                                               const T* __restrict__ part1,
                                                               double* __restrict__ acc2, double* __restrict__ acc1,
                                                               int Mp, int P, int nt, int ns_off, int ns_diag) {
    // grid.x = ntri * (128 / SR_ROWS) + blocks for acc1, grid.y = P
    __shared__ double img[SR_ROWS][TILE + 1];
    constexpr int SUBS = TILE / SR_ROWS;
    const int p = blockIdx.y;
    const int ntri = nt * (nt + 1) / 2;
    const int per_p = (ntri - nt) * ns_off + nt * ns_diag;
    const int tri = blockIdx.x / SUBS, sub = blockIdx.x % SUBS;
    const int t = threadIdx.x;
    if (tri < ntri) {
        int it = 0;
        while ((it + 1) * (it + 2) / 2 <= tri) ++it;
        const int jt = tri - it * (it + 1) / 2;
        const int ns = (it == jt) ? ns_diag : ns_off;
        const int slab0 = (tri - it) * ns_off + it * ns_diag;
        const size_t base = (size_t)p * Mp * Mp;
        const int jj = t & 127, rbase = t >> 7;
        constexpr int RPT = SR_ROWS * 128 / NTHREADS;  // rows per thread (8): their loads of one split are in flight together
        double acc[RPT];
#pragma unroll
        for (int q = 0; q < RPT; ++q) acc[q] = 0.0;
        const T* src = part2 + ((size_t)p * per_p + slab0) * (TILE * TILE) + (size_t)(sub * SR_ROWS + rbase) * TILE + jj;
        for (int sp = 0; sp < ns; ++sp) {
#pragma unroll
            for (int q = 0; q < RPT; ++q) acc[q] += (double)src[(size_t)sp * (TILE * TILE) + q * (NTHREADS / 128) * TILE];
        }
#pragma unroll
        for (int q = 0; q < RPT; ++q) {
            const int r = rbase + q * (NTHREADS / 128), ii = sub * SR_ROWS + r;
            // diagonal tile: only 16x16 blocks with column block <= row block were accumulated; inside the diagonal
            // blocks keep the lower triangle; everything is mirrored below (exactly symmetric output)
            if (it == jt && jj > ii) continue;
            acc2[base + (size_t)(it * TILE + ii) * Mp + jt * TILE + jj] = acc[q];
            img[r][jj] = acc[q];
        }
        __syncthreads();
        // mirrored: row jt * 128 + c, columns it * 128 + 16 sub + (0 .. 15); two threads per row, eight columns each
        const int c = t >> 1, r0 = (t & 1) * (SR_ROWS / 2);
        double* out = acc2 + base + (size_t)(jt * TILE + c) * Mp + it * TILE + sub * SR_ROWS + r0;
#pragma unroll
        for (int r = 0; r < SR_ROWS / 2; ++r) {
            if (it == jt && c > sub * SR_ROWS + r0 + r) continue;  // (the image holds nothing there: that element is a direct write)
            out[r] = img[r0 + r][c];
        }
    } else {
        const int idx = (blockIdx.x - ntri * SUBS) * NTHREADS + t;
        if (idx < Mp) {
            double s = 0.0;
            for (int sp = 0; sp < ns_diag; ++sp) s += (double)part1[((size_t)p * ns_diag + sp) * Mp + idx];
            acc1[(size_t)p * Mp + idx] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Blocked Cholesky of the M x M site matrices (K_uu + jitter I, W, the moments Gram, -2 lambda_2 + jitter I):
// replaces tf.linalg.cholesky of reference src/models/tsvgp.py:270,300 and src/util.py:377-388 on the GPU.
// Right-looking, 128-wide block columns, three kernels per block column k:
//   potrf_diag_kernel   one workgroup: factor the 128x128 diagonal block in LDS / registers (32-wide sub-blocks),
//                       write L_kk, then assemble inv(L_kk) and write it to `work`
//   chol_tile_kernel<PANEL>   A[i,k] <- A[i,k] * inv(L_kk)^T            (one wave per 32x32 tile below the diagonal)
//   chol_tile_kernel<UPDATE>  A[i,j] <- A[i,j] - A[i,k] * A[j,k]^T      (one wave per trailing lower 32x32 tile)  Latency bound by design (M^3/3 flops is
// microseconds of MFMA time): the critical path is 8 diagonal blocks, each ~M/8 dependent column steps.
// ---------------------------------------------------------------------------------------------------------------
#ifndef TSVGP_CHOL_SB
#define TSVGP_CHOL_SB 16
#endif
constexpr int CH_NB = 128;
constexpr int CH_SB = TSVGP_CHOL_SB;  // sub-block of the diagonal block factored in one wave's registers
constexpr int CH_WT = 32;          // wave tile of the panel / update / inverse-assembly products
constexpr int CH_LD = CH_NB + 1;   // odd LDS row stride: one-lane-per-row column sweeps touch 32 different banks
constexpr int CH_THREADS = 512;    // 8 waves

// 1/sqrt(x) to fp64 accuracy: hardware estimate (v_rsq_f64, 2^-24 measured) + ONE third-order step
//   e = 1 - x y^2,  y <- y + y e (1/2 + 3/8 e):  max rel err 1.4e-16 over 2^20 arguments (tools/rsq_probe.hip), four
// dependent fp64 ops instead of the six of two Newton steps (2.4e-16).  x <= 0 gives NaN without a select: v_rsq_f64
// returns NaN for x < 0 and +inf for x = 0, and inf * 0 in the residual is NaN.
__device__ __forceinline__ double rsqrt_nr(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);
}

// value of `v` in lane `src` (compile-time lane), as a wave-uniform double
__device__ __forceinline__ double readlane_d(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// One pass over the block column of the 16-wide sub-block at (s0, s0) of S, one lane per row, the rows in registers.
// Lanes 0..15 of EVERY participating wave hold the sub-block's own rows (factored redundantly per wave, so that no wave
// waits for another); lanes 16..63 of wave w hold 48 further rows: the rows below the sub-block and, behind them, the
// 16 rows of the identity.  Four columns per step: the 4x4 pivot block is broadcast with v_readlane and factored
// redundantly by every lane (reciprocal square roots by v_rsq_f64 + one third-order step, rsqrt_nr: no f64 divide / sqrt sequences on
// the critical path), each lane solves its own row against it, and the rank-4 update of the remaining columns takes
// its second factor from a small per-wave LDS scratch (wave-uniform reads).  For a row below the sub-block that IS the
// substitution x L_ss^T = a, and for row i of the identity it yields column i of inv(L_ss): the row solves and the
// sub-block inverse cost no pass of their own.  No workgroup barriers inside.
// Rows below the sub-block are written back here; the sub-block's rows and the inverse stay in `a` for
// chol_store_sb, which the caller runs behind a barrier (the other waves load the unfactored sub-block).
struct ChRole {
    int sub, below, ident;  // ident: row of the identity held by this lane, -1 for none
};
__device__ __forceinline__ ChRole chol_factor_rows(double* __restrict__ S, double* __restrict__ dinv, double* __restrict__ xs,
                                                   int s0, int lane, int w, int need_inverse, int* fail, int col_base,
                                                   double (&a)[CH_SB]) {
    const int nbelow = CH_NB - s0 - CH_SB;
    const int idx = (64 - CH_SB) * w + lane - CH_SB;
    ChRole role;
    role.sub = lane < CH_SB;
    role.below = !role.sub && idx < nbelow;
    role.ident = (!role.sub && !role.below && need_inverse && idx < nbelow + CH_SB) ? idx - nbelow : -1;
    double* const row = S + (role.sub ? s0 + lane : s0 + CH_SB + idx) * CH_LD + s0;
    if (role.sub || role.below) {
#pragma unroll
        for (int c = 0; c < CH_SB; ++c) a[c] = row[c];
    } else {
#pragma unroll
        for (int c = 0; c < CH_SB; ++c) a[c] = (c == role.ident) ? 1.0 : 0.0;
    }
    int bad = 0;
#ifdef TSVGP_DIAG_POTRF
    extern __shared__ __attribute__((aligned(16))) unsigned char diag_raw[];
    unsigned long long* fst = reinterpret_cast<unsigned long long*>(diag_raw + (size_t)CH_NB * CH_LD * sizeof(double));
    int nf = 0;
#define FSTAMP() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (w == 0 && (s0 == 0 || s0 == 80)) { unsigned long long tt = __builtin_amdgcn_s_memtime(); if (lane == 0) fst[(s0 == 0 ? 0 : 32) + nf] = tt; } ++nf; }
#else
#define FSTAMP()
#endif
    FSTAMP()
#pragma unroll
    for (int j0 = 0; j0 < CH_SB; j0 += 4) {
        const double p00 = readlane_d(a[j0], j0), p10 = readlane_d(a[j0], j0 + 1), p20 = readlane_d(a[j0], j0 + 2),
                     p30 = readlane_d(a[j0], j0 + 3);
        const double p11 = readlane_d(a[j0 + 1], j0 + 1), p21 = readlane_d(a[j0 + 1], j0 + 2),
                     p31 = readlane_d(a[j0 + 1], j0 + 3);
        const double p22 = readlane_d(a[j0 + 2], j0 + 2), p32 = readlane_d(a[j0 + 2], j0 + 3);
        const double p33 = readlane_d(a[j0 + 3], j0 + 3);
        const double i0 = rsqrt_nr(p00);
        const double l10 = p10 * i0, l20 = p20 * i0, l30 = p30 * i0;
        const double d1 = p11 - l10 * l10;
        const double i1 = rsqrt_nr(d1);
        const double l21 = (p21 - l20 * l10) * i1, l31 = (p31 - l30 * l10) * i1;
        const double d2 = p22 - l20 * l20 - l21 * l21;
        const double i2 = rsqrt_nr(d2);
        const double l32 = (p32 - l30 * l20 - l31 * l21) * i2;
        const double d3 = p33 - l30 * l30 - l31 * l31 - l32 * l32;
        const double i3 = rsqrt_nr(d3);
#ifdef TSVGP_DIAG_POTRF
        { double q = i3; asm volatile("" : "+v"(q)); }
        FSTAMP()
#endif
        if (bad == 0) bad = !(p00 > 0.0) ? j0 + 1 : !(d1 > 0.0) ? j0 + 2 : !(d2 > 0.0) ? j0 + 3 : !(d3 > 0.0) ? j0 + 4 : 0;
        // own row against the pivot block; for the pivot rows this reproduces the factor's rows
        // (x_q = d_q * rsqrt(d_q) = sqrt(d_q) on the diagonal)
        const double x0 = a[j0] * i0;
        const double x1 = (a[j0 + 1] - x0 * l10) * i1;
        const double x2 = (a[j0 + 2] - x0 * l20 - x1 * l21) * i2;
        const double x3 = (a[j0 + 3] - x0 * l30 - x1 * l31 - x2 * l32) * i3;
        a[j0] = x0;
        a[j0 + 1] = x1;
        a[j0 + 2] = x2;
        a[j0 + 3] = x3;
#ifdef TSVGP_DIAG_POTRF
        { double q = x3; asm volatile("" : "+v"(q)); }
        FSTAMP()
#endif
        if (lane == 0 && w == 0) {
            dinv[s0 + j0] = i0;
            dinv[s0 + j0 + 1] = i1;
            dinv[s0 + j0 + 2] = i2;
            dinv[s0 + j0 + 3] = i3;
        }
        if (j0 + 4 < CH_SB) {
            // the four new columns of the sub-block's rows go through a 512-byte LDS scratch (one per wave): row c's
            // entries come back as one wave-uniform 32-byte read per column of the update (two LDS reads instead of
            // eight v_readlane per column)
            if (lane < CH_SB) *reinterpret_cast<v4d*>(xs + 4 * lane) = v4d{x0, x1, x2, x3};
            __builtin_amdgcn_wave_barrier();  // one wave, in-order LDS: the reads below see the writes
            // the second factors are requested four columns at a time before the first FMA of the group (8 ds_read_b128
            // in flight: one LDS latency per group, not one per column).  Not all of a step's at once: the kernel has to
            // stay within 112 VGPRs, see potrf_diag_kernel.
#pragma unroll
            for (int c0 = j0 + 4; c0 < CH_SB; c0 += 4) {
                v4d y[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) y[c] = *reinterpret_cast<const v4d*>(xs + 4 * (c0 + c));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    double v = a[c0 + c];
                    v = fma(-x0, y[c][0], v);
                    v = fma(-x1, y[c][1], v);
                    v = fma(-x2, y[c][2], v);
                    v = fma(-x3, y[c][3], v);
                    // pin the update here: otherwise the optimiser sinks the FMAs towards the final stores and the
                    // operands of the whole step stay live
                    asm volatile("" : "+v"(v));
                    a[c0 + c] = v;
                }
            }
            __builtin_amdgcn_wave_barrier();  // the next step's writes come after these reads
            FSTAMP()
        }
    }
    FSTAMP()
#undef FSTAMP
    if (role.below) {
#pragma unroll
        for (int c = 0; c < CH_SB; ++c) row[c] = a[c];
    }
    if (lane == 0 && w == 0 && bad != 0 && *fail == 0) *fail = col_base + s0 + bad;  // first non-positive pivot, 1-based
    return role;
}

// The sub-block's factor (wave 0) and the inverse of it: X[c][i] = a[c] of identity row i, parked transposed in the
// unused strict upper triangle of the sub-block (S[s0 + i][s0 + c], c > i); the diagonal of the inverse is dinv.
__device__ __forceinline__ void chol_store_sb(double* __restrict__ S, int s0, int lane, int w, const ChRole& role,
                                              const double (&a)[CH_SB]) {
    if (role.sub && w == 0) {
        double* const row = S + (s0 + lane) * CH_LD + s0;
#pragma unroll
        for (int c = 0; c < CH_SB; ++c)
            if (c <= lane) row[c] = a[c];
    } else if (role.ident >= 0) {
        double* const prow = S + (s0 + role.ident) * CH_LD + s0;
#pragma unroll
        for (int c = 1; c < CH_SB; ++c)
            if (c > role.ident) prow[c] = a[c];
    }
}

// X[r][c] of the inverse under construction: diagonal in dinv, strict lower part parked transposed above the diagonal
__device__ __forceinline__ double chol_xval(const double* __restrict__ S, const double* __restrict__ dinv, int r, int c) {
    const double* p = (r > c) ? S + c * CH_LD + r : dinv + r;  // one LDS read either way
    const double v = *p;
    return (r >= c) ? v : 0.0;
}

// Off-diagonal part of the inverse by 2x2 recursion,  X_lo = -X_hh (L_hl X_ll),  for the block with rows [h0, h0+n)
// and columns [l0, l0+n), n = 16 NT.  One wave per 16-column tile J of the result: T(:, J) = L_hl X_ll(:, J) stays in
// the accumulators and is fed straight back as the B operand of the second product (for v_mfma_f64_16x16x4 the C
// layout of a 16x16 tile IS the B-operand layout of its four k-steps), so the two stages need no exchange.
template <int NT>
__device__ __forceinline__ void chol_inv_offdiag(double* __restrict__ S, const double* __restrict__ dinv, int h0, int l0,
                                                 int J, int lane) {
    const int li = lane & 15, lk = lane >> 4;
    v4d T[NT];
#pragma unroll
    for (int I = 0; I < NT; ++I) T[I] = v4d{0, 0, 0, 0};
    for (int kk = 4 * J; kk < 4 * NT; ++kk) {  // X_ll is lower triangular: k-steps below column tile J contribute zeros
        const double bq = chol_xval(S, dinv, l0 + 4 * kk + lk, l0 + 16 * J + li);
#pragma unroll
        for (int I = 0; I < NT; ++I)
            T[I] = Mfma<double>::run(S[(h0 + 16 * I + li) * CH_LD + l0 + 4 * kk + lk], bq, T[I]);
    }
    v4d O[NT];
#pragma unroll
    for (int I = 0; I < NT; ++I) {
        O[I] = v4d{0, 0, 0, 0};
#pragma unroll
        for (int Kt = 0; Kt < NT; ++Kt) {
            if (Kt > I) continue;  // X_hh is lower triangular
#pragma unroll
            for (int q = 0; q < 4; ++q)
                O[I] = Mfma<double>::run(-chol_xval(S, dinv, h0 + 16 * I + li, h0 + 16 * Kt + 4 * q + lk), T[Kt][q], O[I]);
        }
    }
#pragma unroll
    for (int I = 0; I < NT; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r)  // X[h0 + i][l0 + j] parked at S[l0 + j][h0 + i]
            S[(l0 + 16 * J + li) * CH_LD + h0 + 16 * I + lk + 4 * r] = O[I][r];
}

// Diagonal block: right-looking over eight 16-wide sub-blocks, everything in LDS / registers.  Round s:
//   (1) waves 0..2 factor the 16x16 diagonal sub-block in registers and, in the same pass, solve the rows below against
//       it and invert it (chol_factor_rows: 64 rows per wave, the sub-block redundantly in each); the other waves
//       meanwhile apply the PREVIOUS phase's update, A_ij -= L_is L_js^T in 16x16 MFMA tiles, to the block columns
//       beyond this one;
//   (2) all waves apply this phase's update to the next block column only (one tile each).
// Then L_kk is written out and inv(L_kk) is assembled from the sub-block inverses by three levels of the 2x2
// recursion (chol_inv_offdiag) and written to `work` for the panel kernel.
// FUSED (round 5): the panel rows below the block are solved by the SAME launch.  grid = (batch, workgroups of eight 16-row
// strips): every workgroup factors the diagonal block for itself -- redundantly: the factorisation is one workgroup's serial
// work either way, and nothing has to cross workgroups -- and then each of its waves takes one strip through the substitution
// of chol_panel2_kernel (tsvgp_chol.hip) with the factor-side operands read straight from the LDS image of the factor.  The
// strip is requested at the kernel's start (its eight tiles wait in registers behind the 26 us of pivot chains) and leaves
// through the LDS image, which is dead by then, as whole row halves.  EXPERIMENT (TSVGP_POTRF_FUSE), measured and not the
// default: one launch fewer per block step, but the substitution fed from LDS (conditional reads of the parked inverses, no
// operand prefetch) takes longer than the panel kernel fed from `work`: 436 against 412 us at M = 1024
// (profiles/r05_chain_ab.txt); ~150 registers per lane.
template <bool FUSED>
__global__ __launch_bounds__(CH_THREADS) void potrf_diag_kernel(double* __restrict__ A, int lda, int64_t stride, int k,
                                                                double* __restrict__ work, int* __restrict__ info,
                                                                int need_inverse, double* __restrict__ Xout,
                                                                double* __restrict__ Xtout, int ldx, int64_t xstride,
                                                                int nstrips) {
    // These few waves are the critical path of the M x M prelude while the K(X, Z) fill of the same step fills every CU
    // from a side stream: ask the SIMD arbiter to issue them first.  The fill (96 VGPRs, three waves per SIMD) leaves 224
    // registers per SIMD lane: with at most 112 VGPRs this workgroup (two waves per SIMD) is placed on a CU the fill
    // occupies without waiting for one of its workgroups to retire -- at 144 a factorisation under the fill took 0.2 ms
    // longer (A/B on one box; tools/kres.py prints the count).
    __builtin_amdgcn_s_setprio(TSVGP_CHOL_PRIO);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* S = reinterpret_cast<double*>(smem_raw);  // [CH_NB][CH_LD]; lower: A -> L, strict upper: inv(L)^T
    __shared__ int fail;
    __shared__ double dinv[CH_NB];
    __shared__ __attribute__((aligned(32))) double xs[4 * 4 * CH_SB];  // one scratch per participating wave
    const int t = threadIdx.x, lane = t & 63, b = blockIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    double* Ab = A + (size_t)b * stride + (size_t)k * CH_NB * lda + (size_t)k * CH_NB;
    // FUSED: this wave's strip of the panel, in the MFMA accumulator layout (lane (n, G), tile s, register r <-> [n][16 s + 4 r + G])
    const int strip = FUSED ? (int)blockIdx.y * (CH_THREADS / 64) + w : 0;
    const bool has_strip = FUSED && strip < nstrips;
    const bool writes_block = !FUSED || blockIdx.y == 0;
    double* Arow0 = Ab + (size_t)(CH_NB + 16 * strip) * lda;  // (row (k + 1) * 128 + 16 strip, column k * 128)
    v4d U[FUSED ? CH_NB / 16 : 1];
    if constexpr (FUSED) {
        if (has_strip) {
#pragma unroll
            for (int s_ = 0; s_ < CH_NB / 16; ++s_)
#pragma unroll
                for (int r = 0; r < 4; ++r) U[s_][r] = Arow0[(size_t)(lane & 15) * lda + 16 * s_ + 4 * r + (lane >> 4)];
        }
    }
#ifdef TSVGP_DIAG_POTRF
    unsigned long long stamp[40];
    int nstamp = 0;
#define PSTAMP() stamp[nstamp++] = __builtin_amdgcn_s_memtime();
#else
#define PSTAMP()
#endif
    PSTAMP()
    if (t == 0) fail = 0;
    // 16-byte loads, eight in flight per thread; the strict upper triangle is read and dropped
#pragma unroll
    for (int it0 = 0; it0 < CH_NB * CH_NB / 2 / CH_THREADS; it0 += 8) {
        v2d v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = t + (it0 + u) * CH_THREADS, r = idx >> 6, c = (idx & 63) * 2;
            v[u] = *reinterpret_cast<const v2d*>(Ab + (size_t)r * lda + c);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int idx = t + (it0 + u) * CH_THREADS, r = idx >> 6, c = (idx & 63) * 2;
            S[r * CH_LD + c] = (c <= r) ? v[u][0] : 0.0;
            S[r * CH_LD + c + 1] = (c + 1 <= r) ? v[u][1] : 0.0;
        }
    }
    __syncthreads();
    PSTAMP()

    // one 16x16 tile of a trailing update: A[i0.., j0..] -= L[i0.., c0..c0+16) L[j0.., c0..c0+16)^T (lower part only: the
    // strict upper triangles of the diagonal tiles hold the sub-block inverses)
    auto update_tile = [&](int c0, int i0, int j0) {
        int lane_ = lane;
        asm volatile("" : "+v"(lane_));
        const int li = lane_ & 15, lk = lane_ >> 4;
        v4d acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = S[(i0 + lk + 4 * r) * CH_LD + j0 + li];
#pragma unroll
        for (int kk = 0; kk < CH_SB / 4; ++kk)
            acc = Mfma<double>::run(-S[(i0 + li) * CH_LD + c0 + 4 * kk + lk], S[(j0 + li) * CH_LD + c0 + 4 * kk + lk], acc);
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (j0 + li <= i0 + lk + 4 * r) S[(i0 + lk + 4 * r) * CH_LD + j0 + li] = acc[r];
    };
    // At the top of round s the block column of sub-block s carries the updates of all earlier phases, the block columns
    // beyond it those of all but the last: that one is applied during the round by the waves that do not factor.
    for (int s0 = 0; s0 < CH_NB; s0 += CH_SB) {
        const int nbelow = CH_NB - s0 - CH_SB, n16 = nbelow / 16;
        const int nrows = nbelow + (need_inverse ? CH_SB : 0);
        const int ntake = nrows > 0 ? (nrows + 63 - CH_SB) / (64 - CH_SB) : 1;  // 16 + 48 rows per factoring wave
        const bool takes = w < ntake;
        double a[CH_SB];
        ChRole role{0, 0, -1};
        if (takes) {
            role = chol_factor_rows(S, dinv, xs + 4 * CH_SB * w, s0, lane, w, need_inverse, &fail, k * CH_NB, a);
        } else if (s0 > 0) {
            const int ntile = n16 * (n16 + 1) / 2;
            for (int ti = w - ntake; ti < ntile; ti += CH_THREADS / 64 - ntake) {
                int I = 0;
                while ((I + 1) * (I + 2) / 2 <= ti) ++I;
                const int J = ti - I * (I + 1) / 2;
                update_tile(s0 - CH_SB, s0 + CH_SB + 16 * I, s0 + CH_SB + 16 * J);
            }
        }
        __syncthreads();
        if (takes) chol_store_sb(S, s0, lane, w, role, a);  // nothing reads these rows before the next barrier
        PSTAMP()
        // this phase's update of the NEXT block column, one tile per wave; the rest waits for the next round
        if (w < n16) update_tile(s0, s0 + CH_SB + 16 * w, s0 + CH_SB);
        __syncthreads();
        PSTAMP()
    }

    // L_kk out (zeros above the diagonal)
    if (writes_block) {
#pragma unroll 4
        for (int it = 0; it < CH_NB * CH_NB / 2 / CH_THREADS; ++it) {
            const int idx = t + it * CH_THREADS, r = idx >> 6, c = (idx & 63) * 2;
            v2d v;
            v[0] = (c <= r) ? S[r * CH_LD + c] : 0.0;
            v[1] = (c + 1 <= r) ? S[r * CH_LD + c + 1] : 0.0;
            *reinterpret_cast<v2d*>(Ab + (size_t)r * lda + c) = v;
        }
        // (block 0 initialises the status word -- no memset launch in front of the factorisation -- later blocks keep the first failure)
        if (t == 0 && (k == 0 || (fail != 0 && info[b] == 0))) info[b] = fail;
    }
    PSTAMP()
    if constexpr (FUSED) {
        // P = A_panel inv(L_kk)^T by substitution over the 16-wide column blocks (chol_panel2_kernel's recurrence):
        //   P_s = U_s inv(L_ss)^T,  U_s' -= P_s L_s's^T (s' > s); operands: lane (n, G), k-step kk <-> row n, column 4 kk + G of the
        //   tile -- inv(L_ss) from the strict upper triangle + dinv (chol_xval), L_s's from the lower part of the image
        static_assert(CH_SB == 16 && CH_NB == 128, "fused panel");
        const int n = lane & 15, G = lane >> 4;
        if (has_strip) {
#pragma unroll
            for (int s_ = 0; s_ < CH_NB / 16; ++s_) {
                v4d ps = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
                    ps = Mfma<double>::run(chol_xval(S, dinv, 16 * s_ + n, 16 * s_ + 4 * kk + G), U[s_][kk], ps);
#pragma unroll
                for (int s2 = s_ + 1; s2 < CH_NB / 16; ++s2)
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        U[s2] = Mfma<double>::run(-S[(16 * s2 + n) * CH_LD + 16 * s_ + 4 * kk + G], ps[kk], U[s2]);
                U[s_] = ps;
            }
        }
        __syncthreads();  // every wave has read its operands: the image is dead and becomes the strips' way out
        if (has_strip) {
            constexpr int OLD = 66;  // [16][66] per wave and 64-column half: 8-byte accesses of a half wave on 32 bank pairs
            double* img = S + (size_t)w * 16 * OLD;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
                    for (int r = 0; r < 4; ++r) img[n * OLD + 16 * s_ + 4 * r + G] = U[4 * h + s_][r];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const double* d = img + (2 * i + (lane >> 5)) * OLD + 2 * (lane & 31);
                    v2d v;
                    v[0] = d[0];
                    v[1] = d[1];
                    *reinterpret_cast<v2d*>(Arow0 + (size_t)(2 * i + (lane >> 5)) * lda + 64 * h + 2 * (lane & 31)) = v;
                }
            }
        }
    } else if (need_inverse == 2) {
        // round 5: no assembled inverse.  The panel kernel (tsvgp_chol.hip: chol_panel2_kernel) solves by substitution over the
        // 16-wide column blocks and wants the factor's off-diagonal tiles in MFMA register layout and the inverted diagonal
        // sub-blocks (which the factor passes left in the strict upper triangles + dinv) -- tsvgp_chol.h: the `work` image
        static_assert(CH_SB == 16 && CH_NB == 128, "work image of the panel kernel");
        double* Wb = work + (size_t)b * CH_NB * CH_NB;
        for (int idx = t; idx < 28 * 256; idx += CH_THREADS) {  // the 28 tiles below the diagonal tiles
            int q = idx >> 8, j = 0;
            while (q >= 7 - j) {
                q -= 7 - j;
                ++j;
            }
            const int i = j + 1 + q, r = (idx >> 6) & 3, l = idx & 63;
            Wb[(size_t)tsvgp_chol::work_tile_index(j, i - j) * 256 + r * 64 + l] = S[(16 * i + (l & 15)) * CH_LD + 16 * j + 4 * r + (l >> 4)];
        }
        for (int idx = t; idx < 8 * 256; idx += CH_THREADS) {
            const int s0 = 16 * (idx >> 8), r = (idx >> 4) & 15, c = idx & 15;
            Wb[(size_t)tsvgp_chol::WORK_TILES * 256 + idx] = chol_xval(S, dinv, s0 + r, s0 + c);
        }
    } else if (need_inverse) {  // the last block column has no panel below it
        // level 1: the two 64x64 diagonal blocks, X_10 = -X_11 (L_10 X_00) with 32x32 blocks; level 2: the 64x64 block
        // levels n = CH_SB .. 64 of the 2x2 recursion; a level has 64 / n block pairs of n / 16 column tiles: four waves
        if constexpr (CH_SB == 16) {
            if (w < 4) chol_inv_offdiag<1>(S, dinv, 32 * w + 16, 32 * w, 0, lane);
            __syncthreads();
        }
        if (w < 4) chol_inv_offdiag<2>(S, dinv, 64 * (w >> 1) + 32, 64 * (w >> 1), w & 1, lane);
        __syncthreads();
        if (w < 4) chol_inv_offdiag<4>(S, dinv, 64, 0, w, lane);
        __syncthreads();
        PSTAMP()
        double* Wb = work + (size_t)b * CH_NB * CH_NB;
        for (int idx = t; idx < CH_NB * CH_NB; idx += CH_THREADS) {
            const int r = idx >> 7, c = idx & 127;
            Wb[idx] = chol_xval(S, dinv, r, c);
        }
        if (Xout) {  // the block of inv(L) and of its transpose, for tsvgp_potrf_inv_f64
            double* Xb = Xout + (size_t)b * xstride + (size_t)k * CH_NB * ldx + (size_t)k * CH_NB;
            double* Xtb = Xtout + (size_t)b * xstride + (size_t)k * CH_NB * ldx + (size_t)k * CH_NB;
            for (int idx = t; idx < CH_NB * CH_NB; idx += CH_THREADS) {
                const int r = idx >> 7, c = idx & 127;
                Xb[(size_t)r * ldx + c] = chol_xval(S, dinv, r, c);
                Xtb[(size_t)r * ldx + c] = chol_xval(S, dinv, c, r);
            }
        }
    }
#ifdef TSVGP_DIAG_POTRF
    __syncthreads();
    PSTAMP()
#ifndef TSVGP_DIAG_POTRF_K
#define TSVGP_DIAG_POTRF_K 0
#endif
    if (t == 0 && k == TSVGP_DIAG_POTRF_K && need_inverse) {
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(work + (size_t)b * CH_NB * CH_NB);
        dst[0] = nstamp;
        for (int i = 0; i < nstamp; ++i) dst[1 + i] = stamp[i];
        const unsigned long long* fst = reinterpret_cast<const unsigned long long*>(smem_raw + (size_t)CH_NB * CH_LD * sizeof(double));
        for (int i = 0; i < 64; ++i) dst[64 + i] = fst[i];
    }
#endif
#undef PSTAMP
}

// Tile products for the panel solve and the trailing update, one WAVE per 32x32 output tile, K = 128, no LDS and no
// barriers in the product: the sum over k does not care which lane group supplies which k, so lane (i = l&15, g = l>>4)
// streams the CONTIGUOUS run k in [64h + 16g, 64h + 16g + 16) of its operand row straight from global memory / L2
// (8 x 16-byte loads) and MFMA step kk of half h uses element kk of every lane's run, for A and B alike.
//   OP 0 (panel):  A[i, k] <- A[i, k] * inv(L_kk)^T   one workgroup = the four 32-column tiles of a 32-row block; the
//                  block is overwritten in place, so all four waves finish reading before any of them stores
//   OP 1 (update): A[i, j] <- A[i, j] - A[i, k] * A[j, k]^T   over the lower 32x32 tiles of the trailing matrix
//   Right-hand-side rows (tsvgp_potrf_solve_f64): `ext32` further 32-row blocks directly below row M of the same buffer ride along
//   as panel rows -- OP 0 takes them as more row blocks of the panel (the grid is simply longer), OP 1 as a rectangle of
//   ext32 x nb32 tiles behind the triangle -- so that after the last block step they hold  B L^-T.
template <int OP>
__global__ __launch_bounds__(NTHREADS) void chol_tile_kernel(double* __restrict__ Amat, int lda, int64_t stride, int k,
                                                             int nt, const double* __restrict__ work, int ext32) {
    __builtin_amdgcn_s_setprio(TSVGP_CHOL_PRIO);
    const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    double* Ab = Amat + (size_t)b * stride;
    const int nb32 = (nt - k - 1) * (CH_NB / CH_WT);
    const int base = (k + 1) * CH_NB;  // first row / column of the trailing matrix
    int ti = 0, tj = 0;
    bool active = true;
    if (OP == 0) {
        ti = blockIdx.x;
        tj = w;
    } else {
        const int wid = blockIdx.x * (NTHREADS / 64) + w;
        const int ntri = nb32 * (nb32 + 1) / 2;
        active = wid < ntri + ext32 * nb32;
        if (wid < ntri) {
            int r = (int)((sqrtf(8.0f * (float)wid + 1.0f) - 1.0f) * 0.5f);
            while (r * (r + 1) / 2 > wid) --r;
            while ((r + 1) * (r + 2) / 2 <= wid) ++r;
            ti = r;
            tj = wid - r * (r + 1) / 2;
        } else {  // right-hand-side rows: row block nb32 + e (the first row behind the matrix is row base + 32 nb32 = M)
            const int r = wid - ntri, q = nb32 > 0 ? r / nb32 : 0;
            ti = nb32 + q;
            tj = r - q * nb32;
        }
    }
    v4d acc[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[s][n] = v4d{0, 0, 0, 0};
    double* Cb = Ab + (size_t)(base + CH_WT * ti) * lda + (OP == 0 ? (size_t)k * CH_NB : (size_t)base) + CH_WT * tj + li;
    if (OP == 1 && active) {
        // round 5: the tile itself is requested first and accumulated on (-A B^T with the sign on the A operand): its 16
        // eight-byte loads per lane used to follow the products as a read-modify-write and sat on the kernel's tail
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[s][n][r] = Cb[(size_t)(16 * s + g + 4 * r) * lda + 16 * n];
    }
    if (active) {
        // Round 5: which k a lane group supplies to which MFMA step is free (both operands use the same map), and the map decides
        // what one load instruction touches.  Lane (i, g) used to stream the contiguous run [64 h + 16 g, + 16) of its row: 64
        // separate 16-byte pieces per instruction.  Now the four lanes of a row take ADJACENT pairs -- load q covers k in
        // [64 h + 8 q, + 8) of 16 rows, 64 contiguous bytes each -- a quarter of the segments (scattered accesses queue on the
        // CU's one address path at ~200 cycles each, profiles/r05_potrf_diag_lab.txt).
        const double* Arow = Ab + (size_t)(base + CH_WT * ti + li) * lda + (size_t)k * CH_NB + 2 * g;
        const double* Brow = (OP == 0) ? work + (size_t)b * CH_NB * CH_NB + (size_t)(CH_WT * tj + li) * CH_NB + 2 * g
                                       : Ab + (size_t)(base + CH_WT * tj + li) * lda + (size_t)k * CH_NB + 2 * g;
        const size_t bstep = (OP == 0) ? (size_t)16 * CH_NB : (size_t)16 * lda;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            v2d ra[2][8], rb[2][8];
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    ra[s][q] = *reinterpret_cast<const v2d*>(Arow + (size_t)16 * s * lda + 64 * h + 8 * q);
                    rb[s][q] = *reinterpret_cast<const v2d*>(Brow + s * bstep + 64 * h + 8 * q);
                }
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[s][n] = Mfma<double>::run(OP == 1 ? -ra[s][kk >> 1][kk & 1] : ra[s][kk >> 1][kk & 1],
                                                      rb[n][kk >> 1][kk & 1], acc[s][n]);
        }
    }
    if (OP == 0) __syncthreads();  // in place: every wave of the row block has its operands in registers
    if (!active) return;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double* Cr = Cb + (size_t)(16 * s + g + 4 * r) * lda;
#pragma unroll
            for (int n = 0; n < 2; ++n) Cr[16 * n] = acc[s][n][r];
        }
}

// Panel solve by SUBSTITUTION (TSVGP_POTRF_SUBST): X L_kk^T = A_ik for the rows below the diagonal block, without the
// inverse of L_kk.  chol_tile_kernel<0> multiplies by inv(L_kk), which costs a factor cond(L_kk) of accuracy in the
// trailing matrix: harmless for the site matrices of a well-conditioned K_uu, but a numerically barely definite matrix
// (cond ~ 1e14, lambda_min ~ 30 eps lambda_max: the new Lambda_2 on a K_uu with cond 1e10) then loses its definiteness
// at the first pivot of the next block where a LAPACK-style factorisation goes through.  Here, as in LAPACK's blocked
// trsm, only 16 x 16 diagonal sub-blocks are solved against directly (one lane per row, forward substitution, IEEE
// division) and the rest of the row block is updated with MFMA tiles:
//   for s in sub-blocks:  X[:, s] = A[:, s] L_ss^-T ;  A[:, >s] -= X[:, s] L[>s, s]^T
// One workgroup per PS_ROWS rows of the panel; the rows and the current 16-wide column block of L_kk live in LDS.
constexpr int PS_ROWS = 64;
constexpr int PS_LD = CH_SB + 1;
__global__ __launch_bounds__(NTHREADS) void chol_panel_subst_kernel(double* __restrict__ Amat, int lda, int64_t stride, int k) {
    static_assert(CH_SB == 16, "sub-block width of the substitution panel");
    __builtin_amdgcn_s_setprio(TSVGP_CHOL_PRIO);
    __shared__ double Pn[PS_ROWS][CH_LD];
    __shared__ double Lc[CH_NB][PS_LD];
    __shared__ double dinv[CH_SB];
    const int t = threadIdx.x, lane = t & 63, li = lane & 15, lk = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    double* Ab = Amat + (size_t)blockIdx.y * stride;
    const size_t row0 = (size_t)(k + 1) * CH_NB + (size_t)blockIdx.x * PS_ROWS, col0 = (size_t)k * CH_NB;
    for (int idx = t; idx < PS_ROWS * CH_NB / 2; idx += NTHREADS) {
        const int r = idx >> 6, c = (idx & 63) * 2;
        const v2d v = *reinterpret_cast<const v2d*>(Ab + (row0 + r) * lda + col0 + c);
        Pn[r][c] = v[0];
        Pn[r][c + 1] = v[1];
    }
    for (int s0 = 0; s0 < CH_NB; s0 += CH_SB) {
        const int nr = CH_NB - s0;  // rows of L_kk from the sub-block down
        __syncthreads();           // the previous update has read Lc; the first pass: Pn is complete
        for (int idx = t; idx < nr * CH_SB; idx += NTHREADS) {
            const int r = idx >> 4, c = idx & 15;
            Lc[r][c] = Ab[(col0 + s0 + r) * lda + col0 + s0 + c];
        }
        __syncthreads();
        if (t < CH_SB) dinv[t] = 1.0 / Lc[t][t];
        __syncthreads();
        if (t < PS_ROWS) {  // one lane per row: x_c = (a_c - sum_{j<c} x_j L_ss[c][j]) / L_ss[c][c]
            double a[CH_SB];
#pragma unroll
            for (int c = 0; c < CH_SB; ++c) a[c] = Pn[t][s0 + c];
#pragma unroll
            for (int c = 0; c < CH_SB; ++c) {
                double v[4] = {a[c], 0.0, 0.0, 0.0};
#pragma unroll
                for (int j = 0; j < c; ++j) v[j & 3] = fma(-a[j], Lc[c][j], v[j & 3]);
                a[c] = ((v[0] + v[1]) + (v[2] + v[3])) * dinv[c];
            }
#pragma unroll
            for (int c = 0; c < CH_SB; ++c) Pn[t][s0 + c] = a[c];
        }
        __syncthreads();
        // A[:, s0 + 16 + 16 cb ..] -= X[:, s0 .. s0 + 16] * L[s0 + 16 + 16 cb .., s0 .. s0 + 16]^T on 16 x 16 MFMA tiles
        const int ncb = (nr - CH_SB) / 16, ntile = (PS_ROWS / 16) * ncb;
        for (int ti = w; ti < ntile; ti += NTHREADS / 64) {
            const int rb = ti / ncb, cb = ti - rb * ncb;
            const int c0 = s0 + CH_SB + 16 * cb;
            v4d acc;
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] = Pn[16 * rb + lk + 4 * r][c0 + li];
#pragma unroll
            for (int kk = 0; kk < CH_SB / 4; ++kk)
                acc = Mfma<double>::run(-Pn[16 * rb + li][s0 + 4 * kk + lk], Lc[CH_SB + 16 * cb + li][4 * kk + lk], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) Pn[16 * rb + lk + 4 * r][c0 + li] = acc[r];
        }
    }
    __syncthreads();
    for (int idx = t; idx < PS_ROWS * CH_NB / 2; idx += NTHREADS) {
        const int r = idx >> 6, c = (idx & 63) * 2;
        v2d v;
        v[0] = Pn[r][c];
        v[1] = Pn[r][c + 1];
        *reinterpret_cast<v2d*>(Ab + (row0 + r) * lda + col0 + c) = v;
    }
}

// inv(L) from the inverted diagonal blocks by the 2x2 recursion, one level per launch pair:
//   [[L00, 0], [L10, L11]]^-1 = [[X00, 0], [-X11 L10 X00, X11]]      (blocks of n rows; the second may be shorter)
// with X (lower, row-major) and Xt = X^T kept side by side so that every product is of the form A * B^T with both
// operands read along contiguous k (the wave-tile scheme of chol_tile_kernel):
//   STAGE 0:  T^T[j, i] =  sum_k Xt00[j, k] L10[i, k]        (stored at block (0, 1) of the scratch matrix T)
//   STAGE 1:  X10[i, j] = -sum_k X11[i, k] T^T[j, k]         (stored to X and, transposed, to Xt)
// The k-ranges skip the zero halves of the triangular operands (Xt00 is upper, X11 lower).
template <int STAGE>
__global__ __launch_bounds__(NTHREADS) void trtri_level_kernel(const double* __restrict__ Lm, int lda, int64_t strideA,
                                                               double* __restrict__ X, double* __restrict__ Xt,
                                                               double* __restrict__ T, int ldx, int64_t strideX, int M,
                                                               int n) {
    // One workgroup per 32x32 output tile; its four waves split the k range (64-wide chunks, round robin) and the
    // partial tiles are summed through LDS: a lone wave per tile would sit out one load latency per chunk.
    __builtin_amdgcn_s_setprio(TSVGP_CHOL_PRIO);
    __shared__ v4d part[NTHREADS / 64][4][64];
    const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair = blockIdx.y, b = blockIdx.z;
    const int r0 = 2 * pair * n, r1 = r0 + n;
    const int n1 = min(n, M - r1);
    if (n1 <= 0) return;
    const int nrow32 = (STAGE == 0 ? n : n1) / CH_WT, ncol32 = (STAGE == 0 ? n1 : n) / CH_WT;
    const int wid = blockIdx.x;
    if (wid >= nrow32 * ncol32) return;
    const int tr = wid / ncol32, tc = wid - tr * ncol32;  // output tile: rows tr, columns tc (units of 32)
    const double* Lb = Lm + (size_t)b * strideA;
    double* Xb = X + (size_t)b * strideX;
    double* Xtb = Xt + (size_t)b * strideX;
    double* Tb = T + (size_t)b * strideX;
    // operands: rows of A (output rows) and rows of B (output columns), both read along k
    const double *Arow, *Brow;
    size_t astep, bstep;
    int kbeg, kend;
    if (STAGE == 0) {
        Arow = Xtb + (size_t)(r0 + CH_WT * tr + li) * ldx + r0;  // Xt00[j, k]: zero for k < j
        Brow = Lb + (size_t)(r1 + CH_WT * tc + li) * lda + r0;    // L10[i, k]
        astep = (size_t)16 * ldx;
        bstep = (size_t)16 * lda;
        kbeg = (CH_WT * tr) & ~63;
        kend = n;
    } else {
        Arow = Xb + (size_t)(r1 + CH_WT * tr + li) * ldx + r1;  // X11[i, k]: zero for k > i
        Brow = Tb + (size_t)(r0 + CH_WT * tc + li) * ldx + r1;  // T^T[j, k]
        astep = (size_t)16 * ldx;
        bstep = (size_t)16 * ldx;
        kbeg = 0;
        kend = min(n1, (CH_WT * (tr + 1) + 63) & ~63);
    }
    v4d acc[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 2; ++q) acc[s][q] = v4d{0, 0, 0, 0};
    for (int k0 = kbeg + 64 * w; k0 < kend; k0 += 64 * (NTHREADS / 64)) {
        v2d ra[2][8], rb[2][8];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                ra[s][q] = *reinterpret_cast<const v2d*>(Arow + s * astep + k0 + 16 * g + 2 * q);
                rb[s][q] = *reinterpret_cast<const v2d*>(Brow + s * bstep + k0 + 16 * g + 2 * q);
            }
#pragma unroll
        for (int kk = 0; kk < 16; ++kk)
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    acc[s][q] = Mfma<double>::run(ra[s][kk >> 1][kk & 1], rb[q][kk >> 1][kk & 1], acc[s][q]);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 2; ++q) part[w][2 * s + q][lane] = acc[s][q];
    __syncthreads();
    // wave w finishes the 16x16 sub-tile (s, q) = (w >> 1, w & 1)
    const int s = w >> 1, q = w & 1;
    v4d sum = part[0][w][lane];
#pragma unroll
    for (int u = 1; u < NTHREADS / 64; ++u) sum += part[u][w][lane];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int orow = CH_WT * tr + 16 * s + g + 4 * r, ocol = CH_WT * tc + 16 * q + li;
        if (STAGE == 0) {
            Tb[(size_t)(r0 + orow) * ldx + r1 + ocol] = sum[r];
        } else {
            const double v = -sum[r];
            Xb[(size_t)(r1 + orow) * ldx + r0 + ocol] = v;
            Xtb[(size_t)(r0 + ocol) * ldx + r1 + orow] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Small fused M x M helpers of the replicated prelude / epilogue (each replaces a chain of ~5 us elementwise launches).
// ---------------------------------------------------------------------------------------------------------------
// 16-byte zero fill (hipMemsetAsync takes 23-31 us for the 8 MB inverse-factor buffers; this takes the time of the stores)
__global__ __launch_bounds__(NTHREADS) void zero_fill_kernel(v2d* __restrict__ p, size_t n2) {
    for (size_t i = (size_t)blockIdx.x * NTHREADS + threadIdx.x; i < n2; i += (size_t)gridDim.x * NTHREADS) p[i] = v2d{0.0, 0.0};
}
inline void zero_fill(void* p, size_t bytes, hipStream_t st) {  // bytes a multiple of 16, p 16-byte aligned
    const size_t n2 = bytes / 16;
    if (n2 == 0) return;
    const unsigned grid = (unsigned)((n2 + NTHREADS - 1) / NTHREADS < 2048 ? (n2 + NTHREADS - 1) / NTHREADS : 2048);
    hipLaunchKernelGGL(zero_fill_kernel, dim3(grid), dim3(NTHREADS), 0, st, reinterpret_cast<v2d*>(p), n2);
}

// dst[b][i][j] = keep(i, j) ? scale * src[b][si][sj] : 0 on the leading M x M block.
//   flip == 0: (si, sj) = (i, j),                 keep = (i >= j)   -- the lower factor out of the factorisation buffer
//   flip == 1: (si, sj) = (M - 1 - i, M - 1 - j), keep = (i <= j)   -- U = J C J, the upper-form factor (util.rev_cholesky)
//   flip == 2: (si, sj) = (M - 1 - i, M - 1 - j), keep everything   -- J A J, the input of that factorisation
__global__ __launch_bounds__(NTHREADS) void tri_copy_kernel(const double* __restrict__ src, int lds_, int64_t sstride,
                                                            double* __restrict__ dst, int ldd, int64_t dstride, int M,
                                                            double scale, int flip, double diag_add) {
    __builtin_amdgcn_s_setprio(1);
    const int b = blockIdx.z, i = blockIdx.y;
    const int j = blockIdx.x * NTHREADS + threadIdx.x;
    if (j >= M) return;
    const bool keep = flip >= 2 ? true : flip ? (i <= j) : (i >= j);
    double v = 0.0;
    if (keep) {
        const bool rev = flip == 1 || flip == 2;
        const int si = rev ? M - 1 - i : i, sj = rev ? M - 1 - j : j;
        v = scale * src[(size_t)b * sstride + (size_t)si * lds_ + sj];
        if (i == j) v += diag_add;
    }
    dst[(size_t)b * dstride + (size_t)i * ldd + j] = v;
}

// dst[b][i][j] = (i <= j) ? src[b][M - 1 - j][M - 1 - i] : 0  -- transpose and index reversal in one pass, through a padded
// LDS tile so that both sides are coalesced.  What turns the solved right-hand-side rows of tsvgp_potrf_solve_f64,
// (J L J) C^-T = (C^-1 J L^T J)^T, into the upper triangular D = U_W^-1 L^T = J (C^-1 J L^T J) J of reference src/util.py:173.
__global__ __launch_bounds__(NTHREADS) void flip_transpose_kernel(const double* __restrict__ src, int lds_, int64_t sstride,
                                                                  double* __restrict__ dst, int ldd, int64_t dstride, int M) {
    __shared__ double tile[32][33];
    const int b = blockIdx.z, i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if (j0 + 31 < i0) {  // wholly below the diagonal: zeros
        for (int r = ty; r < 32; r += 8)
            if (i0 + r < M && j0 + tx < M) dst[(size_t)b * dstride + (size_t)(i0 + r) * ldd + j0 + tx] = 0.0;
        return;
    }
    for (int r = ty; r < 32; r += 8) {  // tile[r][c] = src[M - 1 - (j0 + r)][M - 1 - (i0 + c)]: c runs along a source row
        const int sr = M - 1 - (j0 + r), sc = M - 1 - (i0 + tx);
        tile[r][tx] = (sr >= 0 && sc >= 0) ? src[(size_t)b * sstride + (size_t)sr * lds_ + sc] : 0.0;
    }
    __syncthreads();
    for (int c = ty; c < 32; c += 8) {  // dst[i0 + c][j0 + r] = tile[r][c]
        const int i = i0 + c, j = j0 + tx;
        if (i < M && j < M) dst[(size_t)b * dstride + (size_t)i * ldd + j] = (i <= j) ? tile[tx][c] : 0.0;
    }
}

// target[p][i][j] = c_ll * LLt[p][i][j] + c_g * s * G1s[i][j] + jitter * (i == j),  G1s = (G1 + G1^T) / 2 (also written out),
// s = num_data / rows (rows: a device scalar, the all-reduced row count) or 1 when num_data <= 0.
// The matrix of the final factorisation of one E-step, reference src/models/tsvgp.py:286-300:
//   -2 [(1 - lr) lambda_2 + lr scale G1] + jitter I  with lambda_2 = -1/2 L L^T  ->  c_ll = 1 - lr, c_g = -2 lr.
__global__ __launch_bounds__(NTHREADS) void site_target_kernel(const double* __restrict__ G1, const double* __restrict__ LLt,
                                                               double* __restrict__ target, double* __restrict__ G1s, int M,
                                                               double c_ll, double c_g, double jitter,
                                                               const double* __restrict__ rows, double num_data) {
    __shared__ double tile[32][33];
    const size_t base = (size_t)blockIdx.z * M * M;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    for (int r = ty; r < 32; r += 8) {  // the transposed tile G1[j0.., i0..]
        const int jj = j0 + r, ii = i0 + tx;
        tile[r][tx] = (jj < M && ii < M) ? G1[base + (size_t)jj * M + ii] : 0.0;
    }
    __syncthreads();
    const double s = num_data > 0.0 ? num_data / rows[0] : 1.0;
    for (int r = ty; r < 32; r += 8) {
        const int i = i0 + r, j = j0 + tx;
        if (i < M && j < M) {
            const size_t o = base + (size_t)i * M + j;
            const double g = 0.5 * (G1[o] + tile[tx][r]);
            G1s[o] = g;
            target[o] = c_ll * LLt[o] + c_g * s * g + (i == j ? jitter : 0.0);
        }
    }
}

// The elementwise part of the site update in TWO launches (round 4; it was site_target_kernel + a gemv + ~10 torch launches of
// 4-6 us each on the replicated critical path), reference src/util.py:429-438 and src/models/tsvgp.py:284-300:
//   Gs      = (G1 + G1^T) / 2
//   target  = (1 - lr) L L^T - 2 lr s Gs + jitter I                    (the matrix of the final factorisation, :293-300)
//   l1_new  = (1 - lr) l1_old + lr s (G0 - 2 Gs meanZ)                  (:284, :296; util.py:436)
// with s = num_data / rows (device scalar) or 1.  site_update_kernel: one workgroup per 32 x 32 tile (the transposed tile
// through LDS, as site_target_kernel) writes the tile of `target` and the tile's share of Gs meanZ, 32 partial row sums, to
// `part` [P][M/32 column tiles][M]; site_update_finish_kernel adds the column tiles in a fixed order (bitwise reproducible)
// and finishes lambda_1.  (A first version swept a row block's tiles in ONE workgroup: 32 workgroups, 124 us at M = 1024.)
__global__ __launch_bounds__(NTHREADS) void site_update_kernel(const double* __restrict__ G1, const double* __restrict__ LLt,
                                                               const double* __restrict__ meanZ, double* __restrict__ target,
                                                               double* __restrict__ part, int M, int P, double lr, double jitter,
                                                               const double* __restrict__ rows, double num_data) {
    __shared__ double tile[32][33];
    __shared__ double mz[32];
    const int p = blockIdx.z;
    const size_t base = (size_t)p * M * M;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const double s = num_data > 0.0 ? num_data / rows[0] : 1.0;
    const double c_ll = 1.0 - lr, c_g = -2.0 * lr * s;
    for (int r = ty; r < 32; r += 8) {  // the transposed tile G1[j0.., i0..]
        const int jj = j0 + r, ii = i0 + tx;
        tile[r][tx] = (jj < M && ii < M) ? G1[base + (size_t)jj * M + ii] : 0.0;
    }
    if (threadIdx.x < 32) mz[threadIdx.x] = (j0 + threadIdx.x < M) ? meanZ[(size_t)(j0 + threadIdx.x) * P + p] : 0.0;
    __syncthreads();
    const int ntj = (M + 31) / 32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = ty + 8 * q, i = i0 + r, j = j0 + tx;
        double v = 0.0;
        if (i < M && j < M) {
            const size_t o = base + (size_t)i * M + j;
            const double g = 0.5 * (G1[o] + tile[tx][r]);
            target[o] = c_ll * LLt[o] + c_g * g + (i == j ? jitter : 0.0);
            v = g * mz[tx];
        }
        v += __shfl_xor(v, 16);  // over the 32 columns of the tile (a half wave): fixed order
        v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 4);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 1);
        if (tx == 0 && i < M) part[((size_t)p * ntj + blockIdx.x) * M + i] = v;
    }
}
__global__ __launch_bounds__(NTHREADS) void site_update_finish_kernel(const double* __restrict__ part, const double* __restrict__ G0,
                                                                      const double* __restrict__ l1_old, double* __restrict__ l1_new,
                                                                      int M, int P, double lr, const double* __restrict__ rows,
                                                                      double num_data) {
    const int i = blockIdx.x * NTHREADS + threadIdx.x, p = blockIdx.y;
    if (i >= M) return;
    const double s = num_data > 0.0 ? num_data / rows[0] : 1.0;
    const int ntj = (M + 31) / 32;
    double v = 0.0;
    for (int jb = 0; jb < ntj; ++jb) v += part[((size_t)p * ntj + jb) * M + i];
    const size_t o = (size_t)i * P + p;
    l1_new[o] = (1.0 - lr) * l1_old[o] + lr * s * (G0[o] - 2.0 * v);
}

// beta = l1 - D^T (D v) per latent, D [P][M][M] upper triangular, v = K6 l1 [M][P]  (K6^-1 m of reference src/util.py:176-179 in the
// form of t_SVGP._site_operands): the two triangular matrix-vector products that stand between the factorisation and the
// moments kernel, in three small launches (they were gemv + copy + gemv + copy + multiply + fill + reduce + copy + add: ~75 us of
// 5-14 us launches).  STAGE 0: t[i] = sum_{j >= i} D[i][j] v[j], one wave per row.  STAGE 1: the column sums of D[i][j] t[i]
// over 64-row groups, one workgroup per (64 columns, 64 rows) -- many small workgroups: under the N-sized fill of the same step a
// launch of 16 workgroups took 79 us -- to part [P][M / 64][M].  STAGE 2: beta[j] = l1[j] - sum of the row groups in order.
template <int STAGE>
__global__ __launch_bounds__(NTHREADS) void site_beta_kernel(const double* __restrict__ D, const double* __restrict__ v,
                                                             const double* __restrict__ l1, double* __restrict__ tvec,
                                                             double* __restrict__ part, double* __restrict__ beta, int M, int P) {
    const int p = blockIdx.z;
    const double* Dp = D + (size_t)p * M * M;
    const int nrg = (M + 63) / 64;
    if (STAGE == 0) {
        const int lane = threadIdx.x & 63, i = blockIdx.x * (NTHREADS / 64) + (threadIdx.x >> 6);
        if (i >= M) return;
        double acc = 0.0;
        for (int j = (i & ~63) + lane; j < M; j += 64)
            if (j >= i) acc += Dp[(size_t)i * M + j] * v[(size_t)j * P + p];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) tvec[(size_t)p * M + i] = acc;
    } else if (STAGE == 1) {
        __shared__ double sums[4][64];
        const int c = threadIdx.x & 63, g = threadIdx.x >> 6, j = blockIdx.x * 64 + c, rg = blockIdx.y;
        double acc = 0.0;
        if (rg <= (int)blockIdx.x && j < M) {  // row groups beyond the column strip lie below the diagonal
            // the wave's 16 rows requested at once (round 4 walked them one load at a time: 69 us at M = 1024, all of it latency)
            double d[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = rg * 64 + g + 4 * u;
                d[u] = (i <= j && i < M) ? Dp[(size_t)i * M + j] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int i = rg * 64 + g + 4 * u;
                acc = fma(d[u], i < M ? tvec[(size_t)p * M + i] : 0.0, acc);
            }
        }
        sums[g][c] = acc;
        __syncthreads();
        if (g == 0 && j < M) part[((size_t)p * nrg + rg) * M + j] = ((sums[0][c] + sums[1][c]) + sums[2][c]) + sums[3][c];
    } else {
        const int j = blockIdx.x * NTHREADS + threadIdx.x;
        if (j >= M) return;
        double acc = 0.0;
        for (int rg = 0; rg <= j / 64; ++rg) acc += part[((size_t)p * nrg + rg) * M + j];
        beta[(size_t)j * P + p] = l1[(size_t)j * P + p] - acc;
    }
}

// y[:, p] = A_p v[:, p] for row-major A_p [M x M] (p-th matrix at A + p * strideA; strideA = 0: one matrix for all latents) and
// v, y [M x P]: one wave per row, 16-byte loads, eight in flight per lane.  (rocBLAS takes 24-30 us for this 8 MB read.)
__global__ __launch_bounds__(NTHREADS) void gemv_rows_kernel(const double* __restrict__ A, int64_t strideA, const double* __restrict__ v,
                                                             double* __restrict__ y, int M, int P) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * (NTHREADS / 64) + (threadIdx.x >> 6), p = blockIdx.y;
    if (i >= M) return;
    const double* row = A + (size_t)p * strideA + (size_t)i * M;
    double acc = 0.0;
    if ((M & 1) == 0 && (reinterpret_cast<uintptr_t>(row) & 15) == 0) {
        for (int j0 = 0; j0 < M; j0 += 1024) {
            v2d a[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + 2 * (lane + 64 * u);
                a[u] = j < M ? *reinterpret_cast<const v2d*>(row + j) : v2d{0.0, 0.0};
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + 2 * (lane + 64 * u);
                if (j < M) acc = fma(a[u][0], v[(size_t)j * P + p], fma(a[u][1], v[(size_t)(j + 1) * P + p], acc));
            }
        }
    } else {
        for (int j = lane; j < M; j += 64) acc = fma(row[j], v[(size_t)j * P + p], acc);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) y[(size_t)i * P + p] = acc;
}

// flags[0] = sum |info_a|, flags[1] = nonpos (as is, NaN included), flags[2] = sum |info_b|   (t_SVGP._status_flags)
// ---------------------------------------------------------------------------------------------------------------
// Clock keeper (round 5).  The chip's clock follows its load with a time constant of several milliseconds: behind 1.5 ms of
// the latency-bound M x M chain (a handful of workgroups at a time) the moments kernel starts near 2.05 GHz and climbs back
// towards the 2.3-2.4 GHz it holds back-to-back -- 2.22 ms instead of 1.95 for a 125 000-row launch, 16.5 instead of 15.2 at
// N = 1e6 (profiles/r05_clock_lab.txt, tools/clock_lab.py).  A kernel that keeps the vector ALUs of every CU issuing fp64 FMAs
// on registers over that stretch -- one wave per SIMD, lowest priority, no memory traffic but the poll of a flag -- holds the
// clock (1.95 ms again).  It runs on a side stream beside the chain and leaves when the main stream raises the flag in front of
// the N-pass, or after max_ticks of the 100 MHz real-time counter, whichever comes first: every wave reaches one of the two.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS) void clock_keeper_kernel(const int* __restrict__ flag, unsigned max_ticks) {
    __shared__ int leave;
    __builtin_amdgcn_s_setprio(0);
    if (threadIdx.x == 0) leave = 0;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    const bool poller = threadIdx.x < 64;  // wave 0 polls the flag (one load per workgroup and round), the others read LDS
    double x0 = 1.0 + 1e-9 * threadIdx.x, x1 = x0 + 0.25, x2 = x0 + 0.5, x3 = x0 + 0.75;
    const double a = 1.0 - 0x1p-40, b = 0x1p-40;
    for (;;) {
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {  // 4 x 128 FMAs per lane: ~2 us of the SIMD's fp64 issue slots per round
                x0 = __builtin_fma(x0, a, b);
                x1 = __builtin_fma(x1, a, b);
                x2 = __builtin_fma(x2, a, b);
                x3 = __builtin_fma(x3, a, b);
            }
        }
        if (poller) {
            if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                __builtin_amdgcn_s_memrealtime() - t0 >= max_ticks) {
                if (threadIdx.x == 0) __hip_atomic_store(&leave, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                break;
            }
        } else if (__hip_atomic_load(&leave, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != 0 ||
                   __builtin_amdgcn_s_memrealtime() - t0 >= max_ticks) {
            break;
        }
    }
    asm volatile("" ::"v"(x0), "v"(x1), "v"(x2), "v"(x3));
}
__global__ void keeper_signal_kernel(int* flag, int value) {
    __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ void step_status_kernel(const int* __restrict__ info_a, int na, const int* __restrict__ info_b, int nb,
                                   const double* __restrict__ nonpos, double* __restrict__ flags) {
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < na; ++i) a += fabs((double)info_a[i]);
        for (int i = 0; i < nb; ++i) b += fabs((double)info_b[i]);
        flags[0] = a;
        flags[1] = nonpos ? nonpos[0] : 0.0;
        flags[2] = b;
    }
}

// Lower triangle of the symmetric accumulators, row by row, [P][M (M + 1) / 2]: what the all-reduce of the dual
// accumulators ships (half of P M^2); unpack mirrors it back.
__global__ __launch_bounds__(NTHREADS) void sym_pack_kernel(const double* __restrict__ A, int lda, int64_t stride, int M,
                                                            double* __restrict__ out, int unpack) {
    const int i = blockIdx.y, p = blockIdx.z;
    const int j = blockIdx.x * NTHREADS + threadIdx.x;
    if (j > i) return;
    const size_t tri = (size_t)M * (M + 1) / 2;
    double* o = out + (size_t)p * tri + (size_t)i * (i + 1) / 2 + j;
    double* a = const_cast<double*>(A) + (size_t)p * stride;
    if (!unpack) {
        *o = a[(size_t)i * lda + j];
    } else {
        const double v = *o;
        a[(size_t)i * lda + j] = v;
        a[(size_t)j * lda + i] = v;
    }
}

// single-wave MFMA map self-test
template <typename T>
__global__ void selftest_kernel(const T* a, const T* b, T* c) {
    typedef typename Mfma<T>::acc_t acc_t;
    const int lane = threadIdx.x & 63;
    acc_t acc = acc_t{0, 0, 0, 0};
    const T av = a[(lane & 15) * 4 + (lane >> 4)];
    const T bv = b[(lane >> 4) * 16 + (lane & 15)];
    acc = Mfma<T>::run(av, bv, acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) c[Mfma<T>::row(lane, r) * 16 + (lane & 15)] = acc[r];
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? TSVGP_OK : TSVGP_ELAUNCH; }

// Kernels that use more than the default 64 KB of dynamic LDS have to opt in, per kernel and per DEVICE.  The only
// process-wide state of this library: one "already asked" flag per (kernel, device); setting the attribute twice is
// harmless, so concurrent first calls need no lock.
constexpr int MAX_DEVICES = 64;
struct DynLdsOptIn {
    std::atomic<unsigned char> done[MAX_DEVICES];
    int ensure(const void* fn, size_t bytes) {
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0) return TSVGP_ELAUNCH;
        const bool tracked = dev < MAX_DEVICES;
        if (tracked && done[dev].load(std::memory_order_acquire)) return TSVGP_OK;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return TSVGP_ELAUNCH;
        if (tracked) done[dev].store(1, std::memory_order_release);
        return TSVGP_OK;
    }
};

// K(X, Z) for input dimensions beyond the fill kernel's compile-time sizes (D > 32, e.g. the 784 pixels of the reference's
// MNIST notebook): at that size the scaled distance IS a GEMM, r2 = |x~|^2 + |z~|^2 - 2 x~.z~ (GPflow's own square_distance
// form [ext]), which the host takes from the BLAS library; this kernel turns the Gram block G = x~ z~^T into
// K = variance * k(r2) in place and zero-fills the padding.  One thread per two adjacent columns.
template <typename T, int KIND>
__global__ __launch_bounds__(NTHREADS) void gram_to_kernel_kernel(T* __restrict__ K, const T* __restrict__ xx,
                                                                  const T* __restrict__ zz, T variance, int64_t N, int M,
                                                                  int64_t ldk, int64_t rows_pad, int cols_pad) {
    typedef typename Mfma<T>::pair_t pair_t;
    const int64_t n = blockIdx.y;
    const int m = (blockIdx.x * NTHREADS + threadIdx.x) * 2;
    if (n >= rows_pad || m >= cols_pad) return;
    pair_t* p = reinterpret_cast<pair_t*>(K + n * ldk + m);
    pair_t out;
    out[0] = out[1] = T(0);
    if (n < N) {
        const pair_t g = *p;
        const T x = xx[n];
        if (m < M) out[0] = variance * kernel_profile<KIND>(x + zz[m] - T(2) * g[0]);
        if (m + 1 < M) out[1] = variance * kernel_profile<KIND>(x + zz[m + 1] - T(2) * g[1]);
    }
    *p = out;
}

template <typename T>
int gram_to_kernel(int kind, T* K, const T* xx, const T* zz, T variance, int64_t N, int M, int64_t ldk, void* stream) {
    if (!K || !xx || !zz || N <= 0 || M <= 0 || (kind != TSVGP_KERNEL_SE && kind != TSVGP_KERNEL_MATERN32 && kind != TSVGP_KERNEL_MATERN52))
        return TSVGP_EINVAL;
    const int64_t rows_pad = (N + TILE - 1) / TILE * TILE;
    const int cols_pad = (M + TILE - 1) / TILE * TILE;
    if (ldk < cols_pad || (ldk % 2) != 0 || rows_pad > 0x7fffffff) return TSVGP_EINVAL;
    // grid.y is limited to 65535: rows go through y (and z when there are more)
    const dim3 block(NTHREADS);
    const unsigned gx = (unsigned)((cols_pad / 2 + NTHREADS - 1) / NTHREADS);
    for (int64_t r0 = 0; r0 < rows_pad; r0 += 65535) {
        const unsigned gy = (unsigned)((rows_pad - r0 < 65535) ? rows_pad - r0 : 65535);
        T* Kr = K + r0 * ldk;
        const T* xr = xx + (r0 < N ? r0 : 0);
        const int64_t Nr = N - r0 > 0 ? N - r0 : 0;
#define TSVGP_G2K(KIND_) hipLaunchKernelGGL((gram_to_kernel_kernel<T, KIND_>), dim3(gx, gy), block, 0, (hipStream_t)stream, Kr, xr, zz, variance, Nr, M, ldk, (int64_t)gy, cols_pad)
        if (kind == TSVGP_KERNEL_SE) TSVGP_G2K(TSVGP_KERNEL_SE);
        else if (kind == TSVGP_KERNEL_MATERN32) TSVGP_G2K(TSVGP_KERNEL_MATERN32);
        else TSVGP_G2K(TSVGP_KERNEL_MATERN52);
#undef TSVGP_G2K
    }
    return launch_status();
}

// M-step gradient for input dimensions beyond kgrad_kernel's compile-time sizes (D > 16): the contraction with dK/d(theta, Z)
// in the same GEMM form as the fill above.  With s = |x~|^2 + |z~|^2 - 2 G (G = x~ z~^T from the BLAS library), V = g0 beta^T -
// 2 g1 * U and W = -2 variance V * k'(s), everything N-sized that is left is
//     d variance = sum V k(s);    W^T x~ [M, D], the row and column sums of W     (BLAS GEMM + two reductions, on the host side)
// and dZ, d lengthscales follow in M x D (estep.kernel_grad).  This kernel turns G into W IN PLACE (padding zero) and writes
// one partial sum of V k(s) per workgroup.  One thread per two adjacent columns, one row per blockIdx.y.
template <typename T, int KIND>
__global__ __launch_bounds__(NTHREADS) void gram_to_gradw_kernel(T* __restrict__ G, const T* __restrict__ xx,
                                                                 const T* __restrict__ zz, T variance, const T* __restrict__ U,
                                                                 int64_t ldu, const T* __restrict__ g0, const T* __restrict__ g1,
                                                                 int gstride, const T* __restrict__ beta, int bstride, int64_t N,
                                                                 int M, int64_t ldk, int64_t rows_pad, int cols_pad,
                                                                 double* __restrict__ vpart) {
    typedef typename Mfma<T>::pair_t pair_t;
    __shared__ double red[NTHREADS / 64];
    const int64_t n = blockIdx.y;
    const int m = (blockIdx.x * NTHREADS + threadIdx.x) * 2;
    double va = 0.0;
    if (n < rows_pad && m < cols_pad) {
        pair_t* p = reinterpret_cast<pair_t*>(G + n * ldk + m);
        pair_t out;
        out[0] = out[1] = T(0);
        if (n < N) {
            const pair_t g = *p;
            const pair_t u = *reinterpret_cast<const pair_t*>(U + n * ldu + m);
            const T x = xx[n], gg0 = g0[n * gstride], gg1 = g1[n * gstride];
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (m + q < M) {
                    T f, df;
                    kernel_profile_grad<KIND>(x + zz[m + q] - T(2) * g[q], f, df);
                    const T v = gg0 * beta[(int64_t)(m + q) * bstride] - T(2) * gg1 * u[q];
                    va += (double)(v * f);
                    out[q] = T(-2) * variance * v * df;
                }
        }
        *p = out;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) va += __shfl_xor(va, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = va;
    __syncthreads();
    if (threadIdx.x == 0) vpart[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

template <typename T>
int gram_to_gradw(int kind, T* G, const T* xx, const T* zz, T variance, const T* U, int64_t ldu, const T* g0, const T* g1,
                  int gstride, const T* beta, int bstride, int64_t N, int M, int64_t ldk, double* vpart, void* stream) {
    if (!G || !xx || !zz || !U || !g0 || !g1 || !beta || !vpart || N <= 0 || M <= 0 || gstride <= 0 || bstride <= 0 ||
        (kind != TSVGP_KERNEL_SE && kind != TSVGP_KERNEL_MATERN32 && kind != TSVGP_KERNEL_MATERN52))
        return TSVGP_EINVAL;
    const int64_t rows_pad = (N + TILE - 1) / TILE * TILE;
    const int cols_pad = (M + TILE - 1) / TILE * TILE;
    if (ldk < cols_pad || (ldk % 2) != 0 || ldu < cols_pad || (ldu % 2) != 0 || rows_pad > 0x7fffffff) return TSVGP_EINVAL;
    const dim3 block(NTHREADS);
    const unsigned gx = (unsigned)((cols_pad / 2 + NTHREADS - 1) / NTHREADS);
    for (int64_t r0 = 0; r0 < rows_pad; r0 += 65535) {
        const unsigned gy = (unsigned)((rows_pad - r0 < 65535) ? rows_pad - r0 : 65535);
        const int64_t Nr = N - r0 > 0 ? N - r0 : 0;
        const int64_t ro = r0 < N ? r0 : 0;  // rows past N are only zero-filled: no per-row operand is read
#define TSVGP_G2W(KIND_)                                                                                                         \
    hipLaunchKernelGGL((gram_to_gradw_kernel<T, KIND_>), dim3(gx, gy), block, 0, (hipStream_t)stream, G + r0 * ldk, xx + ro, zz,  \
                       variance, U + ro * ldu, ldu, g0 + ro * gstride, g1 + ro * gstride, gstride, beta, bstride, Nr, M, ldk,      \
                       (int64_t)gy, cols_pad, vpart + r0 * gx)
        if (kind == TSVGP_KERNEL_SE) TSVGP_G2W(TSVGP_KERNEL_SE);
        else if (kind == TSVGP_KERNEL_MATERN32) TSVGP_G2W(TSVGP_KERNEL_MATERN32);
        else TSVGP_G2W(TSVGP_KERNEL_MATERN52);
#undef TSVGP_G2W
    }
    return launch_status();
}

// P latents in one launch: inv_ls [P x D] (device), variance [P] (HOST: the scalars travel as kernel arguments, like the
// single-latent entry point's `variance`), K + p * strideK.
template <typename T>
int kernel_fill(int kind, const T* X, const T* Z, const T* inv_ls, const T* variance, T* K, int64_t strideK, int64_t N,
                int M, int D, int64_t ldk, int P, void* stream) {
    if (!X || !Z || !inv_ls || !variance || !K || N <= 0 || M <= 0 || D <= 0 || P <= 0 || P > TSVGP_MAX_BATCH ||
        (kind != TSVGP_KERNEL_SE && kind != TSVGP_KERNEL_MATERN32 && kind != TSVGP_KERNEL_MATERN52))
        return TSVGP_EINVAL;
    const int64_t rows_pad = (N + TILE - 1) / TILE * TILE;
    const int cols_pad = (M + TILE - 1) / TILE * TILE;
    if (ldk < cols_pad || (ldk % 2) != 0) return TSVGP_EINVAL;
    if (P > 1 && (strideK < rows_pad * ldk || (strideK % 2) != 0)) return TSVGP_EINVAL;
    if (D > 32) return TSVGP_EINVAL;  // input dimensions are padded to a compile-time size (1, 2, 4, 8, 16, 32)
    FillBatch<T> var{};
    for (int q = 0; q < P; ++q) var.v[q] = variance[q];
    // (rows_pad is a multiple of 128: every power of two up to FILL_ROWS divides it)
    static const int rows_big = [] {
        const char* e = getenv("TSVGP_FILL_ROWS_BLK");  // experiment knob: rows per workgroup of the N-sized fill (8, 16, 32, 64)
        const int v = e ? atoi(e) : 0;
        return (v == 8 || v == 16 || v == 32 || v == 64) ? v : FILL_ROWS_DEFAULT;
    }();
    const int rows_blk = rows_pad <= 4096 ? 8 : rows_big;
    // fp32: four columns per thread (16-byte stores) where the layout allows it, else two
    const int DT = D <= 1 ? 1 : D <= 2 ? 2 : D <= 4 ? 4 : D <= 8 ? 8 : D <= 16 ? 16 : 32;
    const bool cpt4 = sizeof(T) == 4 && DT <= 16 && (ldk % 4) == 0 && (strideK % 4) == 0 &&
                      (reinterpret_cast<uintptr_t>(K) & 15) == 0 && (cols_pad % 4) == 0;  // (DT = 32: 128 registers of Z alone)
    const int cols_wg = (cpt4 ? 4 : 2) * NTHREADS;
    dim3 grid((unsigned)((rows_pad + rows_blk - 1) / rows_blk), (unsigned)((cols_pad + cols_wg - 1) / cols_wg), (unsigned)P);
#ifdef TSVGP_FILL_GRID_CAP  // experiment build (tools/exp_overlap2.py): fewer, looping workgroups
    if ((unsigned)(TSVGP_FILL_GRID_CAP) < grid.x) grid.x = (unsigned)(TSVGP_FILL_GRID_CAP);
#endif
    // An output far beyond the caches (4 MB of L2 per XCD, 256 MB of Infinity Cache) is written with non-temporal stores:
    // nothing of it would still be cached when its reader arrives, and the M x M factorisations that run beside the fill of
    // an E-step keep their operands in L2 (under the fill a factorisation call: 0.87 -> 0.80 ms; the step at N = 1e6
    // 36.79 -> 36.45 ms on one box, within noise on two others, never slower).  A small output stays cacheable for its reader.
#ifdef TSVGP_FILL_STREAM  // experiment builds (tools/exp_fill_nt.py, tools/ab_builds.sh): 0 / 1 for every launch
    const int stream_out = TSVGP_FILL_STREAM;
#else
    static const int store_wt = [] { const char* e = getenv("TSVGP_FILL_STORE"); return e && e[0] == 'w' ? 1 : 0; }();
    const int stream_out = (double)rows_pad * (double)ldk * sizeof(T) * P >= 512.0 * 1024 * 1024 ? 1 + store_wt : 0;
#endif
#define TSVGP_FILL_LAUNCH(KIND_, DT_)                                                                                        \
    do {                                                                                                                     \
        if constexpr (sizeof(T) == 4 && DT_ <= 16) {                                                                          \
            if (cpt4) {                                                                                                      \
                hipLaunchKernelGGL((se_fill_kernel<T, KIND_, DT_, 4>), grid, dim3(NTHREADS), 0, (hipStream_t)stream, X, Z,   \
                                   inv_ls, var, K, strideK, N, M, D, ldk, rows_pad, cols_pad, stream_out, rows_blk);         \
                break;                                                                                                       \
            }                                                                                                                \
        }                                                                                                                    \
        hipLaunchKernelGGL((se_fill_kernel<T, KIND_, DT_, 2>), grid, dim3(NTHREADS), 0, (hipStream_t)stream, X, Z, inv_ls,   \
                           var, K, strideK, N, M, D, ldk, rows_pad, cols_pad, stream_out, rows_blk);                         \
    } while (0)
#define TSVGP_FILL_DT(KIND_)                          \
    switch (DT) {                                     \
        case 1: TSVGP_FILL_LAUNCH(KIND_, 1); break;   \
        case 2: TSVGP_FILL_LAUNCH(KIND_, 2); break;   \
        case 4: TSVGP_FILL_LAUNCH(KIND_, 4); break;   \
        case 8: TSVGP_FILL_LAUNCH(KIND_, 8); break;   \
        case 16: TSVGP_FILL_LAUNCH(KIND_, 16); break; \
        default: TSVGP_FILL_LAUNCH(KIND_, 32); break; \
    }
    switch (kind) {
        case TSVGP_KERNEL_SE: TSVGP_FILL_DT(TSVGP_KERNEL_SE) break;
        case TSVGP_KERNEL_MATERN32: TSVGP_FILL_DT(TSVGP_KERNEL_MATERN32) break;
        default: TSVGP_FILL_DT(TSVGP_KERNEL_MATERN52) break;
    }
#undef TSVGP_FILL_DT
#undef TSVGP_FILL_LAUNCH
    return launch_status();
}

// `batch` latents in one launch (grid.y): latent b reads A + b * strideA (0: one shared operand) and Tm + b * strideT and
// writes C + b * strideC.  In place (C == A, strideC == strideA) is allowed for TSVGP_TRI_UPPER only: a workgroup owns
// its 128-row panel and takes the output column tiles in increasing order, tile `it` reads the columns >= 128 * it and
// then overwrites the columns [128 * it, 128 * it + 128), which no later tile reads.
template <typename T>
int trmm(const T* A, int64_t strideA, const T* Tm, int64_t strideT, T* C, int64_t strideC, int64_t Np, int Mp, int mode,
         int batch, void* stream) {
    if (!A || !Tm || !C || Np <= 0 || Mp <= 0 || (Np % TILE) || (Mp % TILE) || mode < 0 || mode > 2 || batch <= 0 ||
        batch > 65535 || strideA < 0 || strideT < 0 || strideC < 0)
        return TSVGP_EINVAL;
    if (batch > 1 && strideC < Np * (int64_t)Mp) return TSVGP_EINVAL;  // outputs of different latents must not overlap
    if ((const void*)A == (const void*)C && (mode != TSVGP_TRI_UPPER || strideA != strideC)) return TSVGP_EINVAL;
    PanelArgs<T> a{};
    a.A = A;
    a.Tm = Tm;
    a.C = C;
    a.N = Np;
    a.Np = Np;
    a.Mp = Mp;
    a.P = 1;
    a.mode = mode;
    a.strideA = strideA;
    a.strideT = strideT;
    a.strideC = strideC;
    const dim3 grid((unsigned)(Np / TILE), (unsigned)batch), block(NTHREADS);
#ifndef TSVGP_LOWER_OLD
    if (mode == TSVGP_TRI_LOWER) {
        static DynLdsOptIn optin1sl;
        if (optin1sl.ensure(reinterpret_cast<const void*>(&panel1_kernel<T, MODE_STORE, TSVGP_TRI_LOWER>), 0) != TSVGP_OK)
            return TSVGP_ELAUNCH;
        hipLaunchKernelGGL((panel1_kernel<T, MODE_STORE, TSVGP_TRI_LOWER>), grid, block, 0, (hipStream_t)stream, a);
    } else
#endif
    if (mode == TSVGP_TRI_LOWER)
        hipLaunchKernelGGL((panel_kernel<T, MODE_STORE, TSVGP_TRI_LOWER>), grid, block, 0, (hipStream_t)stream, a);
#ifndef TSVGP_TRMM_OLD  // (-DTSVGP_TRMM_OLD: A/B builds keep round 2's panel_kernel for the upper product)
    else if (mode == TSVGP_TRI_UPPER) {
        static DynLdsOptIn optin1s;  // 66 KB of static LDS
        if (optin1s.ensure(reinterpret_cast<const void*>(&panel1_kernel<T, MODE_STORE>), 0) != TSVGP_OK) return TSVGP_ELAUNCH;
        hipLaunchKernelGGL((panel1_kernel<T, MODE_STORE>), grid, block, 0, (hipStream_t)stream, a);
    }
#endif
    else if (mode == TSVGP_TRI_UPPER)
        hipLaunchKernelGGL((panel_kernel<T, MODE_STORE, TSVGP_TRI_UPPER>), grid, block, 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((panel_kernel<T, MODE_STORE, TSVGP_TRI_DENSE>), grid, block, 0, (hipStream_t)stream, a);
    return launch_status();
}

template <typename T>
int lik_map(const T* mean, const T* var, const T* Y, int lik, double lik_param, T* g0, T* g1, double* ve_partial,
            int32_t* nonpos_partial, int64_t N, int64_t Np, int P, void* stream) {
    if (!mean || !var || !Y || !g0 || !g1 || !ve_partial || !nonpos_partial || N <= 0 || Np < N || (Np % TILE) || P <= 0)
        return TSVGP_EINVAL;
    if (lik & ~(0xFF | TSVGP_LIK_NOCROP)) return TSVGP_EINVAL;
    const int base = lik & 0xFF;
    if (base != TSVGP_LIK_GAUSSIAN && base != TSVGP_LIK_BERNOULLI) return TSVGP_EINVAL;
    if (base == TSVGP_LIK_GAUSSIAN && !(lik_param > 0.0)) return TSVGP_EINVAL;
    hipLaunchKernelGGL(lik_map_kernel<T>, dim3((unsigned)(Np / TILE)), dim3(NTHREADS), 0, (hipStream_t)stream, mean, var, Y, lik,
                       lik_param, g0, g1, ve_partial, nonpos_partial, N, P);
    return launch_status();
}

// kdiag: HOST array of P values (one kernel variance per latent), or of ONE value with kdiag_uniform (a shared kernel, any P).
template <typename T>
int moments(const T* A, int64_t strideA, const T* Tm, const T* gamma, const T* Y, const double* kdiag, bool kdiag_uniform,
            int lik, double lik_param, T* mean, T* var, T* g0, T* g1, double* ve_partial, int32_t* nonpos_partial,
            int64_t N, int64_t Np, int Mp, int P, int mode, void* stream) {
    const bool mean_only = (lik & TSVGP_LIK_MEANONLY) != 0;
    if (!A || !gamma || !kdiag || N <= 0 || Np < N || (Np % TILE) || Mp <= 0 || (Mp % TILE) || P <= 0 || strideA < 0)
        return TSVGP_EINVAL;
    if (!kdiag_uniform && P > TSVGP_MAX_BATCH) return TSVGP_EINVAL;
    if (strideA != 0 && strideA < Np * (int64_t)Mp) return TSVGP_EINVAL;
    if (!mean_only && (!Tm || mode < 0 || mode > 2)) return TSVGP_EINVAL;
    if (lik & ~(0xFF | TSVGP_LIK_NOCROP | TSVGP_LIK_MEANONLY)) return TSVGP_EINVAL;
    const int lik_base = lik & 0xFF;
    if (lik_base != TSVGP_LIK_NONE && lik_base != TSVGP_LIK_GAUSSIAN && lik_base != TSVGP_LIK_BERNOULLI) return TSVGP_EINVAL;
    // mean only: the Bernoulli gradients do depend on the variance; gamma [P][Mp] has to fit the LDS
    constexpr size_t MEAN_LDS_MAX = 128 * 1024;
    if (mean_only && (lik_base == TSVGP_LIK_BERNOULLI || var || (size_t)Mp * P * sizeof(T) > MEAN_LDS_MAX)) return TSVGP_EINVAL;
    if (mean_only) {
        static DynLdsOptIn optin;  // one per type T
        if (optin.ensure(reinterpret_cast<const void*>(&mean_lik_kernel<T>), MEAN_LDS_MAX) != TSVGP_OK) return TSVGP_ELAUNCH;
    }
    lik &= ~TSVGP_LIK_MEANONLY;
    if (lik_base == TSVGP_LIK_NONE) lik = TSVGP_LIK_NONE;
    if (lik != TSVGP_LIK_NONE && (!Y || !g0 || !g1)) return TSVGP_EINVAL;
    if (lik_base == TSVGP_LIK_GAUSSIAN && !(lik_param > 0.0)) return TSVGP_EINVAL;
    PanelArgs<T> a{};
    a.A = A;
    a.Tm = Tm;
    a.gamma = gamma;
    a.Y = Y;
    a.mean = mean;
    a.var = var;
    a.g0 = g0;
    a.g1 = g1;
    a.ve_partial = ve_partial;
    a.nonpos_partial = nonpos_partial;
    a.kdiag_uniform = kdiag_uniform ? 1 : 0;
    for (int q = 0; q < (kdiag_uniform ? 1 : P); ++q) a.kdiag[q] = kdiag[q];
    a.strideA = strideA;
    a.lik_param = lik_param;
    a.N = N;
    a.Np = Np;
    a.Mp = Mp;
    a.P = P;
    a.mode = mode;
    a.lik = lik;
    const dim3 grid((unsigned)(Np / TILE)), block(NTHREADS);
    if (mean_only)
        hipLaunchKernelGGL((mean_lik_kernel<T>), grid, block, (size_t)Mp * P * sizeof(T), (hipStream_t)stream, a);
#ifndef TSVGP_LOWER_OLD  // (-DTSVGP_LOWER_OLD: A/B builds keep round 2's panel_kernel for the lower-form products)
    else if (mode == TSVGP_TRI_LOWER && (size_t)Mp * sizeof(T) <= P1_GAMMA_LDS_MAX) {
        static DynLdsOptIn optin1l;
        if (optin1l.ensure(reinterpret_cast<const void*>(&panel1_kernel<T, MODE_MOMENTS, TSVGP_TRI_LOWER>), P1_GAMMA_LDS_MAX) !=
            TSVGP_OK)
            return TSVGP_ELAUNCH;
        hipLaunchKernelGGL((panel1_kernel<T, MODE_MOMENTS, TSVGP_TRI_LOWER>), grid, block, (size_t)Mp * sizeof(T),
                           (hipStream_t)stream, a);
    }
#endif
    else if (mode == TSVGP_TRI_LOWER && (size_t)Mp * sizeof(T) <= 8192)
        hipLaunchKernelGGL((panel_kernel<T, MODE_MOMENTS, TSVGP_TRI_LOWER, true>), grid, block, (size_t)Mp * sizeof(T),
                           (hipStream_t)stream, a);
    else if (mode == TSVGP_TRI_LOWER)
        hipLaunchKernelGGL((panel_kernel<T, MODE_MOMENTS, TSVGP_TRI_LOWER>), grid, block, 0, (hipStream_t)stream, a);
#ifndef TSVGP_MOMENTS_OLD  // (-DTSVGP_MOMENTS_OLD: A/B builds keep round 2's panel_kernel on this path)
#if TSVGP_MOMENTS_WIDE
    else if (mode == TSVGP_TRI_UPPER && (sizeof(T) == 8 || !TSVGP_MOMENTS_OLD_F32) && (Mp % (2 * TILE)) == 0 &&
             (size_t)Mp * sizeof(T) <= P1W_GAMMA_LDS_MAX) {
        // upper form, an even number of column tiles: two tiles per pass
        static DynLdsOptIn optin1w;
        if (optin1w.ensure(reinterpret_cast<const void*>(&panel1_kernel<T, MODE_MOMENTS, TSVGP_TRI_UPPER, 2>), P1W_GAMMA_LDS_MAX) !=
            TSVGP_OK)
            return TSVGP_ELAUNCH;
        hipLaunchKernelGGL((panel1_kernel<T, MODE_MOMENTS, TSVGP_TRI_UPPER, 2>), grid, block, (size_t)Mp * sizeof(T),
                           (hipStream_t)stream, a);
    }
#endif
    else if (mode == TSVGP_TRI_UPPER && (sizeof(T) == 8 || !TSVGP_MOMENTS_OLD_F32) && (size_t)Mp * sizeof(T) <= P1_GAMMA_LDS_MAX) {
        // upper form: one workgroup per CU with the hand-laid instruction stream (panel1_kernel)
        static DynLdsOptIn optin1;
        if (optin1.ensure(reinterpret_cast<const void*>(&panel1_kernel<T>), P1_GAMMA_LDS_MAX) != TSVGP_OK) return TSVGP_ELAUNCH;
        hipLaunchKernelGGL((panel1_kernel<T>), grid, block, (size_t)Mp * sizeof(T), (hipStream_t)stream, a);
    }
#endif
    else if (mode == TSVGP_TRI_UPPER && (size_t)Mp * sizeof(T) <= 8192)
        // gamma fits beside the staging buffers without costing the second workgroup per CU: fused mean
        hipLaunchKernelGGL((panel_kernel<T, MODE_MOMENTS, TSVGP_TRI_UPPER, true>), grid, block, (size_t)Mp * sizeof(T),
                           (hipStream_t)stream, a);
    else if (mode == TSVGP_TRI_UPPER)
        hipLaunchKernelGGL((panel_kernel<T, MODE_MOMENTS, TSVGP_TRI_UPPER>), grid, block, 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((panel_kernel<T, MODE_MOMENTS, TSVGP_TRI_DENSE>), grid, block, 0, (hipStream_t)stream, a);
    return launch_status();
}

template <typename T>
int64_t site_accum_work_bytes(int Mp, int P, int nsplit) {
    if (Mp <= 0 || (Mp % TILE) || P <= 0 || nsplit <= 0) return -1;
    const int64_t nt = Mp / TILE, n_off = nt * (nt - 1) / 2;
    const int64_t ns_diag = syrk_ns_diag(nsplit, (int)sizeof(T));
    const int64_t per_p = n_off * nsplit + nt * ns_diag;
    return (int64_t)P * (per_p * TILE * TILE + ns_diag * Mp) * (int64_t)sizeof(T);
}

template <typename T>
int site_accum(const T* B, int64_t strideB, const T* g0, const T* g1, double* acc2, double* acc1, void* work, int64_t Np,
               int Mp, int P, int nsplit, void* stream) {
    if (!B || !g0 || !g1 || !acc2 || !acc1 || !work || Np <= 0 || (Np % TILE) || Mp <= 0 || (Mp % TILE) || P <= 0 ||
        nsplit <= 0 || strideB < 0 || (strideB != 0 && strideB < Np * (int64_t)Mp))
        return TSVGP_EINVAL;
    // the operand rows and the weights of a chunk travel by 16-byte LDS-DMA pieces: B, g0, g1 (contiguous [Np x P]) on
    // 16-byte boundaries, as every allocator hands them out; a sliced view that is not gets an error, not a fault
    if (((uintptr_t)B | (uintptr_t)g0 | (uintptr_t)g1 | (uintptr_t)work) & 15 || ((strideB * (int64_t)sizeof(T)) & 15))
        return TSVGP_EINVAL;
    const int nt = Mp / TILE, ntri = nt * (nt + 1) / 2, n_off = ntri - nt;
    const int64_t total_chunks = Np / KC;
    SyrkArgs<T> a{};
    a.B = B;
    a.g0 = g0;
    a.g1 = g1;
    a.Np = Np;
    a.strideB = strideB;
    a.Mp = Mp;
    a.P = P;
    a.nt = nt;
    a.ns_off = nsplit;
    a.ns_diag = syrk_ns_diag(nsplit, (int)sizeof(T));
    a.chunks_off = (total_chunks + a.ns_off - 1) / a.ns_off;
    a.chunks_diag = (total_chunks + a.ns_diag - 1) / a.ns_diag;
    const int64_t per_p = (int64_t)n_off * a.ns_off + (int64_t)nt * a.ns_diag;
    a.part2 = reinterpret_cast<T*>(work);
    a.part1 = a.part2 + (size_t)P * per_p * TILE * TILE;
    const int64_t nwg = (int64_t)P * per_p;
    if (nwg > 0x7fffffff) return TSVGP_EINVAL;
#ifndef TSVGP_SYRK_OLD  // (-DTSVGP_SYRK_OLD: A/B builds keep round 2's syrk_kernel for fp64 too)
    if (sizeof(T) == 8 && P <= 8) {  // (the weights of a chunk travel as one 1-KiB LDS-DMA piece: 16 P doubles)
      if constexpr (sizeof(T) == 8) {
        static DynLdsOptIn optin_s1;  // 72 KB of static LDS: above the 64 KB a kernel gets without asking
        if (optin_s1.ensure(reinterpret_cast<const void*>(&syrk1_kernel), 0) != TSVGP_OK) return TSVGP_ELAUNCH;
        hipLaunchKernelGGL(syrk1_kernel, dim3((unsigned)nwg), dim3(NTHREADS), 0, (hipStream_t)stream, a);
      }
    } else if (sizeof(T) == 4 && !TSVGP_SYRK_OLD_F32 && P <= 8) {
      if constexpr (sizeof(T) == 4) {
        const int64_t total32 = Np / S1F_KC;  // this kernel's chunks are 32 rows
        a.chunks_off = (total32 + a.ns_off - 1) / a.ns_off;
        a.chunks_diag = (total32 + a.ns_diag - 1) / a.ns_diag;
        static DynLdsOptIn optin_s1f;  // 68 KB of static LDS
        if (optin_s1f.ensure(reinterpret_cast<const void*>(&syrk1f_kernel), 0) != TSVGP_OK) return TSVGP_ELAUNCH;
        hipLaunchKernelGGL(syrk1f_kernel, dim3((unsigned)nwg), dim3(NTHREADS), 0, (hipStream_t)stream, a);
      }
    } else
#endif
        hipLaunchKernelGGL(syrk_kernel<T>, dim3((unsigned)nwg), dim3(NTHREADS), 0, (hipStream_t)stream, a);
    if (launch_status() != TSVGP_OK) return TSVGP_ELAUNCH;
    const int extra = (Mp + NTHREADS - 1) / NTHREADS;
    hipLaunchKernelGGL(syrk_reduce_kernel<T>, dim3((unsigned)(ntri * (TILE / SR_ROWS) + extra), (unsigned)P), dim3(NTHREADS), 0,
                       (hipStream_t)stream, a.part2, a.part1, acc2, acc1, Mp, P, nt, a.ns_off, a.ns_diag);
    return launch_status();
}

template <typename T>
int site_accum_slots() {
    int dev = 0, cus = 0, nb = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
#ifndef TSVGP_SYRK_OLD
    if (sizeof(T) == 8) return cus;  // syrk1_kernel: one workgroup per CU by construction (512 registers per wave)
    if (!TSVGP_SYRK_OLD_F32) {       // syrk1f_kernel: what its registers and 68 KB of LDS allow
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&syrk1f_kernel), NTHREADS, 0) !=
            hipSuccess)
            return -1;
        return cus * (nb > 2 ? 2 : nb < 1 ? 1 : nb);
    }
#endif
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&syrk_kernel<T>), NTHREADS,
                                                     0) != hipSuccess)
        return -1;
    if (nb > 2) nb = 2;
    return cus * nb;
}

int potrf(double* A, int M, int lda, int batch, int64_t stride, int* info, double* work, int flags, void* stream,
          double* X = nullptr, double* Xt = nullptr, double* T = nullptr, int rhs_rows = 0) {
    if (!A || !info || !work || M <= 0 || (M % CH_NB) || lda < M || batch <= 0 ||
        (flags & ~(TSVGP_POTRF_SUBST | TSVGP_POTRF_RHS_UPPER | TSVGP_POTRF_DIAG_V1 | TSVGP_POTRF_DIAG_V2 | TSVGP_POTRF_FUSE)) || rhs_rows < 0 || (rhs_rows % CH_NB) ||
        (rhs_rows > 0 && stride < (int64_t)(M + rhs_rows) * lda))
        return TSVGP_EINVAL;
    const bool rhs_upper = (flags & TSVGP_POTRF_RHS_UPPER) != 0;
    const bool inv = X != nullptr, subst = (flags & TSVGP_POTRF_SUBST) != 0, diag_v1 = (flags & TSVGP_POTRF_DIAG_V1) != 0,
               diag_v2 = (flags & TSVGP_POTRF_DIAG_V2) != 0, fuse = (flags & TSVGP_POTRF_FUSE) != 0;
    if (inv && (!Xt || !T)) return TSVGP_EINVAL;
    const int nt = M / CH_NB;
#ifdef TSVGP_DIAG_POTRF
    const size_t smem = (size_t)CH_NB * CH_LD * sizeof(double) + 64 * sizeof(unsigned long long);  // + in-phase stamps
#else
    const size_t smem = (size_t)CH_NB * CH_LD * sizeof(double);
#endif
    static DynLdsOptIn optin;
    if (optin.ensure(reinterpret_cast<const void*>(&potrf_diag_kernel<false>), smem) != TSVGP_OK) return TSVGP_ELAUNCH;
    static DynLdsOptIn optin_fused;
    if (optin_fused.ensure(reinterpret_cast<const void*>(&potrf_diag_kernel<true>), smem) != TSVGP_OK) return TSVGP_ELAUNCH;
    hipStream_t st = (hipStream_t)stream;
    const int64_t xstride = (int64_t)M * M;
    if (inv) {  // the blocks the recursion does not write stay zero
        if ((reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(Xt)) & 15) return TSVGP_EINVAL;
        zero_fill(X, sizeof(double) * xstride * batch, st);
        zero_fill(Xt, sizeof(double) * xstride * batch, st);
    }
    // Lookahead in ONE launch was built and measured (commit 472dfec, not kept): the factoring workgroup of block k + 1
    // applies column k's update to its own block (S -= P P^T through LDS chunks of the panel block) while extra workgroups
    // of the same launch update the rest of the trailing matrix.  0.464 vs 0.458 ms at M = 1024: that in-kernel update
    // costs 15 us (LDS-read bound: 80 KB of operand reads per 8-column chunk for 10 MFMAs per wave), as much as the
    // 13 us update launch it hides, and the 10 KB chunk buffer brings the workgroup to the edge of what fits beside
    // three workgroups of the K(X, Z) fill on one CU (with 18 KB it no longer fits and waits: 1.22 vs 0.96 ms under the fill).
    const int wpb = NTHREADS / 64;
    for (int k = 0; k < nt; ++k) {
        // right-hand-side rows that column block k can reach: all of them, or (block upper triangular B) its row blocks 0 .. k
        const int ext_rows = rhs_upper ? (rhs_rows < (k + 1) * CH_NB ? rhs_rows : (k + 1) * CH_NB) : rhs_rows;
        const int below = nt - k - 1;
        // what the panel step behind this block needs of it: nothing (last block / substitution panels), the assembled inverse
        // (round 4's step, and tsvgp_potrf_inv_f64), or the `work` image of chol_panel2_kernel (round 5)
        const bool panel2 = !inv && !diag_v1 && !subst;
        const int need_inverse = inv ? 1 : ((below > 0 || ext_rows > 0) && !subst) ? (panel2 ? 2 : 1) : 0;
        const int nstrips = (below * CH_NB + ext_rows) / 16;
        const bool fused = panel2 && !diag_v2 && fuse && nstrips > 0;  // (measured slower than two launches: tsvgp_hip.h)
        if (panel2 && diag_v2) {  // the diagonal block as an MFMA tile dataflow (tsvgp_chol.hip; experimental)
            if (tsvgp_chol::launch_diag2(A, lda, stride, k, work, info, need_inverse, batch, st) != hipSuccess) return TSVGP_ELAUNCH;
        } else if (fused) {  // diagonal block AND the panel rows below it in one launch
            const int per = CH_THREADS / 64;
            hipLaunchKernelGGL(potrf_diag_kernel<true>, dim3(batch, (nstrips + per - 1) / per), dim3(CH_THREADS), smem, st, A, lda, stride,
                               k, work, info, 2, X, Xt, M, xstride, nstrips);
        } else {
            hipLaunchKernelGGL(potrf_diag_kernel<false>, dim3(batch), dim3(CH_THREADS), smem, st, A, lda, stride, k, work, info,
                               need_inverse, X, Xt, M, xstride, 0);
        }
        if (below > 0 || ext_rows > 0) {
            // the panel: the matrix rows below the diagonal block and, contiguous with them (row M on), the right-hand-side rows
            const int nb32 = below * (CH_NB / CH_WT), ext32 = ext_rows / CH_WT;
            const int ntile = nb32 * (nb32 + 1) / 2 + ext32 * nb32;
            if (subst)
                hipLaunchKernelGGL(chol_panel_subst_kernel, dim3((below * CH_NB + ext_rows) / PS_ROWS, batch), dim3(NTHREADS), 0,
                                   st, A, lda, stride, k);
            else if (fused) {
                // (done by the diagonal block's launch)
            } else if (panel2) {  // substitution on tile registers against the diagonal kernel's `work` image
                if (tsvgp_chol::launch_panel2(A, lda, stride, k, work, (below * CH_NB + ext_rows) / 16, batch, st) != hipSuccess)
                    return TSVGP_ELAUNCH;
            } else
                hipLaunchKernelGGL(chol_tile_kernel<0>, dim3(nb32 + ext32, batch), dim3(NTHREADS), 0, st, A, lda, stride, k, nt,
                                   work, ext32);
            if (ntile > 0)
                hipLaunchKernelGGL(chol_tile_kernel<1>, dim3((ntile + wpb - 1) / wpb, batch), dim3(NTHREADS), 0, st, A,
                                   lda, stride, k, nt, work, ext32);
        }
    }
    if (inv) {
        for (int n = CH_NB; n < M; n *= 2) {
            const int npair = (M + 2 * n - 1) / (2 * n);
            const int ntile = (n / CH_WT) * (n / CH_WT);
            const dim3 grid(ntile, npair, batch);
            hipLaunchKernelGGL(trtri_level_kernel<0>, grid, dim3(NTHREADS), 0, st, A, lda, stride, X, Xt, T, M, xstride,
                               M, n);
            hipLaunchKernelGGL(trtri_level_kernel<1>, grid, dim3(NTHREADS), 0, st, A, lda, stride, X, Xt, T, M, xstride,
                               M, n);
        }
    }
    return launch_status();
}

template <typename T>
int kernel_grad(int kind, const T* X, const T* Z, const T* inv_ls, T variance, const T* U, int64_t ldu, const T* g0,
                const T* g1, int gstride, const T* beta, int bstride, int64_t N, int M, int D, double* zpart,
                double* lpart, double* vpart, void* stream) {
    if (!X || !Z || !inv_ls || !U || !g0 || !g1 || !beta || !zpart || !lpart || !vpart || N <= 0 || M <= 0 || D <= 0 ||
        D > 16 || gstride <= 0 || bstride <= 0 || (ldu % 2) != 0)
        return TSVGP_EINVAL;
    if (kind != TSVGP_KERNEL_SE && kind != TSVGP_KERNEL_MATERN32 && kind != TSVGP_KERNEL_MATERN52) return TSVGP_EINVAL;
    const int Mp = (M + TILE - 1) / TILE * TILE;
    if (ldu < Mp) return TSVGP_EINVAL;
    const dim3 grid((unsigned)((N + KG_ROWS - 1) / KG_ROWS), (unsigned)((Mp + FILL_COLS - 1) / FILL_COLS));
    hipStream_t st = (hipStream_t)stream;
    const int DT = D <= 1 ? 1 : D <= 2 ? 2 : D <= 4 ? 4 : D <= 8 ? 8 : 16;
    if (hipMemsetAsync(zpart, 0, sizeof(double) * (size_t)grid.x * Mp * DT, st) != hipSuccess) return TSVGP_ELAUNCH;
#define TSVGP_KG_LAUNCH(KIND_, DT_)                                                                                  \
    hipLaunchKernelGGL((kgrad_kernel<T, KIND_, DT_>), grid, dim3(NTHREADS), 0, st, X, Z, inv_ls, variance, U, ldu, g0, g1, \
                       gstride, beta, bstride, N, M, D, zpart, lpart, vpart, Mp)
#define TSVGP_KG_DT(KIND_)                          \
    switch (DT) {                                   \
        case 1: TSVGP_KG_LAUNCH(KIND_, 1); break;   \
        case 2: TSVGP_KG_LAUNCH(KIND_, 2); break;   \
        case 4: TSVGP_KG_LAUNCH(KIND_, 4); break;   \
        case 8: TSVGP_KG_LAUNCH(KIND_, 8); break;   \
        default: TSVGP_KG_LAUNCH(KIND_, 16); break; \
    }
    if (kind == TSVGP_KERNEL_SE) {
        TSVGP_KG_DT(TSVGP_KERNEL_SE)
    } else if (kind == TSVGP_KERNEL_MATERN32) {
        TSVGP_KG_DT(TSVGP_KERNEL_MATERN32)
    } else {
        TSVGP_KG_DT(TSVGP_KERNEL_MATERN52)
    }
#undef TSVGP_KG_DT
#undef TSVGP_KG_LAUNCH
    return launch_status();
}

}  // namespace

extern "C" {

const char* tsvgp_version(void) { return "tsvgp_hip gfx950 0.3.0"; }
int tsvgp_abi_version(void) { return TSVGP_ABI_VERSION; }

int tsvgp_site_accum_slots_f64(void) { return site_accum_slots<double>(); }
int tsvgp_site_accum_slots_f32(void) { return site_accum_slots<float>(); }

int tsvgp_se_fill_f64(const double* X, const double* Z, const double* inv_ls, double variance, double* K, int64_t N,
                      int M, int D, int64_t ldk, void* stream) {
    return kernel_fill<double>(TSVGP_KERNEL_SE, X, Z, inv_ls, &variance, K, 0, N, M, D, ldk, 1, stream);
}
int tsvgp_se_fill_f32(const float* X, const float* Z, const float* inv_ls, float variance, float* K, int64_t N, int M,
                      int D, int64_t ldk, void* stream) {
    return kernel_fill<float>(TSVGP_KERNEL_SE, X, Z, inv_ls, &variance, K, 0, N, M, D, ldk, 1, stream);
}
int tsvgp_kernel_fill_f64(int kind, const double* X, const double* Z, const double* inv_ls, double variance, double* K,
                          int64_t N, int M, int D, int64_t ldk, void* stream) {
    return kernel_fill<double>(kind, X, Z, inv_ls, &variance, K, 0, N, M, D, ldk, 1, stream);
}
int tsvgp_kernel_fill_f32(int kind, const float* X, const float* Z, const float* inv_ls, float variance, float* K,
                          int64_t N, int M, int D, int64_t ldk, void* stream) {
    return kernel_fill<float>(kind, X, Z, inv_ls, &variance, K, 0, N, M, D, ldk, 1, stream);
}
int tsvgp_kernel_fill_batched_f64(int kind, const double* X, const double* Z, const double* inv_ls,
                                  const double* variance_host, double* K, int64_t strideK, int64_t N, int M, int D,
                                  int64_t ldk, int P, void* stream) {
    return kernel_fill<double>(kind, X, Z, inv_ls, variance_host, K, strideK, N, M, D, ldk, P, stream);
}
int tsvgp_kernel_fill_batched_f32(int kind, const float* X, const float* Z, const float* inv_ls,
                                  const float* variance_host, float* K, int64_t strideK, int64_t N, int M, int D,
                                  int64_t ldk, int P, void* stream) {
    return kernel_fill<float>(kind, X, Z, inv_ls, variance_host, K, strideK, N, M, D, ldk, P, stream);
}

int tsvgp_gram_to_kernel_f64(int kind, double* K, const double* xx, const double* zz, double variance, int64_t N, int M,
                             int64_t ldk, void* stream) {
    return gram_to_kernel<double>(kind, K, xx, zz, variance, N, M, ldk, stream);
}
int tsvgp_gram_to_kernel_f32(int kind, float* K, const float* xx, const float* zz, float variance, int64_t N, int M,
                             int64_t ldk, void* stream) {
    return gram_to_kernel<float>(kind, K, xx, zz, variance, N, M, ldk, stream);
}
// vpart: one double per workgroup = ceil(Mp / 512) * Np of them (tsvgp_gram_to_gradw_parts)
int64_t tsvgp_gram_to_gradw_parts(int64_t N, int M) {
    if (N <= 0 || M <= 0) return -1;
    const int64_t rows_pad = (N + TILE - 1) / TILE * TILE;
    const int64_t cols_pad = (M + TILE - 1) / TILE * TILE;
    return rows_pad * ((cols_pad / 2 + NTHREADS - 1) / NTHREADS);
}
int tsvgp_gram_to_gradw_f64(int kind, double* G, const double* xx, const double* zz, double variance, const double* U, int64_t ldu,
                            const double* g0, const double* g1, int gstride, const double* beta, int bstride, int64_t N, int M,
                            int64_t ldk, double* vpart, void* stream) {
    return gram_to_gradw<double>(kind, G, xx, zz, variance, U, ldu, g0, g1, gstride, beta, bstride, N, M, ldk, vpart, stream);
}
int tsvgp_gram_to_gradw_f32(int kind, float* G, const float* xx, const float* zz, float variance, const float* U, int64_t ldu,
                            const float* g0, const float* g1, int gstride, const float* beta, int bstride, int64_t N, int M,
                            int64_t ldk, double* vpart, void* stream) {
    return gram_to_gradw<float>(kind, G, xx, zz, variance, U, ldu, g0, g1, gstride, beta, bstride, N, M, ldk, vpart, stream);
}

int tsvgp_trmm_f64(const double* A, const double* Tm, double* C, int64_t Np, int Mp, int mode, void* stream) {
    return trmm<double>(A, 0, Tm, 0, C, 0, Np, Mp, mode, 1, stream);
}
int tsvgp_trmm_f32(const float* A, const float* Tm, float* C, int64_t Np, int Mp, int mode, void* stream) {
    return trmm<float>(A, 0, Tm, 0, C, 0, Np, Mp, mode, 1, stream);
}
int tsvgp_trmm_batched_f64(const double* A, int64_t strideA, const double* Tm, int64_t strideT, double* C,
                           int64_t strideC, int64_t Np, int Mp, int mode, int batch, void* stream) {
    return trmm<double>(A, strideA, Tm, strideT, C, strideC, Np, Mp, mode, batch, stream);
}
int tsvgp_trmm_batched_f32(const float* A, int64_t strideA, const float* Tm, int64_t strideT, float* C, int64_t strideC,
                           int64_t Np, int Mp, int mode, int batch, void* stream) {
    return trmm<float>(A, strideA, Tm, strideT, C, strideC, Np, Mp, mode, batch, stream);
}

int tsvgp_lik_map_f64(const double* mean, const double* var, const double* Y, int lik, double lik_param, double* g0, double* g1,
                      double* ve_partial, int32_t* nonpos_partial, int64_t N, int64_t Np, int P, void* stream) {
    return lik_map<double>(mean, var, Y, lik, lik_param, g0, g1, ve_partial, nonpos_partial, N, Np, P, stream);
}
int tsvgp_lik_map_f32(const float* mean, const float* var, const float* Y, int lik, double lik_param, float* g0, float* g1,
                      double* ve_partial, int32_t* nonpos_partial, int64_t N, int64_t Np, int P, void* stream) {
    return lik_map<float>(mean, var, Y, lik, lik_param, g0, g1, ve_partial, nonpos_partial, N, Np, P, stream);
}
int tsvgp_moments_f64(const double* A, const double* Tm, const double* gamma, const double* Y, double kdiag, int lik,
                      double lik_param, double* mean, double* var, double* g0, double* g1, double* ve_partial,
                      int32_t* nonpos_partial, int64_t N, int64_t Np, int Mp, int P, int mode, void* stream) {
    return moments<double>(A, 0, Tm, gamma, Y, &kdiag, true, lik, lik_param, mean, var, g0, g1, ve_partial,
                           nonpos_partial, N, Np, Mp, P, mode, stream);
}
int tsvgp_moments_batched_f64(const double* A, int64_t strideA, const double* Tm, const double* gamma, const double* Y,
                              const double* kdiag_host, int lik, double lik_param, double* mean, double* var, double* g0,
                              double* g1, double* ve_partial, int32_t* nonpos_partial, int64_t N, int64_t Np, int Mp,
                              int P, int mode, void* stream) {
    return moments<double>(A, strideA, Tm, gamma, Y, kdiag_host, false, lik, lik_param, mean, var, g0, g1, ve_partial,
                           nonpos_partial, N, Np, Mp, P, mode, stream);
}
int tsvgp_moments_batched_f32(const float* A, int64_t strideA, const float* Tm, const float* gamma, const float* Y,
                              const double* kdiag_host, int lik, double lik_param, float* mean, float* var, float* g0,
                              float* g1, double* ve_partial, int32_t* nonpos_partial, int64_t N, int64_t Np, int Mp,
                              int P, int mode, void* stream) {
    return moments<float>(A, strideA, Tm, gamma, Y, kdiag_host, false, lik, lik_param, mean, var, g0, g1, ve_partial,
                          nonpos_partial, N, Np, Mp, P, mode, stream);
}
int tsvgp_moments_f32(const float* A, const float* Tm, const float* gamma, const float* Y, double kdiag, int lik,
                      double lik_param, float* mean, float* var, float* g0, float* g1, double* ve_partial,
                      int32_t* nonpos_partial, int64_t N, int64_t Np, int Mp, int P, int mode, void* stream) {
    return moments<float>(A, 0, Tm, gamma, Y, &kdiag, true, lik, lik_param, mean, var, g0, g1, ve_partial,
                          nonpos_partial, N, Np, Mp, P, mode, stream);
}

int64_t tsvgp_site_accum_work_bytes_f64(int Mp, int P, int nsplit) {
    return site_accum_work_bytes<double>(Mp, P, nsplit);
}
int64_t tsvgp_site_accum_work_bytes_f32(int Mp, int P, int nsplit) {
    return site_accum_work_bytes<float>(Mp, P, nsplit);
}
int tsvgp_site_accum_f64(const double* B, const double* g0, const double* g1, double* acc2, double* acc1, void* work,
                         int64_t Np, int Mp, int P, int nsplit, void* stream) {
    return site_accum<double>(B, 0, g0, g1, acc2, acc1, work, Np, Mp, P, nsplit, stream);
}
int tsvgp_site_accum_batched_f64(const double* B, int64_t strideB, const double* g0, const double* g1, double* acc2,
                                 double* acc1, void* work, int64_t Np, int Mp, int P, int nsplit, void* stream) {
    return site_accum<double>(B, strideB, g0, g1, acc2, acc1, work, Np, Mp, P, nsplit, stream);
}
int tsvgp_site_accum_batched_f32(const float* B, int64_t strideB, const float* g0, const float* g1, double* acc2,
                                 double* acc1, void* work, int64_t Np, int Mp, int P, int nsplit, void* stream) {
    return site_accum<float>(B, strideB, g0, g1, acc2, acc1, work, Np, Mp, P, nsplit, stream);
}
int tsvgp_site_accum_f32(const float* B, const float* g0, const float* g1, double* acc2, double* acc1, void* work,
                         int64_t Np, int Mp, int P, int nsplit, void* stream) {
    return site_accum<float>(B, 0, g0, g1, acc2, acc1, work, Np, Mp, P, nsplit, stream);
}

int tsvgp_potrf_f64(double* A, int M, int lda, int batch, int64_t stride, int* info, double* work, int flags,
                    void* stream) {
    return potrf(A, M, lda, batch, stride, info, work, flags, stream);
}
int tsvgp_potrf_solve_f64(double* A, int M, int lda, int batch, int64_t stride, int* info, double* work, int rhs_rows,
                          int flags, void* stream) {
    if (rhs_rows <= 0) return TSVGP_EINVAL;
    return potrf(A, M, lda, batch, stride, info, work, flags, stream, nullptr, nullptr, nullptr, rhs_rows);
}
int tsvgp_flip_transpose_f64(const double* src, int lds, int64_t sstride, double* dst, int ldd, int64_t dstride, int M, int batch,
                             void* stream) {
    if (!src || !dst || M <= 0 || lds < M || ldd < M || batch <= 0 || batch > 65535) return TSVGP_EINVAL;
    const unsigned nb = (unsigned)((M + 31) / 32);
    hipLaunchKernelGGL(flip_transpose_kernel, dim3(nb, nb, (unsigned)batch), dim3(NTHREADS), 0, (hipStream_t)stream, src, lds,
                       sstride, dst, ldd, dstride, M);
    return launch_status();
}
int tsvgp_potrf_inv_f64(double* A, int M, int lda, int batch, int64_t stride, int* info, double* work, double* X,
                        double* Xt, double* T, int flags, void* stream) {
    if (!X || !Xt || !T) return TSVGP_EINVAL;
    return potrf(A, M, lda, batch, stride, info, work, flags, stream, X, Xt, T);
}

int tsvgp_tri_copy_shift_f64(const double* src, int lds, int64_t sstride, double* dst, int ldd, int64_t dstride, int M, int batch,
                             double scale, double diag_add, int flip, void* stream) {
    if (!src || !dst || M <= 0 || lds < M || ldd < M || batch <= 0 || batch > 65535 || M > 65535 || flip < 0 || flip > 3)
        return TSVGP_EINVAL;
    hipLaunchKernelGGL(tri_copy_kernel, dim3((M + NTHREADS - 1) / NTHREADS, M, batch), dim3(NTHREADS), 0, (hipStream_t)stream,
                       src, lds, sstride, dst, ldd, dstride, M, scale, flip, diag_add);
    return launch_status();
}
int tsvgp_tri_copy_f64(const double* src, int lds, int64_t sstride, double* dst, int ldd, int64_t dstride, int M, int batch,
                       double scale, int flip, void* stream) {
    if (flip > 2) return TSVGP_EINVAL;
    return tsvgp_tri_copy_shift_f64(src, lds, sstride, dst, ldd, dstride, M, batch, scale, 0.0, flip, stream);
}

int tsvgp_site_target_f64(const double* G1, const double* LLt, double* target, double* G1s, int M, int P, double c_ll,
                          double c_g, double jitter, const double* rows, double num_data, void* stream) {
    if (!G1 || !LLt || !target || !G1s || M <= 0 || P <= 0 || P > 65535 || (num_data > 0.0 && !rows)) return TSVGP_EINVAL;
    const unsigned nb = (unsigned)((M + 31) / 32);
    hipLaunchKernelGGL(site_target_kernel, dim3(nb, nb, (unsigned)P), dim3(NTHREADS), 0, (hipStream_t)stream, G1, LLt, target,
                       G1s, M, c_ll, c_g, jitter, rows, num_data);
    return launch_status();
}
int tsvgp_site_update_f64(const double* G1, const double* G0, const double* LLt, const double* meanZ, const double* l1_old,
                          double* target, double* l1_new, double* work, int M, int P, double lr, double jitter,
                          const double* rows, double num_data, void* stream) {
    if (!G1 || !G0 || !LLt || !meanZ || !l1_old || !target || !l1_new || !work || M <= 0 || P <= 0 || P > 65535 ||
        (num_data > 0.0 && !rows))
        return TSVGP_EINVAL;
    const unsigned nb = (unsigned)((M + 31) / 32);
    hipLaunchKernelGGL(site_update_kernel, dim3(nb, nb, (unsigned)P), dim3(NTHREADS), 0, (hipStream_t)stream, G1, LLt, meanZ,
                       target, work, M, P, lr, jitter, rows, num_data);
    hipLaunchKernelGGL(site_update_finish_kernel, dim3((unsigned)((M + NTHREADS - 1) / NTHREADS), (unsigned)P), dim3(NTHREADS), 0,
                       (hipStream_t)stream, work, G0, l1_old, l1_new, M, P, lr, rows, num_data);
    return launch_status();
}
int tsvgp_site_beta_f64(const double* D, const double* v, const double* l1, double* work, double* beta, int M, int P,
                        void* stream) {
    if (!D || !v || !l1 || !work || !beta || M <= 0 || P <= 0 || P > 65535) return TSVGP_EINVAL;
    const int wpb = NTHREADS / 64, nrg = (M + 63) / 64;
    double* part = work + (size_t)P * M;  // work: t [P][M], then the partial sums [P][nrg][M]
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(site_beta_kernel<0>, dim3((unsigned)((M + wpb - 1) / wpb), 1, (unsigned)P), dim3(NTHREADS), 0, st, D, v, l1, work,
                       part, beta, M, P);
    hipLaunchKernelGGL(site_beta_kernel<1>, dim3((unsigned)nrg, (unsigned)nrg, (unsigned)P), dim3(NTHREADS), 0, st, D, v, l1, work, part,
                       beta, M, P);
    hipLaunchKernelGGL(site_beta_kernel<2>, dim3((unsigned)((M + NTHREADS - 1) / NTHREADS), 1, (unsigned)P), dim3(NTHREADS), 0, st, D, v,
                       l1, work, part, beta, M, P);
    return launch_status();
}
int tsvgp_gemv_f64(const double* A, int64_t strideA, const double* v, double* y, int M, int P, void* stream) {
    if (!A || !v || !y || M <= 0 || P <= 0 || P > 65535 || strideA < 0 || (strideA != 0 && strideA < (int64_t)M * M)) return TSVGP_EINVAL;
    const int wpb = NTHREADS / 64;
    hipLaunchKernelGGL(gemv_rows_kernel, dim3((unsigned)((M + wpb - 1) / wpb), (unsigned)P), dim3(NTHREADS), 0, (hipStream_t)stream, A,
                       strideA, v, y, M, P);
    return launch_status();
}
int tsvgp_step_status_f64(const int32_t* info_a, int na, const int32_t* info_b, int nb, const double* nonpos, double* flags,
                          void* stream) {
    if (!flags || na < 0 || nb < 0 || (na > 0 && !info_a) || (nb > 0 && !info_b)) return TSVGP_EINVAL;
    hipLaunchKernelGGL(step_status_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, info_a, na, info_b, nb, nonpos, flags);
    return launch_status();
}
int tsvgp_keeper_run(const int32_t* flag, double max_us, int workgroups, void* stream) {
    if (!flag || !(max_us > 0.0) || max_us > 1.0e6 || workgroups < 0) return TSVGP_EINVAL;
    if (workgroups == 0) {  // one workgroup of four waves per CU: one wave per SIMD
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            return TSVGP_ELAUNCH;
        workgroups = cus;
    }
    hipLaunchKernelGGL(clock_keeper_kernel, dim3(workgroups), dim3(NTHREADS), 0, (hipStream_t)stream, flag, (unsigned)(max_us * 100.0));
    return launch_status();
}
int tsvgp_keeper_signal(int32_t* flag, int value, void* stream) {
    if (!flag) return TSVGP_EINVAL;
    hipLaunchKernelGGL(keeper_signal_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, flag, value);
    return launch_status();
}
int tsvgp_sym_pack_f64(const double* A, int lda, int64_t stride, int M, int P, double* packed, void* stream) {
    if (!A || !packed || M <= 0 || lda < M || P <= 0 || P > 65535 || M > 65535) return TSVGP_EINVAL;
    hipLaunchKernelGGL(sym_pack_kernel, dim3((M + NTHREADS - 1) / NTHREADS, M, P), dim3(NTHREADS), 0, (hipStream_t)stream, A,
                       lda, stride, M, packed, 0);
    return launch_status();
}
int tsvgp_sym_unpack_f64(const double* packed, double* A, int lda, int64_t stride, int M, int P, void* stream) {
    if (!A || !packed || M <= 0 || lda < M || P <= 0 || P > 65535 || M > 65535) return TSVGP_EINVAL;
    hipLaunchKernelGGL(sym_pack_kernel, dim3((M + NTHREADS - 1) / NTHREADS, M, P), dim3(NTHREADS), 0, (hipStream_t)stream, A,
                       lda, stride, M, const_cast<double*>(packed), 1);
    return launch_status();
}

int tsvgp_kernel_grad_rows(void) { return KG_ROWS; }
int tsvgp_kernel_grad_dpad(int D) { return D <= 1 ? 1 : D <= 2 ? 2 : D <= 4 ? 4 : D <= 8 ? 8 : 16; }
int tsvgp_kernel_grad_f64(int kind, const double* X, const double* Z, const double* inv_ls, double variance,
                          const double* U, int64_t ldu, const double* g0, const double* g1, int gstride,
                          const double* beta, int bstride, int64_t N, int M, int D, double* zpart, double* lpart,
                          double* vpart, void* stream) {
    return kernel_grad<double>(kind, X, Z, inv_ls, variance, U, ldu, g0, g1, gstride, beta, bstride, N, M, D, zpart, lpart,
                               vpart, stream);
}
int tsvgp_kernel_grad_f32(int kind, const float* X, const float* Z, const float* inv_ls, float variance, const float* U,
                          int64_t ldu, const float* g0, const float* g1, int gstride, const float* beta, int bstride,
                          int64_t N, int M, int D, double* zpart, double* lpart, double* vpart, void* stream) {
    return kernel_grad<float>(kind, X, Z, inv_ls, variance, U, ldu, g0, g1, gstride, beta, bstride, N, M, D, zpart, lpart,
                              vpart, stream);
}
int tsvgp_selftest_mfma_f64(const double* a, const double* b, double* c, void* stream) {
    if (!a || !b || !c) return TSVGP_EINVAL;
    hipLaunchKernelGGL(selftest_kernel<double>, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, c);
    return launch_status();
}
int tsvgp_selftest_mfma_f32(const float* a, const float* b, float* c, void* stream) {
    if (!a || !b || !c) return TSVGP_EINVAL;
    hipLaunchKernelGGL(selftest_kernel<float>, dim3(1), dim3(64), 0, (hipStream_t)stream, a, b, c);
    return launch_status();
}

}  // extern "C"
