// tsvgp_chol.hip -- the 128 x 128 diagonal block of the blocked Cholesky factorisation (tsvgp_potrf*_f64), round 5.
//
// Replaces: the serial part of  tf.linalg.cholesky  in reference src/util.py:376-389 (posterior_from_dense_site: chol of
// W = I + L^T K L) and src/models/tsvgp.py:270, :300 (chol of K_uu + jitter I, chol of -2 lambda_2 + jitter I): two
// dependent M x M factorisations per E-step, each eight 128-wide block steps whose diagonal block is a chain of 128 pivots.
//
// Round 4's kernel (potrf_diag_kernel, tsvgp_kernels.hip) held one matrix ROW per lane and paid 37.5 us per block:
// rank-4 updates on the vector ALU through an LDS scratch, then three levels of a 2 x 2 recursion for the inverse.
// Here every 16 x 16 tile lives in the ACCUMULATOR layout of v_mfma_f64_16x16x4_f64 with the block symmetric,
//     lane (n = l & 15, G = l >> 4), register r   <->   T[n][4 r + G]              (the f64 C/D layout: row = G + 4 r),
// in which a 4-column pivot group q is register q of all four lane groups, i.e. it IS an MFMA A/B operand as it stands:
//     tile step       T' = [T, register q zeroed] + mfma(aop, T[q])   ONE MFMA per tile and pivot group, aop (lane (m, k)) =
//                     0 for the rows m above the pivot block (finished columns untouched), inv(L_pp)[m - 4q][k] for the
//                     pivot rows (register q becomes the solved columns A_q inv(L_pp)^T), -(A_d inv(a_pp))[m][k] below
//                     (the rank-4 update of the columns to the right; A_d inv(a_pp) is itself one MFMA of the diagonal tile)
//     later columns   T -= sum_kk mfma(P_js[kk], P_is[kk])                                           (rank 16)
// so that the vector ALU is left with the 4 x 4 pivot chain only (four dependent v_rsq_f64 + one third-order step each).
// On MI355X an fp64 MFMA takes 64 cycles of its SIMD's matrix pipe (profiles/r02_mfma_peak_microbench.txt), as many FMAs per
// clock as the vector ALU: the matrix instructions buy the data movement, not arithmetic, and the work is spread so that the
// wave on the critical path issues two of them per pivot group:
//   wave 0 (pivot wave)   the diagonal tile D_s of block column s: chain -> aop -> LDS broadcast -> its own step, and the
//                         tile X(s, s) that starts as the identity and ends as inv(L_ss)^T (the panel kernel's operand)
//   waves 1 .. 7          wave i owns tile ROW i: the tile (i, s) stays in its registers from column to column -- behind
//                         barrier G(s, q) it takes group q's step with the broadcast aop, publishes register q (final) in
//                         LDS and adds the look-ahead product of the register finished one group earlier, so that the next
//                         column's tile is there one MFMA after the column ends; the owner of row s + 1 hands the next
//                         diagonal tile to the pivot wave.  Each wave also takes ONE job per barrier interval from the
//                         flattened list of rank-16 updates of column s - 1 on the columns >= s + 1 (column 0's phase: the
//                         staging of those columns from global memory instead).
// Five barriers per block column; finished tiles go to global memory as they finish (no output pass).  The factor's tiles
// and the inverted 16 x 16 diagonal tiles also go to `work` in the register layout for chol_panel2_kernel (below), which
// solves the panel rows by substitution over the eight 16-wide column blocks -- MFMAs on tile registers again.
// tools/emul_diag2.py is the lane-level NumPy model of this file (index algebra and barrier placement checked against
// numpy.linalg); tools/diag2_lab.hip times the kernel alone with in-kernel stamps.

#include "tsvgp_chol.h"

#include <type_traits>

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int NB = 128;         // block size of the factorisation (CH_NB)
constexpr int NTC = NB / 16;    // tile columns
constexpr int NTILES = NTC * (NTC + 1) / 2;  // lower tiles of the block
constexpr int D2_THREADS = 512;
constexpr size_t D2_LDS_BYTES = ((size_t)NTILES * 4 * 64 + 4 * 64) * sizeof(double);  // 36 tiles + 4 operands: 75 776

#ifndef TSVGP_CHOL_PRIO
#define TSVGP_CHOL_PRIO 3
#endif

__device__ __forceinline__ v4d mfma(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

// 1/sqrt(x): v_rsq_f64 + one third-order step (max rel err 1.4e-16, tools/rsq_probe.hip); NaN for x <= 0 (inf * 0).
__device__ __forceinline__ double rsqrt_nr(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);
}

__device__ __forceinline__ double readlane_d(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

// LDS image (and the `work` image for the panel kernel): column j, slot u = tile (j + u, j), columns packed one after the
// other.  A tile is [4 registers][64 lanes]: every lane reads and writes its own words (conflict free, no layout change).
__device__ __forceinline__ int tile_index(int col, int slot) { return tsvgp_chol::work_tile_index(col, slot); }
__device__ __forceinline__ double* tile_ptr(double* S, int col, int slot) { return S + (size_t)tile_index(col, slot) * 256; }
__device__ __forceinline__ v4d tile_read(const double* tp, int lane) {
    v4d t;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = tp[r * 64 + lane];
    return t;
}
__device__ __forceinline__ void tile_write(double* tp, int lane, v4d t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) tp[r * 64 + lane] = t[r];
}
__device__ __forceinline__ v4d tile_identity(int n, int G) {
    v4d t;
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = (n == 4 * r + G) ? 1.0 : 0.0;
    return t;
}
// Global traffic of a tile goes as whole rows -- lane l moves the two doubles (row l / 8 + 8 h, columns 2 (l % 8), + 1), 16 bytes,
// eight 128-byte row segments per wave instruction -- and changes layout in LDS.  In the register layout itself (lane (n, G): 16
// rows x 32 bytes per 8-byte instruction) a tile's four loads or stores took ~160-240 cycles EACH to issue and queued on the
// CU's one address path: 10 us of staging and ~1 us per block column in front of the pivot wave (tools/diag2_lab.hip).
typedef double v2d_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int tile_word(int row, int c) { return (c >> 2) * 64 + row + 16 * (c & 3); }  // [n][c] in the image
// tile (i, j), i >= j, of the block whose lower triangle is stored at Ab -> LDS image tp; a diagonal tile is mirrored
__device__ __forceinline__ void tile_g2l(double* __restrict__ tp, const double* __restrict__ Ab, int lda, int i, int j, int lane) {
    v2d_ v[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
        v[h] = *reinterpret_cast<const v2d_*>(Ab + (size_t)(16 * i + (lane >> 3) + 8 * h) * lda + 16 * j + 2 * (lane & 7));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = (lane >> 3) + 8 * h, c = 2 * (lane & 7);
        if (i != j) {
            tp[tile_word(row, c)] = v[h][0];
            tp[tile_word(row, c + 1)] = v[h][1];
        } else {  // stored: c <= row; the image is the full symmetric tile
            if (c <= row) {
                tp[tile_word(row, c)] = v[h][0];
                tp[tile_word(c, row)] = v[h][0];
            }
            if (c + 1 <= row) {
                tp[tile_word(row, c + 1)] = v[h][1];
                tp[tile_word(c + 1, row)] = v[h][1];
            }
        }
    }
}
// LDS image tp -> tile (i, j) of the matrix at Ab; a diagonal tile goes out with zeros above its diagonal
__device__ __forceinline__ void tile_l2g(const double* __restrict__ tp, double* __restrict__ Ab, int lda, int i, int j, int lane) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = (lane >> 3) + 8 * h, c = 2 * (lane & 7);
        v2d_ v;
        v[0] = tp[tile_word(row, c)];
        v[1] = tp[tile_word(row, c + 1)];
        if (i == j) {
            v[0] = (c <= row) ? v[0] : 0.0;
            v[1] = (c + 1 <= row) ? v[1] : 0.0;
        }
        *reinterpret_cast<v2d_*>(Ab + (size_t)(16 * i + row) * lda + 16 * j + c) = v;
    }
}
__device__ __forceinline__ void tile_zero_global(double* __restrict__ Ab, int lda, int i, int j, int lane) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
        *reinterpret_cast<v2d_*>(Ab + (size_t)(16 * i + (lane >> 3) + 8 * h) * lda + 16 * j + 2 * (lane & 7)) = v2d_{0.0, 0.0};
}

#ifdef TSVGP_DIAG_D2  // tools/diag2_lab.hip: s_memtime stamps of the pivot wave, 16 per block column + 8
__device__ unsigned long long* g_d2_dbg = nullptr;
#define D2_STAMP(i) { if (w == 0 && g_d2_dbg) { const unsigned long long tt_ = __builtin_amdgcn_s_memtime(); if (lane == 0) g_d2_dbg[i] = tt_; } }
#else
#define D2_STAMP(i)
#endif

// v shifted up by SH lanes inside every row of 16 lanes (DPP row_shr: lane l reads lane l - SH), zeros shifted in
template <int SH>
__device__ __forceinline__ double row_shr(double v) {
    if constexpr (SH == 0) return v;
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x110 + SH, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x110 + SH, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// the entry (max(n, G), min(n, G)) of a symmetric 4 x 4 matrix for the lanes n < 4 (sel = 4 max + min), 0 elsewhere (sel < 0).
// A TREE of selects (depth 4) on the two index bits of each coordinate: as a chain of ten dependent v_cndmask pairs it was
// ~160 cycles of the pivot wave's critical path, twice per pivot group.
__device__ __forceinline__ double sel_sym(int sel, double v00, double v10, double v11, double v20, double v21, double v22,
                                          double v30, double v31, double v32, double v33) {
    const int hi = sel >> 2, lo = sel & 3;  // hi = max, lo = min (sel >= 0)
    const bool l1 = lo & 1, l2 = lo & 2;
    // row `hi` of the lower triangle, indexed by lo (entries beyond the diagonal are never selected: lo <= hi)
    const double r0 = v00;
    const double r1 = l1 ? v11 : v10;
    const double r2 = l2 ? v22 : (l1 ? v21 : v20);
    const double r3 = l2 ? (l1 ? v33 : v32) : (l1 ? v31 : v30);
    const double a = (hi & 2) ? ((hi & 1) ? r3 : r2) : ((hi & 1) ? r1 : r0);
    return sel < 0 ? 0.0 : a;
}

// A tile of the column through pivot group Q: T' = [T with register Q zeroed] + mfma(aop, T[Q]) -- rows of aop above the
// pivot block are zero (finished columns untouched), the pivot rows hold inv(L_pp) (register Q becomes the solved columns
// A_q inv(L_pp)^T), the rows below hold -A_d inv(a_pp) (the rank-4 update of the columns to the right).
template <int Q>
__device__ __forceinline__ void tile_step(v4d& t, double aop) {
    const double bq = t[Q];
    t[Q] = 0.0;
    t = mfma(aop, bq, t);
}

// The 4 x 4 pivot chain of group Q on the diagonal tile D (vector ALU, wave-uniform values) and the A operand of the
// group's tile steps.  `fb`: 1-based position of the first non-positive pivot inside the group, 0 for none.
template <int Q>
__device__ __forceinline__ double pivot_chain(const v4d& D, int n, int sel_s, bool lower, int& fb) {
    const double tq = D[Q];
    const double p00 = readlane_d(tq, 4 * Q), p10 = readlane_d(tq, 4 * Q + 1), p20 = readlane_d(tq, 4 * Q + 2),
                 p30 = readlane_d(tq, 4 * Q + 3);
    const double p11 = readlane_d(tq, 4 * Q + 1 + 16), p21 = readlane_d(tq, 4 * Q + 2 + 16), p31 = readlane_d(tq, 4 * Q + 3 + 16);
    const double p22 = readlane_d(tq, 4 * Q + 2 + 32), p32 = readlane_d(tq, 4 * Q + 3 + 32);
    const double p33 = readlane_d(tq, 4 * Q + 3 + 48);
    const double i0 = rsqrt_nr(p00);
    const double l10 = p10 * i0, l20 = p20 * i0, l30 = p30 * i0;
    const double d1 = fma(-l10, l10, p11);
    const double i1 = rsqrt_nr(d1);
    const double l21 = fma(-l20, l10, p21) * i1, l31 = fma(-l30, l10, p31) * i1;
    const double d2 = fma(-l21, l21, fma(-l20, l20, p22));
    const double i2 = rsqrt_nr(d2);
    const double r10 = -l10 * i0 * i1;
    const double l32 = fma(-l31, l21, fma(-l30, l20, p32)) * i2;
    const double d3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, p33)));
    const double i3 = rsqrt_nr(d3);
    const double r21 = -l21 * i1 * i2;
    const double r20 = -fma(l21, r10, l20 * i0) * i2;
    // R = inv(L_pp) (lower), Pinv = R^T R = inv(a_pp)
    const double r32 = -l32 * i2 * i3;
    const double r31 = -fma(l32, r21, l31 * i1) * i3;
    const double r30 = -fma(l32, r20, fma(l31, r10, l30 * i0)) * i3;
    fb = !(p00 > 0.0) ? 1 : !(d1 > 0.0) ? 2 : !(d2 > 0.0) ? 3 : !(d3 > 0.0) ? 4 : 0;
    double aop = row_shr<4 * Q>(lower ? sel_sym(sel_s, i0, r10, i1, r20, r21, i2, r30, r31, r32, i3) : 0.0);
    if constexpr (Q < 3) {
        const double q33 = i3 * i3, q32 = i3 * r32, q31 = i3 * r31, q30 = i3 * r30;
        const double q22 = fma(r32, r32, i2 * i2), q21 = fma(r32, r31, i2 * r21), q20 = fma(r32, r30, i2 * r20);
        const double q11 = fma(r31, r31, fma(r21, r21, i1 * i1)), q10 = fma(r31, r30, fma(r21, r20, i1 * r10));
        const double q00 = fma(r30, r30, fma(r20, r20, fma(r10, r10, i0 * i0)));
        const double pop = sel_sym(sel_s, q00, q10, q11, q20, q21, q22, q30, q31, q32, q33);
        const v4d zero = {0.0, 0.0, 0.0, 0.0};
        const double W = mfma(pop, tq, zero)[0];  // lane (m, k): (A_d inv(a_pp))[m][k]
        aop = (n >= 4 * Q + 4) ? -W : aop;
    }
    return aop;
}

// Workgroup barrier for LDS traffic only: __syncthreads() also waits for the wave's outstanding GLOBAL stores (vmcnt(0)), and the
// finished tiles leave for global memory right in front of the barriers -- every barrier then sat out a store round trip (1-2 us).
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct Helper {  // what a helper wave keeps in registers: its tile row and the look-ahead sum of the next column's tile
    v4d t, la;
    int row;      // 1 .. 7
    bool active;  // until the row's tile has become the diagonal tile
};

// job g of the flattened list of tiles (j, u) = tile (j + u, j), j = p + 2 .. 7, u = 0 .. 7 - j (column by column); false
// past the end.  d = j - p.
__device__ __forceinline__ bool job_decode(int p, int g, int& d, int& u) {
    d = 2;
    while (p + d < NTC && g >= NTC - (p + d)) {
        g -= NTC - (p + d);
        ++d;
    }
    u = g;
    return p + d < NTC;
}

// Trailing job g of column p (p >= 0): the rank-16 update of tile (j, u) -- column j slot u pairs with column p slot u + d, the A
// operand is column p slot d.
__device__ __forceinline__ void helper_job(double* __restrict__ S, int p, int g, int lane) {
    int d, u;
    if (p >= 0 && job_decode(p, g, d, u)) {
        double* tp = tile_ptr(S, p + d, u);
        const v4d a = tile_read(tile_ptr(S, p, d), lane);
        const v4d bq = tile_read(tile_ptr(S, p, u + d), lane);
        v4d c = tile_read(tp, lane);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) c = mfma(-a[kk], bq[kk], c);
        tile_write(tp, lane, c);
    }
}

// What the helper waves do behind barrier G(s, Q): the group's step on their own tile (A operand from the pivot wave through
// LDS), register Q of the tile published at once (it is final), the look-ahead product of the register finished one group
// ago (A operand: the same register of row s + 1's tile, published by its owner one barrier ago), and ONE job.
template <int Q>
__device__ __forceinline__ void helper_group(Helper& h, double* __restrict__ S, const double* __restrict__ aopbuf,
                                             double* __restrict__ Ab, int lda, double* __restrict__ Wb, int s, int w, int lane,
                                             int n, int G) {
    if (h.active) {
        const double aop = aopbuf[Q * 64 + lane];
        if constexpr (Q >= 1) {
            if (s + 1 < NTC) h.la = mfma(-tile_ptr(S, s, 1)[(Q - 1) * 64 + lane], h.t[Q - 1], h.la);
        }
        tile_step<Q>(h.t, aop);
        tile_ptr(S, s, h.row - s)[Q * 64 + lane] = h.t[Q];
        if constexpr (Q == 3) {  // tile (row, s) of L is final: in register layout to the panel kernel (512 contiguous bytes per
            double* wt = Wb + (size_t)tile_index(s, h.row - s) * 256;  // register; the matrix gets it behind barrier E)
#pragma unroll
            for (int r = 0; r < 4; ++r) wt[r * 64 + lane] = h.t[r];
        }
    }
    if constexpr (Q < 3) helper_job(S, s - 1, (w - 1) + 7 * Q, lane);  // (the fourth job of a phase runs behind barrier E)
}

__global__ __launch_bounds__(D2_THREADS) void potrf_diag2_kernel(double* __restrict__ A, int lda, int64_t stride, int k,
                                                                 double* __restrict__ work, int* __restrict__ info,
                                                                 int need_inverse) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* S = reinterpret_cast<double*>(smem_raw);  // 36 tiles: column j slot u = tile (j + u, j)
    double* aopbuf = S + (size_t)NTILES * 256;         // [4 groups][64 lanes]: the pivot wave's A operands of a column
    const int t = threadIdx.x, lane = t & 63, n = lane & 15, G = lane >> 4, b = blockIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);
    double* Ab = A + (size_t)b * stride + (size_t)k * NB * lda + (size_t)k * NB;
    double* Wb = work + (size_t)b * NB * NB;
    const int sel_s = n < 4 ? (n > G ? 4 * n + G : 4 * G + n) : -1;
    const bool lower = G <= n;
    D2_STAMP(0)
    v4d D, X;
    Helper h;
    int bad = 0;
    if (w == 0) {
        __builtin_amdgcn_s_setprio(3);
        tile_g2l(tile_ptr(S, 0, 0), Ab, lda, 0, 0, lane);  // (this wave's own LDS words: no barrier in between)
        D = tile_read(tile_ptr(S, 0, 0), lane);
    } else {
        __builtin_amdgcn_s_setprio(2);
        h.row = w;
        h.active = true;
        h.la = v4d{0.0, 0.0, 0.0, 0.0};
        // staging: the wave's own tile (w, 0), and four of the 28 tiles of the columns >= 1 (job list of p = -1) with zeros to
        // their mirrors above the diagonal of the block -- everything requested before anything is waited for
        tile_g2l(tile_ptr(S, 0, w), Ab, lda, w, 0, lane);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int d, u;
            if (job_decode(-1, (w - 1) + 7 * q, d, u)) {
                const int j = d - 1;
                tile_g2l(tile_ptr(S, j, u), Ab, lda, j + u, j, lane);
                if (u > 0) tile_zero_global(Ab, lda, j, j + u, lane);
            }
        }
        tile_zero_global(Ab, lda, 0, w, lane);
        h.t = tile_read(tile_ptr(S, 0, w), lane);
    }
    for (int s = 0; s < NTC; ++s) {
        D2_STAMP(8 + 16 * s + 0)
        if (w == NTC - 1) X = tile_identity(n, G);  // (the inverse tile rides in wave 7: one MFMA per group off the pivot wave)
#define D2_GROUP(Q)                                                               \
        if (w == 0) {                                                             \
            int fb;                                                               \
            const double aop = pivot_chain<Q>(D, n, sel_s, lower, fb);            \
            aopbuf[Q * 64 + lane] = aop;                                          \
            tile_step<Q>(D, aop);                                                 \
            if (bad == 0 && fb != 0) bad = 16 * s + 4 * Q + fb;                   \
        }                                                                         \
        lds_barrier(); /* G(s, Q) */                                                \
        if (w != 0) helper_group<Q>(h, S, aopbuf, Ab, lda, Wb, s, w, lane, n, G); \
        if (w == NTC - 1) tile_step<Q>(X, aopbuf[Q * 64 + lane]);                 \
        D2_STAMP(8 + 16 * s + 1 + Q)
        D2_GROUP(0)
        D2_GROUP(1)
        D2_GROUP(2)
        D2_GROUP(3)
#undef D2_GROUP
        if (w == 0) {
            tile_write(tile_ptr(S, s, 0), lane, D);  // L_ss: a helper takes it to the matrix behind the barrier
        } else if (s + 1 < NTC && h.active && h.row == s + 1) {
            // the owner of row s + 1 finishes the diagonal tile of the next column from its own registers and hands it over
            h.la = mfma(-h.t[3], h.t[3], h.la);
            double* dp = tile_ptr(S, s + 1, 0);
            tile_write(dp, lane, tile_read(dp, lane) + h.la);
            h.active = false;
        }
        if (w == NTC - 1 && need_inverse) {  // X[n][c] = inv(L_ss)[c][n] row-major to the panel kernel
#pragma unroll
            for (int r = 0; r < 4; ++r) Wb[(size_t)NTILES * 256 + s * 256 + (4 * r + G) * 16 + n] = X[r];
        }
        D2_STAMP(8 + 16 * s + 5)
        lds_barrier();  // E(s)
        D2_STAMP(8 + 16 * s + 6)
        if (w != 0) {
            // column s is complete in LDS: its tiles go to the matrix, one per helper (tile (s + u, s) by wave 1 + u % 7), and the
            // phase's fourth trailing job runs -- both in the shadow of the next column's first pivot chain
            for (int u = w - 1; u < NTC - s; u += NTC - 1) tile_l2g(tile_ptr(S, s, u), Ab, lda, s + u, s, lane);
            helper_job(S, s - 1, (w - 1) + 7 * 3, lane);
        }
        if (s + 1 < NTC) {
            if (w == 0) {
                D = tile_read(tile_ptr(S, s + 1, 0), lane);
            } else if (h.active) {
                // the last look-ahead product; the next column's tile is its LDS copy (which carries the updates of the
                // columns before s) plus the look-ahead sum
                h.la = mfma(-tile_ptr(S, s, 1)[3 * 64 + lane], h.t[3], h.la);
                h.t = tile_read(tile_ptr(S, s + 1, h.row - (s + 1)), lane) + h.la;
                h.la = v4d{0.0, 0.0, 0.0, 0.0};
            }
        }
        D2_STAMP(8 + 16 * s + 7)
    }
    if (t == 0 && (k == 0 || (bad != 0 && info[b] == 0))) info[b] = bad != 0 ? k * NB + bad : 0;  // (block 0 initialises the status word)
    D2_STAMP(1)
}

// Panel rows below the diagonal block (and the right-hand-side rows that ride along, tsvgp_potrf_solve_f64):
//     P = A_panel inv(L_kk)^T     by substitution over the eight 16-wide column blocks, right-looking:
//     P_s = U_s inv(L_ss)^T,   U_s' -= P_s L_s's^T  (s' > s)
// One WAVE per 16-row strip, the strip's eight tiles in registers in the accumulator layout: the product with the inverted
// diagonal tile is four MFMAs whose B operands are the registers of U_s as they stand, and P_s's registers are in turn the
// B operands of the updates -- 144 MFMAs per strip against the 256 of the product with the assembled 128 x 128 inverse
// (round 4's chol_tile_kernel<0>), and no inverse to assemble.  The operands on the factor's side (the tiles L_s's and
// inv(L_ss), 36 KB) come from `work`, where potrf_diag2_kernel left them in register layout: coalesced 512-byte loads.
// The factor-side operands of column block s + 1 are requested before the MFMAs of column block s (they do not depend on
// them): without that every stage sat out one L2 round trip per operand (tools/dev_diag2.py: 40 us instead of 6).
constexpr int P2_THREADS = 256;
template <int S>
struct P2Ops {  // operands of column block S: inv(L_SS) and the tiles (s2, S), s2 > S
    double x[4];
    double l[NTC - 1 - S > 0 ? NTC - 1 - S : 1][4];
};
template <int S>
__device__ __forceinline__ void p2_load(P2Ops<S>& o, const double* __restrict__ Wb, int lane, int n, int G) {
    const double* Xb = Wb + (size_t)NTILES * 256 + S * 256 + n * 16 + G;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) o.x[kk] = Xb[4 * kk];
#pragma unroll
    for (int s2 = S + 1; s2 < NTC; ++s2) {
        const double* lt = Wb + (size_t)tile_index(S, s2 - S) * 256 + lane;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) o.l[s2 - S - 1][kk] = lt[kk * 64];
    }
}
// (The solved tiles stay in registers until the end: vmcnt counts loads and stores in one queue, so a store issued in stage s
// would make the wait for stage s + 1's operands sit out the store's round trip as well.)
template <int S>
__device__ __forceinline__ void p2_stage(v4d (&U)[NTC], const P2Ops<S>& o) {
    v4d ps = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) ps = mfma(o.x[kk], U[S][kk], ps);
#pragma unroll
    for (int s2 = S + 1; s2 < NTC; ++s2)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) U[s2] = mfma(-o.l[s2 - S - 1][kk], ps[kk], U[s2]);
    U[S] = ps;
}
// Strip I/O: a strip's 16 rows x 128 columns travel as whole 512-byte row halves (16 bytes per lane, two rows per
// instruction) and change layout in LDS -- read straight in the register layout (lane (n, G), tile s, register r <->
// [n][16 s + 4 r + G]: 16 x 32-byte pieces per 8-byte wave load) a strip took 6.2 us to load and 3.2 us to store, more than
// its 144 MFMAs (tools/diag2_lab.hip).  One 64-column half at a time through a private [16][66] image per wave: row stride
// 66 doubles = 4 banks (mod 64 dwords), so the 8-byte register-layout accesses of a half wave hit 32 different bank pairs.
constexpr int P2_LD = 66;
typedef double v2d_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void strip_load(v4d (&U)[NTC], const double* __restrict__ Arow0, int lda, double* __restrict__ img, int lane,
                                           int n, int G) {
    v2d_ v[2][8];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 8; ++i)  // rows 2 i, 2 i + 1: lanes 0..31 / 32..63, 16 bytes each
            v[h][i] = *reinterpret_cast<const v2d_*>(Arow0 + (size_t)(2 * i + (lane >> 5)) * lda + 64 * h + 2 * (lane & 31));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            double* d = img + (2 * i + (lane >> 5)) * P2_LD + 2 * (lane & 31);
            d[0] = v[h][i][0];
            d[1] = v[h][i][1];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) U[4 * h + s][r] = img[n * P2_LD + 16 * s + 4 * r + G];
    }
}
__device__ __forceinline__ void strip_store(const v4d (&U)[NTC], double* __restrict__ Arow0, int lda, double* __restrict__ img, int lane,
                                            int n, int G) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int r = 0; r < 4; ++r) img[n * P2_LD + 16 * s + 4 * r + G] = U[4 * h + s][r];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const double* d = img + (2 * i + (lane >> 5)) * P2_LD + 2 * (lane & 31);
            v2d_ v;
            v[0] = d[0];
            v[1] = d[1];
            *reinterpret_cast<v2d_*>(Arow0 + (size_t)(2 * i + (lane >> 5)) * lda + 64 * h + 2 * (lane & 31)) = v;
        }
    }
}

__global__ __launch_bounds__(P2_THREADS) __attribute__((amdgpu_waves_per_eu(1, 1))) void chol_panel2_kernel(
    double* __restrict__ A, int lda, int64_t stride, int k, const double* __restrict__ work, int nstrips) {
    __builtin_amdgcn_s_setprio(TSVGP_CHOL_PRIO);
    __shared__ double imgs[P2_THREADS / 64][16 * P2_LD];
    const int lane = threadIdx.x & 63, n = lane & 15, G = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int strip = blockIdx.x * (P2_THREADS / 64) + w;
    if (strip >= nstrips) return;
    double* Arow0 = A + (size_t)blockIdx.y * stride + (size_t)((k + 1) * NB + 16 * strip) * lda + (size_t)k * NB;
    const double* Wb = work + (size_t)blockIdx.y * NB * NB;
#ifdef TSVGP_DIAG_D2
#define P2_STAMP(i) { if (g_d2_dbg && strip == 0 && blockIdx.y == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long tt_ = __builtin_amdgcn_s_memtime(); if (lane == 0) g_d2_dbg[200 + i] = tt_; } }
#else
#define P2_STAMP(i)
#endif
    P2_STAMP(0)
    P2Ops<0> o0;
    P2Ops<1> o1;
    P2Ops<2> o2;
    P2Ops<3> o3;
    P2Ops<4> o4;
    P2Ops<5> o5;
    P2Ops<6> o6;
    P2Ops<7> o7;
    v4d U[NTC];
    p2_load(o0, Wb, lane, n, G);
    strip_load(U, Arow0, lda, imgs[w], lane, n, G);
    p2_load(o1, Wb, lane, n, G);
    __builtin_amdgcn_sched_barrier(0);
    P2_STAMP(1)
    p2_stage(U, o0);
    P2_STAMP(2)
    p2_load(o2, Wb, lane, n, G);
    __builtin_amdgcn_sched_barrier(0);
    p2_stage(U, o1);
    p2_load(o3, Wb, lane, n, G);
    __builtin_amdgcn_sched_barrier(0);
    p2_stage(U, o2);
    p2_load(o4, Wb, lane, n, G);
    p2_load(o5, Wb, lane, n, G);
    __builtin_amdgcn_sched_barrier(0);
    p2_stage(U, o3);
    p2_load(o6, Wb, lane, n, G);
    p2_load(o7, Wb, lane, n, G);
    __builtin_amdgcn_sched_barrier(0);
    p2_stage(U, o4);
    p2_stage(U, o5);
    p2_stage(U, o6);
    p2_stage(U, o7);
    P2_STAMP(3)
    strip_store(U, Arow0, lda, imgs[w], lane, n, G);
    P2_STAMP(4)
}

}  // namespace

namespace tsvgp_chol {

hipError_t launch_diag2(double* A, int lda, int64_t stride, int k, double* work, int* info, int need_inverse, int batch,
                        hipStream_t stream) {
    static bool opted[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return hipErrorInvalidDevice;
    if (dev < 0 || dev >= 64 || !opted[dev]) {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_diag2_kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)D2_LDS_BYTES);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) opted[dev] = true;
    }
    hipLaunchKernelGGL(potrf_diag2_kernel, dim3(batch), dim3(D2_THREADS), D2_LDS_BYTES, stream, A, lda, stride, k, work, info,
                       need_inverse);
    return hipGetLastError();
}

hipError_t launch_panel2(double* A, int lda, int64_t stride, int k, const double* work, int nstrips, int batch,
                         hipStream_t stream) {
    if (nstrips <= 0) return hipSuccess;
    hipLaunchKernelGGL(chol_panel2_kernel, dim3((nstrips + P2_THREADS / 64 - 1) / (P2_THREADS / 64), batch), dim3(P2_THREADS), 0,
                       stream, A, lda, stride, k, work, nstrips);
    return hipGetLastError();
}

}  // namespace tsvgp_chol
