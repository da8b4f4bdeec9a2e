// tsvgp_chol.h -- internal interface between tsvgp_kernels.hip (the C-ABI, the blocked factorisation's driver) and
// tsvgp_chol.hip (the diagonal-block kernel of round 5).  Not part of the C-ABI: include/tsvgp_hip.h is.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsvgp_chol {

// The `work` image of a factored 128 x 128 diagonal block (what launch_panel2 reads), 44 x 256 doubles per matrix:
//   tiles [36][4 registers][64 lanes]: the factor's lower 16 x 16 tiles, column by column (column j, slot u = tile (j + u, j)),
//          lane (n = l & 15, G = l >> 4), register r  <->  L[16 (j + u) + n][16 j + 4 r + G]   (the MFMA accumulator layout);
//   then   [8][16][16]: inv(L_ss) of the eight diagonal tiles, row-major.
constexpr int WORK_TILES = 36;
__host__ __device__ constexpr int work_tile_index(int col, int slot) { return 8 * col - col * (col - 1) / 2 + slot; }

// Factors the 128 x 128 diagonal block k of each of `batch` matrices in place (lower factor, exact zeros above the
// diagonal of the block) and leaves in work + b * 128 * 128 what launch_panel2 needs: the factor's 36 lower 16 x 16 tiles in
// MFMA register layout and, with need_inverse, the eight inverted diagonal tiles (44 x 256 doubles in all).
// info[b] receives the 1-based column of the first non-positive pivot (only when it is still 0).  Returns hipGetLastError().
hipError_t launch_diag2(double* A, int lda, int64_t stride, int k, double* work, int* info, int need_inverse, int batch,
                        hipStream_t stream);

// The `nstrips` 16-row strips directly below diagonal block k (rows (k + 1) * 128 on; right-hand-side rows included) times
// inv(L_kk)^T, in place, from the `work` image launch_diag2 wrote.
hipError_t launch_panel2(double* A, int lda, int64_t stride, int k, const double* work, int nstrips, int batch,
                         hipStream_t stream);

}  // namespace tsvgp_chol
