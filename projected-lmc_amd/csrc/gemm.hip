// gemm.hip -- the tile engine as a plain batched "TN" product for the host layer:
//     C[b] (=, +=, -=) A[b]^T B[b],   A: K x M, B: K x N (both K-major, row-major), C: M x N.
// Used by the adjoint of the whitened inducing-point interpolation (projectedlmc/_var_engine.py: the m x m x n products
// of the Cholesky adjoint behind VariationalMultitaskGPModel / SGPR, projected_lmc.py:672-683, 302-303), which round 1
// handed to rocBLAS.  Same main loop and coalesced write-back as the trailing update of the sweep (gemm_core.hpp).
#include "api_common.hpp"
#include "../../include/plmc.h"

namespace plmc {

// grid (N / 128, M / 128, batch)
// `tri` (plmc_gemm_tn_tri_*): triangular structure of the operands, so that a tile only walks the contraction range in which both
// have entries (the skipped terms are exact zeros: the result is the same number):
//   PLMC_TRI_A_LOWER  A[k][i] = 0 for k < 128 * (i / 128)   (A a lower triangle stored K-major: W, L = U^T, Ls)       -> k >= 128 ib
//   PLMC_TRI_B_LOWER  B[k][j] = 0 for k < 128 * (j / 128)                                                           -> k >= 128 jb
//   PLMC_TRI_A_UPPER  A[k][i] = 0 for k >= 128 * (i / 128 + 1)   (A an upper triangle stored K-major: the transpose of a lower one) -> k < 128 (ib + 1)
//   PLMC_TRI_C_LOWER  only the tiles on and below the block diagonal are wanted (ib >= jb): the others are left untouched,
//   PLMC_TRI_C_ZERO   ... or, with this bit as well, written as zeros (mode 0 only).
template <typename T, int MODE>
__global__ __launch_bounds__(NTHREADS, TILE_MIN_WAVES<T>) void k_gemm_tn(const T *__restrict__ A, int64_t lda, int64_t strideA,
                                                       const T *__restrict__ B, int64_t ldb, int64_t strideB, T *C, int64_t ldc,
                                                       int64_t strideC, int K, int tri) {
  __shared__ __align__(16) T smem[tile_smem_elems<T>()];
  const int jb = blockIdx.x, ib = blockIdx.y, b = blockIdx.z;
  int k_lo = 0, k_hi = K;
  if (tri) {
    if ((tri & PLMC_TRI_C_LOWER) && ib < jb && !(tri & PLMC_TRI_C_ZERO)) return;
    if (tri & PLMC_TRI_A_LOWER) k_lo = ib * NB;
    if ((tri & PLMC_TRI_B_LOWER) && jb * NB > k_lo) k_lo = jb * NB;
    if ((tri & PLMC_TRI_A_UPPER) && (ib + 1) * NB < k_hi) k_hi = (ib + 1) * NB;
    if ((tri & PLMC_TRI_C_LOWER) && ib < jb) k_hi = 0;                 // (with PLMC_TRI_C_ZERO: an empty range stores zeros)
    if (k_lo > k_hi) k_lo = k_hi;
  }
  Acc<T> acc;
  acc.zero();
  if (k_hi > k_lo)
    tile_mainloop<T, false, false>(acc, A + (int64_t)b * strideA + (int64_t)k_lo * lda + (int64_t)ib * NB, lda,
                                   B + (int64_t)b * strideB + (int64_t)k_lo * ldb + (int64_t)jb * NB, ldb, k_hi - k_lo, smem);
  tile_writeback<T, MODE>(acc, C + (int64_t)b * strideC + (int64_t)ib * NB * ldc + (int64_t)jb * NB, ldc, smem);
}

template <typename T>
int gemm_tn_impl(int mode, int M, int N, int K, const T *A, int64_t lda, int64_t strideA, const T *B, int64_t ldb, int64_t strideB,
                 T *C, int64_t ldc, int64_t strideC, int batch, void *stream, int tri = 0) {
  PLMC_REQUIRE(A && B && C, "null pointer");
  PLMC_REQUIRE(mode >= 0 && mode <= 2, "mode: 0 store, 1 add, 2 subtract");
  PLMC_REQUIRE(tri >= 0 && tri < 32 && (!(tri & PLMC_TRI_C_ZERO) || ((tri & PLMC_TRI_C_LOWER) && mode == 0)), "tri: PLMC_TRI_* bits; C_ZERO needs C_LOWER and mode 0");
  PLMC_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0 && M % NB == 0 && N % NB == 0 && K % BK == 0,
               "M, N must be multiples of plmc_block(), K a multiple of 16 (pad with zeros)");
  constexpr int EPV = Traits<T>::EPV;
  PLMC_REQUIRE(lda >= M && ldb >= N && ldc >= N && lda % EPV == 0 && ldb % EPV == 0 && ldc % EPV == 0, "leading dimensions");
  PLMC_REQUIRE(aligned16(A) && aligned16(B) && aligned16(C) && strideA % EPV == 0 && strideB % EPV == 0 && strideC % EPV == 0,
               "16-byte alignment");
  const dim3 grid(N / NB, M / NB, batch);
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) hipLaunchKernelGGL((k_gemm_tn<T, WB_STORE>), grid, dim3(NTHREADS), 0, st, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, K, tri);
  else if (mode == 1) hipLaunchKernelGGL((k_gemm_tn<T, WB_ADD>), grid, dim3(NTHREADS), 0, st, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, K, tri);
  else hipLaunchKernelGGL((k_gemm_tn<T, WB_SUB>), grid, dim3(NTHREADS), 0, st, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, K, tri);
  return launch_status(__func__);
}

}  // namespace plmc

extern "C" {
int plmc_gemm_tn_f32(int mode, int M, int N, int K, const float *A, int64_t lda, int64_t strideA, const float *B, int64_t ldb,
                     int64_t strideB, float *C, int64_t ldc, int64_t strideC, int batch, void *stream) {
  return plmc::gemm_tn_impl<float>(mode, M, N, K, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, batch, stream);
}
int plmc_gemm_tn_f64(int mode, int M, int N, int K, const double *A, int64_t lda, int64_t strideA, const double *B, int64_t ldb,
                     int64_t strideB, double *C, int64_t ldc, int64_t strideC, int batch, void *stream) {
  return plmc::gemm_tn_impl<double>(mode, M, N, K, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, batch, stream);
}
int plmc_gemm_tn_tri_f32(int mode, int tri, int M, int N, int K, const float *A, int64_t lda, int64_t strideA, const float *B, int64_t ldb,
                         int64_t strideB, float *C, int64_t ldc, int64_t strideC, int batch, void *stream) {
  return plmc::gemm_tn_impl<float>(mode, M, N, K, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, batch, stream, tri);
}
int plmc_gemm_tn_tri_f64(int mode, int tri, int M, int N, int K, const double *A, int64_t lda, int64_t strideA, const double *B, int64_t ldb,
                         int64_t strideB, double *C, int64_t ldc, int64_t strideC, int batch, void *stream) {
  return plmc::gemm_tn_impl<double>(mode, M, N, K, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, batch, stream, tri);
}
}
