// gemm.hip -- the tile engine as a plain batched "TN" product for the host layer:
//     C[b] (=, +=, -=) A[b]^T B[b],   A: K x M, B: K x N (both K-major, row-major), C: M x N.
// Used by the adjoint of the whitened inducing-point interpolation (projectedlmc/_var_engine.py: the m x m x n products
// of the Cholesky adjoint behind VariationalMultitaskGPModel / SGPR, projected_lmc.py:672-683, 302-303), which round 1
// handed to rocBLAS.  Same main loop and coalesced write-back as the trailing update of the sweep (gemm_core.hpp).
#include "api_common.hpp"
#include "../../include/plmc.h"

namespace plmc {

// grid (N / 128, M / 128, batch)
template <typename T, int MODE>
__global__ __launch_bounds__(NTHREADS, TILE_MIN_WAVES<T>) void k_gemm_tn(const T *__restrict__ A, int64_t lda, int64_t strideA,
                                                       const T *__restrict__ B, int64_t ldb, int64_t strideB, T *C, int64_t ldc,
                                                       int64_t strideC, int K) {
  __shared__ __align__(16) T smem[tile_smem_elems<T>()];
  const int jb = blockIdx.x, ib = blockIdx.y, b = blockIdx.z;
  Acc<T> acc;
  acc.zero();
  tile_mainloop<T, false, false>(acc, A + (int64_t)b * strideA + (int64_t)ib * NB, lda, B + (int64_t)b * strideB + (int64_t)jb * NB, ldb, K,
                                 smem);
  tile_writeback<T, MODE>(acc, C + (int64_t)b * strideC + (int64_t)ib * NB * ldc + (int64_t)jb * NB, ldc, smem);
}

template <typename T>
int gemm_tn_impl(int mode, int M, int N, int K, const T *A, int64_t lda, int64_t strideA, const T *B, int64_t ldb, int64_t strideB,
                 T *C, int64_t ldc, int64_t strideC, int batch, void *stream) {
  PLMC_REQUIRE(A && B && C, "null pointer");
  PLMC_REQUIRE(mode >= 0 && mode <= 2, "mode: 0 store, 1 add, 2 subtract");
  PLMC_REQUIRE(M > 0 && N > 0 && K > 0 && batch > 0 && M % NB == 0 && N % NB == 0 && K % BK == 0,
               "M, N must be multiples of plmc_block(), K a multiple of 16 (pad with zeros)");
  constexpr int EPV = Traits<T>::EPV;
  PLMC_REQUIRE(lda >= M && ldb >= N && ldc >= N && lda % EPV == 0 && ldb % EPV == 0 && ldc % EPV == 0, "leading dimensions");
  PLMC_REQUIRE(aligned16(A) && aligned16(B) && aligned16(C) && strideA % EPV == 0 && strideB % EPV == 0 && strideC % EPV == 0,
               "16-byte alignment");
  const dim3 grid(N / NB, M / NB, batch);
  hipStream_t st = (hipStream_t)stream;
  if (mode == 0) hipLaunchKernelGGL((k_gemm_tn<T, WB_STORE>), grid, dim3(NTHREADS), 0, st, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, K);
  else if (mode == 1) hipLaunchKernelGGL((k_gemm_tn<T, WB_ADD>), grid, dim3(NTHREADS), 0, st, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, K);
  else hipLaunchKernelGGL((k_gemm_tn<T, WB_SUB>), grid, dim3(NTHREADS), 0, st, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, K);
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
}
