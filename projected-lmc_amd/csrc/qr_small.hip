// qr_small.hip -- Householder QR of one small mixing matrix in a single launch.
//
// Replaces `torch.linalg.qr(H)` of LMCMixingMatrix.QR in bulk mode (projected_lmc.py:864-875):
// on the device that call is rocSOLVER geqrf + orgqr, ~45 dependent launches (0.37 ms of launch latency for a
// 16 x 16 matrix) at the head of every training step; for a single-latent shard that is 5 % of the step.
// Same convention as LAPACK xGEQR2 / xORG2R (which rocSOLVER and the CPU path of torch follow), so Q and R agree
// with the reference's factors to rounding, signs included:
//     beta = -sign(a_kk) |a_k:m,k|,  tau = (beta - a_kk) / beta,  v = a_k+1:m,k / (a_kk - beta),  H_k = I - tau v v^T,
//     a column that is already zero below the diagonal gets tau = 0 (its R_kk keeps its sign).
// One wave: lane j owns column j for the reflector applications (conflict-free row-major LDS rows), lane i owns
// row i for the norms.  Arithmetic in double for both precisions (the matrices are tiny), rounded once at the end.
#include "api_common.hpp"
#include "../../include/plmc.h"

namespace plmc {

constexpr int QR_MAX = 64;

template <typename T>
__global__ __launch_bounds__(64) void k_qr_small(const T *__restrict__ A, int m, int n, int64_t lda, T *__restrict__ Q,
                                                 int64_t ldq, T *__restrict__ R, int64_t ldr) {
  extern __shared__ __align__(16) unsigned char qr_raw[];
  double *a = reinterpret_cast<double *>(qr_raw);          // [QR_MAX][QR_MAX]: R above the diagonal, reflectors below
  double *qm = a + QR_MAX * QR_MAX;                        // [QR_MAX][QR_MAX]: Q
  double *v = qm + QR_MAX * QR_MAX;                        // [QR_MAX]
  double *tau = v + QR_MAX;                                // [QR_MAX]
  const int t = threadIdx.x;
  for (int e = t; e < QR_MAX * QR_MAX; e += 64) {
    const int i = e / QR_MAX, j = e % QR_MAX;
    a[e] = (i < m && j < n) ? (double)A[(int64_t)i * lda + j] : 0.0;
    qm[e] = (i == j && j < n) ? 1.0 : 0.0;
  }
  __syncthreads();
  auto wave_sum = [](double x) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    return x;
  };
  for (int k = 0; k < n; ++k) {
    const double x = (t > k && t < m) ? a[t * QR_MAX + k] : 0.0;
    const double xn2 = wave_sum(x * x);
    const double alpha = a[k * QR_MAX + k];
    double beta = alpha, tk = 0.0, scale = 0.0;
    if (xn2 > 0.0) {
      beta = -copysign(sqrt(alpha * alpha + xn2), alpha);
      tk = (beta - alpha) / beta;
      scale = 1.0 / (alpha - beta);
    }
    v[t] = t == k ? 1.0 : (t > k && t < m ? x * scale : 0.0);
    __syncthreads();
    if (t == k) { tau[k] = tk; }
    if (t > k && t < n) {                                  // apply H_k to column t
      double w = 0.0;
      for (int i = k; i < m; ++i) w += v[i] * a[i * QR_MAX + t];
      w *= tk;
      for (int i = k; i < m; ++i) a[i * QR_MAX + t] -= w * v[i];
    }
    __syncthreads();
    if (t == k) a[k * QR_MAX + k] = beta;
    if (t > k && t < m) a[t * QR_MAX + k] = v[t];          // keep the reflector (as xGEQR2 does)
    __syncthreads();
  }
  // Q = H_0 H_1 ... H_{n-1} [I; 0]: apply the reflectors in reverse order
  for (int k = n - 1; k >= 0; --k) {
    const double tk = tau[k];
    if (t < n && tk != 0.0) {
      double w = 0.0;
      for (int i = k; i < m; ++i) w += (i == k ? 1.0 : a[i * QR_MAX + k]) * qm[i * QR_MAX + t];
      w *= tk;
      for (int i = k; i < m; ++i) qm[i * QR_MAX + t] -= w * (i == k ? 1.0 : a[i * QR_MAX + k]);
    }
    __syncthreads();
  }
  for (int e = t; e < m * n; e += 64) {
    const int i = e / n, j = e % n;
    Q[(int64_t)i * ldq + j] = (T)qm[i * QR_MAX + j];
  }
  for (int e = t; e < n * n; e += 64) {
    const int i = e / n, j = e % n;
    R[(int64_t)i * ldr + j] = i <= j ? (T)a[i * QR_MAX + j] : T(0);
  }
}

template <typename T>
int qr_small_impl(const T *A, int m, int n, int64_t lda, T *Q, int64_t ldq, T *R, int64_t ldr, void *stream) {
  PLMC_REQUIRE(A && Q && R, "null pointer");
  PLMC_REQUIRE(n >= 1 && m >= n && m <= QR_MAX, "need 1 <= n <= m <= plmc_qr_max()");
  PLMC_REQUIRE(lda >= n && ldq >= n && ldr >= n, "leading dimension too small");
  const size_t smem = (2 * QR_MAX * QR_MAX + 2 * QR_MAX) * sizeof(double);
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_qr_small<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)smem);
  hipLaunchKernelGGL(k_qr_small<T>, dim3(1), dim3(64), smem, (hipStream_t)stream, A, m, n, lda, Q, ldq, R, ldr);
  return launch_status(__func__);
}

}  // namespace plmc

extern "C" {
int plmc_qr_max(void) { return plmc::QR_MAX; }
int plmc_qr_small_f32(const float *A, int m, int n, int64_t lda, float *Q, int64_t ldq, float *R, int64_t ldr, void *stream) {
  return plmc::qr_small_impl<float>(A, m, n, lda, Q, ldq, R, ldr, stream);
}
int plmc_qr_small_f64(const double *A, int m, int n, int64_t lda, double *Q, int64_t ldq, double *R, int64_t ldr, void *stream) {
  return plmc::qr_small_impl<double>(A, m, n, lda, Q, ldq, R, ldr, stream);
}
}
