// potrf.hip -- blocked right-looking Cholesky Khat = U^T U on the augmented factor buffer,
// W = U^-T by blocked forward substitution, and the two small vector kernels around them.
//
// Replaces torch.linalg.cholesky_ex / triangular solves that gpytorch runs behind
// `latent_output.log_prob(proj_target)` (projected_lmc.py:1201) and
// gp.mlls.ExactMarginalLogLikelihood (experiments.py:233); SURVEY.md 8a rows a3/a4.
//
// Per block row k (NB = 128):  k_diag  (factor + invert the diagonal block in LDS)
//                              k_panel (row panel  <- V_kk^T * panel,   MFMA)
//                              k_trail (trailing   -= panel^T * panel,  MFMA, upper tiles + aug)
// The augmented columns ride along, so U^-T y (and U^-T K*^T for prediction) cost nothing extra.
#include "api_common.hpp"
#include "covariance.hpp"
#include "../../include/plmc.h"

namespace plmc {

// ----------------------------------------------------------------------------------------------
// Diagonal block: right-looking elimination of [A_kk | I] in LDS (packed triangles), giving
// U_kk (upper) and W_kk = U_kk^-T (lower) at once.  One workgroup of 1024 threads per latent.
// Row scaling is deferred (sInv) so each step needs a single barrier.
constexpr int DIAG_THREADS = 1024;

__device__ __forceinline__ int rowU(int i) { return i * NB - (i * (i - 1)) / 2 - i; }   // U[i][j] at rowU(i)+j, j>=i
__device__ __forceinline__ int rowL(int i) { return (i * (i + 1)) / 2; }                // W[i][c] at rowL(i)+c, c<=i
constexpr int TRI = NB * (NB + 1) / 2;

template <typename T>
__global__ __launch_bounds__(DIAG_THREADS) void k_diag(T *__restrict__ A, int64_t lda, int64_t strideA, int kblk,
                                                        T *__restrict__ Vd, int64_t strideV,
                                                        double *__restrict__ logdet, int *__restrict__ info) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T *sU = reinterpret_cast<T *>(smem_raw);
  T *sW = sU + TRI;
  T *sInv = sW + TRI;
  const int lat = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = DIAG_THREADS / 64;
  T *blk = A + (int64_t)lat * strideA + (int64_t)kblk * NB * lda + (int64_t)kblk * NB;

  for (int e = tid; e < NB * NB; e += DIAG_THREADS) {
    int i = e >> 7, j = e & 127;
    if (j >= i) sU[rowU(i) + j] = blk[(int64_t)i * lda + j];
    if (j <= i) sW[rowL(i) + j] = (i == j) ? T(1) : T(0);
  }
  __syncthreads();

  double lacc = 0.0;
  int bad = 0;
  for (int k = 0; k < NB; ++k) {
    const T piv = sU[rowU(k) + k];
    const bool ok = piv > T(0);
    const T inv = ok ? T(1) / dsqrt(piv) : T(1);
    if (tid == 0) {
      sInv[k] = inv;
      if (ok) lacc += log((double)piv);
      else if (!bad) bad = kblk * NB + k + 1;
    }
    const int ru_k = rowU(k), rl_k = rowL(k);
    for (int i = k + 1 + wave; i < NB; i += NW) {
      const T uki = sU[ru_k + i] * inv;
      const int ru_i = rowU(i), rl_i = rowL(i);
      for (int j = i + lane; j < NB; j += 64) sU[ru_i + j] -= uki * (sU[ru_k + j] * inv);
      for (int c = lane; c <= k; c += 64) sW[rl_i + c] -= uki * (sW[rl_k + c] * inv);
    }
    __syncthreads();
  }
  // write back: U_kk (upper part only), Vd = W_kk^T as a full block with explicit zeros below.
  T *vd = Vd + (int64_t)lat * strideV + (int64_t)kblk * NB * NB;
  for (int e = tid; e < NB * NB; e += DIAG_THREADS) {
    int i = e >> 7, j = e & 127;
    if (j >= i) {
      blk[(int64_t)i * lda + j] = sU[rowU(i) + j] * sInv[i];
      vd[e] = sW[rowL(j) + i] * sInv[j];          // V[i][j] = W[j][i]
    } else {
      vd[e] = T(0);
    }
  }
  if (tid == 0) {
    if (kblk == 0) { logdet[lat] = lacc; info[lat] = bad; }
    else { logdet[lat] += lacc; if (bad && info[lat] == 0) info[lat] = bad; }
  }
}

// ----------------------------------------------------------------------------------------------
// Row panel solve: P <- V_kk^T P for the 128-row panel right of the diagonal block (and the
// augmented block).  grid (T + Taug, q); in place (each workgroup owns a full column strip).
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_panel(T *A, int64_t n_pad, int64_t lda, int64_t strideA,
                                                     int kblk, int Ttr, const T *__restrict__ Vd, int64_t strideV) {
  __shared__ __align__(16) T smem[tile_smem_elems<T>()];
  const int lat = blockIdx.y;
  const int t = blockIdx.x;
  const int64_t col0 = t < Ttr ? (int64_t)(kblk + 1 + t) * NB : n_pad + (int64_t)(t - Ttr) * NB;
  T *P = A + (int64_t)lat * strideA + (int64_t)kblk * NB * lda + col0;
  const T *V = Vd + (int64_t)lat * strideV + (int64_t)kblk * NB * NB;
  Acc<T> acc;
  acc.zero();
  tile_mainloop<T, false>(acc, V, NB, P, lda, NB, smem);
  tile_store<T>(acc, P, lda);
}

// Trailing update: C[i][j] -= sum_k P[k][i] P[k][j] over upper tiles of the trailing matrix and
// the augmented block.  grid (T + Taug, T, q).
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_trail(T *A, int64_t n_pad, int64_t lda, int64_t strideA,
                                                     int kblk, int Ttr) {
  const int bx = blockIdx.x, by = blockIdx.y, lat = blockIdx.z;
  if (bx < Ttr && bx < by) return;
  __shared__ __align__(16) T smem[tile_smem_elems<T>()];
  const int ib = kblk + 1 + by;
  const int64_t col0 = bx < Ttr ? (int64_t)(kblk + 1 + bx) * NB : n_pad + (int64_t)(bx - Ttr) * NB;
  T *Al = A + (int64_t)lat * strideA;
  const T *Prow = Al + (int64_t)kblk * NB * lda;
  Acc<T> acc;
  acc.zero();
  tile_mainloop<T, true>(acc, Prow + (int64_t)ib * NB, lda, Prow + col0, lda, NB, smem);
  tile_add_store<T>(acc, Al + (int64_t)ib * NB * lda + col0, lda);
}

// ----------------------------------------------------------------------------------------------
// W diagonal blocks: W[kb+a][kb+b] = V_k[b][a] (lower, explicit zeros above).  grid (m, q).
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_wdiag(const T *__restrict__ Vd, int64_t strideV, T *__restrict__ W,
                                                     int64_t ldw, int64_t strideW) {
  __shared__ T s[NB][NB + 1];
  const int k = blockIdx.x, lat = blockIdx.y;
  const T *v = Vd + (int64_t)lat * strideV + (int64_t)k * NB * NB;
  for (int e = threadIdx.x; e < NB * NB; e += NTHREADS) s[e >> 7][e & 127] = v[e];
  __syncthreads();
  T *w = W + (int64_t)lat * strideW + (int64_t)k * NB * ldw + (int64_t)k * NB;
  for (int e = threadIdx.x; e < NB * NB; e += NTHREADS) {
    int a = e >> 7, b = e & 127;
    w[(int64_t)a * ldw + b] = b <= a ? s[b][a] : T(0);
  }
}

// Block row k of W = U^-T (k >= 1), tiles jb < k:  S = sum_{l in [jb*NB, k*NB)} U[l][k-block]^T W[l][jb-block],
// W[k][jb] = -V_kk^T S.  grid (k, q).  S is staged through the output tile itself.
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_trtri_row(const T *__restrict__ A, int64_t lda, int64_t strideA,
                                                         const T *__restrict__ Vd, int64_t strideV, T *W,
                                                         int64_t ldw, int64_t strideW, int kblk) {
  __shared__ __align__(16) T smem[tile_smem_elems<T>()];
  const int jb = blockIdx.x, lat = blockIdx.y;
  const T *Al = A + (int64_t)lat * strideA;
  T *Wl = W + (int64_t)lat * strideW;
  Acc<T> acc;
  acc.zero();
  tile_mainloop<T, false>(acc, Al + (int64_t)jb * NB * lda + (int64_t)kblk * NB, lda,
                          Wl + (int64_t)jb * NB * ldw + (int64_t)jb * NB, ldw, (kblk - jb) * NB, smem);
  T *out = Wl + (int64_t)kblk * NB * ldw + (int64_t)jb * NB;
  tile_store<T>(acc, out, ldw);
  __threadfence_block();
  __syncthreads();
  acc.zero();
  tile_mainloop<T, true>(acc, Vd + (int64_t)lat * strideV + (int64_t)kblk * NB * NB, NB, out, ldw, NB, smem);
  tile_store<T>(acc, out, ldw);
}

// ----------------------------------------------------------------------------------------------
// z[lat][i] = A[i][n_pad + c];  quad[lat] = sum z^2 (double).  grid (q).
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_extract_col(const T *__restrict__ A, int64_t n_pad, int64_t lda,
                                                           int64_t strideA, int c, T *__restrict__ z,
                                                           double *__restrict__ quad) {
  __shared__ double red[NTHREADS];
  const int lat = blockIdx.x;
  const T *Al = A + (int64_t)lat * strideA + n_pad + c;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n_pad; i += NTHREADS) {
    T v = Al[i * lda];
    z[(int64_t)lat * n_pad + i] = v;
    s += (double)v * (double)v;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = NTHREADS / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) quad[lat] = red[0];
}

// alpha[i] = sum_{l >= block(i)} W[l][i] z[l].  grid (n_pad / 64, q); 4 row groups x 64 columns.
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_wt_matvec(const T *__restrict__ W, int64_t n_pad, int64_t ldw,
                                                         int64_t strideW, const T *__restrict__ z,
                                                         T *__restrict__ alpha) {
  __shared__ double red[4][64];
  const int lat = blockIdx.y;
  const int lane = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int64_t col = (int64_t)blockIdx.x * 64 + lane;
  const int64_t l0 = (col / NB) * NB;
  const T *Wl = W + (int64_t)lat * strideW;
  const T *zl = z + (int64_t)lat * n_pad;
  double s = 0.0;
  for (int64_t l = l0 + rg; l < n_pad; l += 4) s += (double)Wl[l * ldw + col] * (double)zl[l];
  red[rg][lane] = s;
  __syncthreads();
  if (rg == 0) alpha[(int64_t)lat * n_pad + col] = (T)(red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane]);
}

// ----------------------------------------------------------------------------------------------
template <typename T>
int potrf_impl(T *A, int64_t n_pad, int64_t lda, int naug, int64_t strideA, T *Vd, double *logdet, int *info, int q,
               void *stream) {
  PLMC_REQUIRE(A && Vd && logdet && info, "null pointer");
  PLMC_REQUIRE(n_pad > 0 && n_pad % NB == 0 && lda % NB == 0 && lda >= n_pad, "n_pad/lda must be multiples of NB");
  PLMC_REQUIRE(naug >= 0 && n_pad + naug <= lda, "naug exceeds the augmented block");
  PLMC_REQUIRE(q > 0 && aligned16(A) && aligned16(Vd), "bad q or unaligned buffer");
  hipStream_t st = (hipStream_t)stream;
  const int m = (int)(n_pad / NB);
  const int Taug = (naug + NB - 1) / NB;
  const int64_t strideV = (int64_t)m * NB * NB;
  const size_t diag_smem = (2 * TRI + NB) * sizeof(T);
  static bool attr_done[2] = {false, false};
  if (!attr_done[sizeof(T) == 8]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_diag<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        (int)diag_smem);
    attr_done[sizeof(T) == 8] = true;
  }
  for (int k = 0; k < m; ++k) {
    const int Ttr = m - k - 1;
    const double nb3 = (double)NB * NB * NB, nt = (double)Ttr * NB;
    {
      ProfScope ps(PK_DIAG, st, q * 2.0 * nb3 / 3.0, q * 3.0 * NB * NB * sizeof(T));
      hipLaunchKernelGGL(k_diag<T>, dim3(q), dim3(DIAG_THREADS), diag_smem, st, A, lda, strideA, k, Vd, strideV,
                         logdet, info);
    }
    if (Ttr + Taug > 0) {
      ProfScope ps(PK_PANEL, st, q * (nt + naug) * NB * NB, q * 2.0 * (nt + naug) * NB * sizeof(T));
      hipLaunchKernelGGL(k_panel<T>, dim3(Ttr + Taug, q), dim3(NTHREADS), 0, st, A, n_pad, lda, strideA, k, Ttr, Vd,
                         strideV);
    }
    if (Ttr > 0) {
      ProfScope ps(PK_TRAIL, st, q * (nt * nt * NB + 2.0 * nt * naug * NB),
                   q * 2.0 * (nt * nt / 2 + nt * naug) * sizeof(T));
      hipLaunchKernelGGL(k_trail<T>, dim3(Ttr + Taug, Ttr, q), dim3(NTHREADS), 0, st, A, n_pad, lda, strideA, k, Ttr);
    }
  }
  return launch_status(__func__);
}

template <typename T>
int trtri_impl(const T *A, int64_t n_pad, int64_t lda, int64_t strideA, const T *Vd, T *W, int64_t ldw,
               int64_t strideW, int q, void *stream) {
  PLMC_REQUIRE(A && Vd && W, "null pointer");
  PLMC_REQUIRE(n_pad > 0 && n_pad % NB == 0 && lda % NB == 0 && ldw % NB == 0 && ldw >= n_pad, "bad leading dims");
  PLMC_REQUIRE(q > 0 && aligned16(A) && aligned16(W) && aligned16(Vd), "bad q or unaligned buffer");
  hipStream_t st = (hipStream_t)stream;
  const int m = (int)(n_pad / NB);
  const int64_t strideV = (int64_t)m * NB * NB;
  {
    ProfScope ps(PK_WDIAG, st, 0.0, q * 2.0 * m * NB * NB * sizeof(T));
    hipLaunchKernelGGL(k_wdiag<T>, dim3(m, q), dim3(NTHREADS), 0, st, Vd, strideV, W, ldw, strideW);
  }
  for (int k = 1; k < m; ++k) {
    const double kn = (double)k * NB;
    ProfScope ps(PK_TRTRI, st, q * (NB * kn * kn + (double)NB * NB * kn), q * (kn * kn / 2 + 2.0 * kn * NB) * sizeof(T));
    hipLaunchKernelGGL(k_trtri_row<T>, dim3(k, q), dim3(NTHREADS), 0, st, A, lda, strideA, Vd, strideV, W, ldw,
                       strideW, k);
  }
  return launch_status(__func__);
}

template <typename T>
int extract_col_impl(const T *A, int64_t n_pad, int64_t lda, int64_t strideA, int c, T *z, double *quad, int q,
                     void *stream) {
  PLMC_REQUIRE(A && z && quad, "null pointer");
  PLMC_REQUIRE(c >= 0 && n_pad + c < lda, "column outside the augmented block");
  ProfScope ps(PK_EXTRACT, (hipStream_t)stream, 0.0, q * 2.0 * n_pad * sizeof(T));
  hipLaunchKernelGGL(k_extract_col<T>, dim3(q), dim3(NTHREADS), 0, (hipStream_t)stream, A, n_pad, lda, strideA, c, z,
                     quad);
  return launch_status(__func__);
}

template <typename T>
int wt_matvec_impl(const T *W, int64_t n_pad, int64_t ldw, int64_t strideW, const T *z, T *alpha, int q,
                   void *stream) {
  PLMC_REQUIRE(W && z && alpha, "null pointer");
  PLMC_REQUIRE(n_pad > 0 && n_pad % NB == 0, "n_pad must be a multiple of NB");
  ProfScope ps(PK_WTMV, (hipStream_t)stream, q * (double)n_pad * n_pad, q * ((double)n_pad * n_pad / 2) * sizeof(T));
  hipLaunchKernelGGL(k_wt_matvec<T>, dim3((unsigned)(n_pad / 64), q), dim3(NTHREADS), 0, (hipStream_t)stream, W,
                     n_pad, ldw, strideW, z, alpha);
  return launch_status(__func__);
}

}  // namespace plmc

extern "C" {
int plmc_potrf_f32(float *A, int64_t n_pad, int64_t lda, int naug, int64_t strideA, float *Vd, double *logdet,
                   int *info, int q, void *stream) {
  return plmc::potrf_impl<float>(A, n_pad, lda, naug, strideA, Vd, logdet, info, q, stream);
}
int plmc_potrf_f64(double *A, int64_t n_pad, int64_t lda, int naug, int64_t strideA, double *Vd, double *logdet,
                   int *info, int q, void *stream) {
  return plmc::potrf_impl<double>(A, n_pad, lda, naug, strideA, Vd, logdet, info, q, stream);
}
int plmc_trtri_f32(const float *A, int64_t n_pad, int64_t lda, int64_t strideA, const float *Vd, float *W,
                   int64_t ldw, int64_t strideW, int q, void *stream) {
  return plmc::trtri_impl<float>(A, n_pad, lda, strideA, Vd, W, ldw, strideW, q, stream);
}
int plmc_trtri_f64(const double *A, int64_t n_pad, int64_t lda, int64_t strideA, const double *Vd, double *W,
                   int64_t ldw, int64_t strideW, int q, void *stream) {
  return plmc::trtri_impl<double>(A, n_pad, lda, strideA, Vd, W, ldw, strideW, q, stream);
}
int plmc_extract_col_f32(const float *A, int64_t n_pad, int64_t lda, int64_t strideA, int c, float *z, double *quad,
                         int q, void *stream) {
  return plmc::extract_col_impl<float>(A, n_pad, lda, strideA, c, z, quad, q, stream);
}
int plmc_extract_col_f64(const double *A, int64_t n_pad, int64_t lda, int64_t strideA, int c, double *z,
                         double *quad, int q, void *stream) {
  return plmc::extract_col_impl<double>(A, n_pad, lda, strideA, c, z, quad, q, stream);
}
int plmc_wt_matvec_f32(const float *W, int64_t n_pad, int64_t ldw, int64_t strideW, const float *z, float *alpha,
                       int q, void *stream) {
  return plmc::wt_matvec_impl<float>(W, n_pad, ldw, strideW, z, alpha, q, stream);
}
int plmc_wt_matvec_f64(const double *W, int64_t n_pad, int64_t ldw, int64_t strideW, const double *z, double *alpha,
                       int q, void *stream) {
  return plmc::wt_matvec_impl<double>(W, n_pad, ldw, strideW, z, alpha, q, stream);
}
}
