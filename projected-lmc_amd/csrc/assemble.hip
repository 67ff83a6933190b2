// assemble.hip -- fused covariance assembly X -> Khat (upper tiles), right-hand sides and
// cross-covariances into the augmented block.  HBM-write bound: one pass, X staged in LDS.
//
// Replaces `self.covar_module(x)` + `likelihood(dist)` of the reference
// (projected_lmc.py:316,1090,1200; kernels from handle_covar_ :151-167), i.e. the ~8 unfused
// ATen launches and q*n*n temporaries gpytorch makes per kernel evaluation.
#include "api_common.hpp"
#include "covariance.hpp"
#include "../../include/plmc.h"

namespace plmc {

// grid (m, m, q): blockIdx.x = column block jb, blockIdx.y = row block ib; tiles with jb < ib exit.
// Each thread produces 8 rows x 2 groups of 4 consecutive columns (16-byte stores for f32).
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_assemble(int kind, const T *__restrict__ X, int n, int d,
                                                        const T *__restrict__ ell, const T *__restrict__ oscale,
                                                        const T *__restrict__ noise, T *__restrict__ A,
                                                        int64_t lda, int64_t strideA, int ib0, int skip) {
  const int jb = blockIdx.x, ib = ib0 + blockIdx.y, lat = blockIdx.z;
  if (jb < ib || (ib < skip && jb < skip)) return;                 // `skip`: the leading skip x skip block triangle was written by another launch
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T *ui = reinterpret_cast<T *>(smem_raw);
  const int ldu = d + 1;
  T *uj = ui + NB * ldu;
  const int tid = threadIdx.x;
  const T *el = ell + (int64_t)lat * d;
  for (int e = tid; e < NB * d; e += NTHREADS) {
    int r = e / d, k = e % d;
    int gi = ib * NB + r, gj = jb * NB + r;
    T inv = T(1) / el[k];
    ui[r * ldu + k] = gi < n ? X[(int64_t)gi * d + k] * inv : T(0);
    uj[r * ldu + k] = gj < n ? X[(int64_t)gj * d + k] * inv : T(0);
  }
  __syncthreads();
  const T os = oscale ? oscale[lat] : T(1);
  const T nz = noise[lat];
  T *Al = A + (int64_t)lat * strideA;
  const int tx = tid & 15, ty = tid >> 4;
#pragma unroll 1
  for (int rr = 0; rr < 8; ++rr) {
    const int r = ty + 16 * rr;
    const int gi = ib * NB + r;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int c0 = tx * 4 + 64 * h;
      T v[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int gj = jb * NB + c0 + c;
        T r2 = T(0), sp = T(1);
        for (int k = 0; k < d; ++k) {
          T df = ui[r * ldu + k] - uj[(c0 + c) * ldu + k];
          r2 += df * df;
          if (kind == K_SPLINE) sp *= spline_factor(ui[r * ldu + k], uj[(c0 + c) * ldu + k]);
        }
        T val;
        if (gi < n && gj < n) {
          val = os * (kind == K_SPLINE ? sp : kern_value<T>(kind, r2));
          if (gi == gj) val += nz;
        } else {
          val = (gi == gj) ? T(1) : T(0);       // identity padding keeps the padded factor trivial
        }
        v[c] = val;
      }
      T *dst = Al + (int64_t)gi * lda + jb * NB + c0;          // 16-byte aligned: lda, NB, c0 multiples of 4
      using vec_t = typename Traits<T>::vec_t;
      constexpr int EPV = Traits<T>::EPV;
#pragma unroll
      for (int c = 0; c < 4; c += EPV) {
        vec_t o;
#pragma unroll
        for (int e = 0; e < EPV; ++e) o[e] = v[c + e];
        *reinterpret_cast<vec_t *>(dst + c) = o;
      }
    }
  }
}

// Up to 8 input dimensions (DCAP in {4, 8}): the thread's 8 columns live in registers as 4 column pairs, the loops
// over dimensions are unrolled, the kernel kind is a compile-time constant and interior tiles (off the diagonal, no
// padding) carry no per-element predicate; pairs of columns go through packed arithmetic.  Same formulas
// and the same accurate exp / sqrt as k_assemble, i.e. the same values up to the order of the distance sum.
template <typename T, int DCAP, int KIND>
__device__ __forceinline__ void assemble_tile(const T *ui, const T *uj, int ldu, T os, T nz, T *Al, int64_t lda, int ib,
                                              int jb, int n, bool edge) {
  typedef Pair<T> T2;
  // 32 lanes x 4 consecutive columns = one full 512-byte (fp32) tile row per half wave: whole-row stores
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  const int c0 = tx * 4;
  T2 u2[2][DCAP];                                         // [pair][k]
#pragma unroll
  for (int pp = 0; pp < 2; ++pp)
#pragma unroll
    for (int k = 0; k < DCAP; ++k) u2[pp][k] = T2{uj[(c0 + 2 * pp) * ldu + k], uj[(c0 + 2 * pp + 1) * ldu + k]};
#pragma unroll 2
  for (int rr = 0; rr < 16; ++rr) {
    const int r = ty + 8 * rr;
    const int gi = ib * NB + r;
    T xi[DCAP];
#pragma unroll
    for (int k = 0; k < DCAP; ++k) xi[k] = ui[r * ldu + k];
    T2 v[2];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      T2 r2 = {T(0), T(0)};
#pragma unroll
      for (int k = 0; k < DCAP; ++k) {
        const T2 df = xi[k] - u2[pp][k];
        r2 += df * df;
      }
      v[pp] = os * kern_value_pair<T>(KIND, r2);
    }
    T o[4] = {v[0].x, v[0].y, v[1].x, v[1].y};
    if (edge) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int gj = jb * NB + c0 + c;
        if (gi < n && gj < n) { if (gi == gj) o[c] += nz; }
        else o[c] = (gi == gj) ? T(1) : T(0);             // identity padding keeps the padded factor trivial
      }
    }
    T *dst = Al + (int64_t)gi * lda + jb * NB + c0;
    using vec_t = typename Traits<T>::vec_t;
    constexpr int EPV = Traits<T>::EPV;
#pragma unroll
    for (int c = 0; c < 4; c += EPV) {
      vec_t w;
#pragma unroll
      for (int e = 0; e < EPV; ++e) w[e] = o[c + e];
      *reinterpret_cast<vec_t *>(dst + c) = w;
    }
  }
}

template <typename T, int DCAP>
__global__ __launch_bounds__(NTHREADS) void k_assemble_small(int kind, const T *__restrict__ X, int n, int d,
                                                              const T *__restrict__ ell, const T *__restrict__ oscale,
                                                              const T *__restrict__ noise, T *__restrict__ A,
                                                              int64_t lda, int64_t strideA, int ib0, int skip) {
  const int jb = blockIdx.x, ib = ib0 + blockIdx.y, lat = blockIdx.z;
  if (jb < ib || (ib < skip && jb < skip)) return;
  constexpr int ldu = DCAP + 1;
  __shared__ T ui[NB * ldu], uj[NB * ldu];
  const int tid = threadIdx.x;
  const T *el = ell + (int64_t)lat * d;
  for (int e = tid; e < NB * DCAP; e += NTHREADS) {      // unused dimensions: zeros (they add 0 to every distance)
    const int r = e / DCAP, k = e % DCAP;
    const int gi = ib * NB + r, gj = jb * NB + r;
    const T inv = k < d ? T(1) / el[k] : T(0);
    ui[r * ldu + k] = (k < d && gi < n) ? X[(int64_t)gi * d + k] * inv : T(0);
    uj[r * ldu + k] = (k < d && gj < n) ? X[(int64_t)gj * d + k] * inv : T(0);
  }
  __syncthreads();
  const T os = oscale ? oscale[lat] : T(1);
  const T nz = noise[lat];
  T *Al = A + (int64_t)lat * strideA;
  const bool edge = ib == jb || (jb + 1) * NB > n;
  if (kind == K_RBF) assemble_tile<T, DCAP, K_RBF>(ui, uj, ldu, os, nz, Al, lda, ib, jb, n, edge);
  else if (kind == K_MATERN12) assemble_tile<T, DCAP, K_MATERN12>(ui, uj, ldu, os, nz, Al, lda, ib, jb, n, edge);
  else if (kind == K_MATERN32) assemble_tile<T, DCAP, K_MATERN32>(ui, uj, ldu, os, nz, Al, lda, ib, jb, n, edge);
  else assemble_tile<T, DCAP, K_MATERN52>(ui, uj, ldu, os, nz, Al, lda, ib, jb, n, edge);
}

// One thread per element of the first `ncols` augmented columns (n_pad x ncols).
template <typename T>
__global__ void k_write_rhs(const T *__restrict__ rhs, int nrhs, int n, T *__restrict__ A, int64_t n_pad,
                            int64_t lda, int64_t strideA, int c0, int64_t naug_pad, int zero_fill) {
  const int lat = blockIdx.y;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_pad * naug_pad) return;
  const int64_t i = idx / naug_pad;
  const int c = (int)(idx % naug_pad);
  T *p = A + (int64_t)lat * strideA + i * lda + n_pad + c;
  if (c >= c0 && c < c0 + nrhs) {
    *p = i < n ? rhs[((int64_t)lat * nrhs + (c - c0)) * n + i] : T(0);
  } else if (zero_fill) {
    *p = T(0);
  }
}

// Out[i][col0 + j] = os * k(x_i, xs_j).  Workgroup = 256 test points (one per thread, its scaled coordinates in registers) x
// CROSS_ROWS training points (scaled coordinates staged in LDS, read as broadcasts); every row of the block is one 1 KB
// store.  grid (ceil(ns / 256), ceil(n_rows / CROSS_ROWS), q).  Rows n .. n_rows - 1 (padding) are written as zeros.
constexpr int CROSS_ROWS = 32;
template <typename T, int DCAP>
__global__ __launch_bounds__(NTHREADS) void k_assemble_cross(int kind, const T *__restrict__ X, int n,
                                                              const T *__restrict__ Xs, int ns, int d,
                                                              const T *__restrict__ ell, const T *__restrict__ oscale,
                                                              T *__restrict__ A, int64_t n_rows, int64_t lda,
                                                              int64_t strideA, int64_t col0) {
  __shared__ T xi[CROSS_ROWS][DCAP + 1];
  const int lat = blockIdx.z;
  const int j = blockIdx.x * NTHREADS + threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.y * CROSS_ROWS;
  const T *el = ell + (int64_t)lat * d;
  for (int e = threadIdx.x; e < CROSS_ROWS * DCAP; e += NTHREADS) {
    const int r = e / DCAP, k = e % DCAP;
    xi[r][k] = (k < d && i0 + r < n) ? X[(i0 + r) * d + k] / el[k] : T(0);
  }
  T xs[DCAP];
#pragma unroll
  for (int k = 0; k < DCAP; ++k) xs[k] = (k < d && j < ns) ? Xs[(int64_t)j * d + k] / el[k] : T(0);
  const T os = oscale ? oscale[lat] : T(1);
  __syncthreads();
  if (j >= ns) return;
  T *out = A + (int64_t)lat * strideA + i0 * lda + col0 + j;
#pragma unroll 4
  for (int r = 0; r < CROSS_ROWS; ++r) {
    if (i0 + r >= n_rows) break;
    T val = T(0);
    if (i0 + r < n) {
      T r2 = T(0), sp = T(1);
#pragma unroll
      for (int k = 0; k < DCAP; ++k) {
        const T df = xi[r][k] - xs[k];
        r2 += df * df;
        if (kind == K_SPLINE && k < d) sp *= spline_factor(xi[r][k], xs[k]);
      }
      val = os * (kind == K_SPLINE ? sp : kern_value<T>(kind, r2));
    }
    out[(int64_t)r * lda] = val;
  }
}

// block rows ib0 .. ib0 + nrows - 1 (nrows < 0: all of them); the tiles right of the diagonal of those rows, the first `ncols` block
// columns only (ncols < 0: all), without the leading skip x skip block triangle
template <typename T>
int assemble_impl(int kind, const T *X, int n, int d, const T *ell, const T *oscale, const T *noise, T *A,
                  int64_t lda, int64_t strideA, int q, void *stream, int ib0 = 0, int nrows = -1, int ncols = -1, int skip = 0) {
  PLMC_REQUIRE(kind >= 0 && kind <= 4, "unknown kernel kind");
  PLMC_REQUIRE(X && ell && noise && A, "null pointer");
  PLMC_REQUIRE(n > 0 && q > 0 && d > 0 && d <= MAX_DIM, "need n>0, q>0, 0<d<=plmc_max_dim()");
  const int64_t n_pad = plmc_pad(n);
  PLMC_REQUIRE(lda >= n_pad && lda % NB == 0, "lda must be a multiple of NB and >= n_pad");
  PLMC_REQUIRE(strideA >= n_pad * lda || q == 1, "strideA too small");
  const int m = (int)(n_pad / NB);
  if (nrows < 0) nrows = m - ib0;
  if (ncols < 0) ncols = m;
  PLMC_REQUIRE(ib0 >= 0 && nrows >= 0 && ib0 + nrows <= m && ncols <= m && skip >= 0, "row / column range outside the matrix");
  if (nrows == 0 || ncols == 0) return 0;
  size_t smem = 2 * NB * (d + 1) * sizeof(T);
  const double tiles = (double)nrows * (m - ib0) - (double)nrows * (nrows - 1) / 2.0;   // upper tiles of these rows
  ProfScope ps(PK_ASSEMBLE, (hipStream_t)stream, 0.0, q * tiles * NB * NB * sizeof(T));
  if (d <= 4 && kind != K_SPLINE)              // the spline kernel is evaluated by the general kernel (not a function of r2)
    hipLaunchKernelGGL((k_assemble_small<T, 4>), dim3(ncols, nrows, q), dim3(NTHREADS), 0, (hipStream_t)stream, kind, X, n, d,
                       ell, oscale, noise, A, lda, strideA, ib0, skip);
  else if (d <= 8 && kind != K_SPLINE)
    hipLaunchKernelGGL((k_assemble_small<T, 8>), dim3(ncols, nrows, q), dim3(NTHREADS), 0, (hipStream_t)stream, kind, X, n, d,
                       ell, oscale, noise, A, lda, strideA, ib0, skip);
  else
    hipLaunchKernelGGL(k_assemble<T>, dim3(ncols, nrows, q), dim3(NTHREADS), smem, (hipStream_t)stream, kind, X, n, d, ell,
                       oscale, noise, A, lda, strideA, ib0, skip);
  return launch_status(__func__);
}

template <typename T>
int write_rhs_impl(const T *rhs, int nrhs, int n, T *A, int64_t lda, int64_t strideA, int c0, int clear_cols, int q,
                   void *stream) {
  PLMC_REQUIRE(A && (rhs || nrhs == 0), "null pointer");
  const int64_t n_pad = plmc_pad(n);
  PLMC_REQUIRE(lda > n_pad && lda % NB == 0, "no augmented block (lda must exceed n_pad)");
  PLMC_REQUIRE(c0 >= 0 && nrhs >= 0 && n_pad + c0 + nrhs <= lda, "rhs columns exceed the augmented block");
  PLMC_REQUIRE(clear_cols >= 0 && n_pad + clear_cols <= lda, "clear_cols exceeds the buffer");
  const int64_t ncols = clear_cols > c0 + nrhs ? clear_cols : c0 + nrhs;
  const int zero_fill = clear_cols > 0;
  const int64_t tot = n_pad * ncols;
  ProfScope ps(PK_WRITE_RHS, (hipStream_t)stream, 0.0, q * (double)tot * sizeof(T));
  hipLaunchKernelGGL(k_write_rhs<T>, dim3((unsigned)((tot + 255) / 256), q), dim3(256), 0, (hipStream_t)stream, rhs,
                     nrhs, n, A, n_pad, lda, strideA, c0, ncols, zero_fill);
  return launch_status(__func__);
}

template <typename T>
int assemble_cross_impl(int kind, const T *X, int n, const T *Xs, int ns, int d, const T *ell, const T *oscale, T *Out,
                        int64_t ldo, int64_t strideO, int64_t col0, int64_t n_rows, int q, void *stream) {
  PLMC_REQUIRE(kind >= 0 && kind <= 4, "unknown kernel kind");
  PLMC_REQUIRE(X && Xs && ell && Out, "null pointer");
  PLMC_REQUIRE(n > 0 && ns > 0 && q > 0 && d > 0 && d <= MAX_DIM, "bad sizes");
  PLMC_REQUIRE(n_rows >= n && col0 >= 0 && col0 + ns <= ldo, "cross block exceeds the output buffer");
  ProfScope ps(PK_CROSS, (hipStream_t)stream, 0.0, q * (double)n_rows * ns * sizeof(T));
  const dim3 grid((ns + NTHREADS - 1) / NTHREADS, (unsigned)((n_rows + CROSS_ROWS - 1) / CROSS_ROWS), q);
#define PLMC_CROSS(DC) \
  hipLaunchKernelGGL((k_assemble_cross<T, DC>), grid, dim3(NTHREADS), 0, (hipStream_t)stream, kind, X, n, Xs, ns, d, ell, oscale, Out, n_rows, ldo, strideO, col0)
  if (d <= 4) PLMC_CROSS(4);
  else if (d <= 8) PLMC_CROSS(8);
  else if (d <= 16) PLMC_CROSS(16);
  else PLMC_CROSS(32);
#undef PLMC_CROSS
  return launch_status(__func__);
}

int assemble_rows(const AssembleJob &job, int elem_bytes, void *A, int64_t lda, int64_t strideA, int q, int ib0, int nrows, void *stream,
                  int ncols, int skip) {
  if (elem_bytes == 4)
    return assemble_impl<float>(job.kind, (const float *)job.X, job.n, job.d, (const float *)job.ell, (const float *)job.oscale, (const float *)job.noise,
                                (float *)A, lda, strideA, q, stream, ib0, nrows, ncols, skip);
  return assemble_impl<double>(job.kind, (const double *)job.X, job.n, job.d, (const double *)job.ell, (const double *)job.oscale,
                               (const double *)job.noise, (double *)A, lda, strideA, q, stream, ib0, nrows, ncols, skip);
}
}  // namespace plmc

extern "C" {
int plmc_assemble_f32(int kind, const float *X, int n, int d, const float *ell, const float *oscale,
                      const float *noise, float *A, int64_t lda, int64_t strideA, int q, void *stream) {
  return plmc::assemble_impl<float>(kind, X, n, d, ell, oscale, noise, A, lda, strideA, q, stream);
}
int plmc_assemble_f64(int kind, const double *X, int n, int d, const double *ell, const double *oscale,
                      const double *noise, double *A, int64_t lda, int64_t strideA, int q, void *stream) {
  return plmc::assemble_impl<double>(kind, X, n, d, ell, oscale, noise, A, lda, strideA, q, stream);
}
int plmc_write_rhs_f32(const float *rhs, int nrhs, int n, float *A, int64_t lda, int64_t strideA, int c0,
                       int clear_cols, int q, void *stream) {
  return plmc::write_rhs_impl<float>(rhs, nrhs, n, A, lda, strideA, c0, clear_cols, q, stream);
}
int plmc_write_rhs_f64(const double *rhs, int nrhs, int n, double *A, int64_t lda, int64_t strideA, int c0,
                       int clear_cols, int q, void *stream) {
  return plmc::write_rhs_impl<double>(rhs, nrhs, n, A, lda, strideA, c0, clear_cols, q, stream);
}
int plmc_assemble_cross_f32(int kind, const float *X, int n, const float *Xs, int ns, int d, const float *ell,
                            const float *oscale, float *Out, int64_t ldo, int64_t strideO, int64_t col0,
                            int64_t n_rows, int q, void *stream) {
  return plmc::assemble_cross_impl<float>(kind, X, n, Xs, ns, d, ell, oscale, Out, ldo, strideO, col0, n_rows, q,
                                          stream);
}
int plmc_assemble_cross_f64(int kind, const double *X, int n, const double *Xs, int ns, int d, const double *ell,
                            const double *oscale, double *Out, int64_t ldo, int64_t strideO, int64_t col0,
                            int64_t n_rows, int q, void *stream) {
  return plmc::assemble_cross_impl<double>(kind, X, n, Xs, ns, d, ell, oscale, Out, ldo, strideO, col0, n_rows, q,
                                           stream);
}
}
