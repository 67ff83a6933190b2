// posterior.hip -- the two reductions of the eval-mode posterior that follow the augmented sweep.
//
// Replaces, for ProjectedGPModel.__call__ in eval mode (projected_lmc.py:1133-1155) and ExactGPModel's
// DefaultPredictionStrategy behind it:
//   k_posterior_moments : mean_i(s) = v_i(s)^T z_i and |v_i(s)|^2 from the augmented columns [ z | V ] = U^-T [ y | K*^T ]
//                         of the factor buffer (the latent posterior is mean, k** - |v|^2)          HBM-bound, one pass
//   k_mix_posterior     : task-space moments mean(s, t) = sum_i mean_i(s) H_it, var(s, t) = sum_i var_i(s) H_it^2 + eps
//                         (projected_lmc.py:1144, :1152; the sharded form passes its local latents and adds eps after
//                         the all-reduce)                                                            HBM-bound
// Sums are carried in fp64 and rounded once.
#include "api_common.hpp"
#include "../../include/plmc.h"

namespace plmc {

// grid (ceil(ns / 64), q), 256 threads = 64 columns x 4 row groups
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_posterior_moments(const T *__restrict__ A, int64_t n_pad, int64_t lda, int64_t strideA, int ns,
                                                                T *__restrict__ mean, T *__restrict__ vsq) {
  __shared__ double red[2][4][64];
  const int lat = blockIdx.y, c = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int s = blockIdx.x * 64 + c;
  const T *Z = A + (int64_t)lat * strideA + n_pad;          // column n_pad: z; columns n_pad + 1 + s: v(s)
  double m = 0.0, v2 = 0.0;
  if (s < ns) {
    for (int64_t r = rg; r < n_pad; r += 4) {
      const double z = (double)Z[r * lda], v = (double)Z[r * lda + 1 + s];
      m += v * z;
      v2 += v * v;
    }
  }
  red[0][rg][c] = m;
  red[1][rg][c] = v2;
  __syncthreads();
  if (rg == 0 && s < ns) {
    mean[(int64_t)lat * ns + s] = (T)(red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
    vsq[(int64_t)lat * ns + s] = (T)(red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
  }
}

// one thread per (s, t); latent moments (q, ns), mixing matrix Ht (q, p); outputs (ns, p)
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_mix_posterior(const T *__restrict__ mean_lat, const T *__restrict__ var_lat, const T *__restrict__ Ht,
                                                            int q, int ns, int p, T eps, T *__restrict__ mean, T *__restrict__ var) {
  const int64_t idx = (int64_t)blockIdx.x * NTHREADS + threadIdx.x;
  if (idx >= (int64_t)ns * p) return;
  const int s = (int)(idx / p), t = (int)(idx % p);
  double m = 0.0, v = 0.0;
  for (int i = 0; i < q; ++i) {
    const double h = (double)Ht[(int64_t)i * p + t];
    m += (double)mean_lat[(int64_t)i * ns + s] * h;
    v += (double)var_lat[(int64_t)i * ns + s] * h * h;
  }
  mean[idx] = (T)m;
  var[idx] = (T)(v + (double)eps);
}

template <typename T>
int posterior_moments_impl(const T *A, int64_t n_pad, int64_t lda, int64_t strideA, int ns, T *mean, T *vsq, int q, void *stream) {
  PLMC_REQUIRE(A && mean && vsq, "null pointer");
  PLMC_REQUIRE(n_pad > 0 && n_pad % NB == 0 && ns > 0 && q > 0 && lda >= n_pad + 1 + ns, "bad sizes (lda must hold the 1 + ns augmented columns)");
  const double bytes = (double)q * n_pad * (1.0 + ns) * sizeof(T);
  ProfScope ps(PK_EXTRACT, (hipStream_t)stream, 0.0, bytes);
  hipLaunchKernelGGL(k_posterior_moments<T>, dim3((ns + 63) / 64, q), dim3(NTHREADS), 0, (hipStream_t)stream, A, n_pad, lda, strideA, ns, mean, vsq);
  return launch_status(__func__);
}

template <typename T>
int mix_posterior_impl(const T *mean_lat, const T *var_lat, const T *Ht, int q, int ns, int p, double eps, T *mean, T *var, void *stream) {
  PLMC_REQUIRE(mean_lat && var_lat && Ht && mean && var, "null pointer");
  PLMC_REQUIRE(q > 0 && ns > 0 && p > 0, "bad sizes");
  const int64_t total = (int64_t)ns * p;
  hipLaunchKernelGGL(k_mix_posterior<T>, dim3((unsigned)((total + NTHREADS - 1) / NTHREADS)), dim3(NTHREADS), 0, (hipStream_t)stream, mean_lat, var_lat,
                     Ht, q, ns, p, (T)eps, mean, var);
  return launch_status(__func__);
}

}  // namespace plmc

extern "C" {
int plmc_posterior_moments_f32(const float *A, int64_t n_pad, int64_t lda, int64_t strideA, int ns, float *mean, float *vsq, int q, void *stream) {
  return plmc::posterior_moments_impl<float>(A, n_pad, lda, strideA, ns, mean, vsq, q, stream);
}
int plmc_posterior_moments_f64(const double *A, int64_t n_pad, int64_t lda, int64_t strideA, int ns, double *mean, double *vsq, int q, void *stream) {
  return plmc::posterior_moments_impl<double>(A, n_pad, lda, strideA, ns, mean, vsq, q, stream);
}
int plmc_mix_posterior_f32(const float *mean_lat, const float *var_lat, const float *Ht, int q, int ns, int p, double eps, float *mean, float *var,
                           void *stream) {
  return plmc::mix_posterior_impl<float>(mean_lat, var_lat, Ht, q, ns, p, eps, mean, var, stream);
}
int plmc_mix_posterior_f64(const double *mean_lat, const double *var_lat, const double *Ht, int q, int ns, int p, double eps, double *mean,
                           double *var, void *stream) {
  return plmc::mix_posterior_impl<double>(mean_lat, var_lat, Ht, q, ns, p, eps, mean, var, stream);
}
}
