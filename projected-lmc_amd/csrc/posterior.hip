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

// grid (ceil((1 + ns) / (16 EPV)), q), 256 threads = 16 column groups of EPV = 16 bytes x 16 row groups; every thread walks its
// rows 8 at a time (8 x 16-byte loads in flight per thread; a workgroup reads 256-byte row pieces of 128 rows at once).  Few
// columns per workgroup: with one latent per rank (the sharded runs) the column groups are all the parallelism there is.
// The augmented block starts at the 16-byte-aligned column n_pad: its column 0 is z, column 1 + s is v(s).
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_posterior_moments(const T *__restrict__ A, int64_t n_pad, int64_t lda, int64_t strideA, int ns,
                                                                T *__restrict__ mean, T *__restrict__ vsq) {
  using vec_t = typename Traits<T>::vec_t;
  constexpr int EPV = Traits<T>::EPV, UNR = 8, CG = 16, RG = NTHREADS / CG;
  __shared__ double red[2][RG][CG][EPV];
  const int lat = blockIdx.y, cg = threadIdx.x % CG, rg = threadIdx.x / CG;
  const int c0 = (blockIdx.x * CG + cg) * EPV;                  // first augmented column of this thread
  const T *Z = A + (int64_t)lat * strideA + n_pad;              // column 0 of the augmented block
  double m[EPV], v2[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) m[e] = v2[e] = 0.0;
  if (c0 < 1 + ns) {
    for (int64_t r0 = rg; r0 < n_pad; r0 += RG * UNR) {
      vec_t v[UNR];
      T z[UNR];
      // unconditional loads from a clamped row (a predicated load compiles to a branch with a full wait behind it: the UNR loads
      // of a thread then go out one round trip at a time -- k_wt_matvec, potrf.hip); rows beyond n_pad contribute exact zeros
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int64_t r = r0 + RG * u;
        const int64_t rc = r < n_pad ? r : n_pad - 1;
        v[u] = *reinterpret_cast<const vec_t *>(Z + rc * lda + c0);
        z[u] = Z[rc * lda];
      }
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const bool ok = r0 + RG * u < n_pad;
#pragma unroll
        for (int e = 0; e < EPV; ++e) {
          const double x = ok ? (double)v[u][e] : 0.0;
          m[e] += x * (ok ? (double)z[u] : 0.0);
          v2[e] += x * x;
        }
      }
    }
  }
#pragma unroll
  for (int e = 0; e < EPV; ++e) { red[0][rg][cg][e] = m[e]; red[1][rg][cg][e] = v2[e]; }
  __syncthreads();
  if (rg == 0) {
#pragma unroll
    for (int e = 0; e < EPV; ++e) {
      const int s = c0 + e - 1;                                  // test point of augmented column c0 + e
      if (s >= 0 && s < ns) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int g = 0; g < RG; ++g) { a += red[0][g][cg][e]; b += red[1][g][cg][e]; }
        mean[(int64_t)lat * ns + s] = (T)a;
        vsq[(int64_t)lat * ns + s] = (T)b;
      }
    }
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
  ProfScope ps(PK_POST, (hipStream_t)stream, 0.0, bytes);
  constexpr int CPW = 16 * Traits<T>::EPV;                      // augmented columns per workgroup
  hipLaunchKernelGGL(k_posterior_moments<T>, dim3((1 + ns + CPW - 1) / CPW, q), dim3(NTHREADS), 0, (hipStream_t)stream, A, n_pad, lda, strideA, ns, mean, vsq);
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
