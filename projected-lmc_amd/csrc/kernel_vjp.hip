// kernel_vjp.hip -- vector-Jacobian product of a dense (cross-)covariance block with respect to its
// inputs and hyper-parameters:  given G = d loss / d K  with  K_i[a][b] = os_i k(|x1_a - x2_b| / ell_i),
//   gX1[i][a][k]  = sum_b G_i[a][b] dK_i[a][b] / d x1_a[k]
//   gEll[i][a][k] = sum_b G_i[a][b] dK_i[a][b] / d ell_i[k]       (row partials; the caller sums over a)
//   gOs[i][a]     = sum_b G_i[a][b] k(...)                         (row partials)
// Used by the variational path: gradients of K_ZZ and K_ZX with respect to the
// learned inducing locations Z and the lengthscales -- what torch autograd derives through gpytorch's
// kernel evaluation chain in `loss.backward()` (experiments.py:270) for VariationalMultitaskGPModel.
// One wave per row a: lanes stride over b with coalesced reads of G; fixed-order reductions, fp64
// accumulation, no atomics.  HBM-bound: reads G once.
#include "api_common.hpp"
#include "covariance.hpp"
#include "../../include/plmc.h"

namespace plmc {

template <typename T, int DCAP>
__global__ __launch_bounds__(NTHREADS) void k_kernel_vjp(int kind, const T *__restrict__ X1, int n1,
                                                          const T *__restrict__ X2, int n2, int d,
                                                          const T *__restrict__ ell, const T *__restrict__ oscale,
                                                          const T *__restrict__ G, int64_t ldg, int64_t strideG,
                                                          double *__restrict__ gX1, double *__restrict__ gEll,
                                                          double *__restrict__ gOs) {
  const int lat = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int a = blockIdx.x * 4 + wave;
  if (a >= n1) return;
  const T *el = ell + (int64_t)lat * d;
  const T os = oscale ? oscale[lat] : T(1);
  T u1[DCAP], il[DCAP];
#pragma unroll
  for (int k = 0; k < DCAP; ++k) {
    il[k] = k < d ? T(1) / el[k] : T(0);
    u1[k] = k < d ? X1[(int64_t)a * d + k] * il[k] : T(0);
  }
  double sx[DCAP], sl[DCAP], so = 0.0;
#pragma unroll
  for (int k = 0; k < DCAP; ++k) { sx[k] = 0.0; sl[k] = 0.0; }
  const T *Grow = G + (int64_t)lat * strideG + (int64_t)a * ldg;
  for (int b = lane; b < n2; b += 64) {
    const T g = Grow[b];
    T df[DCAP];
    T r2 = T(0);
#pragma unroll
    for (int k = 0; k < DCAP; ++k) {
      df[k] = k < d ? u1[k] - X2[(int64_t)b * d + k] * il[k] : T(0);
      r2 += df[k] * df[k];
    }
    T val, base;
    kern_value_base<T>(kind, r2, val, base);
    so += (double)(g * val);
    const T c = g * os * base;
#pragma unroll
    for (int k = 0; k < DCAP; ++k) {
      sl[k] += (double)(c * df[k] * df[k]);           // * 1/ell_k below
      sx[k] -= (double)(c * df[k]);                   // dK/dx1_k = -os base (x1-x2)_k / ell_k^2 = -os base du_k / ell_k
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    so += __shfl_down(so, off, 64);
#pragma unroll
    for (int k = 0; k < DCAP; ++k) {
      sx[k] += __shfl_down(sx[k], off, 64);
      sl[k] += __shfl_down(sl[k], off, 64);
    }
  }
  if (lane == 0) {
    const int64_t o = ((int64_t)lat * n1 + a) * d;
#pragma unroll
    for (int k = 0; k < DCAP; ++k)
      if (k < d) {
        gX1[o + k] = sx[k] * (double)il[k];
        gEll[o + k] = sl[k] * (double)il[k];
      }
    gOs[(int64_t)lat * n1 + a] = so;
  }
}

template <typename T>
int kernel_vjp_impl(int kind, const T *X1, int n1, const T *X2, int n2, int d, const T *ell, const T *oscale,
                    const T *G, int64_t ldg, int64_t strideG, double *gX1, double *gEll, double *gOs, int q,
                    void *stream) {
  PLMC_REQUIRE(kind >= 0 && kind <= 3, "unknown kernel kind");
  PLMC_REQUIRE(X1 && X2 && ell && G && gX1 && gEll && gOs, "null pointer");
  PLMC_REQUIRE(n1 > 0 && n2 > 0 && q > 0 && d > 0 && d <= MAX_DIM && ldg >= n2, "bad sizes");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid((n1 + 3) / 4, q), block(NTHREADS);
  ProfScope ps(PK_VJP, st, 0.0, (double)q * n1 * n2 * sizeof(T));
#define PLMC_LAUNCH_VJP(DC)                                                                                         \
  hipLaunchKernelGGL((k_kernel_vjp<T, DC>), grid, block, 0, st, kind, X1, n1, X2, n2, d, ell, oscale, G, ldg, strideG, \
                     gX1, gEll, gOs)
  if (d <= 4) PLMC_LAUNCH_VJP(4);
  else if (d <= 8) PLMC_LAUNCH_VJP(8);
  else if (d <= 16) PLMC_LAUNCH_VJP(16);
  else PLMC_LAUNCH_VJP(32);
#undef PLMC_LAUNCH_VJP
  return launch_status(__func__);
}

}  // namespace plmc

extern "C" {
int plmc_kernel_vjp_f32(int kind, const float *X1, int n1, const float *X2, int n2, int d, const float *ell,
                        const float *oscale, const float *G, int64_t ldg, int64_t strideG, double *gX1, double *gEll,
                        double *gOs, int q, void *stream) {
  return plmc::kernel_vjp_impl<float>(kind, X1, n1, X2, n2, d, ell, oscale, G, ldg, strideG, gX1, gEll, gOs, q, stream);
}
int plmc_kernel_vjp_f64(int kind, const double *X1, int n1, const double *X2, int n2, int d, const double *ell,
                        const double *oscale, const double *G, int64_t ldg, int64_t strideG, double *gX1, double *gEll,
                        double *gOs, int q, void *stream) {
  return plmc::kernel_vjp_impl<double>(kind, X1, n1, X2, n2, d, ell, oscale, G, ldg, strideG, gX1, gEll, gOs, q,
                                       stream);
}
}
