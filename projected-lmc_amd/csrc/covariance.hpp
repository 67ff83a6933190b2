// covariance.hpp -- stationary ARD kernel profiles shared by assembly and gradient kernels.
// Restates the kernels handle_covar_ builds (projected_lmc.py:151-167): gpytorch RBFKernel and
// MaternKernel(nu in {1/2,3/2,5/2}) on scaled inputs u = x / ell  [gpytorch-knowledge].
#pragma once
#include <hip/hip_runtime.h>

namespace plmc {

enum { K_RBF = 0, K_MATERN12 = 1, K_MATERN32 = 2, K_MATERN52 = 3, K_SPLINE = 4 };

// One factor of the reference's SplineKernel (projected_lmc.py:26-36): k(x, x') = prod_k [1 + m M + m^2 (M - m / 3) / 2],
// m = min(x_k, x'_k), M = max(x_k, x'_k).  Not a function of the distance and without a lengthscale: the kernels that
// take a `kind` evaluate it from the staged inputs themselves (callers pass ell = 1) instead of through r2.
template <typename T> __device__ __forceinline__ T spline_factor(T a, T b) {
  const T mn = a < b ? a : b, mx = a < b ? b : a;
  return T(1) + mn * mx + T(0.5) * mn * mn * (mx - mn * T(1.0 / 3.0));
}

__device__ __forceinline__ float  dexp(float x) { return expf(x); }
__device__ __forceinline__ double dexp(double x) { return exp(x); }
__device__ __forceinline__ float  dsqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double dsqrt(double x) { return sqrt(x); }

// k(r2) for unit output scale, r2 = |u - u'|^2.
template <typename T> __device__ __forceinline__ T kern_value(int kind, T r2) {
  if (kind == K_RBF) return dexp(T(-0.5) * r2);
  T r = dsqrt(r2 > T(0) ? r2 : T(0));
  if (kind == K_MATERN12) return dexp(-r);
  if (kind == K_MATERN32) { T s = T(1.7320508075688772) * r; return (T(1) + s) * dexp(-s); }
  T s = T(2.23606797749979) * r;
  return (T(1) + s + T(5.0 / 3.0) * r2) * dexp(-s);
}

// value and "base" such that  d k / d ell_k = base * (u_k - u'_k)^2 / ell_k   (unit output scale).
template <typename T> __device__ __forceinline__ void kern_value_base(int kind, T r2, T &val, T &base) {
  if (kind == K_RBF) { val = dexp(T(-0.5) * r2); base = val; return; }
  T r = dsqrt(r2 > T(0) ? r2 : T(0));
  if (kind == K_MATERN12) {
    val = dexp(-r);
    base = r > T(1e-15) ? val / r : T(0);
    return;
  }
  if (kind == K_MATERN32) {
    T s = T(1.7320508075688772) * r, e = dexp(-s);
    val = (T(1) + s) * e; base = T(3) * e; return;
  }
  T s = T(2.23606797749979) * r, e = dexp(-s);
  val = (T(1) + s + T(5.0 / 3.0) * r2) * e;
  base = T(5.0 / 3.0) * (T(1) + s) * e;
}

// Gradient-epilogue variant: for fp32 the transcendental pair is taken from the hardware units
// (v_sqrt_f32 / v_exp_f32, ~1-2 ulp) instead of the ~35-instruction IEEE expansions; the values only
// weight a reduction whose fp32 tolerance is 1e-3, the covariance ASSEMBLY keeps the accurate forms.
__device__ __forceinline__ void kern_value_base_fast(int kind, float r2, float &val, float &base) {
  if (kind == K_RBF) { val = __expf(-0.5f * r2); base = val; return; }
  const float r = __builtin_amdgcn_sqrtf(r2 > 0.f ? r2 : 0.f);
  if (kind == K_MATERN12) { val = __expf(-r); base = r > 1e-15f ? val / r : 0.f; return; }
  if (kind == K_MATERN32) { const float s = 1.7320508075688772f * r, e = __expf(-s); val = (1.f + s) * e; base = 3.f * e; return; }
  const float s = 2.23606797749979f * r, e = __expf(-s);
  val = (1.f + s + (5.0f / 3.0f) * r2) * e;
  base = (5.0f / 3.0f) * (1.f + s) * e;
}
__device__ __forceinline__ void kern_value_base_fast(int kind, double r2, double &val, double &base) {
  kern_value_base<double>(kind, r2, val, base);
}


// Two elements at a time (fp32: the polynomial parts compile to packed v_pk_* instructions; only the square root and
// the exponential stay scalar).  Same values as kern_value_base_fast.
__device__ __forceinline__ float  fast_exp(float x) { return __expf(x); }
__device__ __forceinline__ double fast_exp(double x) { return exp(x); }
__device__ __forceinline__ float  fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ double fast_sqrt(double x) { return sqrt(x); }
template <typename T> using Pair = T __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ void kern_value_base_pair(int kind, Pair<T> r2, Pair<T> &val, Pair<T> &base) {
  if (kind == K_RBF) {
    const Pair<T> a = T(-0.5) * r2;
    val = Pair<T>{fast_exp(a.x), fast_exp(a.y)};
    base = val;
    return;
  }
  const Pair<T> r = {fast_sqrt(r2.x > T(0) ? r2.x : T(0)), fast_sqrt(r2.y > T(0) ? r2.y : T(0))};
  if (kind == K_MATERN12) {
    val = Pair<T>{fast_exp(-r.x), fast_exp(-r.y)};
    base = Pair<T>{r.x > T(1e-15) ? val.x / r.x : T(0), r.y > T(1e-15) ? val.y / r.y : T(0)};
    return;
  }
  if (kind == K_MATERN32) {
    const Pair<T> s = T(1.7320508075688772) * r, e = {fast_exp(-s.x), fast_exp(-s.y)};
    val = (T(1) + s) * e;
    base = T(3) * e;
    return;
  }
  const Pair<T> s = T(2.23606797749979) * r, e = {fast_exp(-s.x), fast_exp(-s.y)};
  const Pair<T> ope = (T(1) + s) * e;
  val = ope + T(5.0 / 3.0) * r2 * e;
  base = T(5.0 / 3.0) * ope;
}

// exp / sqrt of the packed assembly path.  fp32: the hardware exp2 on x log2(e) with the rounding error of that
// product carried along (t + e = x log2(e) to ~2^-45; exp2(t) is good to 1 ulp), so the result is within ~1.5 ulp
// for every argument -- the plain __expf loses |x| 2^-24 -- at 6 instructions instead of the ~20 of expf; v_sqrt_f32
// is 1 ulp.  fp64 keeps the library forms.
__device__ __forceinline__ float asm_exp(float x) {
  const float L = 1.44269504088896340736f, Ll = 1.9259629911266175e-8f;   // log2(e) = L + Ll
  const float t = x * L;
  const float e = __builtin_fmaf(x, L, -t) + x * Ll;
  const float r = __builtin_amdgcn_exp2f(t);
  return __builtin_fmaf(r, e * 0.6931471805599453f, r);
}
__device__ __forceinline__ double asm_exp(double x) { return exp(x); }
__device__ __forceinline__ float asm_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ double asm_sqrt(double x) { return sqrt(x); }
#define dexp asm_exp
#define dsqrt asm_sqrt
// Two covariance values at a time (assembly kernels).
template <typename T> __device__ __forceinline__ Pair<T> kern_value_pair(int kind, Pair<T> r2) {
  if (kind == K_RBF) {
    const Pair<T> a = T(-0.5) * r2;
    return Pair<T>{dexp(a.x), dexp(a.y)};
  }
  const Pair<T> r = {dsqrt(r2.x > T(0) ? r2.x : T(0)), dsqrt(r2.y > T(0) ? r2.y : T(0))};
  if (kind == K_MATERN12) return Pair<T>{dexp(-r.x), dexp(-r.y)};
  if (kind == K_MATERN32) {
    const Pair<T> s = T(1.7320508075688772) * r;
    return (T(1) + s) * Pair<T>{dexp(-s.x), dexp(-s.y)};
  }
  const Pair<T> s = T(2.23606797749979) * r;
  return (T(1) + s + T(5.0 / 3.0) * r2) * Pair<T>{dexp(-s.x), dexp(-s.y)};
}
#undef dexp
#undef dsqrt

}  // namespace plmc
