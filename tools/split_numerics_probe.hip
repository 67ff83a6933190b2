// split_numerics_probe.hip -- which split of fp32 operands over the 16-bit matrix cores is as accurate as the fp32 MFMA
// chain?  
//
// C = A^T B for K-major fp32 operands A[K][128], B[K][128] (the engine's TN form), computed by one wave per 16 x 16
// output tile straight from global memory (no LDS: this probe is about ROUNDING, not speed), in these modes:
//   f32      v_mfma_f32_16x16x4_f32 chain (what the fp32 engine does)
//   b{6,8,9}_L   x = hi + mid + lo in bf16 (exact residuals), 6 / 8 / 9 plane products on v_mfma_f32_16x16x32_bf16, the
//            products spread over L accumulator LEVELS by magnitude: L = 1: one accumulator (round 2's opt-in);
//            L = 2: hi.hi | everything else; L = 3: hi.hi | hi.mid + mid.hi | the 2^-16 terms and below.  Levels are
//            summed once at the end (small first).
//   h6_3     x = h0 + 2^-11 h1 + 2^-22 h2 in fp16 (residuals scaled by 2^11 per level so that they stay normal),
//            6 products on v_mfma_f32_16x16x32_f16 into 3 levels, combined with the scales at the end.
//   hh       hi.hi only, compared with the fp64 product of the SAME bf16 values: the accumulation error of the bf16 MFMA
//            alone (round-to-nearest would give ~sqrt(K/32) 2^-25 |sum|; truncation ~ (K/32) 2^-24 |sum|).
// Errors are against an fp64 host product: max |err| / max |C| and rms err / rms C, for K = 1024 (a trailing update of
// the sweep) and K = 8192 (the long products of K^-1 = W^T W), normal and wide-range (log-normal scaled) data.
// Round 4: the fp16 modes scale each operand FAMILY as the product does (csrc/bf3_engine.hpp b3_scale_for: the power of two that
// puts the largest magnitude of A resp. B at 2^13, undone exactly at the end) -- round 3's probe fed them unscaled data, whose
// wide-range case overflowed fp16 (the nan rows of profiles/r03_split_numerics.txt) -- and two COMPONENTWISE figures are reported:
//   cw  = max_ij |err_ij| / sum_k |a_ki b_kj|       (what the a-priori bound below bounds), and
//   rel = max |err_ij| / |C_ij| over the entries with |C_ij| >= 2^-20 max |C|.
// A-priori (K-term products, u = 2^-24):  fp32 chain  cw <= K u;   fp16x2 / 3 products / 2 levels
//   cw <= 2 * 2^-23 (each operand rounded to 22 bits) + 2^-22 (the dropped h1.h1 product) + (K / 32 + 2) u (roundings of the level-0
//   sum, one per MFMA) -- for operands within 2^-27 of their family's bound; smaller ones lose bits (h0 goes subnormal).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/split_numerics_probe tools/split_numerics_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <random>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

enum Mode { M_F32 = 0, M_B6_1, M_B6_2, M_B6_3, M_B8_1, M_B8_2, M_B8_3, M_B9_2, M_B9_3, M_H6_3, M_H6_2, M_HH, M_B3_1, M_H3_2, M_H4_2, M_COUNT };
static const char *mode_name[M_COUNT] = {"f32 mfma chain", "bf16x3 6 prod 1 lvl", "bf16x3 6 prod 2 lvl", "bf16x3 6 prod 3 lvl", "bf16x3 8 prod 1 lvl",
                                         "bf16x3 8 prod 2 lvl", "bf16x3 8 prod 3 lvl", "bf16x3 9 prod 2 lvl", "bf16x3 9 prod 3 lvl",
                                         "fp16x3 6 prod 3 lvl", "fp16x3 6 prod 2 lvl", "bf16 hi.hi only   ", "bf16x2 3 prod 1 lvl",
                                         "fp16x2 3 prod 2 lvl", "fp16x2 4 prod 2 lvl"};

__device__ __forceinline__ void split_bf(const float (&x)[8], bf16x8 &h, bf16x8 &m, bf16x8 &l) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const __bf16 a = (__bf16)x[i];
    const float r1 = x[i] - (float)a;
    const __bf16 b = (__bf16)r1;
    const float r2 = r1 - (float)b;
    h[i] = a; m[i] = b; l[i] = (__bf16)r2;
  }
}
// fp16 levels: h0 = fp16(x), h1 = fp16((x - h0) 2^11), h2 = fp16(((x - h0) 2^11 - h1) 2^11)   (x pre-scaled into range)
__device__ __forceinline__ void split_h(const float (&x)[8], f16x8 &h, f16x8 &m, f16x8 &l) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const _Float16 a = (_Float16)x[i];
    const float r1 = (x[i] - (float)a) * 2048.0f;
    const _Float16 b = (_Float16)r1;
    const float r2 = (r1 - (float)b) * 2048.0f;
    h[i] = a; m[i] = b; l[i] = (_Float16)r2;
  }
}

// one wave per 16 x 16 tile; grid (8, 8); C[128][128]
template <int MODE>
__global__ __launch_bounds__(64) void k_probe(const float *__restrict__ A, const float *__restrict__ B, int K, float *__restrict__ C, float *__restrict__ Chh,
                                              float sA, float sB) {
  const int lane = threadIdx.x, ti = blockIdx.y, tj = blockIdx.x;
  const int fr = lane & 15, fg = lane >> 4;
  f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
  if constexpr (MODE == M_F32) {
    for (int k = 0; k < K; k += 4) {
      const float a = A[(int64_t)(k + fg) * 128 + ti * 16 + fr], b = B[(int64_t)(k + fg) * 128 + tj * 16 + fr];
      acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc0, 0, 0, 0);
    }
  } else {
    for (int k = 0; k < K; k += 32) {
      float xa[8], xb[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xa[j] = A[(int64_t)(k + 8 * fg + j) * 128 + ti * 16 + fr];
        xb[j] = B[(int64_t)(k + 8 * fg + j) * 128 + tj * 16 + fr];
      }
      if constexpr (MODE == M_H6_3 || MODE == M_H6_2 || MODE == M_H3_2 || MODE == M_H4_2) {
        f16x8 ah, am, al, bh, bm, bl;
#pragma unroll
        for (int j = 0; j < 8; ++j) { xa[j] *= sA; xb[j] *= sB; }      // the family scales of the product (powers of two: exact)
        split_h(xa, ah, am, al);
        split_h(xb, bh, bm, bl);
#define MMH(x, y, c) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, c, 0, 0, 0)
        if constexpr (MODE == M_H3_2 || MODE == M_H4_2) {     // two fp16 planes (22 bits): h0.h0 | (h1.h0 + h0.h1 [+ h1.h1 2^-11])
          if constexpr (MODE == M_H4_2) { f32x4 t = {0, 0, 0, 0}; MMH(am, bm, t); acc1 += t * (1.0f / 2048.0f); }
          MMH(am, bh, acc1); MMH(ah, bm, acc1);
          MMH(ah, bh, acc0);
        } else if constexpr (MODE == M_H6_3) {
          MMH(al, bh, acc2); MMH(ah, bl, acc2); MMH(am, bm, acc2);
          MMH(am, bh, acc1); MMH(ah, bm, acc1);
          MMH(ah, bh, acc0);
        } else {   // level 1 and level 2 products share one accumulator: the level-2 operand pre-scaled by 2^-11 is not
                   // representable, so this mode adds the level-2 products (already 2^11 too large relative to level 1)
                   // into acc2 and folds it into acc1 every step -- i.e. two MFMA levels + one VALU add per step
          f32x4 t = {0, 0, 0, 0};
          MMH(al, bh, t); MMH(ah, bl, t); MMH(am, bm, t);
          MMH(am, bh, acc1); MMH(ah, bm, acc1);
          acc1 += t * (1.0f / 2048.0f);
          MMH(ah, bh, acc0);
        }
#undef MMH
      } else {
        bf16x8 ah, am, al, bh, bm, bl;
        split_bf(xa, ah, am, al);
        split_bf(xb, bh, bm, bl);
#define MMB(x, y, c) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, c, 0, 0, 0)
        constexpr int NPROD = (MODE == M_B6_1 || MODE == M_B6_2 || MODE == M_B6_3) ? 6 : (MODE == M_B8_1 || MODE == M_B8_2 || MODE == M_B8_3) ? 8 : (MODE == M_B9_2 || MODE == M_B9_3) ? 9 : (MODE == M_B3_1 ? 3 : 1);
        constexpr int LV = (MODE == M_B6_1 || MODE == M_B8_1 || MODE == M_HH || MODE == M_B3_1) ? 1 : (MODE == M_B6_2 || MODE == M_B8_2 || MODE == M_B9_2) ? 2 : 3;
        // level of a product of planes (i, j) is i + j; LV = 1: everything into acc0; 2: level 0 -> acc0, rest -> acc1;
        // 3: level 0 -> acc0, level 1 -> acc1, level >= 2 -> acc2.  Small terms first.
        f32x4 &L0 = acc0;
        f32x4 &L1 = LV >= 2 ? acc1 : acc0;
        f32x4 &L2 = LV >= 3 ? acc2 : L1;
        if (NPROD >= 9) MMB(al, bl, L2);
        if (NPROD >= 8) { MMB(am, bl, L2); MMB(al, bm, L2); }
        if (NPROD >= 6) { MMB(al, bh, L2); MMB(ah, bl, L2); MMB(am, bm, L2); }
        if (NPROD >= 3) { MMB(am, bh, L1); MMB(ah, bm, L1); }
        MMB(ah, bh, L0);
#undef MMB
      }
    }
  }
  f32x4 r;
  if constexpr (MODE == M_H6_3) r = (acc0 + (acc1 + acc2 * (1.0f / 2048.0f)) * (1.0f / 2048.0f)) * (1.0f / (sA * sB));
  else if constexpr (MODE == M_H6_2 || MODE == M_H3_2 || MODE == M_H4_2) r = (acc0 + acc1 * (1.0f / 2048.0f)) * (1.0f / (sA * sB));
  else r = acc0 + (acc1 + acc2);
  // C/D layout: col = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
  for (int q = 0; q < 4; ++q) C[(int64_t)(ti * 16 + fg * 4 + q) * 128 + tj * 16 + fr] = r[q];
  (void)Chh;
}

static float bf16_round(float x) {   // round to nearest even to bf16, as the device cast does
  uint32_t u;
  memcpy(&u, &x, 4);
  u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000u;
  float y;
  memcpy(&y, &u, 4);
  return y;
}

template <int MODE> static void launch(const float *A, const float *B, int K, float *C, float sA, float sB) {
  hipLaunchKernelGGL((k_probe<MODE>), dim3(8, 8), dim3(64), 0, 0, A, B, K, C, (float *)nullptr, sA, sB);
}
static float scale_for(float bound) {       // csrc/bf3_engine.hpp b3_scale_for
  int e;
  (void)frexpf(bound, &e);
  return ldexpf(1.0f, 13 - e);
}

int main() {
  const int Ks[2] = {1024, 8192};
  for (int dist = 0; dist < 4; ++dist)
    for (int ki = 0; ki < 2; ++ki) {
      const int K = Ks[ki];
      if (dist == 3 && ki == 1) continue;
      std::mt19937_64 rng(1234 + dist * 7 + ki);
      std::normal_distribution<double> nd(0.0, 1.0);
      std::vector<float> A((size_t)K * 128), B((size_t)K * 128);
      if (dist == 3) {
        // The group panel of the sweep, with its cancellation: A = V = U^-1 (upper, K-major) of a smooth kernel matrix
        // (Matern-5/2, 1024 points in 8 dimensions, lengthscale 1, noise 1e-4), B = U^T P for a random P, so that the
        // exact product V^T B = P is O(1) while the summands are not.  Reference: fp64 product of the fp32-rounded inputs.
        const int n = K;
        std::vector<double> X((size_t)n * 8), Km((size_t)n * n), U((size_t)n * n, 0.0), V((size_t)n * n, 0.0);
        std::uniform_real_distribution<double> ud(-1.0, 1.0);
        for (auto &v : X) v = ud(rng);
        for (int i = 0; i < n; ++i)
          for (int j = 0; j < n; ++j) {
            double r2 = 0; for (int k = 0; k < 8; ++k) { double dd = X[(size_t)i * 8 + k] - X[(size_t)j * 8 + k]; r2 += dd * dd; }
            const double r = sqrt(5.0 * r2);
            Km[(size_t)i * n + j] = (1.0 + r + r * r / 3.0) * exp(-r) + (i == j ? 1e-4 : 0.0);
          }
        for (int j = 0; j < n; ++j) {                       // K = U^T U, U upper
          for (int i = 0; i <= j; ++i) {
            double sum = Km[(size_t)i * n + j];
            for (int k = 0; k < i; ++k) sum -= U[(size_t)k * n + i] * U[(size_t)k * n + j];
            U[(size_t)i * n + j] = i == j ? sqrt(sum) : sum / U[(size_t)i * n + i];
          }
        }
        for (int c = 0; c < 128; ++c) {                      // V[:, c'] for the LAST 128 columns only (the heaviest rows of the panel)
          const int col = n - 128 + c;
          std::vector<double> x(n, 0.0);
          for (int i = col; i >= 0; --i) {                   // solve U x = e_col
            double sum = (i == col) ? 1.0 : 0.0;
            for (int k = i + 1; k <= col; ++k) sum -= U[(size_t)i * n + k] * x[k];
            x[i] = sum / U[(size_t)i * n + i];
          }
          for (int k = 0; k < n; ++k) A[(size_t)k * 128 + c] = (float)x[k];
        }
        std::vector<double> Pt((size_t)n * 128);
        for (auto &v : Pt) v = nd(rng);
        for (int k = 0; k < n; ++k)                          // B = U^T P: B[k][j] = sum_{l <= k} U[l][k] P[l][j]
          for (int j = 0; j < 128; ++j) {
            double sum = 0; for (int l = 0; l <= k; ++l) sum += U[(size_t)l * n + k] * Pt[(size_t)l * 128 + j];
            B[(size_t)k * 128 + j] = (float)sum;
          }
        double amax = 0; for (auto v : A) amax = fmax(amax, fabs((double)v));
        printf("   (largest |V| entry %.3e)\n", amax);
      }
      for (size_t i = 0; dist < 3 && i < A.size(); ++i) {
        double a = nd(rng), b = nd(rng);
        if (dist == 1) { a *= exp(3.0 * nd(rng)); b *= exp(3.0 * nd(rng)); }      // wide dynamic range
        if (dist == 2) { a = fabs(a) + 0.5; b = fabs(b) + 0.5; }                   // one sign: sums grow like K, no cancellation
        A[i] = (float)a; B[i] = (float)b;
      }
      std::vector<double> ref((size_t)128 * 128, 0.0), refhh((size_t)128 * 128, 0.0), refabs((size_t)128 * 128, 0.0);
      for (int k = 0; k < K; ++k)
        for (int i = 0; i < 128; ++i) {
          const double a = A[(size_t)k * 128 + i], ah = bf16_round(A[(size_t)k * 128 + i]);
          for (int j = 0; j < 128; ++j) {
            ref[(size_t)i * 128 + j] += a * (double)B[(size_t)k * 128 + j];
            refabs[(size_t)i * 128 + j] += fabs(a * (double)B[(size_t)k * 128 + j]);
            refhh[(size_t)i * 128 + j] += ah * (double)bf16_round(B[(size_t)k * 128 + j]);
          }
        }
      float amaxA = 0, amaxB = 0;
      for (auto v : A) amaxA = fmaxf(amaxA, fabsf(v));
      for (auto v : B) amaxB = fmaxf(amaxB, fabsf(v));
      const float sA = scale_for(amaxA), sB = scale_for(amaxB);
      {   // how much of the data sits below 2^-27 of its family's bound (h0 subnormal: bits are lost)
        size_t lowA = 0, lowB = 0;
        for (auto v : A) lowA += fabsf(v) * sA < ldexpf(1.0f, -14) && v != 0.0f;
        for (auto v : B) lowB += fabsf(v) * sB < ldexpf(1.0f, -14) && v != 0.0f;
        printf("   family bounds: max|A| %.3e (scale 2^%d), max|B| %.3e (scale 2^%d); entries below 2^-27 of the bound: %.3f %% / %.3f %%\n", amaxA,
               (int)log2f(sA), amaxB, (int)log2f(sB), 100.0 * lowA / A.size(), 100.0 * lowB / B.size());
      }
      float *dA, *dB, *dC;
      CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dC, 128 * 128 * 4));
      CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
      CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
      printf("== data %s, K = %d\n", dist == 0 ? "N(0,1)" : (dist == 1 ? "N(0,1) x exp(3 N(0,1)) (wide range)" : (dist == 2 ? "|N(0,1)| + 0.5 (one sign)" : "group panel V^T (U^T P), Matern-5/2 + 1e-4 I (cancellation)")), K);
      std::vector<float> C((size_t)128 * 128);
      double e_f32_max = 0, e_f32_rms = 0;
      for (int mode = 0; mode < M_COUNT; ++mode) {
        switch (mode) {
#define CASE(M) case M: launch<M>(dA, dB, K, dC, sA, sB); break;
          CASE(M_F32) CASE(M_B6_1) CASE(M_B6_2) CASE(M_B6_3) CASE(M_B8_1) CASE(M_B8_2) CASE(M_B8_3) CASE(M_B9_2) CASE(M_B9_3) CASE(M_H6_3) CASE(M_H6_2) CASE(M_HH) CASE(M_B3_1) CASE(M_H3_2) CASE(M_H4_2)
#undef CASE
        }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
        const std::vector<double> &R = mode == M_HH ? refhh : ref;
        double emax = 0, cmax = 0, e2 = 0, c2 = 0, bias = 0, cw = 0, rel = 0;
        for (size_t i = 0; i < C.size(); ++i) cmax = fmax(cmax, fabs(R[i]));
        for (size_t i = 0; i < C.size(); ++i) {
          const double e = (double)C[i] - R[i];
          emax = fmax(emax, fabs(e));
          e2 += e * e; c2 += R[i] * R[i];
          bias += e * (R[i] >= 0 ? 1.0 : -1.0);      // > 0: magnitudes too large; < 0: truncation toward zero
          if (mode != M_HH) cw = fmax(cw, fabs(e) / refabs[i]);
          if (fabs(R[i]) >= ldexp(cmax, -20)) rel = fmax(rel, fabs(e) / fabs(R[i]));
        }
        const double rmax = emax / cmax, rrms = sqrt(e2 / c2);
        if (mode == M_F32) { e_f32_max = rmax; e_f32_rms = rrms; }
        printf("  %-22s max|err|/max|C| %.3e (%.2fx f32)   rms err/rms C %.3e (%.2fx f32)   signed mean err/rms C %+.2e   cw %.3e   rel(>=2^-20 max) %.3e\n",
               mode_name[mode], rmax, rmax / e_f32_max, rrms, rrms / e_f32_rms, bias / C.size() / sqrt(c2 / C.size()), cw, rel);
      }
      CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
    }
  return 0;
}
