// mfma32_probe.hip -- operand and accumulator layout of v_mfma_f32_32x32x16_{f16,bf16} on gfx950, checked numerically: one wave,
// C[m][n] = sum_k A[k][m] B[k][n] with K = 16, M = N = 32, against the host product.  (Round 4: the bare rate of this shape is 2.4 PF
// against 1.6 PF for 16x16x32 -- tools/dev/mfma_shapes.py -- so the split engine's main loop moves to it; bf3_engine.hpp.)
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/mfma32_probe tools/mfma32_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float *A, const float *B, float *C) {
  const int l = threadIdx.x;
  f16x8 a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (_Float16)A[(8 * (l / 32) + j) * 32 + (l % 32)];      // assumed: lane = (k group of 8, m), 8 consecutive k per lane
    b[j] = (_Float16)B[(8 * (l / 32) + j) * 32 + (l % 32)];
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int j = 0; j < 16; ++j) {
    const int row = 8 * (j / 4) + 4 * (l / 32) + (j % 4), col = l % 32;       // assumed accumulator layout
    C[row * 32 + col] = c[j];
  }
}
// ---- bare issue rate of the two shapes (no memory traffic), 4 waves per SIMD, operands either a few small values or random
// bit patterns of normal magnitude (the power drawn, hence the clock held, depends on the data)
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ inline unsigned rnd(unsigned x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }
template <bool BIG, bool RANDOM> __global__ __launch_bounds__(256) void k_rate(float *sink, int iters) {
  bf16x8 a[4], b[4];
  unsigned st = 0x9e3779b9u * (threadIdx.x + 1) + blockIdx.x;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) {
      st = rnd(st);
      const float ra = RANDOM ? ((float)(st & 0xffff) / 32768.0f - 1.0f) : (1.0f + 0.01f * (float)((threadIdx.x + j) & 7) + (float)i);
      st = rnd(st);
      const float rb = RANDOM ? ((float)(st & 0xffff) / 32768.0f - 1.0f) * 0.01f : (0.5f - 0.01f * (float)(i + j));
      a[i][j] = (__bf16)ra; b[i][j] = (__bf16)rb;
    }
  float s = 0.f;
  if constexpr (BIG) {
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i >> 1) + 2 * (u & 1)], b[(i & 1) + (u & 2)], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  } else {
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i >> 2], b[i & 3], acc[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  }
  sink[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}
template <bool BIG, bool RANDOM> static void rate(const char *what) {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int grid = 4 * prop.multiProcessorCount, iters = 4000;
  float *sink; hipMalloc(&sink, (size_t)grid * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double best = 0;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k_rate<BIG, RANDOM>), dim3(grid), dim3(256), 0, 0, sink, iters);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)grid * 4 * iters * 32.0 * 2.0 * 16 * 16 * 32;      // both variants: 32 x (16x16x32) flops per iteration and wave
    if (rep) best = fmax(best, fl / (ms * 1e-3) / 1e12);
  }
  printf("%-44s %8.1f TFLOP/s\n", what, best);
  hipFree(sink);
}
int main() {
  rate<false, false>("16x16x32 bf16, few small operand values");
  rate<true, false>("32x32x16 bf16, few small operand values");
  rate<false, true>("16x16x32 bf16, random operands");
  rate<true, true>("32x32x16 bf16, random operands");
  float hA[16 * 32], hB[16 * 32], hC[32 * 32];
  for (int i = 0; i < 16 * 32; ++i) { hA[i] = (float)((i * 7) % 13 - 6) * 0.25f; hB[i] = (float)((i * 5) % 11 - 5) * 0.5f; }
  float *dA, *dB, *dC;
  hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dC, sizeof(hC));
  hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  hipMemcpy(hC, dC, sizeof(hC), hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int m = 0; m < 32; ++m)
    for (int n = 0; n < 32; ++n) {
      double s = 0;
      for (int kk = 0; kk < 16; ++kk) s += (double)hA[kk * 32 + m] * hB[kk * 32 + n];
      maxerr = fmax(maxerr, fabs(s - hC[m * 32 + n]));
    }
  printf("32x32x16 f16 layout check: max |err| = %g (%s)\n", maxerr, maxerr < 1e-3 ? "layout as assumed" : "LAYOUT WRONG");
  return maxerr < 1e-3 ? 0 : 1;
}
