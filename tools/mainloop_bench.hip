// mainloop_bench.hip -- dev microbenchmark: variants of the TN tile main loop (fp32, 128x128 tiles).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>
#include "../projected-lmc_amd/csrc/gemm_core.hpp"
using namespace plmc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// ---- candidate: whole-slab fragment prefetch, scheduling pinned
template <int VAR>
__device__ __forceinline__ void mainloop_x(Acc<float> &acc, const float *__restrict__ Ag, int64_t lda,
                                           const float *__restrict__ Bg, int64_t ldb, int K, float *smem) {
  using vec_t = f32x4;
  constexpr int EPV = 4, CPR = 32, NCH = 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  float *sA = smem, *sB = smem + 2 * BK * LDT;
  vec_t ra[NCH], rb[NCH];
  const int nkt = K / BK;
  auto gload = [&](int kt) {
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      int c = tid + h * NTHREADS, row = c / CPR, col = (c % CPR) * EPV;
      ra[h] = *reinterpret_cast<const vec_t *>(Ag + (int64_t)(kt * BK + row) * lda + col);
      rb[h] = *reinterpret_cast<const vec_t *>(Bg + (int64_t)(kt * BK + row) * ldb + col);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      int c = tid + h * NTHREADS, row = c / CPR, col = (c % CPR) * EPV;
      *reinterpret_cast<vec_t *>(sA + (buf * BK + row) * LDT + col) = ra[h];
      *reinterpret_cast<vec_t *>(sB + (buf * BK + row) * LDT + col) = rb[h];
    }
  };
  gload(0); sstore(0); __syncthreads();
  const int fk = lane >> 4, fm = lane & 15;
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) gload(kt + 1);
    const float *pa = sA + buf * BK * LDT + wm * 64 + fm, *pb = sB + buf * BK * LDT + wn * 64 + fm;
    float a[4][4], b[4][4];
    if (VAR == 1) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < 4; ++t) { a[ks][t] = pa[(ks * 4 + fk) * LDT + t * 16]; b[ks][t] = pb[(ks * 4 + fk) * LDT + t * 16]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc.v[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks][mt], b[ks][nt], acc.v[mt][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    } else {
      // two halves: fragments of ks 2,3 are fetched while the MFMAs of ks 0,1 run
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int t = 0; t < 4; ++t) { a[ks][t] = pa[(ks * 4 + fk) * LDT + t * 16]; b[ks][t] = pb[(ks * 4 + fk) * LDT + t * 16]; }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 2; ks < 4; ++ks)
#pragma unroll
        for (int t = 0; t < 4; ++t) { a[ks][t] = pa[(ks * 4 + fk) * LDT + t * 16]; b[ks][t] = pb[(ks * 4 + fk) * LDT + t * 16]; }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc.v[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks][mt], b[ks][nt], acc.v[mt][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 2; ks < 4; ++ks)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc.v[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks][mt], b[ks][nt], acc.v[mt][nt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (kt + 1 < nkt) sstore(buf ^ 1);
    __syncthreads();
  }
}

// ---- candidate: v_mfma_f32_32x32x2_f32 (2 x 2 MFMA tiles of 32 x 32 per wave), same LDS staging
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void k_gemm32(const float *A, const float *B, float *C, int M, int N, int K) {
  __shared__ __align__(16) float smem[tile_smem_elems<float>()];
  using vec_t = f32x4;
  constexpr int EPV = 4, CPR = 32, NCH = 2;
  const int bi = blockIdx.y, bj = blockIdx.x;
  const float *Ag = A + bi * 128, *Bg = B + bj * 128;
  const int64_t lda = M, ldb = N;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  float *sA = smem, *sB = smem + 2 * BK * LDT;
  vec_t ra[NCH], rb[NCH];
  const int nkt = K / BK;
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  auto gload = [&](int kt) {
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      int c = tid + h * NTHREADS, row = c / CPR, col = (c % CPR) * EPV;
      ra[h] = *reinterpret_cast<const vec_t *>(Ag + (int64_t)(kt * BK + row) * lda + col);
      rb[h] = *reinterpret_cast<const vec_t *>(Bg + (int64_t)(kt * BK + row) * ldb + col);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      int c = tid + h * NTHREADS, row = c / CPR, col = (c % CPR) * EPV;
      *reinterpret_cast<vec_t *>(sA + (buf * BK + row) * LDT + col) = ra[h];
      *reinterpret_cast<vec_t *>(sB + (buf * BK + row) * LDT + col) = rb[h];
    }
  };
  gload(0); sstore(0); __syncthreads();
  const int fk = lane >> 5, fm = lane & 31;
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) gload(kt + 1);
    const float *pa = sA + buf * BK * LDT + wm * 64 + fm, *pb = sB + buf * BK * LDT + wn * 64 + fm;
#pragma unroll
    for (int ks = 0; ks < BK / 2; ++ks) {
      float a[2], b[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) { a[t] = pa[(ks * 2 + fk) * LDT + t * 32]; b[t] = pb[(ks * 2 + fk) * LDT + t * 32]; }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
    }
    if (kt + 1 < nkt) sstore(buf ^ 1);
    __syncthreads();
  }
  float *Cg = C + (int64_t)bi * 128 * N + bj * 128;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = wm * 64 + mt * 32 + (i / 4) * 8 + (lane >> 5) * 4 + (i % 4), col = wn * 64 + nt * 32 + (lane & 31);
        Cg[(int64_t)row * N + col] = acc[mt][nt][i];
      }
}


// ---- candidate: 256 x 256 macro tile, 4 waves (2 x 2) of 128 x 128 = 8 x 8 MFMA tiles, one workgroup per CU
__device__ long long g_clk[4];
constexpr int BLD = 272;                                   // LDS row stride of a 256-wide slab (== 16 mod 32)
template <int BKB, int SKIP = 0>
__global__ __launch_bounds__(256, 1) void k_big(const float *A, const float *B, float *C, int M, int N, int K) {
  extern __shared__ __align__(16) float smem[];
  using vec_t = f32x4;
  constexpr int CPR = 64;                                  // 16-byte chunks per 256-wide row
  constexpr int NCH = BKB * CPR / 256;                     // chunks per thread per operand per slab
  const int bi = blockIdx.y, bj = blockIdx.x;
  const float *Ag = A + bi * 256, *Bg = B + bj * 256;
  const int64_t lda = M, ldb = N;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  float *sA = smem, *sB = smem + 2 * BKB * BLD;
  vec_t ra[NCH], rb[NCH];
  const int nkt = K / BKB;
  f32x4 acc[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  auto gload = [&](int kt) {
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      const int c = tid + h * 256, row = c / CPR, col = (c % CPR) * 4;
      ra[h] = *reinterpret_cast<const vec_t *>(Ag + (int64_t)(kt * BKB + row) * lda + col);
      rb[h] = *reinterpret_cast<const vec_t *>(Bg + (int64_t)(kt * BKB + row) * ldb + col);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      const int c = tid + h * 256, row = c / CPR, col = (c % CPR) * 4;
      *reinterpret_cast<vec_t *>(sA + (buf * BKB + row) * BLD + col) = ra[h];
      *reinterpret_cast<vec_t *>(sB + (buf * BKB + row) * BLD + col) = rb[h];
    }
  };
  gload(0); sstore(0); __syncthreads();
  const bool stamp = blockIdx.x == 3 && blockIdx.y == 5 && tid == 0;
  if (stamp) { g_clk[0] = (long long)__builtin_readcyclecounter(); g_clk[1] = (long long)wall_clock64(); }
  const int fk = lane >> 4, fm = lane & 15;
  float a[2][8], b[2][8];
  auto fload = [&](int buf, int ks, int slot) {
    const float *pa = sA + (buf * BKB + ks * 4 + fk) * BLD + wm * 128 + fm;
    const float *pb = sB + (buf * BKB + ks * 4 + fk) * BLD + wn * 128 + fm;
#pragma unroll
    for (int t = 0; t < 8; ++t) { a[slot][t] = pa[t * 16]; b[slot][t] = pb[t * 16]; }
  };
  fload(0, 0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (!(SKIP & 1) && kt + 1 < nkt) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < BKB / 4; ++ks) {
      const int cur = (SKIP & 2) ? 0 : (ks & 1);
      if (!(SKIP & 2) && ks + 1 < BKB / 4) fload(buf, ks + 1, cur ^ 1);
      if (!(SKIP & 1) && ks == BKB / 4 - 1 && kt + 1 < nkt) sstore(buf ^ 1);
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][mt], b[cur][nt], acc[mt][nt], 0, 0, 0);
      if (SKIP & 8) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // 4 MFMA
        }
      }
      if (SKIP & 16) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);   // 8 MFMA first, reads trail
          __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
      }
    }
    if (!(SKIP & 4)) __syncthreads();
    if (!(SKIP & 2) && kt + 1 < nkt) fload(buf ^ 1, 0, 0);
  }
  if (stamp) { g_clk[2] = (long long)__builtin_readcyclecounter(); g_clk[3] = (long long)wall_clock64(); }
  float *Cg = C + (int64_t)bi * 256 * N + bj * 256;
#pragma unroll
  for (int mt = 0; mt < 8; ++mt)
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Cg[(int64_t)(wm * 128 + mt * 16 + fk * 4 + r) * N + wn * 128 + nt * 16 + fm] = acc[mt][nt][r];
}
template <int BKB, int SKIP = 0> void run_big(const char *name, const float *A, const float *B, float *C, int M, int N, int K) {
  const size_t sm = (size_t)2 * 2 * BKB * BLD * 4;
  CK(hipFuncSetAttribute((const void *)(k_big<BKB, SKIP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
  dim3 g(N / 256, M / 256);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_big<BKB, SKIP>), g, dim3(256), sm, 0, A, B, C, M, N, K);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_big<BKB, SKIP>), g, dim3(256), sm, 0, A, B, C, M, N, K);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms, h; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemcpy(&h, C + 12345, 4, hipMemcpyDeviceToHost));
  long long ck[4]; CK(hipMemcpyFromSymbol(ck, HIP_SYMBOL(g_clk), 32));
  printf("%-28s M=N=%d K=%5d: %7.1f TFLOP/s  (C[12345]=%.4f)  in-kernel clock %.2f GHz, %.1f clk per MFMA\n", name, M, K, 2.0 * M * N * K / (ms / 5 * 1e-3) / 1e12, h,
         (double)(ck[2] - ck[0]) / ((double)(ck[3] - ck[1]) * 10.0), (double)(ck[2] - ck[0]) / ((double)K / 4 * 64));
}

// ---- candidate: 256 x 256 macro tile, permuted columns (lane owns an 8 x 8 sub-block of each 128 x 128 wave tile):
// fragments are two ds_read_b128 per operand, addresses advance by pointer increments, epilogue is direct float4.
template <int SKIP = 0>
__global__ __launch_bounds__(256, 1) void k_big2(const float *A, const float *B, float *C, int M, int N, int K) {
  extern __shared__ __align__(16) float smem[];
  using vec_t = f32x4;
  constexpr int BKB = 16, CPR = 64, NCH = BKB * CPR / 256;          // 4 chunks per thread per operand per slab
  const int bi = blockIdx.y, bj = blockIdx.x;
  const int64_t lda = M, ldb = N;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  float *sA = smem, *sB = smem + 2 * BKB * BLD;
  vec_t ra[NCH], rb[NCH];
  const int nkt = K / BKB;
  f32x4 acc[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  // this thread's 4 chunks of a slab: rows row0 + 4 h, column col0 (fixed) -> one pointer per operand, advanced per slab
  const int row0 = tid / CPR, col0 = (tid % CPR) * 4;
  const float *ga = A + bi * 256 + (int64_t)row0 * lda + col0, *gb = B + bj * 256 + (int64_t)row0 * ldb + col0;
  const int64_t stepA = (int64_t)BKB * lda, stepB = (int64_t)BKB * ldb, rsA = 4 * lda, rsB = 4 * ldb;
  auto gload = [&]() {
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      ra[h] = *reinterpret_cast<const vec_t *>(ga + h * rsA);
      rb[h] = *reinterpret_cast<const vec_t *>(gb + h * rsB);
    }
    ga += stepA; gb += stepB;
  };
  const int soff = row0 * BLD + col0;
  auto sstore = [&](int buf) {
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      *reinterpret_cast<vec_t *>(sA + buf * BKB * BLD + soff + h * 4 * BLD) = ra[h];
      *reinterpret_cast<vec_t *>(sB + buf * BKB * BLD + soff + h * 4 * BLD) = rb[h];
    }
  };
  gload(); sstore(0); __syncthreads();
  const bool stamp = blockIdx.x == 3 && blockIdx.y == 5 && tid == 0;
  if (stamp) { g_clk[0] = (long long)__builtin_readcyclecounter(); g_clk[1] = (long long)wall_clock64(); }
  const int fk = lane >> 4, fm = lane & 15;
  vec_t a[2][2], b[2][2];
  const int foffA = fk * BLD + wm * 128 + fm * 8, foffB = fk * BLD + wn * 128 + fm * 8;
  auto fload = [&](int buf, int ks, int slot) {
    const float *pa = sA + (buf * BKB + ks * 4) * BLD + foffA, *pb = sB + (buf * BKB + ks * 4) * BLD + foffB;
    a[slot][0] = *reinterpret_cast<const vec_t *>(pa); a[slot][1] = *reinterpret_cast<const vec_t *>(pa + 4);
    b[slot][0] = *reinterpret_cast<const vec_t *>(pb); b[slot][1] = *reinterpret_cast<const vec_t *>(pb + 4);
  };
  fload(0, 0, 0);
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (!(SKIP & 1) && kt + 1 < nkt) gload();
#pragma unroll
    for (int ks = 0; ks < BKB / 4; ++ks) {
      const int cur = ks & 1;
      if (ks + 1 < BKB / 4) fload(buf, ks + 1, cur ^ 1);
      if (!(SKIP & 1) && ks == BKB / 4 - 1 && kt + 1 < nkt) sstore(buf ^ 1);
#pragma unroll
      for (int mt = 0; mt < 8; ++mt)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][mt >> 2][mt & 3], b[cur][nt >> 2][nt & 3], acc[mt][nt], 0, 0, 0);
    }
    __syncthreads();
    if (kt + 1 < nkt) fload(buf ^ 1, 0, 0);
  }
  if (stamp) { g_clk[2] = (long long)__builtin_readcyclecounter(); g_clk[3] = (long long)wall_clock64(); }
  // lane (fk, fm), register r of tile (mt, nt): C row = 8 (4 fk + r) + mt, col = 8 fm + nt  (inside the wave tile)
  float *Cg = C + (int64_t)(bi * 256 + wm * 128) * N + bj * 256 + wn * 128 + fm * 8;
#pragma unroll
  for (int r = 0; r < 4; ++r)
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
      float *p = Cg + (int64_t)(8 * (4 * fk + r) + mt) * N;
      vec_t v0, v1;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) { v0[nt] = acc[mt][nt][r]; v1[nt] = acc[mt][4 + nt][r]; }
      *reinterpret_cast<vec_t *>(p) = v0;
      *reinterpret_cast<vec_t *>(p + 4) = v1;
    }
}
template <int SKIP = 0> void run_big2(const char *name, const float *A, const float *B, float *C, int M, int N, int K) {
  const size_t sm = (size_t)2 * 2 * 16 * BLD * 4;
  CK(hipFuncSetAttribute((const void *)(k_big2<SKIP>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
  dim3 g(N / 256, M / 256);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_big2<SKIP>), g, dim3(256), sm, 0, A, B, C, M, N, K);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_big2<SKIP>), g, dim3(256), sm, 0, A, B, C, M, N, K);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms, h; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemcpy(&h, C + 12345, 4, hipMemcpyDeviceToHost));
  long long ck[4]; CK(hipMemcpyFromSymbol(ck, HIP_SYMBOL(g_clk), 32));
  printf("%-28s M=N=%d K=%5d: %7.1f TFLOP/s  (C[12345]=%.4f)  in-kernel clock %.2f GHz, %.1f clk per MFMA\n", name, M, K, 2.0 * M * N * K / (ms / 5 * 1e-3) / 1e12, h,
         (double)(ck[2] - ck[0]) / ((double)(ck[3] - ck[1]) * 10.0), (double)(ck[2] - ck[0]) / ((double)K / 4 * 64));
}

// ---- candidate: 256 x 256 macro tile, 8 waves (4 x 2) of 64 x 128 = 4 x 8 MFMA tiles, two waves per SIMD
template <int BKB>
__global__ __launch_bounds__(512, 1) void k_big8(const float *A, const float *B, float *C, int M, int N, int K) {
  extern __shared__ __align__(16) float smem[];
  using vec_t = f32x4;
  constexpr int CPR = 64;
  constexpr int NCH = BKB * CPR / 512;                     // chunks per thread per operand per slab (2 for BK 16)
  const int bi = blockIdx.y, bj = blockIdx.x;
  const int64_t lda = M, ldb = N;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  float *sA = smem, *sB = smem + 2 * BKB * BLD;
  vec_t ra[NCH], rb[NCH];
  const int nkt = K / BKB;
  f32x4 acc[4][8];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int row0 = tid / CPR, col0 = (tid % CPR) * 4;      // rows row0 + 8 h
  const float *ga = A + bi * 256 + (int64_t)row0 * lda + col0, *gb = B + bj * 256 + (int64_t)row0 * ldb + col0;
  const int64_t stepA = (int64_t)BKB * lda, stepB = (int64_t)BKB * ldb, rsA = 8 * lda, rsB = 8 * ldb;
  auto gload = [&]() {
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      ra[h] = *reinterpret_cast<const vec_t *>(ga + h * rsA);
      rb[h] = *reinterpret_cast<const vec_t *>(gb + h * rsB);
    }
    ga += stepA; gb += stepB;
  };
  const int soff = row0 * BLD + col0;
  auto sstore = [&](int buf) {
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      *reinterpret_cast<vec_t *>(sA + buf * BKB * BLD + soff + h * 8 * BLD) = ra[h];
      *reinterpret_cast<vec_t *>(sB + buf * BKB * BLD + soff + h * 8 * BLD) = rb[h];
    }
  };
  gload(); sstore(0); __syncthreads();
  const bool stamp = blockIdx.x == 3 && blockIdx.y == 5 && tid == 0;
  if (stamp) { g_clk[0] = (long long)__builtin_readcyclecounter(); g_clk[1] = (long long)wall_clock64(); }
  const int fk = lane >> 4, fm = lane & 15;
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) gload();
    const float *pa = sA + buf * BKB * BLD + wm * 64 + fm, *pb = sB + buf * BKB * BLD + wn * 128 + fm;
#pragma unroll
    for (int ks = 0; ks < BKB / 4; ++ks) {
      float a[4], b[8];
#pragma unroll
      for (int t = 0; t < 4; ++t) a[t] = pa[(ks * 4 + fk) * BLD + t * 16];
#pragma unroll
      for (int t = 0; t < 8; ++t) b[t] = pb[(ks * 4 + fk) * BLD + t * 16];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 8; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mt], b[nt], acc[mt][nt], 0, 0, 0);
    }
    if (kt + 1 < nkt) sstore(buf ^ 1);
    __syncthreads();
  }
  if (stamp) { g_clk[2] = (long long)__builtin_readcyclecounter(); g_clk[3] = (long long)wall_clock64(); }
  float *Cg = C + (int64_t)bi * 256 * N + bj * 256;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 8; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Cg[(int64_t)(wm * 64 + mt * 16 + fk * 4 + r) * N + wn * 128 + nt * 16 + fm] = acc[mt][nt][r];
}
template <int BKB> void run_big8(const char *name, const float *A, const float *B, float *C, int M, int N, int K) {
  const size_t sm = (size_t)2 * 2 * BKB * BLD * 4;
  CK(hipFuncSetAttribute((const void *)(k_big8<BKB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm));
  dim3 g(N / 256, M / 256);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_big8<BKB>), g, dim3(512), sm, 0, A, B, C, M, N, K);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k_big8<BKB>), g, dim3(512), sm, 0, A, B, C, M, N, K);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms, h; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipMemcpy(&h, C + 12345, 4, hipMemcpyDeviceToHost));
  long long ck[4]; CK(hipMemcpyFromSymbol(ck, HIP_SYMBOL(g_clk), 32));
  printf("%-28s M=N=%d K=%5d: %7.1f TFLOP/s  (C[12345]=%.4f)  in-kernel clock %.2f GHz, %.1f clk per MFMA (wave 0)\n", name, M, K, 2.0 * M * N * K / (ms / 5 * 1e-3) / 1e12, h,
         (double)(ck[2] - ck[0]) / ((double)(ck[3] - ck[1]) * 10.0), (double)(ck[2] - ck[0]) / ((double)K / 4 * 32));
}

template <int VAR>
__global__ __launch_bounds__(256) void k_gemm(const float *A, const float *B, float *C, int M, int N, int K) {
  __shared__ __align__(16) float smem[tile_smem_elems<float>()];
  Acc<float> acc; acc.zero();
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (VAR == 0) tile_mainloop<float, false>(acc, A + bi * 128, M, B + bj * 128, N, K, smem);
  else if (VAR == 9) tile_mainloop<float, true>(acc, A + bi * 128, M, B + bj * 128, N, K, smem);
  else mainloop_x<VAR>(acc, A + bi * 128, M, B + bj * 128, N, K, smem);
  tile_writeback<float, WB_STORE>(acc, C + (int64_t)bi * 128 * N + bj * 128, N, smem);
}

template <int VAR> void run(const char *name, const float *A, const float *B, float *C, int M, int N, int K) {
  dim3 g(N / 128, M / 128);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_gemm<VAR>, g, dim3(256), 0, 0, A, B, C, M, N, K);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int R = 5;
  CK(hipEventRecord(e0));
  for (int r = 0; r < R; ++r) hipLaunchKernelGGL(k_gemm<VAR>, g, dim3(256), 0, 0, A, B, C, M, N, K);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<float> h(4);
  CK(hipMemcpy(h.data(), C + 12345, 16, hipMemcpyDeviceToHost));
  printf("%-28s M=N=%d K=%5d: %7.1f TFLOP/s  (C[12345]=%.4f)\n", name, M, K, 2.0 * M * N * K / (ms / R * 1e-3) / 1e12, h[0]);
}
int main(int argc, char **argv) {
  const int M = 8192, N = 8192, Kmax = 8192;
  const bool only_v0 = argc > 1;        // counter passes: just the product engine at K = 8192
  std::vector<float> ha((size_t)Kmax * M), hb((size_t)Kmax * N);
  const bool gauss = getenv("MB_GAUSS") != nullptr;
  if (gauss) {
    unsigned long long st = 88172645463325252ULL;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
    for (size_t i = 0; i < ha.size(); ++i) { ha[i] = (float)(sqrt(-2.0 * log(rnd() + 1e-300)) * cos(6.283185307179586 * rnd())); hb[i] = (float)(sqrt(-2.0 * log(rnd() + 1e-300)) * cos(6.283185307179586 * rnd())); }
  } else
  for (size_t i = 0; i < ha.size(); ++i) { ha[i] = (float)((i * 2654435761u >> 20) & 255) / 256.f - 0.5f; hb[i] = (float)((i * 40503u >> 8) & 255) / 256.f - 0.5f; }
  float *A, *B, *C;
  CK(hipMalloc(&A, ha.size() * 4)); CK(hipMalloc(&B, hb.size() * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
  CK(hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  if (only_v0) { run<0>("v0 product engine", A, B, C, M, N, 8192); return 0; }
  {
    dim3 g(N / 128, M / 128);
    for (int K : {8192, 1024}) {
      for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_gemm32, g, dim3(256), 0, 0, A, B, C, M, N, K);
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      CK(hipEventRecord(e0));
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k_gemm32, g, dim3(256), 0, 0, A, B, C, M, N, K);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms, h; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(&h, C + 12345, 4, hipMemcpyDeviceToHost));
      printf("%-28s M=N=%d K=%5d: %7.1f TFLOP/s  (C[12345]=%.4f)\n", "mfma 32x32x2", M, K, 2.0 * M * N * K / (ms / 5 * 1e-3) / 1e12, h);
    }
  }
  for (int K : {8192, 1024}) { run_big8<16>("big8 (8 waves) BK16", A, B, C, M, N, K); run_big8<32>("big8 (8 waves) BK32", A, B, C, M, N, K); run_big2<0>("big2 permuted b128", A, B, C, M, N, K); run_big2<1>("big2 no global", A, B, C, M, N, K); run_big<16>("big 256x256 BK16", A, B, C, M, N, K); run_big<32>("big 256x256 BK32", A, B, C, M, N, K);
    run_big<16, 1>("big BK16 no global/sstore", A, B, C, M, N, K); run_big<16, 9>("  + 1 read : 4 mfma pinned", A, B, C, M, N, K); run_big<16, 17>("  + 8 mfma : 2 reads pinned", A, B, C, M, N, K); run_big<16, 8>("full + 1 read : 4 mfma", A, B, C, M, N, K); run_big<16, 2>("big BK16 no fragment loads", A, B, C, M, N, K);
    run_big<16, 3>("big BK16 no global, no frag", A, B, C, M, N, K); run_big<16, 7>("big BK16 bare + no barrier", A, B, C, M, N, K); run_big<16, 4>("big BK16 no barrier only", A, B, C, M, N, K); }
  for (int K : {8192, 1024}) {
    run<0>("v0 product engine", A, B, C, M, N, K);
    run<9>("v0 with NEG", A, B, C, M, N, K);
    run<1>("v1 slab prefetch", A, B, C, M, N, K);
    run<2>("v2 two-half prefetch", A, B, C, M, N, K);
  }
  return 0;
}
