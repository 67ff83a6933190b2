// macro_tile_probe.hip -- VERDICT r1 item 4 ("256 x 128 macro-tile for the tail update"): what would it buy?
//
// The trailing update's shape (C[tile] -= A^T B, K-major fp32 operands, depth 1024, 48 x 48 blocks of 128 x 128) computed
// (a) by the library's engine (128 x 128 tile, 4 waves, v_mfma_f32_16x16x4_f32) and (b) by a 256 x 128 macro-tile with
// 8 waves (4 x 2, the same 64 x 64 per wave, the same LDS row order), which reads the B-side slab once for twice the
// rows.  Reports time, TFLOP/s and checks the two results against each other.  Probe only -- not part of the product.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/macro_tile_probe tools/macro_tile_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include <vector>
#include "../projected-lmc_amd/csrc/gemm_core.hpp"

using namespace plmc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(NTHREADS, 1) void k_tiles_128(float *M, int64_t ld, int K, int tiles_per_row) {
  __shared__ __align__(16) float smem[tile_smem_elems<float>()];
  const int t = blockIdx.x, ib = t / tiles_per_row, jb = t % tiles_per_row;
  Acc<float> acc;
  acc.zero();
  tile_mainloop<float, false, false>(acc, M + (int64_t)ib * NB, ld, M + (int64_t)jb * NB, ld, K, smem);
  tile_writeback<float, WB_SUB>(acc, M + ((int64_t)K + (int64_t)ib * NB) * ld + (int64_t)jb * NB, ld, smem);
}

constexpr int LDTA = 260, LDTB = 132;                      // LDS row strides (4 (mod 32) words, as the engine's 132)
constexpr int STAGE = BK * (LDTA + LDTB);                  // floats per stage
// 512 threads = 8 waves (wm = wave >> 1 in 0..3, wn = wave & 1); C block = rows [256 ib2, +256) x cols [128 jb, +128)
__global__ __launch_bounds__(512, 2) void k_tiles_256(float *M, int64_t ld, int K, int tiles_per_row) {
  extern __shared__ __align__(16) float sm[];              // 2 stages; reused by the epilogue (64 x 132)
  const int t = blockIdx.x, ib2 = t / tiles_per_row, jb = t % tiles_per_row;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const float *Ag = M + (int64_t)ib2 * 256, *Bg = M + (int64_t)jb * NB;
  // global -> registers: A slab 16 x 256 = 1024 chunks of 16 B (2 per thread: rows r, r + 8), B slab 16 x 128 = 512 (1 per thread)
  const int ra_row = tid >> 6, ra_col = (tid & 63) * 4, rb_row = tid >> 5, rb_col = (tid & 31) * 4;
  float4 va[2], vb;
  auto gload = [&](int s) {
    const float *pa = Ag + ((int64_t)s * BK + ra_row) * ld + ra_col;
    va[0] = *reinterpret_cast<const float4 *>(pa);
    va[1] = *reinterpret_cast<const float4 *>(pa + 8 * ld);
    vb = *reinterpret_cast<const float4 *>(Bg + ((int64_t)s * BK + rb_row) * ld + rb_col);
  };
  auto rp = [](int r) { return (r & 3) * 4 + (r >> 2); };   // contraction row r = 4 ks + fk -> LDS row 4 fk + ks
  auto sstore = [&](int buf) {
    float *sA = sm + buf * STAGE, *sB = sA + BK * LDTA;
    *reinterpret_cast<float4 *>(sA + rp(ra_row) * LDTA + ra_col) = va[0];
    *reinterpret_cast<float4 *>(sA + rp(ra_row + 8) * LDTA + ra_col) = va[1];
    *reinterpret_cast<float4 *>(sB + rp(rb_row) * LDTB + rb_col) = vb;
  };
  const int fk = lane >> 4, fm = lane & 15;
  Acc<float> acc;
  acc.zero();
  const int nkt = K / BK;
  gload(0);
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) gload(kt + 1);
    const float *pa = sm + buf * STAGE + fk * 4 * LDTA + wm * 64 + fm;
    const float *pb = sm + buf * STAGE + BK * LDTA + fk * 4 * LDTB + wn * 64 + fm;
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      float a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = pa[ks * LDTA + i * 16];
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = pb[ks * LDTB + i * 16];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc.v[mt][nt] = Traits<float>::mfma(a[mt], b[nt], acc.v[mt][nt]);
    }
    if (kt + 1 < nkt) sstore(buf ^ 1);
    __syncthreads();
  }
  // epilogue: four passes of 64 rows through LDS, coalesced read-modify-write (16-byte accesses, full 512-byte rows)
  float *stg = sm;
  float *C = M + ((int64_t)K + (int64_t)ib2 * 256) * ld + (int64_t)jb * NB;
  const int crow = tid >> 5, ccol = (tid & 31) * 4;
#pragma unroll 1
  for (int p = 0; p < 4; ++p) {
    float4 cv[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) cv[h] = *reinterpret_cast<const float4 *>(C + (int64_t)(p * 64 + crow + 16 * h) * ld + ccol);
    if (wm == p) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) stg[(mt * 16 + Traits<float>::acc_row(lane, r)) * 132 + wn * 64 + nt * 16 + fm] = acc.v[mt][nt][r];
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const float4 sv = *reinterpret_cast<const float4 *>(stg + (crow + 16 * h) * 132 + ccol);
      const float4 o = {cv[h].x - sv.x, cv[h].y - sv.y, cv[h].z - sv.z, cv[h].w - sv.w};
      *reinterpret_cast<float4 *>(C + (int64_t)(p * 64 + crow + 16 * h) * ld + ccol) = o;
    }
    __syncthreads();
  }
}

int main() {
  const int n = 8192, K = 1024, TPR = 48;
  const int64_t ld = n + 128;
  const size_t elems = (size_t)ld * n;
  std::vector<float> h(elems);
  uint64_t st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) * (1.0 / 9007199254740992.0)) * 2.f - 1.f; };
  for (size_t i = 0; i < elems; ++i) h[i] = rnd();
  float *M0 = nullptr, *M1 = nullptr;
  CK(hipMalloc(&M0, elems * 4));
  CK(hipMalloc(&M1, elems * 4));
  CK(hipMemcpy(M0, h.data(), elems * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(M1, h.data(), elems * 4, hipMemcpyHostToDevice));
  const int smem256 = 2 * STAGE * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tiles_256), hipFuncAttributeMaxDynamicSharedMemorySize, smem256));
  hipLaunchKernelGGL(k_tiles_128, dim3(TPR * TPR), dim3(NTHREADS), 0, 0, M0, ld, K, TPR);
  hipLaunchKernelGGL(k_tiles_256, dim3(TPR / 2 * TPR), dim3(512), smem256, 0, M1, ld, K, TPR);
  CK(hipDeviceSynchronize());
  {  // same arithmetic in the same order per element: the two C regions must agree bit for bit
    std::vector<float> c0((size_t)NB * ld), c1((size_t)NB * ld);
    size_t ndiff = 0;
    for (int ib : {0, 17, 47}) {
      CK(hipMemcpy(c0.data(), M0 + ((int64_t)K + ib * NB) * ld, (size_t)NB * ld * 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(c1.data(), M1 + ((int64_t)K + ib * NB) * ld, (size_t)NB * ld * 4, hipMemcpyDeviceToHost));
      for (int i = 0; i < NB; ++i)
        for (int j = 0; j < TPR * NB; ++j) ndiff += c0[(size_t)i * ld + j] != c1[(size_t)i * ld + j];
    }
    printf("elements differing between the two engines on 3 block rows: %zu\n", ndiff);
  }
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double fl = 2.0 * NB * NB * (double)K * TPR * TPR;
  float ms = 0.f;
  for (int which = 0; which < 2; ++which) {
    auto launch = [&]() {
      if (which == 0) hipLaunchKernelGGL(k_tiles_128, dim3(TPR * TPR), dim3(NTHREADS), 0, 0, M0, ld, K, TPR);
      else hipLaunchKernelGGL(k_tiles_256, dim3(TPR / 2 * TPR), dim3(512), smem256, 0, M1, ld, K, TPR);
    };
    for (int w = 0; w < 2; ++w) launch();
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-40s %8.3f ms per launch  %7.1f TFLOP/s\n", which == 0 ? "128 x 128 tile, 4 waves (library engine)" : "256 x 128 macro-tile, 8 waves", ms / 5,
           fl / (ms / 5 * 1e-3) / 1e12);
  }
  return 0;
}
