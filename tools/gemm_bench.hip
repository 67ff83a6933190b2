// gemm_bench.hip -- dev microbenchmark for the TN tile engine variants (not part of the product).
// C[i][j] = sum_k A[k][i] * B[k][j], A: K x M, B: K x N row-major, fp32.
// hipcc --offload-arch=gfx950 -O3 -o gemm_bench tools/gemm_bench.hip && ./gemm_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>
#include "../projected-lmc_amd/csrc/gemm_core.hpp"

using namespace plmc;
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// ---------------- V0: product engine (16x16x4, BK16, register staging)
__global__ __launch_bounds__(256) void k_v0(const float *A, const float *B, float *C, int M, int N, int K) {
  __shared__ __align__(16) float smem[tile_smem_elems<float>()];
  Acc<float> acc; acc.zero();
  int bi = blockIdx.y, bj = blockIdx.x;
  tile_mainloop<float, false>(acc, A + bi * 128, M, B + bj * 128, N, K, smem);
  tile_store<float>(acc, C + (int64_t)bi * 128 * N + bj * 128, N);
}

// ---------------- V0r: product engine, C -= A^T B (read-modify-write epilogue, as the sweep's k_update)
__global__ __launch_bounds__(256) void k_v0_rmw(const float *A, const float *B, float *C, int M, int N, int K) {
  __shared__ __align__(16) float smem[tile_smem_elems<float>()];
  Acc<float> acc; acc.zero();
  int bi = blockIdx.y, bj = blockIdx.x;
  tile_mainloop<float, true>(acc, A + bi * 128, M, B + bj * 128, N, K, smem);
  tile_add_store<float>(acc, C + (int64_t)bi * 128 * N + bj * 128, N);
}

// ---------------- V0n: product engine, epilogue suppressed (one element per lane stored) -> prologue cost only
__global__ __launch_bounds__(256) void k_v0_nostore(const float *A, const float *B, float *C, int M, int N, int K) {
  __shared__ __align__(16) float smem[tile_smem_elems<float>()];
  Acc<float> acc; acc.zero();
  int bi = blockIdx.y, bj = blockIdx.x;
  tile_mainloop<float, true>(acc, A + bi * 128, M, B + bj * 128, N, K, smem);
  float s = 0.f;
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int r = 0; r < 4; ++r) s += acc.v[a][b][r];
  C[((int64_t)bi * 128 + (threadIdx.x >> 1)) * N + bj * 128 + (threadIdx.x & 1)] = s;
}

// ---------------- V0s: as V0r, but the first 1024 workgroups start staggered (0, 1/4, 1/2, 3/4 of a tile
// time for the 4 co-resident workgroups of a CU) so that prologues/epilogues stop coinciding.
__global__ __launch_bounds__(256) void k_v0_rmw_stagger(const float *A, const float *B, float *C, int M, int N, int K, int ticks) {
  __shared__ __align__(16) float smem[tile_smem_elems<float>()];
  const int w = blockIdx.y * gridDim.x + blockIdx.x;
  if (w < 1024) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long wait = (unsigned long long)(w >> 8) * ticks;
    while (__builtin_amdgcn_s_memrealtime() - t0 < wait) __builtin_amdgcn_s_sleep(32);
  }
  Acc<float> acc; acc.zero();
  int bi = blockIdx.y, bj = blockIdx.x;
  tile_mainloop<float, true>(acc, A + bi * 128, M, B + bj * 128, N, K, smem);
  tile_writeback<float, WB_ADD>(acc, C + (int64_t)bi * 128 * N + bj * 128, N, smem);
}
__global__ __launch_bounds__(256) void k_v0_rmw_wb(const float *A, const float *B, float *C, int M, int N, int K) {
  __shared__ __align__(16) float smem[tile_smem_elems<float>()];
  Acc<float> acc; acc.zero();
  int bi = blockIdx.y, bj = blockIdx.x;
  tile_mainloop<float, true>(acc, A + bi * 128, M, B + bj * 128, N, K, smem);
  tile_writeback<float, WB_ADD>(acc, C + (int64_t)bi * 128 * N + bj * 128, N, smem);
}

// ---------------- V1: 32x32x2 MFMA, register staging, unpadded LDS [BK][128]
template <int BKT>
__global__ __launch_bounds__(256) void k_v1(const float *__restrict__ A, const float *__restrict__ B, float *C, int M, int N, int K) {
  __shared__ __align__(16) float sA[2][BKT][128];
  __shared__ __align__(16) float sB[2][BKT][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const float *Ag = A + blockIdx.y * 128, *Bg = B + blockIdx.x * 128;
  constexpr int NCH = BKT * 32 / 256;
  f32x4 ra[NCH], rb[NCH];
  auto gload = [&](int kt) {
#pragma unroll
    for (int h = 0; h < NCH; ++h) { int c = tid + h * 256; int row = c >> 5, col = (c & 31) * 4;
      ra[h] = *(const f32x4 *)(Ag + (int64_t)(kt * BKT + row) * M + col);
      rb[h] = *(const f32x4 *)(Bg + (int64_t)(kt * BKT + row) * N + col); } };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int h = 0; h < NCH; ++h) { int c = tid + h * 256; int row = c >> 5, col = (c & 31) * 4;
      *(f32x4 *)&sA[buf][row][col] = ra[h]; *(f32x4 *)&sB[buf][row][col] = rb[h]; } };
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int nkt = K / BKT;
  gload(0); sstore(0); __syncthreads();
  const int fk = lane >> 5, fm = lane & 31;
  for (int kt = 0; kt < nkt; ++kt) {
    int buf = kt & 1;
    if (kt + 1 < nkt) gload(kt + 1);
#pragma unroll
    for (int ks = 0; ks < BKT / 2; ++ks) {
      float a0 = sA[buf][ks * 2 + fk][wm * 64 + fm], a1 = sA[buf][ks * 2 + fk][wm * 64 + 32 + fm];
      float b0 = sB[buf][ks * 2 + fk][wn * 64 + fm], b1 = sB[buf][ks * 2 + fk][wn * 64 + 32 + fm];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (kt + 1 < nkt) sstore(buf ^ 1);
    __syncthreads();
  }
  float *Cg = C + (int64_t)blockIdx.y * 128 * N + blockIdx.x * 128;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) {
    int row = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    int col = wn * 64 + b * 32 + (lane & 31);
    Cg[(int64_t)row * N + col] = acc[a][b][r];
  }
}

// ---------------- V3: 32x32x2, direct global->LDS (LDS-DMA), NST stages, counted vmcnt + raw barrier
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
template <int BKT, int NST>
__global__ __launch_bounds__(256) void k_v3(const float *__restrict__ A, const float *__restrict__ B, float *C, int M, int N, int K) {
  extern __shared__ __align__(16) float smem[];          // [NST][2][BKT][128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const float *Ag = A + blockIdx.y * 128, *Bg = B + blockIdx.x * 128;
  constexpr int IPW = BKT / 8;          // wave-instructions per operand per wave per k-tile (BKT rows, 2 rows/instr, 4 waves)
  auto stage = [&](int kt, int buf) {
    float *dA = smem + (buf * 2 + 0) * BKT * 128, *dB = smem + (buf * 2 + 1) * BKT * 128;
#pragma unroll
    for (int h = 0; h < IPW; ++h) {
      int row0 = (wave * IPW + h) * 2;                 // 2 rows per instruction
      int row = row0 + (lane >> 5), col = (lane & 31) * 4;
      __builtin_amdgcn_global_load_lds((gbl_void *)(Ag + (int64_t)(kt * BKT + row) * M + col), (lds_void *)(dA + row0 * 128), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gbl_void *)(Bg + (int64_t)(kt * BKT + row) * N + col), (lds_void *)(dB + row0 * 128), 16, 0, 0);
    }
  };
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  const int nkt = K / BKT;
  const int fk = lane >> 5, fm = lane & 31;
  // prologue: NST-1 tiles in flight
  for (int s = 0; s < NST - 1; ++s) if (s < nkt) stage(s, s);
  for (int kt = 0; kt < nkt; ++kt) {
    // wait for tile kt: allow (NST-2) younger tiles outstanding
    if (NST == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (NST == 3) { if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * IPW) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    else { if (kt + 2 < nkt) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(4 * IPW) : "memory"); else if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * IPW) : "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __builtin_amdgcn_s_barrier();
    // issue next stage into the buffer freed at the previous iteration
    if (kt + NST - 1 < nkt) stage(kt + NST - 1, (kt + NST - 1) % NST);
    const int buf = kt % NST;
    const float *pA = smem + (buf * 2 + 0) * BKT * 128 + wm * 64 + fm;
    const float *pB = smem + (buf * 2 + 1) * BKT * 128 + wn * 64 + fm;
#pragma unroll
    for (int ks = 0; ks < BKT / 2; ++ks) {
      float a0 = pA[(ks * 2 + fk) * 128], a1 = pA[(ks * 2 + fk) * 128 + 32];
      float b0 = pB[(ks * 2 + fk) * 128], b1 = pB[(ks * 2 + fk) * 128 + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
  float *Cg = C + (int64_t)blockIdx.y * 128 * N + blockIdx.x * 128;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int r = 0; r < 16; ++r) {
    int row = wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    int col = wn * 64 + b * 32 + (lane & 31);
    Cg[(int64_t)row * N + col] = acc[a][b][r];
  }
}

// ---------------- triangular pattern: out tile (ib, jb>=ib) = sum_{l >= jb*128} W[l][ib cols]^T W[l][jb cols]
// ord = 0: grid (jb, ib) as in the product; ord = 1: blockIdx.x = ib, blockIdx.y = jb (same-K tiles adjacent)
template <bool REV>
__global__ __launch_bounds__(256) void k_tri(const float *W, float *O, int n, int ord) {
  int jb = ord ? blockIdx.y : blockIdx.x, ib = ord ? blockIdx.x : blockIdx.y, lat = blockIdx.z;
  if (jb < ib) return;
  __shared__ __align__(16) float smem[tile_smem_elems<float>()];
  Acc<float> acc; acc.zero();
  const float *Wl = W + (size_t)lat * n * n + (size_t)jb * 128 * n;
  tile_mainloop<float, false, REV>(acc, Wl + ib * 128, n, Wl + jb * 128, n, n - jb * 128, smem);
  tile_store<float>(acc, O + (size_t)lat * n * n + (size_t)ib * 128 * n + jb * 128, n);
}

int main(int argc, char **argv) {
  int M = 8192, N = 8192, K = argc > 1 ? atoi(argv[1]) : 4096;
  size_t nA = (size_t)K * M, nB = (size_t)K * N, nC = (size_t)M * N;
  std::vector<float> hA(nA), hB(nB);
  srand(1);
  for (auto &v : hA) v = (rand() / (float)RAND_MAX) * 2 - 1;
  for (auto &v : hB) v = (rand() / (float)RAND_MAX) * 2 - 1;
  float *dA, *dB, *dC;
  CK(hipMalloc(&dA, nA * 4)); CK(hipMalloc(&dB, nB * 4)); CK(hipMalloc(&dC, nC * 4));
  CK(hipMemcpy(dA, hA.data(), nA * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB.data(), nB * 4, hipMemcpyHostToDevice));
  dim3 grid(N / 128, M / 128), block(256);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<float> hC(nC);
  auto check = [&](const char *name) {
    CK(hipMemcpy(hC.data(), dC, nC * 4, hipMemcpyDeviceToHost));
    double maxerr = 0;
    for (int t = 0; t < 64; ++t) { int i = (t * 977) % M, j = (t * 1543 + 7) % N; double s = 0;
      for (int k = 0; k < K; ++k) s += (double)hA[(size_t)k * M + i] * hB[(size_t)k * N + j];
      maxerr = fmax(maxerr, fabs(s - hC[(size_t)i * N + j])); }
    printf("  %s check max abs err %.3e\n", name, maxerr);
  };
  auto timeit = [&](const char *name, auto launch) {
    CK(hipMemset(dC, 0, nC * 4));
    launch(); CK(hipDeviceSynchronize());
    check(name);
    for (int w = 0; w < 2; ++w) launch();
    CK(hipEventRecord(e0)); const int R = 5; for (int r = 0; r < R; ++r) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= R;
    printf("%-28s %8.3f ms  %7.2f TFLOP/s\n", name, ms, 2.0 * M * N * K / ms / 1e9); fflush(stdout);
  };
  {  // the regime of the sweep: short K, RMW epilogue (timed without the correctness check)
    hipEvent_t a0, a1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_v0_rmw, grid, block, 0, 0, dA, dB, dC, M, N, K);
    CK(hipEventRecord(a0)); for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_v0_rmw, grid, block, 0, 0, dA, dB, dC, M, N, K);
    CK(hipEventRecord(a1)); CK(hipEventSynchronize(a1)); float ms; CK(hipEventElapsedTime(&ms, a0, a1)); ms /= 10;
    printf("v0 RMW epilogue K=%d          %8.3f ms  %7.2f TFLOP/s\n", K, ms, 2.0 * M * N * K / ms / 1e9);
  }
  {
    hipEvent_t a0, a1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k_v0_nostore, grid, block, 0, 0, dA, dB, dC, M, N, K);
    CK(hipEventRecord(a0)); for (int r = 0; r < 10; ++r) hipLaunchKernelGGL(k_v0_nostore, grid, block, 0, 0, dA, dB, dC, M, N, K);
    CK(hipEventRecord(a1)); CK(hipEventSynchronize(a1)); float ms; CK(hipEventElapsedTime(&ms, a0, a1)); ms /= 10;
    printf("v0 no-store epilogue K=%d     %8.3f ms  %7.2f TFLOP/s\n", K, ms, 2.0 * M * N * K / ms / 1e9);
  }
  {
    hipEvent_t a0, a1; CK(hipEventCreate(&a0)); CK(hipEventCreate(&a1));
    auto tm = [&](const char *nm, auto launch) {
      for (int w = 0; w < 2; ++w) launch();
      CK(hipEventRecord(a0)); for (int r = 0; r < 10; ++r) launch();
      CK(hipEventRecord(a1)); CK(hipEventSynchronize(a1)); float ms; CK(hipEventElapsedTime(&ms, a0, a1)); ms /= 10;
      printf("%-30s %8.3f ms  %7.2f TFLOP/s\n", nm, ms, 2.0 * M * N * K / ms / 1e9); };
    tm("v0 RMW coalesced writeback", [&] { hipLaunchKernelGGL(k_v0_rmw_wb, grid, block, 0, 0, dA, dB, dC, M, N, K); });
    for (int ticks : {400, 800, 1600, 2400})
      { char nm[64]; snprintf(nm, 64, "v0 RMW wb stagger %d ticks", ticks);
        tm(nm, [&] { hipLaunchKernelGGL(k_v0_rmw_stagger, grid, block, 0, 0, dA, dB, dC, M, N, K, ticks); }); }
  }
  timeit("v0 16x16x4 BK16 reg", [&] { hipLaunchKernelGGL(k_v0, grid, block, 0, 0, dA, dB, dC, M, N, K); });
  timeit("v1 32x32x2 BK16 reg", [&] { hipLaunchKernelGGL(k_v1<16>, grid, block, 0, 0, dA, dB, dC, M, N, K); });
  timeit("v1 32x32x2 BK32 reg", [&] { hipLaunchKernelGGL(k_v1<32>, grid, block, 0, 0, dA, dB, dC, M, N, K); });
#define V3(BKT, NST) { size_t sm = (size_t)NST * 2 * BKT * 128 * 4; CK(hipFuncSetAttribute((const void *)k_v3<BKT, NST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm)); \
    timeit("v3 dma BK" #BKT " st" #NST, [&] { hipLaunchKernelGGL((k_v3<BKT, NST>), grid, block, sm, 0, dA, dB, dC, M, N, K); }); }
  V3(16, 2) V3(16, 3) V3(16, 4) V3(32, 2) V3(32, 3) V3(32, 4)
  // ---- triangular tile pattern of K^-1 = W^T W: tiles jb >= ib, K range [jb*128, n), nlat matrices
  {
    const int n = 8192, m = n / 128, nlat = argc > 2 ? atoi(argv[2]) : 8;
    float *dW, *dO;
    CK(hipMalloc(&dW, (size_t)nlat * n * n * 4)); CK(hipMalloc(&dO, (size_t)nlat * n * n * 4));
    for (int l = 0; l < nlat; ++l) CK(hipMemcpy(dW + (size_t)l * n * n, hA.data(), (size_t)std::min((size_t)n * n, nA) * 4, hipMemcpyHostToDevice));
    double fl = 0; for (int jb = 0; jb < m; ++jb) fl += (double)(jb + 1) * 2.0 * 128 * 128 * (n - jb * 128);
    for (int rev = 0; rev < 2; ++rev) for (int ord = 0; ord < 2; ++ord) {
      auto launch = [&] { if (rev) hipLaunchKernelGGL((k_tri<true>), dim3(m, m, nlat), block, 0, 0, dW, dO, n, ord);
                          else hipLaunchKernelGGL((k_tri<false>), dim3(m, m, nlat), block, 0, 0, dW, dO, n, ord); };
      for (int w = 0; w < 2; ++w) launch();
      CK(hipEventRecord(e0)); for (int r = 0; r < 3; ++r) launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
      printf("tri rev=%d ord=%d nlat=%d        %8.3f ms  %7.2f TFLOP/s\n", rev, ord, nlat, ms, nlat * fl / ms / 1e9); fflush(stdout);
    }
  }
  return 0;
}
