// bf3v2_probe.hip -- the second bf16-split tile engine (csrc/bf3_engine.hpp) against the fp32 engine, alone on the GPU.
//
// C[tile] -= A^T B for a 48 x 48 grid of 128 x 128 tiles at depth K (the shape of the sweep's trailing update): the fp32
// engine (v_mfma_f32_16x16x4_f32, 128 x 128 tile per workgroup) and the bf16 engine (six plane products into two
// accumulator levels, 256 x 128 macro tile per workgroup, k8-ordered planes through LDS-DMA).  Reports time per launch,
// TFLOP/s on 2 * 128^2 * K per tile, and the error of both against an fp64 host product on sample tiles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/bf3v2_probe tools/bf3v2_probe.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <random>
#include <vector>
#include "../projected-lmc_amd/csrc/bf3_engine.hpp"

using namespace plmc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(NTHREADS, 1) void k_f32_tiles(const float *__restrict__ P, float *C, int64_t ld, int K, int tiles_per_row) {
  __shared__ __align__(16) float smem[tile_smem_elems<float>()];
  const int t = blockIdx.x, ib = t / tiles_per_row, jb = t % tiles_per_row;
  Acc<float> acc;
  acc.zero();
  tile_mainloop<float, false, false>(acc, P + (int64_t)ib * NB, ld, P + (int64_t)jb * NB, ld, K, smem);
  tile_writeback<float, WB_SUB>(acc, C + (int64_t)ib * NB * ld + (int64_t)jb * NB, ld, smem);
}

// fp32 panel [K][ld] -> k8-ordered planes; grid (ld / 128, K / 128)
__global__ __launch_bounds__(NTHREADS) void k_split(const float *__restrict__ X, int64_t ld, unsigned short *__restrict__ P) {
  const int cb = blockIdx.x, rb = blockIdx.y;
  b3_split_block<false>(X + (int64_t)rb * NB * ld + (int64_t)cb * NB, ld, P + b3_index((int64_t)rb * NB, 0, (int64_t)cb * NB, ld), ld, nullptr, 0,
                        threadIdx.x);
}

// ORDER 0: macro tile = blockIdx.x in row-major order (jb fastest); 1: ib fastest (workgroups that run together share the
// B strip ... and walk the A strips); 2: XCD-local 4 x 8 blocks of macro tiles dealt by blockIdx % 8 (speed only)
template <int ORDER>
__global__ __launch_bounds__(B3_NT, 2) void k_bf3v2_tiles(const unsigned short *__restrict__ P, float *C, int64_t ld, int K, int mrows, int tcols) {
  __shared__ __align__(16) unsigned char lds[B3_LDS_BYTES];
  int t = blockIdx.x, mb, jb;
  if (ORDER == 0) { mb = t / tcols; jb = t % tcols; }
  else if (ORDER == 1) { mb = t % mrows; jb = t / mrows; }
  else {
    // 8 XCDs; XCD x takes the super-blocks x, x + 8, ... of 4 macro rows x 8 tile columns (32 workgroups = one per CU)
    const int xcd = t & 7, slot = t >> 3;
    const int sbc = tcols / 8;                           // super-block columns
    const int sb = xcd + 8 * (slot >> 5), in = slot & 31;
    if (sb >= sbc * (mrows / 4)) return;
    mb = (sb / sbc) * 4 + (in >> 3);
    jb = (sb % sbc) * 8 + (in & 7);
  }
  Acc<float> acc0, acc1;
  acc0.zero();
  acc1.zero();
  b3_mainloop(acc0, acc1, P + (int64_t)mb * 256 * 8, ld, P + (int64_t)jb * NB * 8, ld, K, lds);
  const int tid = threadIdx.x & 255, half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc0.v[a][b] += acc1.v[a][b];
  tile_writeback<float, WB_SUB>(acc0, C + ((int64_t)mb * 256 + half * 128) * ld + (int64_t)jb * NB, ld,
                                reinterpret_cast<float *>(lds + half * B3_WB_BYTES), tid);
}

int main() {
  const int K = 1024, TPR = 48;
  const int64_t ld = (int64_t)TPR * NB + 128;            // an odd number of 128-blocks, as the factor buffers
  const size_t pe = (size_t)K * ld, ce = (size_t)TPR * NB * ld;
  std::vector<float> hp(pe), hc(ce);
  std::mt19937_64 rng(42);
  std::normal_distribution<float> nd(0.f, 1.f);
  for (auto &v : hp) v = nd(rng);
  for (auto &v : hc) v = 30.f * nd(rng);
  float *P32 = nullptr, *C0 = nullptr, *C1 = nullptr;
  unsigned short *Pl = nullptr;
  CK(hipMalloc(&P32, pe * 4));
  CK(hipMalloc(&C0, ce * 4));
  CK(hipMalloc(&C1, ce * 4));
  CK(hipMalloc(&Pl, (size_t)b3_elems(K, ld) * 2));
  CK(hipMemcpy(P32, hp.data(), pe * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(C0, hc.data(), ce * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(C1, hc.data(), ce * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms = 0.f;
  const double fl = 2.0 * NB * NB * (double)K * TPR * TPR;
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_split, dim3((unsigned)(ld / NB), K / NB), dim3(NTHREADS), 0, 0, P32, ld, Pl);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("split of a %d x %lld panel into k8-ordered bf16 planes: %.1f us (%.0f GB/s on 10 bytes per element)\n", K, (long long)ld, 1e3 * ms,
         10.0 * pe / (ms * 1e-3) / 1e9);
  hipLaunchKernelGGL(k_f32_tiles, dim3(TPR * TPR), dim3(NTHREADS), 0, 0, P32, C0, ld, K, TPR);
  hipLaunchKernelGGL((k_bf3v2_tiles<0>), dim3(TPR / 2 * TPR), dim3(B3_NT), 0, 0, Pl, C1, ld, K, TPR / 2, TPR);
  CK(hipDeviceSynchronize());
  std::vector<float> c0((size_t)NB * ld), c1((size_t)NB * ld);
  double e32 = 0, ebf = 0, s32 = 0, sbf = 0, scale = 0;
  size_t cnt = 0;
  const int tiles[6][2] = {{0, 0}, {3, 17}, {47, 47}, {20, 5}, {1, 1}, {46, 0}};
  for (auto &tt : tiles) {
    const int ib = tt[0], jb = tt[1];
    CK(hipMemcpy(c0.data(), C0 + (int64_t)ib * NB * ld, (size_t)NB * ld * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c1.data(), C1 + (int64_t)ib * NB * ld, (size_t)NB * ld * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < NB; i += 3)
      for (int j = 0; j < NB; j += 5) {
        double acc = 0.0;
        for (int k = 0; k < K; ++k) acc += (double)hp[(size_t)k * ld + ib * NB + i] * (double)hp[(size_t)k * ld + jb * NB + j];
        const double ref = (double)hc[((size_t)ib * NB + i) * ld + jb * NB + j] - acc;
        const double d0 = c0[(size_t)i * ld + jb * NB + j] - ref, d1 = c1[(size_t)i * ld + jb * NB + j] - ref;
        e32 = fmax(e32, fabs(d0)); ebf = fmax(ebf, fabs(d1));
        s32 += d0 * d0; sbf += d1 * d1; ++cnt;
        scale = fmax(scale, fabs(ref));
      }
  }
  printf("error vs fp64 on sample tiles (|C| up to %.1f): fp32 engine max %.3e rms %.3e | bf16 engine max %.3e rms %.3e\n", scale, e32,
         sqrt(s32 / cnt), ebf, sqrt(sbf / cnt));
  if (!(ebf < 1e-3 * scale)) { printf("bf16 engine result is WRONG\n"); return 1; }
  auto time_it = [&](const char *name, auto launch) -> int {
    for (int w = 0; w < 2; ++w) launch();
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-40s %8.3f ms per launch  %7.1f TFLOP/s (fp32-equivalent)\n", name, ms / 5, fl / (ms / 5 * 1e-3) / 1e12);
    return 0;
  };
  for (int rep = 0; rep < 2; ++rep) {
    if (time_it("fp32 engine (16x16x4 f32)", [&]() { hipLaunchKernelGGL(k_f32_tiles, dim3(TPR * TPR), dim3(NTHREADS), 0, 0, P32, C0, ld, K, TPR); })) return 1;
    if (time_it("bf16 engine, row-major macro tiles", [&]() { hipLaunchKernelGGL((k_bf3v2_tiles<0>), dim3(TPR / 2 * TPR), dim3(B3_NT), 0, 0, Pl, C1, ld, K, TPR / 2, TPR); })) return 1;
    if (time_it("bf16 engine, column-major macro tiles", [&]() { hipLaunchKernelGGL((k_bf3v2_tiles<1>), dim3(TPR / 2 * TPR), dim3(B3_NT), 0, 0, Pl, C1, ld, K, TPR / 2, TPR); })) return 1;
    if (time_it("bf16 engine, XCD-local 4 x 8 blocks", [&]() { hipLaunchKernelGGL((k_bf3v2_tiles<2>), dim3(8 * ((TPR / 8 * TPR / 8 + 7) / 8) * 32), dim3(B3_NT), 0, 0, Pl, C1, ld, K, TPR / 2, TPR); })) return 1;
  }
  return 0;
}
