// chain_contention_probe.hip -- why do the chain kernels of the sweep take twice as long beside the bulk kernels?
//
// Foreground (what the chain does): 64 dependent launches on one stream of either k_diag (one 512-thread workgroup) or a
// K = 128 half-tile product with burst loads and a read-modify-write epilogue (32 workgroups, the shape of the chain's
// rank-128 update).  Background (what the bulk does): a long launch of depth-1024 tile products with the same engine,
// or a pure copy stream, on a second stream.  Each case reports the average foreground launch time in us.
// Cases: background off / MFMA tiles at 4 workgroups per CU / at 1 per CU / copy only / MFMA tiles on a stream whose CU
// mask leaves 32 CUs (or 8) free, with the foreground stream unmasked or masked onto exactly those CUs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/chain_contention_probe tools/chain_contention_probe.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "../projected-lmc_amd/csrc/diag_block.hpp"

using namespace plmc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// background: C[tile] -= A^T B over depth K, tiles of a 128 x 128 grid over a scratch matrix (reads and writes stay inside it)
__global__ __launch_bounds__(NTHREADS, 1) void k_bg_tiles(float *M, int64_t ld, int K, int tiles_per_row, int mode) {
  extern __shared__ __align__(16) float dyn[];
  __shared__ __align__(16) float smem[tile_smem_elems<float>()];
  // mode 0: every tile its own operand panels (the sweep's bulk); 1: all tiles read the SAME two panels (operands stay in
  // L2: MFMA load without fabric traffic, C still read-modify-written); 2: as 1 and every tile rewrites the same C tile
  const int t = blockIdx.x, ib = t / tiles_per_row, jb = t % tiles_per_row;
  const int oi = mode ? 0 : ib, oj = mode ? 1 : jb, ci = mode == 2 ? 0 : ib, cj = mode == 2 ? 0 : jb;
  Acc<float> acc;
  acc.zero();
  tile_mainloop<float, false, false>(acc, M + (int64_t)oi * NB, ld, M + (int64_t)oj * NB, ld, K, smem);
  tile_writeback<float, WB_SUB>(acc, M + ((int64_t)K + (int64_t)ci * NB) * ld + (int64_t)cj * NB, ld, smem);
  if (dyn[0] == 12345.f && threadIdx.x == 9999) M[0] = dyn[1];     // keeps the dynamic LDS request alive
}
__global__ __launch_bounds__(256) void k_bg_copy(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n, int reps) {
  for (int r = 0; r < reps; ++r)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
// foreground 2: the chain's rank-128 update, half tiles with burst loads
__global__ __launch_bounds__(NTHREADS, 1) void k_fg_update(float *M, int64_t ld) {
  __builtin_amdgcn_s_setprio(3);
  __shared__ __align__(16) float smem[tile_smem_elems<float>()];
  const int t = blockIdx.x, ib = t / 8, jb = t % 8, h0 = (blockIdx.y & 1) * 64;
  Acc<float, 2> acc;
  acc.zero();
  tile_mainloop_burst<float, 2, 4, 8>(acc, M + (int64_t)ib * NB + h0, ld, M + (int64_t)jb * NB, ld, NB, smem);
  tile_writeback<float, WB_SUB, 2>(acc, M + ((int64_t)NB + (int64_t)ib * NB + h0) * ld + (int64_t)jb * NB, ld, smem);
}

int main() {
  const int n = 8192;
  const int64_t ld = n + 128;
  float *A = nullptr, *Vd = nullptr, *Wg = nullptr, *Mbg = nullptr, *Mfg = nullptr, *cs = nullptr, *cd = nullptr;
  CK(hipMalloc(&A, sizeof(float) * ld * n));
  CK(hipMalloc(&Vd, sizeof(float) * 64 * NB * NB));
  CK(hipMalloc(&Wg, sizeof(float) * 64 * NB * NB));
  CK(hipMalloc(&Mbg, sizeof(float) * ld * n));
  CK(hipMalloc(&Mfg, sizeof(float) * ld * 2048));
  const size_t copy_n = (size_t)64 << 20;                  // float4 elements: 1 GiB each way
  CK(hipMalloc(&cs, copy_n * 16));
  CK(hipMalloc(&cd, copy_n * 16));
  CK(hipMemset(Mbg, 0, sizeof(float) * ld * n));
  CK(hipMemset(Mfg, 0, sizeof(float) * ld * 2048));
  CK(hipMemset(cs, 0, copy_n * 16));
  {                                                        // diagonal blocks: 4 I + 0.01 (SPD), everything else 0
    std::vector<float> h((size_t)ld * n, 0.f);
    for (int b = 0; b < 64; ++b)
      for (int i = 0; i < NB; ++i)
        for (int j = 0; j < NB; ++j) h[(size_t)(b * NB + i) * ld + b * NB + j] = (i == j ? 4.f : 0.f) + 0.01f;
    CK(hipMemcpy(A, h.data(), sizeof(float) * ld * n, hipMemcpyHostToDevice));
  }
  int ncu = 0;
  CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0));
  printf("CUs: %d\n", ncu);
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t fg, bg, fg_m32, bg_m32, fg_m8, bg_m8;
  CK(hipStreamCreateWithPriority(&fg, hipStreamNonBlocking, hi));
  CK(hipStreamCreateWithFlags(&bg, hipStreamNonBlocking));
  auto mask_streams = [&](int nfree, bool spread, hipStream_t *f, hipStream_t *b) -> int {
    std::vector<uint32_t> mf(ncu / 32, 0u), mb(ncu / 32, 0xffffffffu);
    for (int k = 0; k < nfree; ++k) {
      const int cu = spread ? k * (ncu / nfree) : ncu - 1 - k;
      mf[cu / 32] |= 1u << (cu % 32);
      mb[cu / 32] &= ~(1u << (cu % 32));
    }
    CK(hipExtStreamCreateWithCUMask(f, (uint32_t)mf.size(), mf.data()));
    CK(hipExtStreamCreateWithCUMask(b, (uint32_t)mb.size(), mb.data()));
    return 0;
  };
  if (mask_streams(32, false, &fg_m32, &bg_m32)) return 1;
  if (mask_streams(8, true, &fg_m8, &bg_m8)) return 1;

  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](const char *name, int fgkind, hipStream_t fs, int bgkind, hipStream_t bs, unsigned bg_lds) -> int {
    // background first (long enough to cover the foreground), then the 64 foreground launches
    if (bgkind == 1) {
      if (bg_lds & 0xffffffu) CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bg_tiles), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bg_lds & 0xffffffu)));
      for (int rep = 0; rep < 8; ++rep)
        hipLaunchKernelGGL(k_bg_tiles, dim3(48 * 48), dim3(NTHREADS), bg_lds & 0xffffffu, bs, Mbg, ld, 1024, 48, (int)(bg_lds >> 24));
    } else if (bgkind == 2) {
      hipLaunchKernelGGL(k_bg_copy, dim3(2048), dim3(256), 0, bs, (const float4 *)cs, (float4 *)cd, copy_n, 12);
    }
    CK(hipEventRecord(e0, fs));
    for (int r = 0; r < 64; ++r) {
      if (fgkind == 0)
        hipLaunchKernelGGL((k_diag<float>), dim3(1), dim3(DIAG_NT), 0, fs, A, ld, (int64_t)0, r, Vd, (int64_t)0, Wg + (size_t)r * NB * NB, (int64_t)NB, (int64_t)0);
      else
        hipLaunchKernelGGL(k_fg_update, dim3(32, 2), dim3(NTHREADS), 0, fs, Mfg, ld);
    }
    CK(hipEventRecord(e1, fs));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    hipError_t q = hipStreamQuery(bs);
    CK(hipDeviceSynchronize());
    printf("%-28s %-58s %7.1f us per launch%s\n", fgkind == 0 ? "k_diag x 64" : "rank-128 update (64 WG) x 64", name, 1e3 * ms / 64,
           (bgkind && q == hipSuccess) ? "   [background ended early]" : "");
    // restore the diagonal blocks k_diag overwrote
    return 0;
  };
  for (int fgkind = 0; fgkind < 2; ++fgkind) {
    for (int warm = 0; warm < 2; ++warm) if (run("(warm-up)", fgkind, fg, 0, bg, 0)) return 1;
    if (run("alone", fgkind, fg, 0, bg, 0)) return 1;
    if (run("beside MFMA tiles, 4 WG/CU", fgkind, fg, 1, bg, 0)) return 1;
    if (run("beside MFMA tiles, 3 WG/CU (16 KB LDS pad)", fgkind, fg, 1, bg, 16000)) return 1;
    if (run("beside MFMA tiles, 1 WG/CU (45 KB LDS pad)", fgkind, fg, 1, bg, 45000)) return 1;
    if (run("beside MFMA tiles, 4 WG/CU, operands L2-resident", fgkind, fg, 1, bg, 1u << 24)) return 1;
    if (run("beside MFMA tiles, 4 WG/CU, operands + C L2-resident", fgkind, fg, 1, bg, 2u << 24)) return 1;
    if (run("beside a copy stream", fgkind, fg, 2, bg, 0)) return 1;
    if (run("MFMA tiles masked off 32 CUs, foreground unmasked", fgkind, fg, 1, bg_m32, 0)) return 1;
    if (run("MFMA tiles masked off 32 CUs, foreground ON those 32", fgkind, fg_m32, 1, bg_m32, 0)) return 1;
    if (run("MFMA tiles masked off 8 CUs (spread), foreground unmasked", fgkind, fg, 1, bg_m8, 0)) return 1;
    if (run("MFMA tiles masked off 8 CUs (spread), foreground ON those 8", fgkind, fg_m8, 1, bg_m8, 0)) return 1;
  }
  return 0;
}
