// bf3v2_probe.hip -- the second bf16-split tile engine (csrc/bf3_engine.hpp) against the fp32 engine, alone on the GPU.
//
// C[tile] -= A^T B for a 48 x 48 grid of 128 x 128 tiles at depth K (the shape of the sweep's trailing update): the fp32
// engine (v_mfma_f32_16x16x4_f32, 128 x 128 tile per workgroup) and the bf16 engine (six plane products into two
// accumulator levels, 256 x 128 macro tile per workgroup, k8-ordered planes through LDS-DMA).  Reports time per launch,
// TFLOP/s on 2 * 128^2 * K per tile, and the error of both against an fp64 host product on sample tiles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/bin/bf3v2_probe tools/bf3v2_probe.hip
// HISTORICAL: written against the first form of bf3_engine.hpp (commit 'bf16x3 engine, second form'; its outputs are
// profiles/r03_bf3v2_probe_*.txt).  The engine's interface has moved on (scheme template parameter, pre-load hook);
// tools/engine_rate_probe.hip is the probe that builds against the current header.
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

// ---- speed prototype: TWO fp16 planes (x = h0 + 2^-11 h1), three plane products into two levels (h0.h0 | h1.h0 + h0.h1):
// tools/split_numerics_probe.hip measures that arithmetic at 0.36-0.43 x the fp32 chain's error.  Same macro tile and k8
// layout (two planes per k8 row); 48 KB per stage, NST stages (3: counted vmcnt, the DMA of stage s + 1 stays in flight).
typedef _Float16 h2_f16x8 __attribute__((ext_vector_type(8)));
constexpr int H2_A_PLANE = 4 * 256 * 16, H2_B_PLANE = 4 * 128 * 16, H2_STAGE = 2 * (H2_A_PLANE + H2_B_PLANE);   // 49 152
__global__ __launch_bounds__(NTHREADS) void k_split_h2(const float *__restrict__ X, int64_t ld, unsigned short *__restrict__ P, float scale) {
  const int cb = blockIdx.x, rb = blockIdx.y;
  const float *S = X + (int64_t)rb * NB * ld + (int64_t)cb * NB;
  for (int w = threadIdx.x; w < 16 * 32; w += NTHREADS) {
    const int k8 = w >> 5, c4 = (w & 31) * 4;
    float x[8][4];
    for (int r = 0; r < 8; ++r) { const float4 v = *reinterpret_cast<const float4 *>(S + (int64_t)(k8 * 8 + r) * ld + c4); x[r][0] = v.x; x[r][1] = v.y; x[r][2] = v.z; x[r][3] = v.w; }
    unsigned short *dst = P + ((((int64_t)rb * 16 + k8) * 2) * ld + (int64_t)cb * NB + c4) * 8;
    for (int c = 0; c < 4; ++c) {
      h2_f16x8 h, m;
      for (int r = 0; r < 8; ++r) { const float v = x[r][c] * scale; const _Float16 a = (_Float16)v; h[r] = a; m[r] = (_Float16)((v - (float)a) * 2048.0f); }
      *reinterpret_cast<h2_f16x8 *>(dst + c * 8) = h;
      *reinterpret_cast<h2_f16x8 *>(dst + ld * 8 + c * 8) = m;
    }
  }
}
template <int NST>
__global__ __launch_bounds__(B3_NT, 2) void k_h2_tiles(const unsigned short *__restrict__ P, float *C, int64_t ld, int K, int mrows, int tcols, float inv_scale2) {
  __shared__ __align__(16) unsigned char lds[NST * H2_STAGE];
  const int t = blockIdx.x, mb = t / tcols, jb = t % tcols;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = w >> 2, wm = (w >> 1) & 1, wn = w & 1;
  // 48 pieces per stage: A 32 (plane, k-group, 64-column segment) = 4 per wave, B 16 = 2 per wave
  const unsigned RS = (unsigned)(2 * ld * 16), PL = (unsigned)(ld * 16);
  const unsigned gA0 = (unsigned)(w >> 2) * RS + (unsigned)(w & 3) * 1024u;           // A piece j (0..3): plane j >> 1, k-group (w >> 2) + 2 (j & 1)
  const unsigned gB0 = (unsigned)((w >> 1) & 3) * RS + (unsigned)(w & 1) * 1024u;     // B piece j (0..1): plane j
  const unsigned lA0 = (unsigned)(((w >> 2) * 256 + (w & 3) * 64) * 16);
  const unsigned lB0 = (unsigned)(2 * H2_A_PLANE + (((w >> 1) & 3) * 128 + (w & 1) * 64) * 16);
  const unsigned voff = (unsigned)lane * 16u;
  const char *baseA = reinterpret_cast<const char *>(P + (int64_t)mb * 256 * 8), *baseB = reinterpret_cast<const char *>(P + (int64_t)jb * NB * 8);
  const int64_t step = 4 * (int64_t)RS;
  typedef __attribute__((address_space(3))) void lds_void;
  auto issue = [&](int buf) {
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(baseA), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(baseB), 0, 0x7fffffff, 0x00020000);
    unsigned char *sb = lds + buf * H2_STAGE;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void *)(sb + lA0 + (j >> 1) * H2_A_PLANE + (j & 1) * (2 * 256 * 16)), 16, voff,
                                               gA0 + (unsigned)(j & 1) * 2u * RS + (unsigned)(j >> 1) * PL, 0, 0);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void *)(sb + lB0 + j * H2_B_PLANE), 16, voff, gB0 + (unsigned)j * PL, 0, 0);
    baseA += step;
    baseB += step;
  };
  const int kg = lane >> 4, fr = lane & 15;
  const unsigned aA = (unsigned)((kg * 256 + half * 128 + wm * 64 + fr) * 16);
  const unsigned aB = (unsigned)(2 * H2_A_PLANE + (kg * 128 + wn * 64 + fr) * 16);
  Acc<float> acc0, acc1;
  acc0.zero();
  acc1.zero();
  auto compute = [&](int buf) {
    const unsigned char *sa = lds + buf * H2_STAGE + aA, *sb = lds + buf * H2_STAGE + aB;
    h2_f16x8 a0[4], a1[4], b0[4], b1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) b0[u] = *reinterpret_cast<const h2_f16x8 *>(sb + u * 256);
#pragma unroll
    for (int u = 0; u < 4; ++u) a1[u] = *reinterpret_cast<const h2_f16x8 *>(sa + H2_A_PLANE + u * 256);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc1.v[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[mt], b0[nt], acc1.v[mt][nt], 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 4; ++u) a0[u] = *reinterpret_cast<const h2_f16x8 *>(sa + u * 256);
#pragma unroll
    for (int u = 0; u < 4; ++u) b1[u] = *reinterpret_cast<const h2_f16x8 *>(sb + H2_B_PLANE + u * 256);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc1.v[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[mt], b1[nt], acc1.v[mt][nt], 0, 0, 0);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc0.v[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0[mt], b0[nt], acc0.v[mt][nt], 0, 0, 0);
  };
  const int nst = K / 32;
  if constexpr (NST == 2) {
    issue(0);
#pragma unroll 1
    for (int s = 0; s < nst; s += 2) {
      __syncthreads();
      issue(1);
      compute(0);
      __syncthreads();
      if (s + 2 < nst) issue(0);
      compute(1);
    }
    __syncthreads();
  } else {
    // three stages: before stage s is read, the pieces of stage s + 1 may still be in flight (6 per wave)
    issue(0);
    issue(1);
    int buf = 0;
#pragma unroll 1
    for (int s = 0; s < nst; ++s) {
      if (s + 1 < nst) asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (s + 2 < nst) issue(buf >= 1 ? buf - 1 : 2);                // buffer of stage s - 1 = (s + 2) % 3
      compute(buf);
      buf = buf == 2 ? 0 : buf + 1;
    }
    __syncthreads();
  }
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc0.v[a][b] = (acc0.v[a][b] + acc1.v[a][b] * (1.0f / 2048.0f)) * inv_scale2;
  const int hf = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  tile_writeback<float, WB_SUB>(acc0, C + ((int64_t)mb * 256 + hf * 128) * ld + (int64_t)jb * NB, ld, reinterpret_cast<float *>(lds + hf * B3_WB_BYTES),
                                threadIdx.x & 255);
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
  // ---- two-plane fp16 prototype: accuracy on a fresh copy of C, then speed
  unsigned short *Ph2 = nullptr;
  float *C2 = nullptr;
  CK(hipMalloc(&Ph2, (size_t)K * 2 * ld * 2));
  CK(hipMalloc(&C2, ce * 4));
  CK(hipMemcpy(C2, hc.data(), ce * 4, hipMemcpyHostToDevice));
  const float h2_scale = 256.0f;                          // |x| <~ 6 here: 6 * 256 = 1536 << 65504
  hipLaunchKernelGGL(k_split_h2, dim3((unsigned)(ld / NB), K / NB), dim3(NTHREADS), 0, 0, P32, ld, Ph2, h2_scale);
  for (int nstv = 2; nstv <= 3; ++nstv) {
    CK(hipMemcpy(C2, hc.data(), ce * 4, hipMemcpyHostToDevice));
    if (nstv == 2) hipLaunchKernelGGL((k_h2_tiles<2>), dim3(TPR / 2 * TPR), dim3(B3_NT), 0, 0, Ph2, C2, ld, K, TPR / 2, TPR, 1.0f / (h2_scale * h2_scale));
    else hipLaunchKernelGGL((k_h2_tiles<3>), dim3(TPR / 2 * TPR), dim3(B3_NT), 0, 0, Ph2, C2, ld, K, TPR / 2, TPR, 1.0f / (h2_scale * h2_scale));
    CK(hipDeviceSynchronize());
    double eh = 0, sh = 0; size_t ch = 0;
    for (auto &tt : tiles) {
      const int ib = tt[0], jb = tt[1];
      CK(hipMemcpy(c1.data(), C2 + (int64_t)ib * NB * ld, (size_t)NB * ld * 4, hipMemcpyDeviceToHost));
      for (int i = 0; i < NB; i += 3)
        for (int j = 0; j < NB; j += 5) {
          double acc = 0.0;
          for (int k = 0; k < K; ++k) acc += (double)hp[(size_t)k * ld + ib * NB + i] * (double)hp[(size_t)k * ld + jb * NB + j];
          const double ref = (double)hc[((size_t)ib * NB + i) * ld + jb * NB + j] - acc;
          const double d1 = c1[(size_t)i * ld + jb * NB + j] - ref;
          eh = fmax(eh, fabs(d1)); sh += d1 * d1; ++ch;
        }
    }
    printf("fp16x2 prototype (%d stages): error vs fp64 max %.3e rms %.3e\n", nstv, eh, sqrt(sh / ch));
  }
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
    if (time_it("fp16x2 prototype, 2 stages", [&]() { hipLaunchKernelGGL((k_h2_tiles<2>), dim3(TPR / 2 * TPR), dim3(B3_NT), 0, 0, Ph2, C2, ld, K, TPR / 2, TPR, 1.0f / (h2_scale * h2_scale)); })) return 1;
    if (time_it("fp16x2 prototype, 3 stages", [&]() { hipLaunchKernelGGL((k_h2_tiles<3>), dim3(TPR / 2 * TPR), dim3(B3_NT), 0, 0, Ph2, C2, ld, K, TPR / 2, TPR, 1.0f / (h2_scale * h2_scale)); })) return 1;
    if (time_it("bf16 engine, XCD-local 4 x 8 blocks", [&]() { hipLaunchKernelGGL((k_bf3v2_tiles<2>), dim3(8 * ((TPR / 8 * TPR / 8 + 7) / 8) * 32), dim3(B3_NT), 0, 0, Pl, C1, ld, K, TPR / 2, TPR); })) return 1;
  }
  return 0;
}
