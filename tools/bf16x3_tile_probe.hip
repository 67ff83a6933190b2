// bf16x3_tile_probe.hip -- DESIGN.md 9.1: can the bf16 matrix cores carry the fp32 tile products?
//
// C[tile] -= A^T B (K-major fp32 operands, depth 1024, a 48 x 48 grid of 128 x 128 tiles -- the shape of the sweep's
// trailing update) computed (a) by the fp32 engine of the library (v_mfma_f32_16x16x4_f32) and (b) from operands split
// ONCE into three bf16 planes x = hi + mid + lo (24 significand bits) with six plane products per tile on
// v_mfma_f32_16x16x32_bf16, fp32 accumulate: hi.hi + hi.mid + mid.hi + mid.mid + hi.lo + lo.hi (the dropped terms are
// below 2^-24 of the product).  The planes keep the K-major layout of the operands; LDS holds them K-major too and the
// MFMA fragments (8 consecutive k per lane) come out of ds_read_b64_tr_b16.  Reports time per launch, TFLOP/s on
// 2 * 128^2 * K per tile, and the error of both results against an fp64 host product on sample tiles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/bf16x3_tile_probe tools/bf16x3_tile_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <math.h>
#include <vector>
#include "../projected-lmc_amd/csrc/gemm_core.hpp"

using namespace plmc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(NTHREADS, 1) void k_f32_tiles(float *M, int64_t ld, int K, int tiles_per_row) {
  __shared__ __align__(16) float smem[tile_smem_elems<float>()];
  const int t = blockIdx.x, ib = t / tiles_per_row, jb = t % tiles_per_row;
  Acc<float> acc;
  acc.zero();
  tile_mainloop<float, false, false>(acc, M + (int64_t)ib * NB, ld, M + (int64_t)jb * NB, ld, K, smem);
  tile_writeback<float, WB_SUB>(acc, M + ((int64_t)K + (int64_t)ib * NB) * ld + (int64_t)jb * NB, ld, smem);
}

// x -> three bf16 planes (round to nearest even at every level; the residuals are exact in fp32)
__global__ __launch_bounds__(256) void k_split(const float *__restrict__ X, int64_t ld, int rows, int cols, unsigned short *__restrict__ P,
                                               int64_t ldp, int64_t plane_stride) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i % cols);
  const float x = X[(int64_t)r * ld + c];
  const __bf16 h = (__bf16)x;
  const float r1 = x - (float)h;
  const __bf16 m = (__bf16)r1;
  const float r2 = r1 - (float)m;
  const __bf16 l = (__bf16)r2;
  unsigned short *o = P + (int64_t)r * ldp + c;
  o[0] = __builtin_bit_cast(unsigned short, h);
  o[plane_stride] = __builtin_bit_cast(unsigned short, m);
  o[2 * plane_stride] = __builtin_bit_cast(unsigned short, l);
}

constexpr int BKH = 32;                     // k rows per LDS stage
constexpr int ROWB = 256;                   // bytes per LDS row (128 bf16), 32-byte blocks XOR-swizzled by (row & 7)
constexpr int PLANE_LDS = BKH * ROWB;       // 8 KB per plane and operand

// LDS row of contraction row k inside a stage: bits 2 and 3 of k swapped, so that the eight rows one 32-lane half reads
// together (k = 8g + 4j + q for lane groups g = 0, 1) are eight consecutive LDS rows -> eight different 32-byte slots
__device__ __forceinline__ int lds_row(int k) { return (k & 19) | ((k >> 1) & 4) | ((k << 1) & 8); }

// (a 3-waves-per-SIMD register bound spills 220 bytes per lane: 48 staging + 64 accumulator + 48 fragment registers)
template <int NPROD, int ABL = 0>
__global__ __launch_bounds__(NTHREADS, 1) void k_bf3_tiles(const unsigned short *__restrict__ P, int64_t ldp, int64_t plane_stride, float *M,
                                                           int64_t ld, int K, int tiles_per_row) {
  __shared__ __align__(16) unsigned char lds[6 * PLANE_LDS];              // [A hi, mid, lo | B hi, mid, lo][32 rows][256 B]
  const int t = blockIdx.x, ib = t / tiles_per_row, jb = t % tiles_per_row;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  // global -> registers: per plane and operand 512 chunks of 16 bytes per stage, two per thread (rows r and r + 16)
  const int lrow = tid >> 4, lch = tid & 15;
  const unsigned short *gA = P + (int64_t)lrow * ldp + (int64_t)ib * NB + lch * 8;
  const unsigned short *gB = P + (int64_t)lrow * ldp + (int64_t)jb * NB + lch * 8;
  i32x4 ra[3][2], rb[3][2];
  auto gload = [&](int s) {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int64_t off = (int64_t)p * plane_stride + ((int64_t)s * BKH + h * 16) * ldp;
        ra[p][h] = *reinterpret_cast<const i32x4 *>(gA + off);
        rb[p][h] = *reinterpret_cast<const i32x4 *>(gB + off);
      }
  };
  int woff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int R = lds_row(lrow + 16 * h);
    woff[h] = R * ROWB + ((((lch >> 1) ^ (R & 7)) << 5) | ((lch & 1) << 4));
  }
  auto sstore = [&]() {
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        *reinterpret_cast<i32x4 *>(lds + p * PLANE_LDS + woff[h]) = ra[p][h];
        *reinterpret_cast<i32x4 *>(lds + (3 + p) * PLANE_LDS + woff[h]) = rb[p][h];
      }
  };
  // fragment addresses: lane group g = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3; read j (0, 1) takes k = 8 g + 4 j + q
  const int g = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
  int roff[2];                                                            // byte offset of the row part, per read
#pragma unroll
  for (int j = 0; j < 2; ++j) roff[j] = ((g >> 1) * 16 + j * 8 + (g & 1) * 4 + q) * ROWB;
  const int rsw = (g & 1) * 4 + q;                                        // (LDS row & 7), the same for both reads
  auto frag = [&](int plane_op, int blk) -> bf16x8 {                       // blk = 16-column block inside the tile's 128
    const int col = ((blk ^ rsw) << 5) + p4 * 8;
    const unsigned char *b = lds + plane_op * PLANE_LDS + col;
    s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(b + roff[0]));
    s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(b + roff[1]));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, v);
  };
  Acc<float> acc;
  acc.zero();
  const int nst = K / BKH;
  gload(0);
  for (int s = 0; s < nst; ++s) {
    if (!(ABL & 2) || s == 0) sstore();
    __syncthreads();
    if (!(ABL & 1) && s + 1 < nst) gload(s + 1);
    if constexpr (NPROD == 66) {
      // six products with every fragment read once per role: B.hi stays live, A planes walk lo -> mid -> hi
      bf16x8 a[4], b[4], bh[4];
      auto mm = [&](bf16x8 (&x)[4], bf16x8 (&y)[4]) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc.v[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[mt], y[nt], acc.v[mt][nt], 0, 0, 0);
      };
#pragma unroll
      for (int i = 0; i < 4; ++i) bh[i] = frag(3 + 0, wn * 4 + i);
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = frag(2, wm * 4 + i);      // A.lo
      mm(a, bh);
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = frag(1, wm * 4 + i);      // A.mid
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = frag(3 + 1, wn * 4 + i);  // B.mid
      mm(a, b);
      mm(a, bh);
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = frag(0, wm * 4 + i);      // A.hi
      mm(a, b);
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = frag(3 + 2, wn * 4 + i);  // B.lo
      mm(a, b);
      mm(a, bh);
    } else {
    // plane products, small terms first: (A plane, B plane)
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int pr = 6 - (NPROD > 6 ? 6 : NPROD); pr < 6; ++pr) {
      bf16x8 a[4], b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) a[i] = frag(PA[pr], wm * 4 + i);
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = frag(3 + PB[pr], wn * 4 + i);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc.v[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt], b[nt], acc.v[mt][nt], 0, 0, 0);
    }
    }
    __syncthreads();
  }
  tile_writeback<float, WB_SUB>(acc, M + ((int64_t)K + (int64_t)ib * NB) * ld + (int64_t)jb * NB, ld, reinterpret_cast<float *>(lds));
}

// Second form: 512 threads = 8 waves (2 x 4, 64 x 32 per wave), TWO LDS stages (96 KB, one workgroup per CU), one barrier
// per stage: the stores of stage s + 1 go to the other buffer at the top of iteration s and overlap the MFMAs of the
// other waves; 24 staging + 32 accumulator registers per lane.  Epilogue: the whole tile through LDS, coalesced
// read-modify-write with 16-byte accesses.
__global__ __launch_bounds__(512, 2) void k_bf3_tiles8(const unsigned short *__restrict__ P, int64_t ldp, int64_t plane_stride, float *M,
                                                       int64_t ld, int K, int tiles_per_row) {
  extern __shared__ __align__(16) unsigned char lds[];                    // 2 x 6 x PLANE_LDS
  const int t = blockIdx.x, ib = t / tiles_per_row, jb = t % tiles_per_row;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
  const int lrow = tid >> 4, lch = tid & 15;                              // one 16-byte chunk per plane and operand
  const unsigned short *gA = P + (int64_t)lrow * ldp + (int64_t)ib * NB + lch * 8;
  const unsigned short *gB = P + (int64_t)lrow * ldp + (int64_t)jb * NB + lch * 8;
  i32x4 ra[3], rb[3];
  auto gload = [&](int s) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const int64_t off = (int64_t)p * plane_stride + (int64_t)s * BKH * ldp;
      ra[p] = *reinterpret_cast<const i32x4 *>(gA + off);
      rb[p] = *reinterpret_cast<const i32x4 *>(gB + off);
    }
  };
  const int R = lds_row(lrow);
  const int woff = R * ROWB + ((((lch >> 1) ^ (R & 7)) << 5) | ((lch & 1) << 4));
  auto sstore = [&](int buf) {
    unsigned char *b = lds + buf * 6 * PLANE_LDS + woff;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      *reinterpret_cast<i32x4 *>(b + p * PLANE_LDS) = ra[p];
      *reinterpret_cast<i32x4 *>(b + (3 + p) * PLANE_LDS) = rb[p];
    }
  };
  const int g = lane >> 4, q = (lane & 15) >> 2, p4 = lane & 3;
  const int rsw = (g & 1) * 4 + q;
  const int r0 = ((g >> 1) * 16 + (g & 1) * 4 + q) * ROWB + p4 * 8, r1 = r0 + 8 * ROWB;
  f32x4 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int nst = K / BKH;
  gload(0);
  sstore(0);
  __syncthreads();
  if (nst > 1) gload(1);
  for (int s = 0; s < nst; ++s) {
    const unsigned char *cur = lds + (s & 1) * 6 * PLANE_LDS;
    if (s + 1 < nst) sstore((s + 1) & 1);
    if (s + 2 < nst) gload(s + 2);
    auto frag = [&](int plane_op, int blk) -> bf16x8 {
      const unsigned char *b = cur + plane_op * PLANE_LDS + ((blk ^ rsw) << 5);
      s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(b + r0));
      s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)(b + r1));
      typedef short s16x8 __attribute__((ext_vector_type(8)));
      return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    bf16x8 a[4], b[2], bh[2];
    auto mm = [&](bf16x8 (&x)[4], bf16x8 (&y)[2]) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[mt], y[nt], acc[mt][nt], 0, 0, 0);
    };
#pragma unroll
    for (int i = 0; i < 2; ++i) bh[i] = frag(3, wn * 2 + i);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = frag(2, wm * 4 + i);
    mm(a, bh);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = frag(1, wm * 4 + i);
#pragma unroll
    for (int i = 0; i < 2; ++i) b[i] = frag(4, wn * 2 + i);
    mm(a, b);
    mm(a, bh);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = frag(0, wm * 4 + i);
    mm(a, b);
#pragma unroll
    for (int i = 0; i < 2; ++i) b[i] = frag(5, wn * 2 + i);
    mm(a, b);
    mm(a, bh);
    __syncthreads();
  }
  // epilogue: tile -> LDS ([128][132] floats), then C -= tile with 16-byte accesses, a full 512-byte row per 32 lanes
  float *stg = reinterpret_cast<float *>(lds);
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        stg[(wm * 64 + mt * 16 + (lane >> 4) * 4 + r) * 132 + wn * 32 + nt * 16 + (lane & 15)] = acc[mt][nt][r];
  __syncthreads();
  float *C = M + ((int64_t)K + (int64_t)ib * NB) * ld + (int64_t)jb * NB;
  const int crow = tid >> 5, ccol = (tid & 31) * 4;
  float4 cv[8];
#pragma unroll
  for (int h = 0; h < 8; ++h) cv[h] = *reinterpret_cast<const float4 *>(C + (int64_t)(crow + 16 * h) * ld + ccol);
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    const float4 sv = *reinterpret_cast<const float4 *>(stg + (crow + 16 * h) * 132 + ccol);
    float4 o = {cv[h].x - sv.x, cv[h].y - sv.y, cv[h].z - sv.z, cv[h].w - sv.w};
    *reinterpret_cast<float4 *>(C + (int64_t)(crow + 16 * h) * ld + ccol) = o;
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
  unsigned short *P = nullptr;
  const int64_t ldp = ld, plane_stride = (int64_t)K * ldp;
  CK(hipMalloc(&M0, elems * 4));
  CK(hipMalloc(&M1, elems * 4));
  CK(hipMalloc(&P, 3 * plane_stride * 2));
  CK(hipMemcpy(M0, h.data(), elems * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(M1, h.data(), elems * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  float ms = 0.f;
  const double fl = 2.0 * NB * NB * (double)K * TPR * TPR;
  // split (timed once: it is paid once per panel, not per tile)
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_split, dim3((unsigned)(((int64_t)K * (TPR * NB) + 255) / 256)), dim3(256), 0, 0, M1, ld, K, TPR * NB, P, ldp, plane_stride);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("split of a %d x %d panel into 3 bf16 planes: %.1f us\n", K, TPR * NB, 1e3 * ms);
  // one launch each for the accuracy comparison (C = C0 - A^T B)
  hipLaunchKernelGGL(k_f32_tiles, dim3(TPR * TPR), dim3(NTHREADS), 0, 0, M0, ld, K, TPR);
  hipLaunchKernelGGL((k_bf3_tiles<6>), dim3(TPR * TPR), dim3(NTHREADS), 0, 0, P, ldp, plane_stride, M1, ld, K, TPR);
  CK(hipDeviceSynchronize());
  std::vector<float> c0((size_t)NB * ld), c1((size_t)NB * ld);
  double e32 = 0, ebf = 0, scale = 0;
  const int tiles[4][2] = {{0, 0}, {3, 17}, {47, 47}, {20, 5}};
  for (auto &tt : tiles) {
    const int ib = tt[0], jb = tt[1];
    CK(hipMemcpy(c0.data(), M0 + ((int64_t)K + ib * NB) * ld, (size_t)NB * ld * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(c1.data(), M1 + ((int64_t)K + ib * NB) * ld, (size_t)NB * ld * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < NB; i += 5)
      for (int j = 0; j < NB; j += 3) {
        double ref = h[((size_t)K + ib * NB + i) * ld + jb * NB + j];
        for (int k = 0; k < K; ++k) ref -= (double)h[(size_t)k * ld + ib * NB + i] * (double)h[(size_t)k * ld + jb * NB + j];
        e32 = fmax(e32, fabs(c0[(size_t)i * ld + jb * NB + j] - ref));
        ebf = fmax(ebf, fabs(c1[(size_t)i * ld + jb * NB + j] - ref));
        scale = fmax(scale, fabs(ref));
      }
  }
  printf("max |error| vs fp64 on sample tiles (|C| up to %.1f): fp32 engine %.3e, bf16x3 (6 products) %.3e\n", scale, e32, ebf);
  auto time_it = [&](const char *name, auto launch) -> int {
    for (int w = 0; w < 2; ++w) launch();
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-34s %8.3f ms per launch  %7.1f TFLOP/s (fp32-equivalent)\n", name, ms / 5, fl / (ms / 5 * 1e-3) / 1e12);
    return 0;
  };
  if (time_it("fp32 engine (16x16x4 f32)", [&]() { hipLaunchKernelGGL(k_f32_tiles, dim3(TPR * TPR), dim3(NTHREADS), 0, 0, M0, ld, K, TPR); })) return 1;
  if (time_it("bf16x3, 6 plane products", [&]() { hipLaunchKernelGGL((k_bf3_tiles<6>), dim3(TPR * TPR), dim3(NTHREADS), 0, 0, P, ldp, plane_stride, M1, ld, K, TPR); })) return 1;
  if (time_it("bf16x3, 6 products, fragments reused", [&]() { hipLaunchKernelGGL((k_bf3_tiles<66>), dim3(TPR * TPR), dim3(NTHREADS), 0, 0, P, ldp, plane_stride, M1, ld, K, TPR); })) return 1;
  CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bf3_tiles8), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 6 * PLANE_LDS));
  {  // accuracy of the 8-wave form on a fresh copy of C
    CK(hipMemcpy(M1, h.data(), elems * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_bf3_tiles8, dim3(TPR * TPR), dim3(512), 2 * 6 * PLANE_LDS, 0, P, ldp, plane_stride, M1, ld, K, TPR);
    CK(hipDeviceSynchronize());
    double e8 = 0;
    for (auto &tt : tiles) {
      const int ib = tt[0], jb = tt[1];
      CK(hipMemcpy(c1.data(), M1 + ((int64_t)K + ib * NB) * ld, (size_t)NB * ld * 4, hipMemcpyDeviceToHost));
      for (int i = 0; i < NB; i += 5)
        for (int j = 0; j < NB; j += 3) {
          double ref = h[((size_t)K + ib * NB + i) * ld + jb * NB + j];
          for (int k = 0; k < K; ++k) ref -= (double)h[(size_t)k * ld + ib * NB + i] * (double)h[(size_t)k * ld + jb * NB + j];
          e8 = fmax(e8, fabs(c1[(size_t)i * ld + jb * NB + j] - ref));
        }
    }
    printf("8-wave double-buffered form: max |error| vs fp64 %.3e\n", e8);
  }
  if (time_it("bf16x3, 8 waves, 2 LDS stages", [&]() { hipLaunchKernelGGL(k_bf3_tiles8, dim3(TPR * TPR), dim3(512), 2 * 6 * PLANE_LDS, 0, P, ldp, plane_stride, M1, ld, K, TPR); })) return 1;
  if (time_it("  ablation: no global loads in the loop", [&]() { hipLaunchKernelGGL((k_bf3_tiles<66, 1>), dim3(TPR * TPR), dim3(NTHREADS), 0, 0, P, ldp, plane_stride, M1, ld, K, TPR); })) return 1;
  if (time_it("  ablation: no loads, no LDS stores", [&]() { hipLaunchKernelGGL((k_bf3_tiles<66, 3>), dim3(TPR * TPR), dim3(NTHREADS), 0, 0, P, ldp, plane_stride, M1, ld, K, TPR); })) return 1;
  if (time_it("bf16x3, 3 products (speed only)", [&]() { hipLaunchKernelGGL((k_bf3_tiles<3>), dim3(TPR * TPR), dim3(NTHREADS), 0, 0, P, ldp, plane_stride, M1, ld, K, TPR); })) return 1;
  if (time_it("bf16, 1 product (speed only)", [&]() { hipLaunchKernelGGL((k_bf3_tiles<1>), dim3(TPR * TPR), dim3(NTHREADS), 0, 0, P, ldp, plane_stride, M1, ld, K, TPR); })) return 1;
  return 0;
}
