// bf3_engine.hpp -- fp32 tile products on the 16-bit matrix cores: the "split engine" (round 3).
//
// What it computes.  C[i][j] (+)= sum_k A[k][i] B[k][j] for fp32 operands that were split ONCE into 16-bit planes, as a few
// plane products on v_mfma_f32_16x16x32_{bf16,f16} with fp32 accumulation into TWO accumulator levels (level 0: the
// leading product, level 1: everything 2^-8 / 2^-11 of it and below), combined once at the end.  Two split schemes:
//   SplitB3  x = hi + mid + lo, three bf16 planes (exact: all 24 significand bits), SIX plane products.  bf16 has the fp32
//            exponent range: works for any data, no scaling.
//   SplitH2  s x = h0 + 2^-11 h1, two fp16 planes (22 significand bits), THREE plane products: h0.h0 | h1.h0 + h0.h1.
//            fp16 has a 5-bit exponent, so every operand FAMILY is scaled by a power of two s that puts a rigorous bound of
//            its magnitude at 2^13 (potrf.hip derives the bounds from the largest diagonal entry and a lower bound of the
//            smallest eigenvalue, i.e. the noise); the product is unscaled in the epilogue (exact).
// tools/split_numerics_probe.hip (profiles/r03_split_numerics.txt) is why: against fp64 either product has 0.3-0.45 x the
// error of the v_mfma_f32_16x16x4_f32 chain of the fp32 engine (K = 1024 and 8192; normal, one-signed, wide-range data and
// the cancelling product of the group panel) -- the fp32 chain rounds its running sum K times, the split engines K / 32
// times, and that rounding, not the operand precision, is what the error consists of: eight or nine bf16 plane products
// or a third fp16 plane give the same digits.  ONE accumulator for all products (round 2's opt-in) carries 2.5 x the
// error of two levels (the small products are rounded against the large running sum).
//
// How.  A workgroup of 512 threads = 8 waves owns a 256 x 128 macro tile (two 128 x 128 tiles of one block column: the
// rows ib, ib + 1 share the B strip), each wave a 64 x 64 block of it as 4 x 4 MFMA tiles -- the accumulator layout of the
// fp32 engine (gemm_core.hpp), so tile_writeback and the gradient epilogues are reused per half.  Operand planes live in
// HBM in "k8" order,
//     P[k / 8][plane][column][k % 8]        (16 bytes = the 8 contraction values one MFMA lane needs),
// so that (a) a stage of 32 contraction rows of a tile is a few runs of 2-4 KB, copied to LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds: no staging registers, no ds_write, one 1 KB piece per wave instruction) and (b) an MFMA
// fragment is ONE conflict-free ds_read_b128 per lane (lanes 16 g .. 16 g + 15 read 256 contiguous bytes of k-group g):
// no transposing reads, no swizzle.  Two LDS stages (SplitB3: 72 KB each, SplitH2: 48 KB); one barrier per stage: the DMA
// of stage s + 1 is issued right after the barrier that ends the reads of stage s - 1 and lands while stage s is multiplied.
// Alone on the GPU (tools/bf3v2_probe.hip, depth-1024 update tiles): fp32 engine 127 TF, SplitB3 223 TF, SplitH2 353 TF
// fp32-equivalent.
#pragma once
#include "gemm_core.hpp"

namespace plmc {

constexpr int B3_NT = 512;                                 // threads per workgroup
constexpr int B3_K = 32;                                   // contraction rows per LDS stage
constexpr int B3_AW = 256, B3_BW = 128;                    // macro tile: A side (rows of C) x B side (columns of C)
constexpr int B3_A_PLANE = (B3_K / 8) * B3_AW * 16;        // bytes of one A plane of a stage: [4 k-groups][256][16 B]
constexpr int B3_B_PLANE = (B3_K / 8) * B3_BW * 16;
constexpr int B3_WB_BYTES = 64 * 132 * 4;                  // staging of tile_writeback, per 128 x 128 half

typedef __bf16 b3_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 b3_f16x8 __attribute__((ext_vector_type(8)));
typedef short b3_s16x8 __attribute__((ext_vector_type(8)));

struct SplitB3 {
  static constexpr int NPL = 3;
  static constexpr float LEVEL1 = 1.0f;                    // weight of the level-1 accumulator in the final sum
  typedef b3_bf16x8 frag_t;
  static __device__ __forceinline__ f32x4 mfma(frag_t a, frag_t b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ void split(float v, short (&o)[3]) {
    const __bf16 a = (__bf16)v;
    const float r1 = v - (float)a;
    const __bf16 b = (__bf16)r1;
    const __bf16 c = (__bf16)(r1 - (float)b);
    o[0] = __builtin_bit_cast(short, a); o[1] = __builtin_bit_cast(short, b); o[2] = __builtin_bit_cast(short, c);
  }
};
struct SplitH2 {
  static constexpr int NPL = 2;
  static constexpr float LEVEL1 = 1.0f / 2048.0f;
  typedef b3_f16x8 frag_t;
  static __device__ __forceinline__ f32x4 mfma(frag_t a, frag_t b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ void split(float v, short (&o)[2]) {      // v is already scaled into the fp16 range
    const _Float16 a = (_Float16)v;
    const _Float16 b = (_Float16)((v - (float)a) * 2048.0f);
    o[0] = __builtin_bit_cast(short, a); o[1] = __builtin_bit_cast(short, b);
  }
};
template <class S> constexpr int b3_stage_bytes() { return S::NPL * (B3_A_PLANE + B3_B_PLANE); }    // 73 728 / 49 152
template <class S> constexpr int b3_lds_bytes() { return 2 * b3_stage_bytes<S>() > 2 * B3_WB_BYTES ? 2 * b3_stage_bytes<S>() : 2 * B3_WB_BYTES; }

// element (k, plane, column) of a k8-ordered plane buffer with `ld` columns, in 16-bit elements
template <class S> __host__ __device__ inline int64_t b3_index(int64_t k, int plane, int64_t col, int64_t ld) {
  return (((k >> 3) * S::NPL + plane) * ld + col) * 8 + (k & 7);
}
// 16-bit elements of a plane buffer of `rows` contraction rows (a multiple of 8) and `ld` columns
template <class S> __host__ __device__ inline int64_t b3_elems(int64_t rows, int64_t ld) { return rows * S::NPL * ld; }

// Power-of-two scale that puts `bound` (> 0) at 2^13: fp16 overflows at 2^16, so a bound that fp32 rounding exceeds by
// less than 8 x is still safe; values down to 2^-27 of the bound keep their 22 bits.
__host__ __device__ inline float b3_scale_for(float bound) {
  int e;
  (void)frexpf(bound, &e);                                  // bound = f 2^e, f in [0.5, 1)
  return ldexpf(1.0f, 13 - e);
}

// Eight contraction rows x four columns of fp32 -> the planes' 16-byte groups (one per column and plane).
// x[r][c]: row r (0..7) of the k8 group, column c (0..3); every value is multiplied by `scale` first (SplitB3: 1).
// `dst` = &P[k8][0][col0][0]; plane stride = ld * 8 elements.
template <class S>
__device__ __forceinline__ void b3_split_store(const float (&x)[8][4], unsigned short *dst, int64_t ld, float scale) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    b3_s16x8 pl[S::NPL];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      short o[S::NPL];
      S::split(x[r][c] * scale, o);
#pragma unroll
      for (int p = 0; p < S::NPL; ++p) pl[p][r] = o[p];
    }
#pragma unroll
    for (int p = 0; p < S::NPL; ++p) *reinterpret_cast<b3_s16x8 *>(dst + (int64_t)p * ld * 8 + c * 8) = pl[p];
  }
}

// One 128 x 128 fp32 block (rows = contraction index, leading dimension lds_) -> planes.  256 threads; thread = one
// k8 group x four columns per pass (8 loads of 16 bytes, 4 NPL stores of 16 bytes; whole 512-byte rows / 64-byte runs).
// `P` points at element (k = first row of the block, plane 0, first column of the block).  COPY: also write the block
// to D (leading dimension ldd).
template <class S, bool COPY>
__device__ __forceinline__ void b3_split_block(const float *__restrict__ Sp, int64_t lds_, unsigned short *__restrict__ P, int64_t ld, float scale,
                                               float *D, int64_t ldd, int tid) {
#pragma unroll 1
  for (int w = tid; w < 16 * 32; w += NTHREADS) {
    const int k8 = w >> 5, c4 = (w & 31) * 4;
    float x[8][4];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float4 v = *reinterpret_cast<const float4 *>(Sp + (int64_t)(k8 * 8 + r) * lds_ + c4);
      x[r][0] = v.x; x[r][1] = v.y; x[r][2] = v.z; x[r][3] = v.w;
      if (COPY) *reinterpret_cast<float4 *>(D + (int64_t)(k8 * 8 + r) * ldd + c4) = v;
    }
    b3_split_store<S>(x, P + ((int64_t)k8 * S::NPL * ld + c4) * 8, ld, scale);
  }
}

// acc0 / acc1 += the two levels of  sum_{k < K} A[k][a-columns]^T B[k][b-columns]  for this wave's 64 x 64 block.
//   Ap: plane buffer at (k = first row of the K range, plane 0, first of the 256 A columns), lda_ columns per plane row;
//   Bp: likewise, first of the 128 B columns, ldb_ columns;  K % 64 == 0;  lds: b3_lds_bytes<S>(), 16-byte aligned, the
//   kernel's ONLY __shared__ object (a second one makes hipcc drain the DMA before every fragment read).
// All 512 threads must call it; ends with a barrier (LDS free for the epilogue).
// `pre` (optional): called once, right behind the LAST DMA issue of the loop, two stages before its end -- the place for the
// epilogue's first C loads (b3_preload): NPRE = the vector-memory loads it issues (0 or 8).  They are the youngest memory
// operations of the wave, so the remaining barriers wait for "all but the NPRE youngest" and the loads stay in flight
// through the last two stages (an older load would hold back every DMA piece behind it: vmcnt counts in issue order).
struct B3NoPre { __device__ __forceinline__ void operator()() const {} };
// REV: walk the K range from its last stage to its first -- tiles whose ranges END at the same row (the K^-1 tiles: every
// range ends at n) then read the same operand rows at the same time, whatever their length.
template <class S, int NSTG = 2, int NPRE = 0, class PRE = B3NoPre, bool REV = false>
__device__ __forceinline__ void b3_mainloop(Acc<float> &acc0, Acc<float> &acc1, const unsigned short *__restrict__ Ap, int64_t lda_,
                                            const unsigned short *__restrict__ Bp, int64_t ldb_, int K, unsigned char *lds, PRE pre = PRE()) {
  static_assert(NPRE == 0 || (NPRE == 8 && NSTG == 2), "pre-loads: eight, two-stage loop only");
  constexpr int NPL = S::NPL, STAGE = b3_stage_bytes<S>();
  typedef typename S::frag_t frag_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave 0..7, provably uniform (DMA destinations are scalar)
  const int half = w >> 2, wm = (w >> 1) & 1, wn = w & 1;
  // ---- DMA plan of this wave: per plane 16 A pieces (k-group, 64-column segment) and 8 B pieces, i.e. per wave and plane
  // two A pieces (k-groups (w >> 2) and (w >> 2) + 2, columns 64 (w & 3) ..) and one B piece (k-group (w >> 1) & 3,
  // columns 64 (w & 1) ..)
  const unsigned RSA = (unsigned)(NPL * lda_ * 16), RSB = (unsigned)(NPL * ldb_ * 16);   // bytes per k8 row (all planes)
  const unsigned PLA = (unsigned)(lda_ * 16), PLB = (unsigned)(ldb_ * 16);               // bytes per plane inside a k8 row
  const unsigned gA0 = (unsigned)(w >> 2) * RSA + (unsigned)(w & 3) * 1024u;
  const unsigned gB0 = (unsigned)((w >> 1) & 3) * RSB + (unsigned)(w & 1) * 1024u;
  const unsigned lA0 = (unsigned)(((w >> 2) * B3_AW + (w & 3) * 64) * 16);
  const unsigned lB0 = (unsigned)(NPL * B3_A_PLANE + (((w >> 1) & 3) * B3_BW + (w & 1) * 64) * 16);
  const unsigned voff = (unsigned)lane * 16u;
  const int64_t fwdA = (int64_t)(B3_K / 8) * RSA, fwdB = (int64_t)(B3_K / 8) * RSB;
  const int64_t stepA = REV ? -fwdA : fwdA, stepB = REV ? -fwdB : fwdB;
  const char *baseA = reinterpret_cast<const char *>(Ap) + (REV ? (int64_t)(K / B3_K - 1) * fwdA : 0);
  const char *baseB = reinterpret_cast<const char *>(Bp) + (REV ? (int64_t)(K / B3_K - 1) * fwdB : 0);
  typedef __attribute__((address_space(3))) void lds_void;
  auto issue = [&](int buf) {
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(baseA), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(baseB), 0, 0x7fffffff, 0x00020000);
    unsigned char *sb = lds + buf * STAGE;
#pragma unroll
    for (int j = 0; j < 2 * NPL; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void *)(sb + lA0 + (j >> 1) * B3_A_PLANE + (j & 1) * (2 * B3_AW * 16)), 16, voff,
                                               gA0 + (unsigned)(j & 1) * 2u * RSA + (unsigned)(j >> 1) * PLA, 0, 0);
#pragma unroll
    for (int j = 0; j < NPL; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void *)(sb + lB0 + j * B3_B_PLANE), 16, voff, gB0 + (unsigned)j * PLB, 0, 0);
    baseA += stepA;
    baseB += stepB;
  };
  // ---- fragment addresses: lane (kg = lane >> 4, fr = lane & 15) reads 16 bytes of k-group kg, row / column .. + fr
  const int kg = lane >> 4, fr = lane & 15;
  const unsigned aA = (unsigned)((kg * B3_AW + half * 128 + wm * 64 + fr) * 16);
  const unsigned aB = (unsigned)(NPL * B3_A_PLANE + (kg * B3_BW + wn * 64 + fr) * 16);
  auto compute = [&](int buf) {
    const unsigned char *sa = lds + buf * STAGE + aA, *sb = lds + buf * STAGE + aB;
    auto fa = [&](int plane, frag_t (&f)[4]) {
#pragma unroll
      for (int t = 0; t < 4; ++t) f[t] = *reinterpret_cast<const frag_t *>(sa + plane * B3_A_PLANE + t * 256);
    };
    auto fb = [&](int plane, frag_t (&f)[4]) {
#pragma unroll
      for (int t = 0; t < 4; ++t) f[t] = *reinterpret_cast<const frag_t *>(sb + plane * B3_B_PLANE + t * 256);
    };
    auto mm = [&](Acc<float> &acc, const frag_t (&x)[4], const frag_t (&y)[4]) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc.v[mt][nt] = S::mfma(x[mt], y[nt], acc.v[mt][nt]);
    };
    // small products first; every fragment is read once
    if constexpr (NPL == 3) {
      frag_t bh[4], bx[4], ax[4], ay[4];
      fb(0, bh);
      fa(2, ax);
      mm(acc1, ax, bh);               // lo . hi
      fa(1, ay);
      mm(acc1, ay, bh);               // mid . hi
      fb(1, bx);
      mm(acc1, ay, bx);               // mid . mid
      fa(0, ax);
      mm(acc1, ax, bx);               // hi . mid
      fb(2, bx);
      mm(acc1, ax, bx);               // hi . lo
      mm(acc0, ax, bh);               // hi . hi
    } else {
      frag_t b0[4], b1[4], a0[4], a1[4];
      fb(0, b0);
      fa(1, a1);
      mm(acc1, a1, b0);               // h1 . h0
      fa(0, a0);
      fb(1, b1);
      mm(acc1, a0, b1);               // h0 . h1
      mm(acc0, a0, b0);               // h0 . h0
    }
  };
  const int nst = K / B3_K;                                        // even
  if constexpr (NSTG == 2) {
    issue(0);
#pragma unroll 1
    for (int s = 0; s + 2 < nst; s += 2) {
      __syncthreads();               // (vmcnt(0) + barrier) stage s has landed for every wave; stage s - 1 is read out
      issue(1);                      // nst is even: stage s + 1 always exists
      compute(0);
      __syncthreads();
      issue(0);
      compute(1);
    }
    __syncthreads();                 // the last two stages
    issue(1);
    pre();
    compute(0);
    if constexpr (NPRE == 0) {
      __syncthreads();
      compute(1);
    } else {
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");             // the last stage has landed; the pre-loads may still fly
      __builtin_amdgcn_s_barrier();
      compute(1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                // LDS free for the epilogue (no vmcnt wait)
      return;
    }
  } else {
    // three stages: the DMA of stage s + 2 is issued behind the barrier that ends the reads of stage s - 1, so a stage has
    // two stage times to land (fabric / HBM latency under load is longer than one); before stage s is read only ITS pieces
    // must have landed: the 3 NPL younger ones of stage s + 1 may stay in flight (vmcnt counts in issue order)
    static_assert(NSTG == 3 && 3 * STAGE <= 160 * 1024, "three stages must fit the LDS");
    issue(0);
    issue(1);
    int buf = 0;
#pragma unroll 1
    for (int s = 0; s < nst; ++s) {
      if (s + 1 < nst) {
        if constexpr (NPL == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      if (s + 2 < nst) issue(buf == 0 ? 2 : buf - 1);               // (s + 2) % 3 = the buffer of stage s - 1
      compute(buf);
      buf = buf == 2 ? 0 : buf + 1;
    }
  }
  __syncthreads();
}

// acc0 <- (acc0 + LEVEL1 acc1) * unscale   (unscale = 1 / (scale of the A family x scale of the B family); SplitB3: 1)
template <class S> __device__ __forceinline__ void b3_combine(Acc<float> &acc0, const Acc<float> &acc1, float unscale) {
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      if constexpr (S::NPL == 3) acc0.v[a][b] += acc1.v[a][b];
      else acc0.v[a][b] = (acc0.v[a][b] + acc1.v[a][b] * S::LEVEL1) * unscale;
    }
}

// Epilogue of one 128 x 128 half of the macro tile (tid = threadIdx.x & 255): the tile goes through the half's staging area
// in two passes of 64 rows (pass h staged by the wave row wm == h) and leaves as 16-byte row chunks, 8 per thread and pass
// (the layout of tile_writeback<float>, gemm_core.hpp).  One workgroup per CU means nothing else covers the latency of the C
// loads of the read-modify-write modes, so ALL 16 chunks of a thread are in flight at once: those of pass 0 from
// b3_preload (issued inside the main loop, PRE) or at entry, those of pass 1 at entry (tile_writeback keeps two in flight:
// 8 dependent round trips per tile, 9 us of the 55 a depth-1024 tile took; tools/engine_rate_probe.hip).
// With PLANES the FINAL values (times `pscale`) are also written as k8-ordered planes -- the next consumer's operand, while
// the tile is still in LDS instead of by a separate split pass: `Pp` = plane buffer at (k = the tile's first row, plane 0,
// the tile's first column), `pld` its columns; per pass the finals go back into the staging area, then every thread splits
// 8 rows x 4 columns.  `live`: a half without a tile takes part in the barriers only (its descriptor has zero records: loads
// give 0, stores are dropped); `planes_live`: this half also writes planes (both wave-uniform per half).
constexpr int B3_WB_NCH = 8;
// Cache policy of the C tile's loads and stores: nt (aux = 2).  A C tile is touched once per launch and streams through
// (208 MB per latent in the tail of the metric shape, 16 GB at n = 44 484); with the default policy it pushes the operand
// planes -- which every macro row re-reads -- out of the Infinity Cache.  Measured (variant builds, -DB3_C_AUX / -DB3_C_AUX_ST):
// loads and stores nt together 18.07 -> 17.87 ms/step at the metric shape, 304 -> 291 ms at the C5 share; either one alone: no
// gain.  The planes' own stores keep the default policy (they are the next launch's operands).
#ifndef B3_C_AUX
#define B3_C_AUX 2
#endif
#ifndef B3_C_AUX_ST
#define B3_C_AUX_ST 2
#endif
__device__ __forceinline__ void b3_preload(f32x4 (&vc)[B3_WB_NCH], const float *Cg, int64_t ldc, int tid, bool live) {
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(Cg), 0, __builtin_amdgcn_readfirstlane(live ? 0x7fffffff : 0), 0x00020000);
  const unsigned voff = (unsigned)(((int64_t)(tid / 32) * ldc + (tid % 32) * 4) * 4);
  const unsigned rstep = (unsigned)((int64_t)8 * ldc * 4);
#pragma unroll
  for (int h = 0; h < B3_WB_NCH; ++h) vc[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rC, voff, (unsigned)h * rstep, B3_C_AUX));
}
template <class S, int MODE, bool PLANES, bool PRE = false>
__device__ __forceinline__ void b3_writeback(const Acc<float> &acc, float *Cg, int64_t ldc, float *smem, int tid, bool live,
                                             unsigned short *Pp = nullptr, int64_t pld = 0, bool planes_live = true, float pscale = 1.0f,
                                             const f32x4 *vc0 = nullptr) {
  constexpr bool ADD = MODE == WB_ADD || MODE == WB_SUB;
  typedef int i32x4_t __attribute__((ext_vector_type(4)));
  constexpr int LDW = 132, CPR = 32, NCH = B3_WB_NCH, RSTEP = 8;
  const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  const int row0 = tid / CPR, col0 = (tid % CPR) * 4;
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(Cg, 0, __builtin_amdgcn_readfirstlane(live ? 0x7fffffff : 0), 0x00020000);
  const unsigned voff = (unsigned)(((int64_t)row0 * ldc + col0) * 4);
  const unsigned rstep = (unsigned)((int64_t)RSTEP * ldc * 4);
  f32x4 vc[2][NCH];
  if (ADD) {
#pragma unroll
    for (int h = 0; h < NCH; ++h)
      vc[0][h] = PRE ? vc0[h] : __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rC, voff, (unsigned)h * rstep, B3_C_AUX));
#pragma unroll
    for (int h = 0; h < NCH; ++h) vc[1][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rC, voff, (unsigned)(8 + h) * rstep, B3_C_AUX));
  }
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();                                   // the previous pass is done with the staging area
    if (wm == half) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = mt * 16 + Traits<float>::acc_row(lane, r);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) smem[row * LDW + wn * 64 + nt * 16 + (lane & 15)] = acc.v[mt][nt][r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      float *sp = smem + (row0 + h * RSTEP) * LDW + col0;
      const f32x4 sv = *reinterpret_cast<const f32x4 *>(sp);
      const f32x4 o = MODE == WB_ADD ? vc[half][h] + sv : (MODE == WB_SUB ? vc[half][h] - sv : (MODE == WB_STORE_NEG ? -sv : sv));
      if (PLANES && MODE != WB_STORE) *reinterpret_cast<f32x4 *>(sp) = o;   // each thread owns its chunks: the staging area now holds the finals
      // row-chunk offset in voffset, soffset = 0: the store-data hazard note of gemm_core.hpp
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, o), rC, voff + (unsigned)(half * 8 + h) * rstep, 0, B3_C_AUX_ST);
    }
    if constexpr (PLANES) {
      __syncthreads();
      if (live && planes_live) {                                   // 64 rows = 8 k8 groups x 32 column quads: one item per thread
        const int k8 = tid >> 5, c4 = (tid & 31) * 4;
        float x[8][4];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const f32x4 v = *reinterpret_cast<const f32x4 *>(smem + (k8 * 8 + r) * LDW + c4);
          x[r][0] = v[0]; x[r][1] = v[1]; x[r][2] = v[2]; x[r][3] = v[3];
        }
        b3_split_store<S>(x, Pp + ((int64_t)(half * 8 + k8) * S::NPL * pld + c4) * 8, pld, pscale);
      }
    }
  }
}

}  // namespace plmc
