// bf3_engine.hpp -- fp32 tile products on the bf16 matrix cores, second engine (round 3).
//
// What it computes.  C[i][j] (+)= sum_k A[k][i] B[k][j] for fp32 operands that were split ONCE into three bf16 planes
// x = hi + mid + lo (round to nearest even at every level, exact residuals: all 24 significand bits), as six plane
// products on v_mfma_f32_16x16x32_bf16 with fp32 accumulation -- into TWO accumulator levels:
//     level 0:  hi.hi                                  (magnitude of the product itself)
//     level 1:  lo.hi + hi.lo + mid.mid + mid.hi + hi.mid   (2^-8 of it and below)
// summed once at the end.  tools/split_numerics_probe.hip (profiles/r03_split_numerics.txt) is why: the error of such a
// product against fp64 is 0.3-0.4 x that of the v_mfma_f32_16x16x4_f32 chain of the fp32 engine (K = 1024 and 8192;
// normal, one-signed and wide-range data) and it is the rounding of level 0 alone -- eight or nine plane products give
// the same digits, a third level changes nothing -- whereas ONE accumulator for all six products (round 2's opt-in)
// carries 2.5 x the error of two levels (the small products are rounded against the large running sum).  The dropped
// products mid.lo, lo.mid, lo.lo are below 2^-24 of a product.
//
// How.  A workgroup of 512 threads = 8 waves owns a 256 x 128 macro tile (two 128 x 128 tiles of one block column: the
// rows ib, ib + 1 share the B strip), each wave a 64 x 64 block of it as 4 x 4 MFMA tiles -- the accumulator layout of the
// fp32 engine (gemm_core.hpp), so tile_writeback and the gradient epilogues are reused per half.  Operand planes live in
// HBM in "k8" order,
//     P[k / 8][plane][column][k % 8]        (16 bytes = the 8 contraction values one MFMA lane needs),
// so that (a) a stage of 32 contraction rows of a tile is 12 runs of 2-4 KB, copied to LDS by LDS-DMA
// (buffer_load_dwordx4 ... lds: no staging registers, no ds_write, one 1 KB piece per wave instruction) and (b) an MFMA
// fragment is ONE conflict-free ds_read_b128 per lane (lanes 16 g .. 16 g + 15 read 256 contiguous bytes of k-group g):
// no transposing reads, no swizzle.  Two LDS stages of 72 KB (A: 3 planes x 16 KB, B: 3 x 8 KB); per stage and wave
// 9 DMA pieces, 24 fragment reads and 96 MFMAs; one barrier per stage: the DMA of stage s + 1 is issued right after the
// barrier that ends the reads of stage s - 1 and lands while stage s is multiplied.
#pragma once
#include "gemm_core.hpp"

namespace plmc {

constexpr int B3_NT = 512;                                 // threads per workgroup
constexpr int B3_K = 32;                                   // contraction rows per LDS stage
constexpr int B3_AW = 256, B3_BW = 128;                    // macro tile: A side (rows of C) x B side (columns of C)
constexpr int B3_A_PLANE = (B3_K / 8) * B3_AW * 16;        // bytes of one A plane of a stage: [4 k-groups][256][16 B]
constexpr int B3_B_PLANE = (B3_K / 8) * B3_BW * 16;
constexpr int B3_STAGE = 3 * (B3_A_PLANE + B3_B_PLANE);    // 73 728
constexpr int B3_LDS_BYTES = 2 * B3_STAGE;                 // 147 456: one workgroup per CU
constexpr int B3_WB_BYTES = 64 * 132 * 4;                  // staging of tile_writeback, per 128 x 128 half

typedef __bf16 b3_bf16x8 __attribute__((ext_vector_type(8)));
typedef short b3_s16x8 __attribute__((ext_vector_type(8)));

// element (k, plane, column) of a k8-ordered plane buffer with `ld` columns, in 16-bit elements
__host__ __device__ inline int64_t b3_index(int64_t k, int plane, int64_t col, int64_t ld) {
  return (((k >> 3) * 3 + plane) * ld + col) * 8 + (k & 7);
}
// 16-bit elements of a plane buffer of `rows` contraction rows (a multiple of 8) and `ld` columns
__host__ __device__ inline int64_t b3_elems(int64_t rows, int64_t ld) { return rows * 3 * ld; }

// Eight contraction rows x four columns of fp32 -> the three planes' 16-byte groups (one per column and plane).
// x[r][c]: row r (0..7) of the k8 group, column c (0..3).  `dst` = &P[k8][0][col0][0]; plane stride = ld * 8 elements.
__device__ __forceinline__ void b3_split_store(const float (&x)[8][4], unsigned short *dst, int64_t ld) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    b3_s16x8 h, m, l;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float v = x[r][c];
      const __bf16 a = (__bf16)v;
      const float r1 = v - (float)a;
      const __bf16 b = (__bf16)r1;
      const __bf16 cc = (__bf16)(r1 - (float)b);
      h[r] = __builtin_bit_cast(short, a);
      m[r] = __builtin_bit_cast(short, b);
      l[r] = __builtin_bit_cast(short, cc);
    }
    *reinterpret_cast<b3_s16x8 *>(dst + c * 8) = h;
    *reinterpret_cast<b3_s16x8 *>(dst + ld * 8 + c * 8) = m;
    *reinterpret_cast<b3_s16x8 *>(dst + 2 * ld * 8 + c * 8) = l;
  }
}

// One 128 x 128 fp32 block (rows = contraction index, leading dimension lds) -> planes.  256 threads; thread = one
// k8 group x four columns per pass (8 loads of 16 bytes, 12 stores of 16 bytes; whole 512-byte rows / 64-byte runs).
// `P` points at element (k = first row of the block, plane 0, first column of the block).  COPY: also write the block
// to D (leading dimension ldd): the panel copy of the sweep does both in one pass.
template <bool COPY>
__device__ __forceinline__ void b3_split_block(const float *__restrict__ S, int64_t lds_, unsigned short *__restrict__ P, int64_t ld, float *D,
                                               int64_t ldd, int tid) {
#pragma unroll 1
  for (int w = tid; w < 16 * 32; w += NTHREADS) {
    const int k8 = w >> 5, c4 = (w & 31) * 4;
    float x[8][4];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float4 v = *reinterpret_cast<const float4 *>(S + (int64_t)(k8 * 8 + r) * lds_ + c4);
      x[r][0] = v.x; x[r][1] = v.y; x[r][2] = v.z; x[r][3] = v.w;
      if (COPY) *reinterpret_cast<float4 *>(D + (int64_t)(k8 * 8 + r) * ldd + c4) = v;
    }
    b3_split_store(x, P + ((int64_t)k8 * 3 * ld + c4) * 8, ld);
  }
}

// acc0 / acc1 += the two levels of  sum_{k < K} A[k][a-columns]^T B[k][b-columns]  for this wave's 64 x 64 block.
//   Ap: plane buffer at (k = first row of the K range, plane 0, first of the 256 A columns), lda_ columns per plane row;
//   Bp: likewise, first of the 128 B columns, ldb_ columns;  K % 64 == 0;  lds: B3_LDS_BYTES, 16-byte aligned, the kernel's
//   ONLY __shared__ object (a second one makes hipcc drain the DMA before every fragment read).
// All 512 threads must call it; ends with a barrier (LDS free for the epilogue).
__device__ __forceinline__ void b3_mainloop(Acc<float> &acc0, Acc<float> &acc1, const unsigned short *__restrict__ Ap, int64_t lda_,
                                            const unsigned short *__restrict__ Bp, int64_t ldb_, int K, unsigned char *lds) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave 0..7, provably uniform (DMA destinations are scalar)
  const int half = w >> 2, wm = (w >> 1) & 1, wn = w & 1;
  // ---- DMA plan of this wave: pieces i = w + 8 j of the 72 per stage.  j = 0..5: A, plane j >> 1, k-group (w >> 2) + 2 (j & 1),
  // columns 64 (w & 3) ..; j = 6..8: B, plane j - 6, k-group (w >> 1) & 3, columns 64 (w & 1) ..
  const unsigned RSA = (unsigned)(3 * lda_ * 16), RSB = (unsigned)(3 * ldb_ * 16);   // bytes per k8 row (all planes)
  const unsigned PLA = (unsigned)(lda_ * 16), PLB = (unsigned)(ldb_ * 16);           // bytes per plane inside a k8 row
  const unsigned gA0 = (unsigned)(w >> 2) * RSA + (unsigned)(w & 3) * 1024u;
  const unsigned gB0 = (unsigned)((w >> 1) & 3) * RSB + (unsigned)(w & 1) * 1024u;
  const unsigned lA0 = (unsigned)(((w >> 2) * B3_AW + (w & 3) * 64) * 16);
  const unsigned lB0 = (unsigned)(3 * B3_A_PLANE + (((w >> 1) & 3) * B3_BW + (w & 1) * 64) * 16);
  const unsigned voff = (unsigned)lane * 16u;
  const char *baseA = reinterpret_cast<const char *>(Ap), *baseB = reinterpret_cast<const char *>(Bp);
  const int64_t stepA = (int64_t)(B3_K / 8) * RSA, stepB = (int64_t)(B3_K / 8) * RSB;
  typedef __attribute__((address_space(3))) void lds_void;
  auto issue = [&](int buf) {
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(baseA), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(baseB), 0, 0x7fffffff, 0x00020000);
    unsigned char *sb = lds + buf * B3_STAGE;
#pragma unroll
    for (int j = 0; j < 6; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_void *)(sb + lA0 + (j >> 1) * B3_A_PLANE + (j & 1) * (2 * B3_AW * 16)), 16, voff,
                                           gA0 + (unsigned)(j & 1) * 2u * RSA + (unsigned)(j >> 1) * PLA, 0, 0);
#pragma unroll
    for (int j = 0; j < 3; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (lds_void *)(sb + lB0 + j * B3_B_PLANE), 16, voff, gB0 + (unsigned)j * PLB, 0, 0);
    baseA += stepA;
    baseB += stepB;
  };
  // ---- fragment addresses: lane (kg = lane >> 4, fr = lane & 15) reads 16 bytes of k-group kg, row / column .. + fr
  const int kg = lane >> 4, fr = lane & 15;
  const unsigned aA = (unsigned)((kg * B3_AW + half * 128 + wm * 64 + fr) * 16);
  const unsigned aB = (unsigned)(3 * B3_A_PLANE + (kg * B3_BW + wn * 64 + fr) * 16);
  auto compute = [&](int buf) {
    const unsigned char *sa = lds + buf * B3_STAGE + aA, *sb = lds + buf * B3_STAGE + aB;
    auto fa = [&](int plane, b3_bf16x8 (&f)[4]) {
#pragma unroll
      for (int t = 0; t < 4; ++t) f[t] = *reinterpret_cast<const b3_bf16x8 *>(sa + plane * B3_A_PLANE + t * 256);
    };
    auto fb = [&](int plane, b3_bf16x8 (&f)[4]) {
#pragma unroll
      for (int t = 0; t < 4; ++t) f[t] = *reinterpret_cast<const b3_bf16x8 *>(sb + plane * B3_B_PLANE + t * 256);
    };
    auto mm = [&](Acc<float> &acc, const b3_bf16x8 (&x)[4], const b3_bf16x8 (&y)[4]) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc.v[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x[mt], y[nt], acc.v[mt][nt], 0, 0, 0);
    };
    // small products first; every fragment is read once
    b3_bf16x8 bh[4], bx[4], ax[4], ay[4];
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
  };
  const int nst = K / B3_K;                                        // even
  issue(0);
#pragma unroll 1
  for (int s = 0; s < nst; s += 2) {
    __syncthreads();               // (vmcnt(0) + barrier) stage s has landed for every wave; stage s - 1 is read out
    issue(1);                      // nst is even: stage s + 1 always exists
    compute(0);
    __syncthreads();
    if (s + 2 < nst) issue(0);
    compute(1);
  }
  __syncthreads();
}

// Epilogue of one 128 x 128 half of the macro tile: tile_writeback (gemm_core.hpp) for fp32 plus, with PLANES, the FINAL
// values of the tile also as k8-ordered bf16 planes -- the next consumer's operand, written while the tile is still in
// LDS instead of by a separate split pass.  `Pp` = plane buffer at (k = the tile's first row, plane 0, the tile's first
// column), `pld` its columns.  Per 64-row pass the finals go back into the staging area (for the read-modify-write modes),
// then every thread splits 8 rows x 4 columns.  tid = threadIdx.x & 255; `live` as in tile_writeback; `planes_live`: this half
// also writes planes (both wave-uniform per half).
template <int MODE, bool PLANES>
__device__ __forceinline__ void b3_writeback(const Acc<float> &acc, float *Cg, int64_t ldc, float *smem, int tid, bool live,
                                             unsigned short *Pp = nullptr, int64_t pld = 0, bool planes_live = true) {
  if constexpr (!PLANES) {
    tile_writeback<float, MODE>(acc, Cg, ldc, smem, tid, live);
  } else {
    constexpr bool ADD = MODE == WB_ADD || MODE == WB_SUB;
    typedef int i32x4_t __attribute__((ext_vector_type(4)));
    constexpr int LDW = 132, CPR = 32, NCH = 8, RSTEP = 8;       // as tile_writeback<float>: 16-byte chunks, 8 per thread and pass
    const int lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int row0 = tid / CPR, col0 = (tid % CPR) * 4;
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(Cg, 0, __builtin_amdgcn_readfirstlane(live ? 0x7fffffff : 0), 0x00020000);
    const unsigned voff = (unsigned)(((int64_t)row0 * ldc + col0) * 4);
    const unsigned rstep = (unsigned)((int64_t)RSTEP * ldc * 4);
    f32x4 vc[NCH];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (ADD) {                                                   // the C rows of this pass, all in flight while the pass is staged
#pragma unroll
        for (int h = 0; h < NCH; ++h)
          vc[h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rC, voff, (unsigned)(half * 8 + h) * rstep, 0));
      }
      if (half) __syncthreads();                                   // plane pass of the previous half is done with the staging area
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
        const f32x4 o = MODE == WB_ADD ? vc[h] + sv : (MODE == WB_SUB ? vc[h] - sv : (MODE == WB_STORE_NEG ? -sv : sv));
        if (MODE != WB_STORE) *reinterpret_cast<f32x4 *>(sp) = o;  // each thread owns its chunks: the staging area now holds the finals
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, o), rC, voff + (unsigned)(half * 8 + h) * rstep, 0, 0);
      }
      __syncthreads();
      if (live && planes_live) {                                   // 64 rows = 8 k8 groups x 32 column quads: one item per thread
        const int k8 = tid >> 5, c4 = (tid & 31) * 4;
        float x[8][4];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const f32x4 v = *reinterpret_cast<const f32x4 *>(smem + (k8 * 8 + r) * LDW + c4);
          x[r][0] = v[0]; x[r][1] = v[1]; x[r][2] = v[2]; x[r][3] = v[3];
        }
        b3_split_store(x, Pp + ((int64_t)(half * 8 + k8) * 3 * pld + c4) * 8, pld);
      }
    }
  }
}

}  // namespace plmc
