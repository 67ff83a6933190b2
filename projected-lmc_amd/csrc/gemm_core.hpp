// gemm_core.hpp -- the one MFMA tile engine every blocked routine is built from (gfx950 only).
//
// All dense work on the hot path is expressed as "TN" products  C[i][j] (+)= sum_k A[k][i] * B[k][j]
// where BOTH operands are stored K-major (row index = contraction index).  That is what the
// HBM layout was chosen for: Khat = U^T U with U upper / row-major makes the Cholesky trailing
// update, the panel solve, the forward substitution W = U^-T and K^-1 = W^T W all TN products,
// so tiles stream from global to LDS without a transpose and MFMA fragments are read
// conflict-free (LDS row order / stride: see tile_mainloop).
//
// Workgroup = 256 threads = 4 waves (2 x 2), block tile 128 x 128, wave tile 64 x 64 =
// 4 x 4 MFMA tiles of v_mfma_{f32,f64}_16x16x4 (the f32 form runs at the f32 matrix peak,
// 64 FLOP/clk/SIMD; MI355X_MICROARCH.md "Matrix cores").
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace plmc {

constexpr int NB = 128;    // block edge of all blocked algorithms == tile edge
constexpr int BK = 16;     // k-depth of one LDS stage
constexpr int LDT = 132;   // LDS row stride in elements (see tile_mainloop for the row order)
constexpr int NTHREADS = 256;
// start of the B stages inside the staging buffer: the A stages (2 * BK * LDT = 4224 elements) rounded up to a
// multiple of 256 elements, which keeps the B fragments of two steps within one ds_read2 offset window
constexpr int SB_OFF = 4352;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

template <typename T> struct Traits;
template <> struct Traits<float> {
  using acc_t = f32x4;
  using vec_t = f32x4;                 // 16-byte global/LDS vector
  static constexpr int EPV = 4;        // elements per 16 B
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f32_16x16x4_f32: col = lane & 15, row = (lane >> 4) * 4 + reg
  static __device__ __forceinline__ int acc_row(int lane, int r) { return ((lane >> 4) << 2) + r; }
};
template <> struct Traits<double> {
  using acc_t = f64x4;
  using vec_t = f64x2;
  static constexpr int EPV = 2;
  static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  static __device__ __forceinline__ int acc_row(int lane, int r) { return (lane >> 4) + (r << 2); }
};

// Accumulators of one wave: MT x 4 MFMA tiles.  MT = 4: the 128 x 128 block tile (64 x 64 per wave); MT = 2: a
// 64-row x 128-column half tile (32 x 64 per wave) for launches that would not fill the CUs with full tiles;
// NT = 2: a 128-row x 64-column half tile (64 x 32 per wave), for the in-place row panel.
template <typename T, int MT = 4, int NT = 4> struct Acc {
  typename Traits<T>::acc_t v[MT][NT];
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
      for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[a][b][r] = T(0);
  }
};

// Occupancy floor of the tile kernels (second __launch_bounds__ argument = waves per SIMD): the fp64 kernels need
// 128 accumulator registers; without a floor hipcc spends 230+ more and leaves one wave per SIMD.
template <typename T> constexpr int TILE_MIN_WAVES = sizeof(T) == 8 ? 2 : 1;

// LDS owned by a tile kernel: at least the 2 stages x (A,B) x BK x LDT elements of tile_mainloop.
// (kept at the 144-stride size: the epilogues of the gradient kernels stage up to 2 x 128 x 33 + 256 elements in it)
template <typename T> constexpr int tile_smem_elems() { return 2 * 2 * BK * 144; }

// Position of accumulator element (mt, nt, r) of this lane inside the 128x128 block tile.
template <typename T, int MT = 4> __device__ __forceinline__ int tile_row(int wm, int mt, int lane, int r) {
  return wm * (16 * MT) + mt * 16 + Traits<T>::acc_row(lane, r);
}
template <int NT = 4> __device__ __forceinline__ int tile_col(int wn, int nt, int lane) { return wn * (16 * NT) + nt * 16 + (lane & 15); }

// acc += sum_{k < K} Ag[k][0..127]^T * Bg[k][0..127]      (NEG: the A fragments are negated -- dev benches only;
// the product kernels subtract in the epilogue instead, tile_writeback<.., WB_SUB>)
// Ag/Bg point at the first row of the K range and the first of the 128 columns; K % BK == 0.
// All 256 threads must call it; ends with a barrier (LDS free for reuse on return).
// REV walks the K range from its last BK-slab to its first: tiles whose ranges END together then
// read the same operand rows at the same time (L2 sharing for the triangular products).
//
// Everything that is not an MFMA is kept off the vector ALU inside the loop: non-MFMA VALU instructions
// compete with the MFMAs of the co-resident waves for the SIMD's issue port (SQ counters on a plain 8192^3
// GEMM: 0.58 VALU per MFMA -> 130 TF; pointer-increment global addresses alone -> 141 TF).  Global addresses
// advance by pointer increments, LDS positions are loop-invariant registers plus one stage offset per slab, and the
// sign of C -= A^T B is applied in the epilogue.  (Unrolling the slab loop by two to make the stage a compile-time
// constant cost 50 more registers and an occupancy step: 128 TF.)
template <typename T, bool NEG, bool REV = false, int MT = 4, int NT = 4>
__device__ __forceinline__ void tile_mainloop(Acc<T, MT, NT> &acc, const T *__restrict__ Ag, int64_t lda,
                                              const T *__restrict__ Bg, int64_t ldb, int K, T *smem) {
  using Tr = Traits<T>;
  using vec_t = typename Tr::vec_t;
  constexpr int EPV = Tr::EPV;
  // B slab: BK rows x BW = 32 NT columns (the tile's columns); A slab: BK rows x AW = 32 MT columns (the tile's rows)
  constexpr int AW = 32 * MT, BW = 32 * NT;
  constexpr int CPR = BW / EPV, CPRA = AW / EPV;     // 16-byte chunks per slab row
  constexpr int NCH = BK * CPR / NTHREADS;           // chunks per thread, B (2 / 4)
  constexpr int NCHA = BK * CPRA / NTHREADS;         // chunks per thread, A (MT = 4: as B; MT = 2: 1 / 2)
  constexpr int RSTEP = NTHREADS / CPR, RSTEPA = NTHREADS / CPRA;   // rows between a thread's chunks
  static_assert(NCHA >= 1 && NCH >= 1 && RSTEP % 4 == 0 && RSTEPA % 4 == 0, "slab split");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  T *sA = smem;
  T *sB = smem + SB_OFF;

  vec_t ra[NCHA], rb[NCH];
  const int nkt = K / BK;
  const int row0 = tid / CPR, col0 = (tid % CPR) * EPV;
  const int row0a = tid / CPRA, col0a = (tid % CPRA) * EPV;
  // global loads are raw buffer loads: descriptor = wave-uniform slab base (advanced with scalar adds and rebuilt
  // per slab: a 32-bit soffset would wrap on the 16 GB factor buffers of n = 44 484), voffset = a loop-invariant
  // byte offset per chunk -- no vector instruction is spent on addresses
  const char *baseA = reinterpret_cast<const char *>(Ag) + (int64_t)(REV ? (nkt - 1) * BK : 0) * lda * (int64_t)sizeof(T);
  const char *baseB = reinterpret_cast<const char *>(Bg) + (int64_t)(REV ? (nkt - 1) * BK : 0) * ldb * (int64_t)sizeof(T);
  const int64_t stepA = (REV ? -(int64_t)BK : (int64_t)BK) * lda * (int64_t)sizeof(T);
  const int64_t stepB = (REV ? -(int64_t)BK : (int64_t)BK) * ldb * (int64_t)sizeof(T);
  unsigned offA[NCHA], offB[NCH];
#pragma unroll
  for (int h = 0; h < NCHA; ++h) offA[h] = (unsigned)(((int64_t)(row0a + h * RSTEPA) * lda + col0a) * (int64_t)sizeof(T));
#pragma unroll
  for (int h = 0; h < NCH; ++h) offB[h] = (unsigned)(((int64_t)(row0 + h * RSTEP) * ldb + col0) * (int64_t)sizeof(T));
  auto gload = [&]() {
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(baseA), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(baseB), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int h = 0; h < NCHA; ++h) ra[h] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rA, offA[h], 0, 0));
#pragma unroll
    for (int h = 0; h < NCH; ++h) rb[h] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rB, offB[h], 0, 0));
    baseA += stepA;
    baseB += stepB;
  };
  // LDS layout of a slab: contraction row r = 4 ks + fk (MFMA step ks, lane group fk) is stored as LDS row
  // 4 fk + ks, stride LDT = 132.  A lane's fragments of steps ks and ks + 1 are then 132 elements apart and, with the
  // 16-element tile offsets, within the 8-bit offset range of ds_read2 from ONE base register (two bases per
  // operand and slab instead of a new one per step); lane groups sit 4 * 132 = 16 (mod 32) words apart, so the two
  // half-waves of a read still cover all banks.  A stage adds BK * LDT elements (one add per operand and slab).
  // a thread's rows row0 + h * RSTEP land on LDS rows rp0 + h * RSTEP / 4 (RSTEP is a multiple of 4): one base
  const int rp0 = (row0 & 3) * 4 + (row0 >> 2), rp0a = (row0a & 3) * 4 + (row0a >> 2);
  T *swB = sB + rp0 * LDT + col0, *swA = sA + rp0a * LDT + col0a;
  auto sstore = [&](int buf) {
    T *wa = swA + buf * (BK * LDT), *wb = swB + buf * (BK * LDT);
#pragma unroll
    for (int h = 0; h < NCHA; ++h) *reinterpret_cast<vec_t *>(wa + h * (RSTEPA / 4) * LDT) = ra[h];
#pragma unroll
    for (int h = 0; h < NCH; ++h) *reinterpret_cast<vec_t *>(wb + h * (RSTEP / 4) * LDT) = rb[h];
  };
  const int fk = lane >> 4, fm = lane & 15;
  const T *pa0 = sA + fk * 4 * LDT + wm * (16 * MT) + fm, *pb0 = sB + fk * 4 * LDT + wn * (16 * NT) + fm;   // stage 0, step 0

  gload();
  sstore(0);
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nkt) gload();
    const T *pa = pa0 + buf * (BK * LDT), *pb = pb0 + buf * (BK * LDT);
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      T a[MT], b[NT];
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        a[t] = pa[ks * LDT + t * 16];
        if (NEG) a[t] = -a[t];
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) b[t] = pb[ks * LDT + t * 16];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc.v[mt][nt] = Tr::mfma(a[mt], b[nt], acc.v[mt][nt]);
    }
    if (kt + 1 < nkt) sstore(buf ^ 1);
    __syncthreads();
  }
}

// Burst form of the main loop for the LATENCY-BOUND chain kernels of the sweep (row panel, rank-128 updates inside a
// group's triangle: K = 128, a handful of tiles per launch, running beside MFMA-saturating bulk kernels).  The
// double-buffered loop above keeps ONE slab in flight, so a K = 128 tile is 8 dependent global round trips -- ~1 us each
// on an idle GPU, 5-10 us when the bulk kernels fill the memory system: that, not the arithmetic, is what made a
// 10 us panel take 50-100 us in situ.  Here all loads of BURST slabs are issued before the first is used (register
// staged: BURST x (NCHA + NCH) x 16 bytes per thread; fp32 half tiles: 96 registers for the whole K = 128), so a tile
// costs K / (16 BURST) round trips.  Same LDS layout, same accumulation order as tile_mainloop (bit-identical results).
// K % (BK * BURST) == 0.  All 256 threads must call it; ends with a barrier.
template <typename T, int MT, int NT, int BURST>
__device__ __forceinline__ void tile_mainloop_burst(Acc<T, MT, NT> &acc, const T *__restrict__ Ag, int64_t lda,
                                                    const T *__restrict__ Bg, int64_t ldb, int K, T *smem) {
  using Tr = Traits<T>;
  using vec_t = typename Tr::vec_t;
  constexpr int EPV = Tr::EPV;
  constexpr int AW = 32 * MT, BW = 32 * NT;
  constexpr int CPR = BW / EPV, CPRA = AW / EPV;
  constexpr int NCH = BK * CPR / NTHREADS, NCHA = BK * CPRA / NTHREADS;
  constexpr int RSTEP = NTHREADS / CPR, RSTEPA = NTHREADS / CPRA;
  static_assert(NCHA >= 1 && NCH >= 1 && RSTEP % 4 == 0 && RSTEPA % 4 == 0, "slab split");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  T *sA = smem;
  T *sB = smem + SB_OFF;
  const int row0 = tid / CPR, col0 = (tid % CPR) * EPV;
  const int row0a = tid / CPRA, col0a = (tid % CPRA) * EPV;
  const char *baseA = reinterpret_cast<const char *>(Ag);
  const char *baseB = reinterpret_cast<const char *>(Bg);
  const int64_t stepA = (int64_t)BK * lda * (int64_t)sizeof(T), stepB = (int64_t)BK * ldb * (int64_t)sizeof(T);
  unsigned offA[NCHA], offB[NCH];
#pragma unroll
  for (int h = 0; h < NCHA; ++h) offA[h] = (unsigned)(((int64_t)(row0a + h * RSTEPA) * lda + col0a) * (int64_t)sizeof(T));
#pragma unroll
  for (int h = 0; h < NCH; ++h) offB[h] = (unsigned)(((int64_t)(row0 + h * RSTEP) * ldb + col0) * (int64_t)sizeof(T));
  const int rp0 = (row0 & 3) * 4 + (row0 >> 2), rp0a = (row0a & 3) * 4 + (row0a >> 2);
  T *swB = sB + rp0 * LDT + col0, *swA = sA + rp0a * LDT + col0a;
  const int fk = lane >> 4, fm = lane & 15;
  const T *pa0 = sA + fk * 4 * LDT + wm * (16 * MT) + fm, *pb0 = sB + fk * 4 * LDT + wn * (16 * NT) + fm;
  const int nburst = K / (BK * BURST);
#pragma unroll 1
  for (int bt = 0; bt < nburst; ++bt) {
    vec_t ra[BURST][NCHA], rb[BURST][NCH];
#pragma unroll
    for (int s = 0; s < BURST; ++s) {                  // every load of the burst in flight before the first use
      const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(baseA), 0, 0x7fffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(baseB), 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int h = 0; h < NCHA; ++h) ra[s][h] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rA, offA[h], 0, 0));
#pragma unroll
      for (int h = 0; h < NCH; ++h) rb[s][h] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rB, offB[h], 0, 0));
      baseA += stepA;
      baseB += stepB;
    }
#pragma unroll
    for (int s = 0; s < BURST; ++s) {
      const int buf = s & 1;
      T *wa = swA + buf * (BK * LDT), *wb = swB + buf * (BK * LDT);
#pragma unroll
      for (int h = 0; h < NCHA; ++h) *reinterpret_cast<vec_t *>(wa + h * (RSTEPA / 4) * LDT) = ra[s][h];
#pragma unroll
      for (int h = 0; h < NCH; ++h) *reinterpret_cast<vec_t *>(wb + h * (RSTEP / 4) * LDT) = rb[s][h];
      __syncthreads();                                 // slab s visible; every wave is past the reads of slab s - 1
      const T *pa = pa0 + buf * (BK * LDT), *pb = pb0 + buf * (BK * LDT);
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks) {
        T a[MT], b[NT];
#pragma unroll
        for (int t = 0; t < MT; ++t) a[t] = pa[ks * LDT + t * 16];
#pragma unroll
        for (int t = 0; t < NT; ++t) b[t] = pb[ks * LDT + t * 16];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc.v[mt][nt] = Tr::mfma(a[mt], b[nt], acc.v[mt][nt]);
      }
    }
    __syncthreads();                                   // both stages free (next burst / the caller's epilogue)
  }
}

// Coalesced epilogue: C[tile] (+)= acc through LDS.  The MFMA accumulator layout gives each lane 4-byte
// pieces of 64-byte row segments (64 scalar loads + 64 scalar stores per lane for a read-modify-write);
// transposing the tile through LDS in two 64-row halves turns that into 16-byte accesses of full
// 512-byte rows (8 + 8 per lane for fp32).  `smem` = the mainloop's staging buffer (free after its
// final barrier; needs 64 x 132 elements).  All 256 threads must call it.
// The C tile is read with raw buffer loads (descriptor on the tile, per-thread voffset fixed, row-chunk offset in
// soffset: no vector instruction for addresses) in chunks of 2 x 16 bytes per thread, software-pipelined: the
// first chunk of the tile is requested before the accumulators are staged, every later one while the previous is
// combined and stored, across the half boundary as well -- an un-prefetched read-modify-write left the workgroup
// idle for two HBM round trips per tile, ~10 % of a depth-1024 update tile.
//
// STORES carry their row-chunk offset in the VECTOR offset (soffset = 0), never in an SGPR.  gfx950 needs one wait
// state between a `buffer_store_dwordx4 ..., sN offen` and a VALU write to its first data register, or lanes 12-15 of
// every 16-lane row store the NEW register value (tools/store_hazard_probe.hip: 61 456 marker dwords at 0 wait
// states, 0 at 1; with an immediate soffset 2 wait states are needed).  hipcc's hazard recognizer
// (GCNHazardRecognizer::createsVALUHazard) inserts those wait states only for MUBUF stores WITHOUT an SGPR soffset
// and emits nothing for the SGPR form, so whether a kernel was correct depended on what the scheduler happened to
// place behind the store: with the cross-half prefetch it placed `v_and_b32 v16, 0x80, v0` (the wave-row test of the
// second staging pass) directly behind `buffer_store_dwordx4 v[16:19], ..., s15 offen` and every update kernel
// produced tiles with wrong elements in columns 48/52/56/60 (+64) from row 48 on -- the "wrong factors under the
// look-ahead" of round 1 ("Write-back hazard").  With soffset = 0 the compiler sees the hazard it knows
// and pads it itself; tests/test_isa_hazards.py scans the built library for the unprotected pattern.
enum { WB_STORE = 0, WB_ADD = 1, WB_SUB = 2, WB_STORE_NEG = 3 };   // C = acc | C += acc | C -= acc | C = -acc
// `tid_` / `live`: for the 512-thread macro-tile kernels (bf3_engine.hpp), whose two halves of 256 threads each write one
// 128 x 128 tile through their own staging area: tid_ = threadIdx.x & 255; a half without a tile (live = false) takes
// part in the barriers only.
template <typename T, int MODE, int MT = 4>
__device__ __forceinline__ void tile_writeback(const Acc<T, MT> &acc, T *Cg, int64_t ldc, T *smem, int tid_ = -1, bool live = true) {
  constexpr bool ADD = MODE == WB_ADD || MODE == WB_SUB;
  using vec_t = typename Traits<T>::vec_t;
  typedef int i32x4_t __attribute__((ext_vector_type(4)));
  constexpr int EPV = Traits<T>::EPV;
  constexpr int LDW = 132;
  constexpr int CPR = 128 / EPV;                       // 16-byte chunks per row
  constexpr int NCH = 64 * CPR / NTHREADS;             // chunks per thread per half (8 fp32 / 16 fp64)
  constexpr int RSTEP = NTHREADS / CPR;                // rows between a thread's chunks (8 / 4)
  constexpr int CH = 2;                                // chunks per pipeline stage
  constexpr int NHALF = MT / 2;
  const int tid = tid_ < 0 ? (int)threadIdx.x : tid_, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int row0 = tid / CPR, col0 = (tid % CPR) * EPV;
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(Cg, 0, __builtin_amdgcn_readfirstlane(live ? 0x7fffffff : 0), 0x00020000);   // 0 records: loads give 0, stores are dropped
  const unsigned voff = (unsigned)(((int64_t)row0 * ldc + col0) * (int64_t)sizeof(T));
  const unsigned rstep = (unsigned)((int64_t)RSTEP * ldc * (int64_t)sizeof(T));      // wave-uniform
  auto cload = [&](int half, int h0, vec_t (&v)[CH]) {
#pragma unroll
    for (int h = 0; h < CH; ++h)
      v[h] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rC, voff, (unsigned)(half * (64 / RSTEP) + h0 + h) * rstep, 0));
  };
  vec_t vcur[CH], vnext[CH];
  if (ADD) cload(0, 0, vcur);
  // MT = 4: two passes of 64 rows, pass h staged by the wave row wm == h; MT = 2: one pass, both wave rows stage
#pragma unroll
  for (int half = 0; half < NHALF; ++half) {
    if (half) __syncthreads();                         // previous half fully read back
    if (MT == 2 || wm == half) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = (MT == 2 ? wm * 32 : 0) + mt * 16 + Traits<T>::acc_row(lane, r);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) smem[row * LDW + wn * 64 + nt * 16 + (lane & 15)] = acc.v[mt][nt][r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int h0 = 0; h0 < NCH; h0 += CH) {
      if (ADD) {
        if (h0 + CH < NCH) cload(half, h0 + CH, vnext);
        else if (half + 1 < NHALF) cload(half + 1, 0, vnext);          // across the half boundary
      }
#pragma unroll
      for (int h = 0; h < CH; ++h) {
        const vec_t sv = *reinterpret_cast<const vec_t *>(smem + (row0 + (h0 + h) * RSTEP) * LDW + col0);
        const vec_t o = MODE == WB_ADD ? vcur[h] + sv : (MODE == WB_SUB ? vcur[h] - sv : (MODE == WB_STORE_NEG ? -sv : sv));
        // row-chunk offset in voffset, soffset = 0: see the hazard note above
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, o), rC,
                                               voff + (unsigned)(half * (64 / RSTEP) + h0 + h) * rstep, 0, 0);
      }
#pragma unroll
      for (int h = 0; h < CH; ++h) vcur[h] = vnext[h];
    }
  }
}

// XCD-aware enumeration of the upper-triangular tile set {(ib, jb): ib <= jb < m} of `nlat` matrices.
// Tiles are grouped in 8 x 8 super-blocks (64 tiles that share 8 + 8 operand strips); super-blocks
// are ordered by decreasing K-depth (jb ascending) and dealt round-robin to the 8 XCDs; workgroup w is
// observed to land on XCD w % 8 (MI355X_MICROARCH.md "Workgroup dispatch" -- speed only, never
// correctness), so the 64 tiles of a super-block stream their strips through ONE L2.
// Launch with grid = xcd_tri_grid(m, nlat) workgroups; returns false for padding / dead tiles.
__host__ __device__ inline int xcd_tri_grid(int m, int nlat) {
  const int M = (m + 7) / 8, G = nlat * (M * (M + 1) / 2);
  return 8 * ((G + 7) / 8) * 64;
}
__device__ __forceinline__ bool xcd_tri_decode(int w, int m, int nlat, int &lat, int &ib, int &jb) {
  const int M = (m + 7) / 8, NSB = M * (M + 1) / 2, G = nlat * NSB;
  const int xcd = w & 7, slot = w >> 3;
  const int g = xcd + 8 * (slot >> 6);
  if (g >= G) return false;
  lat = g / NSB;
  const int k = g - lat * NSB;
  int JB = (int)((sqrtf(8.0f * (float)k + 1.0f) - 1.0f) * 0.5f);
  while ((JB + 1) * (JB + 2) / 2 <= k) ++JB;
  while (JB * (JB + 1) / 2 > k) --JB;
  const int IB = k - JB * (JB + 1) / 2;
  const int t = slot & 63;
  ib = IB * 8 + (t >> 3);
  jb = JB * 8 + (t & 7);
  return ib <= jb && jb < m;
}

// C[tile] = acc (plain store of the 128x128 tile at Cg, leading dimension ldc).
template <typename T, int MT = 4, int NT = 4>
__device__ __forceinline__ void tile_store(const Acc<T, MT, NT> &acc, T *Cg, int64_t ldc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int row = tile_row<T, MT>(wm, mt, lane, r);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) Cg[(int64_t)row * ldc + tile_col<NT>(wn, nt, lane)] = acc.v[mt][nt][r];
    }
}

// C[tile] += acc  (used with NEG mainloop for C -= A^T B).
template <typename T>
__device__ __forceinline__ void tile_add_store(const Acc<T> &acc, T *Cg, int64_t ldc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1;
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      int row = tile_row<T>(wm, mt, lane, r);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        T *p = Cg + (int64_t)row * ldc + tile_col(wn, nt, lane);
        *p = *p + acc.v[mt][nt][r];
      }
    }
}

}  // namespace plmc
