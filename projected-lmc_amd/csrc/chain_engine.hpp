// chain_engine.hpp -- building blocks of the RESIDENT group-chain kernel of the sweep (potrf.hip, k_chain).
//
// The chain of one group of 8 block rows used to be 24 dependent launches (k_diag, k_panel, rank-128 k_update per block row).
// k_chain keeps a handful of workgroups per latent resident for the whole 1024 x 1024 triangle instead: every 128 x 128 tile
// operation of those launches (same operands, same K = 128 products in the same k order: bit-identical results) becomes one
// OP of a static, topologically ordered list; an op is handed to a workgroup statically, its dependencies are version counters
// of the tiles it reads and writes, kept in device memory:
//     producer: 16-byte sc1 (write-through) stores of the tile -> every wave s_waitcnt vmcnt(0) -> barrier -> lane 0: relaxed
//               agent-scope store of the tile's new version;
//     consumer: lane 0 polls the versions it needs (relaxed agent loads, s_sleep, BOUNDED: a spin that runs out raises the
//               abort word and every workgroup leaves) -> barrier -> sc1 loads of the tiles (served by L2, never by a stale L1).
// That is the placement-independent hand-off of /opt/skills/guides (cdna_hip_programming.md Guideline 16, write-through form):
// nothing depends on which CU or XCD a workgroup landed on.
//
// This header: the 512-thread (8-wave) K = 128 tile product with all operand loads in flight at once, and its write-back.
#pragma once
#include "gemm_core.hpp"

namespace plmc {

constexpr int CH_NT = 512;                                  // threads of a chain workgroup (= DIAG_NT: the diagonal-block body needs 8 waves)
constexpr int CH_AUX = 16;                                  // sc1: loads bypass the CU's L1, stores write through to memory
// slabs of 16 contraction rows in flight per burst: fp32 all 8 of K = 128 (64 staging registers), fp64 4 (64 registers)
template <typename T> constexpr int CH_BURST = sizeof(T) == 8 ? 4 : 8;

// Wave -> (wave row wm, wave column wn) of the 128 x 128 tile: waves 0..3 take rows 0..63, waves 4..7 rows 127..64 (wm = 3, 3, 2, 2):
// waves w and w + 4 share a SIMD, so each SIMD carries a top and a bottom wave row -- what balances the triangular panel
// products (TRI below), whose wave row wm multiplies only 2 wm + 2 of the 8 slabs.  Every element still sees the same k order.
__device__ __forceinline__ void chain_wave_pos(int wave, int &wm, int &wn) {
  wm = wave < 4 ? wave >> 1 : 3 - ((wave - 4) >> 1);
  wn = wave & 1;
}

// acc += sum_{k < 128} Ag[k][0..127]^T Bg[k][0..127]   for the whole 128 x 128 tile on 8 waves: wave (wm, wn) = (w >> 1, w & 1)
// owns rows 32 wm .., columns 64 wn .. (2 x 4 MFMA tiles).  Same LDS layout and the same accumulation order per element as
// tile_mainloop / tile_mainloop_burst (gemm_core.hpp): bit-identical results.  smem: tile_smem_elems<T>() elements.
// All 512 threads must call it; ends with a barrier.
// tri (run-time, workgroup-uniform): A is UPPER triangular with exact zeros below its diagonal (the inverse diagonal block V_r of
// the panel products): row i of the product only has terms k <= i, so wave row wm skips the slabs beyond 2 wm + 1 -- the skipped
// terms are exact zeros times finite numbers, the sums are unchanged bit for bit (a non-finite operand already means a failed
// factorisation, reported through `info`).
// `pre`: called once, right behind the loads of the first burst -- the place for the C tile's loads of a read-modify-write
// operation (chain_cload), so that they are in flight during the product instead of being one more round trip behind it.
struct ChainNoPre { __device__ __forceinline__ void operator()() const {} };
template <typename T, class PRE = ChainNoPre>
__device__ __forceinline__ void chain_mainloop(Acc<T, 2, 4> &acc, const T *__restrict__ Ag, int64_t lda, const T *__restrict__ Bg, int64_t ldb, T *smem,
                                               PRE pre = PRE(), const bool tri = false) {
  using Tr = Traits<T>;
  using vec_t = typename Tr::vec_t;
  constexpr int EPV = Tr::EPV, BURST = CH_BURST<T>;
  constexpr int CPR = 128 / EPV;                            // 16-byte chunks per slab row (32 fp32 / 64 fp64)
  constexpr int NCH = BK * CPR / CH_NT;                     // chunks per thread, slab and operand (1 / 2)
  constexpr int RSTEP = CH_NT / CPR;                        // rows between a thread's chunks (16 / 8)
  static_assert(NCH >= 1 && RSTEP % 4 == 0, "slab split");
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));                             // (see chain_writeback)
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wm, wn;
  chain_wave_pos(wave, wm, wn);
  const int nslab = tri ? 2 * wm + 2 : NB / BK;             // slabs this wave multiplies
  T *sA = smem, *sB = smem + SB_OFF;
  const int row0 = tid / CPR, col0 = (tid % CPR) * EPV;
  const char *baseA = reinterpret_cast<const char *>(Ag), *baseB = reinterpret_cast<const char *>(Bg);
  const int64_t stepA = (int64_t)BK * lda * (int64_t)sizeof(T), stepB = (int64_t)BK * ldb * (int64_t)sizeof(T);
  unsigned offA[NCH], offB[NCH];
#pragma unroll
  for (int h = 0; h < NCH; ++h) {
    offA[h] = (unsigned)(((int64_t)(row0 + h * RSTEP) * lda + col0) * (int64_t)sizeof(T));
    offB[h] = (unsigned)(((int64_t)(row0 + h * RSTEP) * ldb + col0) * (int64_t)sizeof(T));
  }
  const int rp0 = (row0 & 3) * 4 + (row0 >> 2);
  T *swA = sA + rp0 * LDT + col0, *swB = sB + rp0 * LDT + col0;
  const int fk = lane >> 4, fm = lane & 15;
  const T *pa0 = sA + fk * 4 * LDT + wm * 32 + fm, *pb0 = sB + fk * 4 * LDT + wn * 64 + fm;
#pragma unroll 1
  for (int bt = 0; bt < NB / (BK * BURST); ++bt) {
    vec_t ra[BURST][NCH], rb[BURST][NCH];
#pragma unroll
    for (int s = 0; s < BURST; ++s) {                       // every load of the burst in flight before the first use
      const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(baseA), 0, 0x7fffffff, 0x00020000);
      const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(baseB), 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int h = 0; h < NCH; ++h) ra[s][h] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rA, offA[h], 0, CH_AUX));
#pragma unroll
      for (int h = 0; h < NCH; ++h) rb[s][h] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rB, offB[h], 0, CH_AUX));
      baseA += stepA;
      baseB += stepB;
    }
    if (bt == 0) pre();
#pragma unroll
    for (int s = 0; s < BURST; ++s) {
      const int buf = s & 1;
      T *wa = swA + buf * (BK * LDT), *wb = swB + buf * (BK * LDT);
#pragma unroll
      for (int h = 0; h < NCH; ++h) {
        *reinterpret_cast<vec_t *>(wa + h * (RSTEP / 4) * LDT) = ra[s][h];
        *reinterpret_cast<vec_t *>(wb + h * (RSTEP / 4) * LDT) = rb[s][h];
      }
      __syncthreads();                                      // slab s visible; every wave is past the reads of slab s - 1
      const T *pa = pa0 + buf * (BK * LDT), *pb = pb0 + buf * (BK * LDT);
      if (bt * BURST + s < nslab) {
#pragma unroll
        for (int ks = 0; ks < BK / 4; ++ks) {
          T a[2], b[4];
#pragma unroll
          for (int t = 0; t < 2; ++t) a[t] = pa[ks * LDT + t * 16];
#pragma unroll
          for (int t = 0; t < 4; ++t) b[t] = pb[ks * LDT + t * 16];
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc.v[mt][nt] = Tr::mfma(a[mt], b[nt], acc.v[mt][nt]);
        }
      }
    }
    __syncthreads();                                        // both stages free (next burst / the caller's epilogue)
  }
}

// The C tile of a read-modify-write operation as 16-byte row chunks (the chunk layout of chain_writeback), all in flight at once.
template <typename T> struct ChainC { typename Traits<T>::vec_t v[2][64 * (128 / Traits<T>::EPV) / CH_NT]; };
template <typename T>
__device__ __forceinline__ void chain_cload(ChainC<T> &vc, const T *Cg, int64_t ldc) {
  using vec_t = typename Traits<T>::vec_t;
  constexpr int EPV = Traits<T>::EPV, CPR = 128 / EPV, NCH = 64 * CPR / CH_NT, RSTEP = CH_NT / CPR;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int row0 = tid / CPR, col0 = (tid % CPR) * EPV;
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(const_cast<T *>(Cg), 0, 0x7fffffff, 0x00020000);
  const unsigned voff = (unsigned)(((int64_t)row0 * ldc + col0) * (int64_t)sizeof(T));
  const unsigned rstep = (unsigned)((int64_t)RSTEP * ldc * (int64_t)sizeof(T));
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int h = 0; h < NCH; ++h)
      vc.v[half][h] = __builtin_bit_cast(vec_t, __builtin_amdgcn_raw_buffer_load_b128(rC, voff + (unsigned)(half * (64 / RSTEP) + h) * rstep, 0, CH_AUX));
}

// C[tile] (op)= acc through the LDS staging area, in two passes of 64 rows (pass h staged by the wave rows wm = 2 h, 2 h + 1), as
// whole 16-byte row chunks; `vc`: the C tile (chain_cload; read only by WB_SUB).  Stores are sc1 (write-through): the tile is
// complete in memory once every wave's vmcnt is 0.
// MODE (run-time, workgroup-uniform, so that the kernel holds ONE copy of this body): WB_STORE / WB_SUB / WB_STORE_NEG as in
// tile_writeback.  The element arithmetic is that of tile_writeback (bit-identical).
template <typename T>
__device__ __forceinline__ void chain_writeback(const Acc<T, 2, 4> &acc, T *Cg, int64_t ldc, T *smem, const int MODE, const ChainC<T> &vc) {
  using vec_t = typename Traits<T>::vec_t;
  typedef int i32x4_t __attribute__((ext_vector_type(4)));
  constexpr int EPV = Traits<T>::EPV, LDW = 132;
  constexpr int CPR = 128 / EPV;                            // chunks per row
  constexpr int NCH = 64 * CPR / CH_NT;                     // chunks per thread and pass (4 fp32 / 8 fp64)
  constexpr int RSTEP = CH_NT / CPR;                        // rows between a thread's chunks (16 / 8)
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));                             // per-operation index arithmetic stays inside the operation (not hoisted out of the op loop into registers)
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wm, wn;
  chain_wave_pos(wave, wm, wn);
  const int row0 = tid / CPR, col0 = (tid % CPR) * EPV;
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(Cg, 0, 0x7fffffff, 0x00020000);
  const unsigned voff = (unsigned)(((int64_t)row0 * ldc + col0) * (int64_t)sizeof(T));
  const unsigned rstep = (unsigned)((int64_t)RSTEP * ldc * (int64_t)sizeof(T));
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();                              // the previous pass is read back
    if ((wm >> 1) == half) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = (wm & 1) * 32 + mt * 16 + Traits<T>::acc_row(lane, r);
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) smem[row * LDW + wn * 64 + nt * 16 + (lane & 15)] = acc.v[mt][nt][r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < NCH; ++h) {
      const vec_t sv = *reinterpret_cast<const vec_t *>(smem + (row0 + h * RSTEP) * LDW + col0);
      const vec_t o = MODE == WB_SUB ? vc.v[half][h] - sv : (MODE == WB_STORE_NEG ? -sv : sv);
      // row-chunk offset in voffset, soffset = 0: the store-data hazard note of gemm_core.hpp
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, o), rC, voff + (unsigned)(half * (64 / RSTEP) + h) * rstep, 0, CH_AUX);
    }
  }
}

// The critical workgroup's fused step between two diagonal blocks: `acc` holds the finished panel tile P = U(r, r+1) (the result of
// chain_mainloop with V_r).  In two passes of 64 rows P goes through the LDS staging area -- in the main loop's slab layout -- and
// from there (a) to memory (write-through, for the pool's updates of the rows below) and (b) straight back into the matrix cores
// as BOTH operands of the next diagonal block's update:  acc2 += P^T P, slab by slab in the order chain_mainloop would take them
// from memory (bit-identical), without the store -> flag -> load round trip of two separate operations.  smem: 64 x LDT elements.
template <typename T>
__device__ __forceinline__ void chain_panel_then_update(const Acc<T, 2, 4> &acc, Acc<T, 2, 4> &acc2, T *Pg, int64_t ldp, T *smem) {
  using Tr = Traits<T>;
  using vec_t = typename Tr::vec_t;
  typedef int i32x4_t __attribute__((ext_vector_type(4)));
  constexpr int EPV = Tr::EPV, CPR = 128 / EPV, NCH = 64 * CPR / CH_NT, RSTEP = CH_NT / CPR;
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int wm, wn;
  chain_wave_pos(wave, wm, wn);
  const int row0 = tid / CPR, col0 = (tid % CPR) * EPV;
  const int fk = lane >> 4, fm = lane & 15;
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(Pg, 0, 0x7fffffff, 0x00020000);
  const unsigned voff = (unsigned)(((int64_t)row0 * ldp + col0) * (int64_t)sizeof(T));
  const unsigned rstep = (unsigned)((int64_t)RSTEP * ldp * (int64_t)sizeof(T));
  // contraction row kk (0..63 inside a pass) -> LDS row: slab kk / 16, inside it (kk % 16 = 4 ks + fk) -> 4 fk + ks (tile_mainloop)
  auto lrow = [](int kk) { return (kk & ~15) + ((kk & 3) << 2) + ((kk & 15) >> 2); };
  const T *pa0 = smem + fk * 4 * LDT + wm * 32 + fm, *pb0 = smem + fk * 4 * LDT + wn * 64 + fm;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (half) __syncthreads();                              // every wave is past the reads of the previous pass
    if ((wm >> 1) == half) {
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int lr = lrow((wm & 1) * 32 + mt * 16 + Tr::acc_row(lane, r));
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) smem[lr * LDT + wn * 64 + nt * 16 + (lane & 15)] = acc.v[mt][nt][r];
        }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < NCH; ++h) {                         // (a) the rows of P to memory
      const vec_t sv = *reinterpret_cast<const vec_t *>(smem + lrow(row0 + h * RSTEP) * LDT + col0);
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(i32x4_t, sv), rP, voff + (unsigned)(half * (64 / RSTEP) + h) * rstep, 0, CH_AUX);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {                           // (b) four slabs of 16 contraction rows
      const T *pa = pa0 + s * (BK * LDT), *pb = pb0 + s * (BK * LDT);
#pragma unroll
      for (int ks = 0; ks < BK / 4; ++ks) {
        T a[2], b[4];
#pragma unroll
        for (int t = 0; t < 2; ++t) a[t] = pa[ks * LDT + t * 16];
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = pb[ks * LDT + t * 16];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int nt = 0; nt < 4; ++nt) acc2.v[mt][nt] = Tr::mfma(a[mt], b[nt], acc2.v[mt][nt]);
      }
    }
  }
  __syncthreads();                                          // staging area free for the write-back of the update
}

// ---- version counters (ints in device memory, one per tile; all accesses relaxed, agent scope)
__device__ __forceinline__ int chain_ld(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void chain_st(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Wait (lane 0 of wave 0 polls, everybody else sits in the barrier) until *p0 >= v0, *p1 >= v1 and *p2 >= v2 (null pointers
// are skipped).  Bounded: after CH_SPIN_TICKS of the 100 MHz wall clock, or as soon as the abort word is raised, the abort
// word is raised and `false` comes back for every thread -- the caller leaves the kernel (the sweep then reports it through
// `info`).  `flag_lds`: one int of LDS.
constexpr long long CH_SPIN_TICKS = 400000000LL;            // 4 s
__device__ __forceinline__ bool chain_wait(const int *p0, int v0, const int *p1, int v1, const int *p2, int v2, int *abort_word, int *flag_lds) {
  if (threadIdx.x == 0) {
    int ok = 1;
    const long long t0 = wall_clock64();
    while (true) {
      const bool r0 = !p0 || chain_ld(p0) >= v0, r1 = !p1 || chain_ld(p1) >= v1, r2 = !p2 || chain_ld(p2) >= v2;
      if (r0 && r1 && r2) break;
      if (chain_ld(abort_word) != 0 || wall_clock64() - t0 > CH_SPIN_TICKS) { chain_st(abort_word, 1); ok = 0; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    *flag_lds = ok;
  }
  __syncthreads();
  const int ok = *flag_lds;
  __syncthreads();                                          // everybody has read the flag before the next wait rewrites it
  return ok != 0;
}
// Publish: every store of this workgroup is complete (sc1: in memory), then the tile's new version.
__device__ __forceinline__ void chain_post(int *p, int v) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) chain_st(p, v);
}

}  // namespace plmc
