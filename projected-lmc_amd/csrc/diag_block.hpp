// diag_block.hpp -- factor AND invert one NB x NB (128 x 128) diagonal block.
//
// This is the latency-critical step of the blocked Cholesky (one workgroup per latent GP, on the
// critical path of every block row).  The block is processed as 8 x 8 sub-blocks of 16 x 16 by
// right-looking elimination of the augmented matrix [A_kk | I]:
//   (a) the 16 x 16 diagonal sub-block is factored and inverted by ONE wave entirely in registers:
//       lane j < 16 owns column j of the sub-block, lane 16 + c owns column c of the identity part;
//       pivots and multipliers are broadcast with v_readlane (no LDS, no barrier inside);
//   (b) the 7 other tiles of that sub-block row are multiplied by the 16 x 16 inverse on MFMA;
//   (c) the rows below are updated with rank-16 MFMA products.
// All 64 tiles that change live in the accumulator registers of 7 worker waves for the whole kernel;
// LDS (19 KB fp32) only carries the finished sub-block row to the waves that need it as an operand.
// Result: U_kk (upper) in place, W_kk = U_kk^-T (lower) and its transpose.  2 barriers per sub-block row.
#pragma once
#include "gemm_core.hpp"
#include "covariance.hpp"

namespace plmc {

constexpr int SB = 16;               // sub-block edge
constexpr int NSB = NB / SB;         // 8
constexpr int DIAG_NT = 512;         // threads of the diagonal-block kernel: 8 waves -> 256 registers each

__device__ __forceinline__ float lane_bcast(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ double lane_bcast(double v, int lane) {
  long long b = __builtin_bit_cast(long long, v);
  int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), lane);
  int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  long long r = ((long long)hi << 32) | (unsigned int)lo;
  return __builtin_bit_cast(double, r);
}

// Reciprocal and reciprocal square root of a pivot from the hardware units (v_rcp / v_rsq, 1 ulp in fp32)
// instead of the ~40-instruction IEEE divide / sqrt expansions; fp64 adds Newton steps to full precision.
__device__ __forceinline__ float pivot_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double pivot_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = y * (2.0 - x * y);
  return y * (2.0 - x * y);
}
__device__ __forceinline__ float pivot_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ double pivot_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  return y * (1.5 - 0.5 * x * y * y);
}

// LDS plan of k_diag (elements of T): two row buffers of 8 tiles (tile 0 = W16 of that row, tiles 1..7 the
// other finished tiles), one tile for the factored diagonal sub-block, one all-zero tile, and the
// hand-over buffer [16][32] = [ next diagonal tile | I ].
constexpr int DIAG_TILE = SB * SB;
constexpr int DIAG_ROWBUF = NSB * DIAG_TILE;
constexpr int DIAG_SLD = 2 * SB;
constexpr int DIAG_UBUF = 2 * DIAG_ROWBUF, DIAG_ZERO = DIAG_UBUF + DIAG_TILE, DIAG_SCR = DIAG_ZERO + DIAG_TILE;
constexpr int DIAG_LDS = DIAG_SCR + SB * DIAG_SLD;
constexpr int DIAG_NWK = DIAG_NT / 64 - 1;                 // 7 worker waves (wave 0 factors)

// Wave 0: factor + invert the 16 x 16 sub-block handed over in `S` = [ tile | I ] (row-major 16 x 32; the
// tile's entries below the diagonal are finite garbage and only ever meet other garbage).
// Lane j < 16 owns column j of the sub-block, lane 16 + c column c of the identity; pivots and multipliers
// travel by v_readlane, so there is no LDS traffic and no barrier inside.  Branch-free; a non-positive
// pivot turns into NaN / inf and is found from the diagonal of U afterwards (k_logdet).
// The elimination runs on UNSCALED rows (root-free LDL^T form): the serial chain per pivot is only
// readlane -> v_rcp -> mul -> fma; the 16 rows are scaled by rsqrt(pivot) at the end, off the chain.
//   ub <- U16 as a row-major 16 x 16 tile (entries below the diagonal are garbage)
//   wb <- W16 = U16^-T as a row-major 16 x 16 tile (zeros above the diagonal included)
template <typename T>
__device__ __forceinline__ void factor16(const T *S, T *ub, T *wb, int lane) {
  const int l32 = lane & 31;
  T x[SB], dk[SB];
#pragma unroll
  for (int i = 0; i < SB; ++i) x[i] = S[i * DIAG_SLD + l32];
#pragma unroll
  for (int k = 0; k < SB; ++k) {
    // all broadcasts of step k back to back into distinct SGPRs (left to itself the scheduler emits
    // readlane / wait states / fma one pair at a time through a single SGPR), then the arithmetic
    T m[SB];
    dk[k] = lane_bcast(x[k], k);               // pivot d_k (wave-uniform)
#pragma unroll
    for (int i = k + 1; i < SB; ++i) m[i] = lane_bcast(x[k], i);   // unscaled U~[k][i]
    __builtin_amdgcn_sched_barrier(0);
    const T tk = x[k] * pivot_rcp(dk[k]);      // row k / d_k
#pragma unroll
    for (int i = k + 1; i < SB; ++i) x[i] -= m[i] * tk;
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int k = 0; k < SB; ++k) x[k] *= pivot_rsqrt(dk[k]);
  if (lane < 32) {
    T *dst = (lane < 16 ? ub : wb) + (lane & 15);
#pragma unroll
    for (int i = 0; i < SB; ++i) dst[i * SB] = x[i];
  }
}

// grid (q); 512 threads = wave 0 (the serial 16 x 16 factor/invert steps) + 7 worker waves.
//
// The block is the augmented matrix [A_kk | I] of 8 x 8 sub-blocks of 16 x 16, eliminated right-looking.
// Sub-block row t has 8 tiles that ever change: the diagonal tile U(t,t) and 7 others (U(t,t+1..7) and
// W(t,0..t-1); W(t,t) comes straight out of the 16 x 16 inversion).  Worker w owns ONE of the 7 others in
// every row (accumulator slot t: tile j = 1 + (w + t) mod 7 of row t) plus one diagonal tile (slot 8:
// t_d = w, or 7 for worker 0), all kept NEGATED in MFMA accumulators for the whole kernel, so that a
// rank-16 update is a plain accumulate and the work of every step is spread evenly.  The worker code is
// fully unrolled over the sub-block rows: every accumulator index is static, and a tile that is not live
// in some step is pointed at an all-zero LDS tile instead of being branched around (a conditionally
// updated accumulator costs register copies at every join).  Per sub-block row s:
//   wave 0 : waits for the hand-over flag, factors/inverts tile (s,s) -> LDS                     | barrier 1
//   workers: apply finished row s-1 to their tiles of rows >= s                                  |
//   workers: multiply their row-s tile by W16 (the accumulator is fed back as the B operand, k permuted
//            consistently in A), publish it in LDS; wave 0 stores the diagonal tiles             | barrier 2
//   owner of (s+1,s+1): applies row s to it, hands it to wave 0 through LDS and raises the flag;
//   then every worker stores its finished row-s tile to global memory (off the critical path).
// Result: U_kk (upper) in place, Vd = W_kk^T and (optionally) W_kk: lower / diagonal tiles only -- the
// caller zeroes the other tiles once per sweep.  log det and the pivot check are done by k_logdet.
// SC1 forms of the body's global accesses, for a caller whose inputs / outputs cross workgroups inside one launch (k_chain): loads
// that bypass the CU's L1 and write-through stores (relaxed, agent scope: global_load / global_store ... sc1), so that neither an
// acquire nor a release fence is needed around the body (chain_engine.hpp).
template <bool SC1, typename T> __device__ __forceinline__ T dg_ld(const T *p) {
  if constexpr (SC1) {
    if constexpr (sizeof(T) == 4) return __builtin_bit_cast(T, __hip_atomic_load(reinterpret_cast<const int *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    else return __builtin_bit_cast(T, __hip_atomic_load(reinterpret_cast<const long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  } else return *p;
}
template <bool SC1, typename T> __device__ __forceinline__ void dg_st(T *p, T v) {
  if constexpr (SC1) {
    if constexpr (sizeof(T) == 4) __hip_atomic_store(reinterpret_cast<int *>(p), __builtin_bit_cast(int, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else __hip_atomic_store(reinterpret_cast<long long *>(p), __builtin_bit_cast(long long, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else *p = v;
}

// The body as a device function (all 512 threads of the workgroup; `smem`: DIAG_LDS elements, `handover_p`: one int of LDS):
// k_diag below is one call of it per launch, the resident chain kernel of the sweep (potrf.hip, k_chain) calls it once per
// block row of a group.  blk: the diagonal block (leading dimension ldu), vd: its 128 x 128 inverse-transpose output
// (leading dimension NB), wo: W_kk output (leading dimension ldwu) or nullptr.  Both roles (factor wave / workers) leave
// through the end of the function with the same number of barriers executed.
template <typename T, int DBG = 0, bool SC1 = false>
__device__ __forceinline__ void diag_body(T *blk, const unsigned ldu, T *__restrict__ vd, T *wo, const unsigned ldwu, T *smem, int *handover_p) {
  using Tr = Traits<T>;
  using acc_t = typename Tr::acc_t;
  int &handover = *handover_p;         // highest sub-block row whose diagonal tile is ready in the hand-over buffer
  T *scr = smem + DIAG_SCR;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6) - 1;          // worker id, -1 = factor wave
  const int fm = lane & 15, fk = lane >> 4, lo = fk * SB + fm;

  if (w < 0) {
    // =========================== wave 0: the serial chain ===========================
    __builtin_amdgcn_s_setprio(3);     // above its own workers, which are above co-resident update tiles
    {                                  // hand-over buffer for s = 0: [ A(0,0) | I ]
      T v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = dg_ld<SC1>(blk + ((unsigned)(fk + 4 * r) * ldu + (unsigned)fm));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        scr[(fk + 4 * r) * DIAG_SLD + fm] = v[r];
        scr[(fk + 4 * r) * DIAG_SLD + SB + fm] = (fk + 4 * r) == fm ? T(1) : T(0);
        smem[DIAG_ZERO + lo + r * 4 * SB] = T(0);
      }
      if (lane == 0) handover = 0;
    }
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < NSB; ++s) {
      const int o = SB * s;
      int ls = lane;
      asm volatile("" : "+v"(ls));     // keeps factor16's lane masks out of the loop-invariant registers
      T *rb_cur = smem + (s & 1) * DIAG_ROWBUF, *ub = smem + DIAG_UBUF;
      if (s > 0) {
        while (__atomic_load_n(&handover, __ATOMIC_ACQUIRE) < s) __builtin_amdgcn_s_sleep(1);
      }
      if (!(DBG & 1)) factor16<T>(scr, ub, rb_cur, ls);
      __syncthreads();
      // idle until the next hand-over: store the diagonal tiles U16 (upper), W16 and V16 = W16^T
      if (!(DBG & 1)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = fk + 4 * r, col = fm;
          const T uv = ub[row * SB + col], wv = rb_cur[row * SB + col], wt = rb_cur[col * SB + row];
          if (col >= row) blk[(unsigned)(o + row) * ldu + (unsigned)(o + col)] = uv;
          dg_st<SC1>(vd + ((o + row) * NB + o + col), wt);
          if (wo) dg_st<SC1>(wo + ((unsigned)(o + row) * ldwu + (unsigned)(o + col)), wv);
        }
      }
      __syncthreads();
    }
    __builtin_amdgcn_s_setprio(0);
    return;
  }

  // =========================== workers ===========================
  __builtin_amdgcn_s_setprio(2);
  const int td = w == 0 ? NSB - 1 : w;                                  // row of this worker's diagonal tile
  // tile j (1..7) of row t owned by this worker
  auto tile_j = [&](int t) { const int x = w + t; return 1 + (x >= DIAG_NWK ? (x >= 2 * DIAG_NWK ? x - 2 * DIAG_NWK : x - DIAG_NWK) : x); };
  acc_t acc[NSB + 1];
  {
    T ld[NSB + 1][4];
#pragma unroll
    for (int t = 0; t <= NSB; ++t) {
      const int tt = t < NSB ? t : td, jj = t < NSB ? tile_j(t) : 0;
      const bool isU = jj < NSB - tt;
      const int u = isU ? tt + jj : 0;                                  // W tiles: any valid address, value unused
#pragma unroll
      for (int r = 0; r < 4; ++r)
        ld[t][r] = dg_ld<SC1>(blk + ((unsigned)(SB * tt + Tr::acc_row(lane, r)) * ldu + (unsigned)(SB * u + fm)));
    }
#pragma unroll
    for (int t = 0; t <= NSB; ++t) {
      const int tt = t < NSB ? t : td, jj = t < NSB ? tile_j(t) : 0;
      const bool isU = jj < NSB - tt;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool keep = isU && (jj > 0 || fm >= Tr::acc_row(lane, r));
        acc[t][r] = keep ? -ld[t][r] : T(0);
      }
    }
  }
  __syncthreads();

#pragma unroll
  for (int s = 0; s < NSB; ++s) {
    const int o = SB * s;
    const int RBC = (s & 1) * DIAG_ROWBUF, RBP = ((s ^ 1) & 1) * DIAG_ROWBUF;
    if (s > 0 && !(DBG & 4)) {
      // ---- apply finished row sr = s - 1:  N(t, j) += U(sr, t)^T * [ U(sr, u) | W(sr, c) ]
      const int sr = s - 1;
      T av[NSB + 1][4], bv[NSB + 1][4];
#pragma unroll
      for (int t = s; t <= NSB; ++t) {
        const int tt = t < NSB ? t : td, jj = t < NSB ? tile_j(t) : 0;
        const bool isU = jj < NSB - tt;
        const int uc = isU ? tt + jj : jj - (NSB - tt);
        // the diagonal tile of row s was brought up to date at the hand-over; rows < s are finished
        const bool live = (t < NSB || td > s) && (isU || uc <= sr);
        const int bt = isU ? uc - sr : (uc == sr ? 0 : NSB - sr + uc);
        const int aoff = live ? RBP + (tt - sr) * DIAG_TILE : DIAG_ZERO;
        const int boff = live ? RBP + bt * DIAG_TILE : DIAG_ZERO;
        const T *pa = smem + aoff + lo, *pb = smem + boff + lo;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { av[t][ks] = pa[ks * 4 * SB]; bv[t][ks] = pb[ks * 4 * SB]; }
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int t = s; t <= NSB; ++t) acc[t] = Tr::mfma(av[t][ks], bv[t][ks], acc[t]);
    }
    __syncthreads();
    // ---- row panel: tile <- W16 * tile = (-W16) * N, published for the rows below
    const int pj = tile_j(s);
    if (!(DBG & 2)) {
      acc_t p;
      T wv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) { p[r] = T(0); wv[r] = -smem[RBC + fm * SB + Tr::acc_row(lane, r)]; }
#pragma unroll
      for (int r = 0; r < 4; ++r) p = Tr::mfma(wv[r], acc[s][r], p);
#pragma unroll
      for (int r = 0; r < 4; ++r) smem[RBC + pj * DIAG_TILE + Tr::acc_row(lane, r) * SB + fm] = p[r];
      acc[s] = p;                                                      // finished: kept for the global store
    }
    __syncthreads();
    // ---- hand the next diagonal tile to wave 0 (no barrier: wave 0 polls the flag, the workers move on)
    if (s + 1 < NSB && td == s + 1) {
      if (!(DBG & 4)) {
        const T *pa = smem + RBC + DIAG_TILE + lo;                     // U(s, s+1) is both operands
        T av[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) av[ks] = pa[ks * 4 * SB];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc[NSB] = Tr::mfma(av[ks], av[ks], acc[NSB]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) scr[Tr::acc_row(lane, r) * DIAG_SLD + fm] = -acc[NSB][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __atomic_store_n(&handover, s + 1, __ATOMIC_RELEASE);
    }
    // ---- store this worker's finished tile of row s
    if (!(DBG & 2)) {
      const bool isU = pj < NSB - s;
      const int uc = isU ? s + pj : pj - (NSB - s);
      if (isU) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          blk[(unsigned)(o + Tr::acc_row(lane, r)) * ldu + (unsigned)(SB * uc + fm)] = acc[s][r];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) dg_st<SC1>(vd + ((SB * uc + fm) * NB + o + Tr::acc_row(lane, r)), acc[s][r]);
        if (wo) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            dg_st<SC1>(wo + ((unsigned)(o + Tr::acc_row(lane, r)) * ldwu + (unsigned)(SB * uc + fm)), acc[s][r]);
        }
      }
    }
  }
  __builtin_amdgcn_s_setprio(0);
}

template <typename T, int DBG = 0, int NT = DIAG_NT>
__global__ __launch_bounds__(NT) void k_diag(T *A, int64_t lda, int64_t strideA, int kblk, T *__restrict__ Vd,
                                             int64_t strideV, T *Wout, int64_t ldw, int64_t strideW) {
  static_assert(NT == 512, "tile ownership is laid out for 8 waves");
  __shared__ __align__(16) T smem[DIAG_LDS];
  __shared__ int handover;
  const int lat = blockIdx.x;
  T *blk = A + (int64_t)lat * strideA + (int64_t)kblk * NB * lda + (int64_t)kblk * NB;
  T *vd = Vd + (int64_t)lat * strideV + (int64_t)kblk * NB * NB;
  T *wo = Wout ? Wout + (int64_t)lat * strideW : nullptr;
  diag_body<T, DBG>(blk, (unsigned)lda, vd, wo, (unsigned)ldw, smem, &handover);   // offsets inside the block fit 32 bits
}

// log det = 2 sum log(U_ii) and the pivot check of a finished sweep, from the diagonal of U (a pivot that
// was not positive left NaN / inf / <= 0 behind; info = 1 + index of the first one, 0 if none).
// grid (q), 256 threads; fixed-order reduction.
// `chain_ctl` (optional): the control words of the resident chain kernel in the sweep's scratch (potrf.hip, k_chain), one
// set every `ctl_stride` elements for `nlat` latents (a launch of the chain kernel uses the set of its first latent): a raised abort word (a bounded spin ran out -- never seen; it would mean a workgroup of the
// chain was not resident) is reported as info = PLMC_INFO_CHAIN_ABORT instead of a pivot index.
constexpr int INFO_CHAIN_ABORT = 0x7ffffff0;
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_logdet(const T *__restrict__ A, int64_t n_pad, int64_t lda, int64_t strideA,
                                                      double *__restrict__ logdet, int *__restrict__ info, const T *chain_ctl = nullptr,
                                                      int64_t ctl_stride = 0, int nlat = 0) {
  __shared__ double red[NTHREADS];
  __shared__ int redb[NTHREADS];
  const int lat = blockIdx.x, tid = threadIdx.x;
  const T *Al = A + (int64_t)lat * strideA;
  double lg = 0.0;
  int bad = 0x7fffffff;
  for (int64_t i0 = tid; i0 < n_pad; i0 += 8 * NTHREADS) {       // eight diagonal entries in flight per thread (26 -> 19 us)
    T dv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t i = i0 + u * NTHREADS;
      const int64_t ic = i < n_pad ? i : n_pad - 1;
      dv[u] = Al[ic * lda + ic];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t i = i0 + u * NTHREADS;
      if (i < n_pad) {
        const T d = dv[u];
        const bool ok = d > T(0) && d < T(3.0e38);
        if (ok) lg += 2.0 * log((double)d);
        else bad = (int)i + 1 < bad ? (int)i + 1 : bad;
      }
    }
  }
  red[tid] = lg;
  redb[tid] = bad;
  __syncthreads();
  for (int o = NTHREADS / 2; o > 0; o >>= 1) {
    if (tid < o) { red[tid] += red[tid + o]; redb[tid] = redb[tid + o] < redb[tid] ? redb[tid + o] : redb[tid]; }
    __syncthreads();
  }
  if (tid == 0) {
    logdet[lat] = red[0];
    info[lat] = redb[0] == 0x7fffffff ? 0 : redb[0];
    if (chain_ctl) {                       // any launch of the chain kernel that gave up poisons the whole batch
      bool ab = false;
      for (int l = 0; l < nlat; ++l) ab = ab || reinterpret_cast<const int *>(chain_ctl + (int64_t)l * ctl_stride)[1] != 0;
      if (ab) info[lat] = INFO_CHAIN_ABORT;
    }
  }
}

// Zero the tiles of the diagonal-block outputs that k_diag never writes: Vd tiles (a, b) with a > b (blocks
// kb < nvd) and W_kk tiles (b, a) above the diagonal (16 x 16 tiles of the diagonal blocks kb < nwd of a matrix with
// leading dimension ldw, one block every wdiag_step elements).  grid (max(nvd, nwd), q).
// `pad_col` > 0: also zero the control area of the resident chain kernel (potrf.hip, k_chain) -- the first 128 elements of rows 0
// and 1 of the matrix at Wd behind column pad_col (the pad column of the group scratch).
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_zero_diag_out(T *__restrict__ Vd, int64_t strideV, int nvd, T *Wd, int64_t ldw,
                                                             int64_t strideW, int64_t wdiag_step, int nwd, int64_t pad_col = 0) {
  const int kb = blockIdx.x, lat = blockIdx.y;
  if (Wd && pad_col > 0 && kb == 0) {
    T *c0 = Wd + (int64_t)lat * strideW + pad_col;
    if (threadIdx.x < 128) { c0[threadIdx.x] = T(0); c0[ldw + threadIdx.x] = T(0); }
  }
  T *vd = kb < nvd ? Vd + (int64_t)lat * strideV + (int64_t)kb * NB * NB : nullptr;
  T *wo = (Wd && kb < nwd) ? Wd + (int64_t)lat * strideW + (int64_t)kb * wdiag_step : nullptr;
  for (int e = threadIdx.x; e < NB * NB; e += NTHREADS) {
    const int i = e >> 7, j = e & 127;
    if (vd && (i >> 4) > (j >> 4)) vd[e] = T(0);
    if (wo && (j >> 4) > (i >> 4)) wo[(int64_t)i * ldw + j] = T(0);
  }
}

}  // namespace plmc
