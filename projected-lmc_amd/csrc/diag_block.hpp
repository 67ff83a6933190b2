// diag_block.hpp -- factor AND invert one NB x NB (128 x 128) diagonal block inside LDS.
//
// This is the latency-critical step of the blocked Cholesky (one workgroup per latent GP, on the
// critical path of every block row).  The block is processed as 8 x 8 sub-blocks of 16 x 16 by
// right-looking elimination of the augmented matrix [A_kk | I]:
//   (a) the 16 x 16 diagonal sub-block is factored and inverted by ONE wave entirely in registers:
//       lane j < 16 owns column j of the sub-block, lane 16 + c owns column c of the identity part;
//       pivots and multipliers are broadcast with v_readlane (no LDS, no barrier inside);
//   (b) the 16-row panel right of it (and the already-started columns of the inverse) is multiplied
//       by the 16 x 16 inverse on MFMA (v_mfma_*_16x16x4, 4 instructions per 16 x 16 tile);
//   (c) the remaining rows are updated with rank-16 MFMA products (upper tiles of U, live tiles of W).
// Result: U_kk (upper) and W_kk = U_kk^-T (lower), both kept packed-triangular in LDS
// (66 KB fp32 / 132 KB fp64).  3 barriers per sub-block row, 24 in total.
#pragma once
#include "gemm_core.hpp"
#include "covariance.hpp"

namespace plmc {

__device__ __forceinline__ int rowU(int i) { return i * NB - (i * (i - 1)) / 2 - i; }   // U[i][j] at rowU(i)+j, j>=i
__device__ __forceinline__ int rowL(int i) { return (i * (i + 1)) / 2; }                // W[i][c] at rowL(i)+c, c<=i
constexpr int TRI = NB * (NB + 1) / 2;
constexpr int SB = 16;               // sub-block edge
constexpr int NSB = NB / SB;         // 8
constexpr int DIAG_NT = 1024;        // threads of the diagonal-block kernel

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

// 1/sqrt(x) on the serial pivot path: hardware v_rsq seed + Newton steps (y <- y (1.5 - 0.5 x y^2))
// instead of the ~40-instruction IEEE sqrt + divide expansion; result is within ~1 ulp.
__device__ __forceinline__ float rsqrt_refined(float x) {
  float y = __builtin_amdgcn_rsqf(x);
  return y * (1.5f - 0.5f * x * y * y);
}
__device__ __forceinline__ double rsqrt_refined(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - 0.5 * x * y * y);
  return y * (1.5 - 0.5 * x * y * y);
}

// One wave: factor + invert the 16 x 16 sub-block s held in LDS.  Branch-free: every lane runs the
// same instruction stream; loads / stores that do not apply to a lane are redirected to a dummy LDS
// slot (an exec-masked branch per element costs ~10 instructions on the serial path).  A non-positive
// pivot turns into NaN/inf here and is detected from the stored diagonal afterwards.
template <typename T>
__device__ __forceinline__ void factor16(T *sU, int s, int lane) {
  const int o = SB * s;
  const int col = lane & 15;
  const bool isU = lane < 16, isW = (lane >= 16) & (lane < 32);
  const int dummy = 2 * TRI + NB + lane;                // private scratch word of this lane
  T x[SB];
#pragma unroll
  for (int i = 0; i < SB; ++i) {
    const bool vu = isU & (i <= col);
    const T v = sU[vu ? rowU(o + i) + o + col : dummy];
    x[i] = vu ? v : ((isW & (i == col)) ? T(1) : T(0));
  }
#pragma unroll
  for (int k = 0; k < SB; ++k) {
    const T piv = lane_bcast(x[k], k);
    const T inv = rsqrt_refined(piv);
    const T rowk = x[k] * inv;
    x[k] = rowk;
#pragma unroll
    for (int i = k + 1; i < SB; ++i) {
      const T m = lane_bcast(rowk, i);      // U[k][i]
      x[i] -= m * rowk;
    }
  }
#pragma unroll
  for (int i = 0; i < SB; ++i) {
    const bool vu = isU & (i <= col), vw = isW & (i >= col);
    const int idx = vu ? rowU(o + i) + o + col : (vw ? TRI + rowL(o + i) + o + col : dummy);
    sU[idx] = x[i];
  }
}

// grid (q); NT threads (DIAG_NT = 1024: 16 waves share the rank-16 updates).  Wout (may be null): where to store W_kk as a full lower block (ldw).
template <typename T, int DBG = 0, int NT = DIAG_NT>
__global__ __launch_bounds__(NT) void k_diag(T *A, int64_t lda, int64_t strideA, int kblk, T *__restrict__ Vd,
                                                   int64_t strideV, T *Wout, int64_t ldw, int64_t strideW,
                                                   double *__restrict__ logdet, int *__restrict__ info) {
  using Tr = Traits<T>;
  __builtin_amdgcn_s_setprio(3);       // critical path: win issue arbitration against co-resident update tiles
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T *sU = reinterpret_cast<T *>(smem_raw);
  T *sW = sU + TRI;
  const int lat = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  T *blk = A + (int64_t)lat * strideA + (int64_t)kblk * NB * lda + (int64_t)kblk * NB;

  // thread owns column j = tid & 127 and rows (tid >> 7) + 2 * it; the global loads are issued in groups
  // of 8 before their LDS stores so that their latencies overlap instead of adding up
  {
    constexpr int RPP = NT / 128;                        // rows covered per pass
    const int j = tid & 127, i0 = tid >> 7;
#pragma unroll 1
    for (int it0 = 0; it0 < 128 / RPP; it0 += 8) {
      T v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + RPP * (it0 + u);
        v[u] = (j >= i) ? blk[(int64_t)i * lda + j] : T(0);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int i = i0 + RPP * (it0 + u);
        if (j >= i) sU[rowU(i) + j] = v[u];
        if (j <= i) sW[rowL(i) + j] = (i == j) ? T(1) : T(0);
      }
    }
  }
  __syncthreads();

  const int fm = lane & 15, fk = lane >> 4;
  // Schedule: wave 0 owns the critical path -- it updates the next 16 x 16 diagonal tile first and
  // factors it right away, while waves 1..3 apply the rest of the rank-16 update; 2 barriers per step.
  if (wave == 0 && !(DBG & 1)) factor16<T>(sU, 0, lane);
  __syncthreads();
  for (int s = 0; s < NSB; ++s) {
    const int o = SB * s;
    // ---- (b) row panel of sub-block row s: P <- W16 * P   (7 tiles: U columns right, W columns left)
    for (int t = wave; t < ((DBG & 2) ? 0 : NSB - 1); t += NT / 64) {
      const bool isU = t < NSB - 1 - s;
      const int cb = isU ? s + 1 + t : t - (NSB - 1 - s);
      typename Tr::acc_t acc;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = T(0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int k = ks * 4 + fk;
        const T a = (k <= fm) ? sW[rowL(o + fm) + o + k] : T(0);
        const T b = isU ? sU[rowU(o + k) + SB * cb + fm] : sW[rowL(o + k) + SB * cb + fm];
        acc = Tr::mfma(a, b, acc);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = Tr::acc_row(lane, r);
        if (isU) sU[rowU(o + row) + SB * cb + fm] = acc[r];
        else sW[rowL(o + row) + SB * cb + fm] = acc[r];
      }
    }
    __syncthreads();
    // ---- (c) rank-16 update of the rows below: U tiles (t,u), s<t<=u ; W tiles (t,c), c<=s<t
    // round-robin over waves 1..NW-1 with a wrapping counter (an integer modulo per candidate tile
    // cost more than the tile itself)
    int rr = 0;
    auto mine = [&]() { const bool m = (rr + 1 == wave); rr = (rr + 1 == NT / 64 - 1) ? 0 : rr + 1; return m; };
    for (int t = s + 1; t < ((DBG & 4) ? 0 : NSB); ++t) {
      for (int u = t; u < NSB; ++u) {
        const bool crit = (t == s + 1) && (u == t);        // next diagonal tile: wave 0
        if (crit ? wave != 0 : !mine()) continue;
        typename Tr::acc_t acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = Tr::acc_row(lane, r);
          acc[r] = (t < u || fm >= row) ? sU[rowU(SB * t + row) + SB * u + fm] : T(0);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int k = ks * 4 + fk;
          const T a = -sU[rowU(o + k) + SB * t + fm];
          const T b = sU[rowU(o + k) + SB * u + fm];
          acc = Tr::mfma(a, b, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = Tr::acc_row(lane, r);
          if (t < u || fm >= row) sU[rowU(SB * t + row) + SB * u + fm] = acc[r];
        }
      }
      for (int c = 0; c <= s; ++c) {
        if (!mine()) continue;
        typename Tr::acc_t acc;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = sW[rowL(SB * t + Tr::acc_row(lane, r)) + SB * c + fm];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int k = ks * 4 + fk;
          const T a = -sU[rowU(o + k) + SB * t + fm];
          const T b = (c < s || fm <= k) ? sW[rowL(o + k) + SB * c + fm] : T(0);
          acc = Tr::mfma(a, b, acc);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sW[rowL(SB * t + Tr::acc_row(lane, r)) + SB * c + fm] = acc[r];
      }
    }
    if (wave == 0 && s + 1 < NSB && !(DBG & 1)) factor16<T>(sU, s + 1, lane);
    __syncthreads();
  }

  // ---- write back: U_kk (upper part), Vd = W_kk^T (full block, zeros below), optional W_kk (lower)
  T *vd = Vd + (int64_t)lat * strideV + (int64_t)kblk * NB * NB;
  T *wo = Wout ? Wout + (int64_t)lat * strideW : nullptr;
  for (int e = tid; e < NB * NB; e += NT) {
    int i = e >> 7, j = e & 127;
    if (j >= i) {
      blk[(int64_t)i * lda + j] = sU[rowU(i) + j];
      vd[e] = sW[rowL(j) + i];                     // V[i][j] = W[j][i]
    } else {
      vd[e] = T(0);
    }
    if (wo) wo[(int64_t)i * ldw + j] = (j <= i) ? sW[rowL(i) + j] : T(0);
  }
  // log det of the block = 2 sum log(U_ii): one log per thread, fixed-order reduction; a pivot that was
  // not positive left NaN / inf / <= 0 on the diagonal -> report the first one through info.
  const T udiag = tid < NB ? sU[rowU(tid) + tid] : T(1);
  const bool okp = udiag > T(0) && udiag < T(3.0e38);
  double lg = (tid < NB && okp) ? 2.0 * log((double)udiag) : 0.0;
  int badi = (tid < NB && !okp) ? kblk * NB + tid + 1 : 0x7fffffff;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    lg += __shfl_down(lg, off, 64);
    const int ob = __shfl_down(badi, off, 64);
    badi = ob < badi ? ob : badi;
  }
  __syncthreads();
  double *red = reinterpret_cast<double *>(sU);
  int *redb = reinterpret_cast<int *>(red + NT / 64);
  if (lane == 0) { red[wave] = lg; redb[wave] = badi; }
  __syncthreads();
  if (tid == 0) {
    const double lacc = red[0] + red[1];                  // pivots live in threads 0..127 = waves 0, 1
    int bad = redb[0] < redb[1] ? redb[0] : redb[1];
    bad = bad == 0x7fffffff ? 0 : bad;
    if (kblk == 0) { logdet[lat] = lacc; info[lat] = bad; }
    else { logdet[lat] += lacc; if (bad && info[lat] == 0) info[lat] = bad; }
  }
}

}  // namespace plmc
