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

// One wave: factor + invert the 16 x 16 sub-block s held in LDS.  Returns through lacc/bad.
template <typename T>
__device__ __forceinline__ void factor16(T *sU, T *sW, T *sPiv, int s, int lane, int gbase, int &bad) {
  const int o = SB * s;
  T x[SB];
  const int col = lane & 15;
#pragma unroll
  for (int i = 0; i < SB; ++i) {
    T v = T(0);
    if (lane < 16) { if (i <= col) v = sU[rowU(o + i) + o + col]; }
    else if (lane < 32) { v = (i == col) ? T(1) : T(0); }
    x[i] = v;
  }
#pragma unroll
  for (int k = 0; k < SB; ++k) {
    const T piv = lane_bcast(x[k], k);
    const bool ok = piv > T(0);
    const T inv = ok ? T(1) / dsqrt(piv) : T(1);
    if (lane == 0) sPiv[o + k] = ok ? piv : T(1);      // logs are taken in parallel after the sweep
    if (!ok && !bad) bad = gbase + o + k + 1;
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
    if (lane < 16) { if (i <= col) sU[rowU(o + i) + o + col] = x[i]; }
    else if (lane < 32) { if (i >= col) sW[rowL(o + i) + o + col] = x[i]; }
  }
}

// grid (q); 256 threads.  Wout (may be null): where to store W_kk as a full lower block (ldw).
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_diag(T *A, int64_t lda, int64_t strideA, int kblk, T *__restrict__ Vd,
                                                   int64_t strideV, T *Wout, int64_t ldw, int64_t strideW,
                                                   double *__restrict__ logdet, int *__restrict__ info) {
  using Tr = Traits<T>;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T *sU = reinterpret_cast<T *>(smem_raw);
  T *sW = sU + TRI;
  T *sPiv = sW + TRI;
  const int lat = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  T *blk = A + (int64_t)lat * strideA + (int64_t)kblk * NB * lda + (int64_t)kblk * NB;

  for (int e = tid; e < NB * NB; e += NTHREADS) {
    int i = e >> 7, j = e & 127;
    if (j >= i) sU[rowU(i) + j] = blk[(int64_t)i * lda + j];
    if (j <= i) sW[rowL(i) + j] = (i == j) ? T(1) : T(0);
  }
  __syncthreads();

  int bad = 0;
  const int fm = lane & 15, fk = lane >> 4;
  for (int s = 0; s < NSB; ++s) {
    const int o = SB * s;
    if (wave == 0) factor16<T>(sU, sW, sPiv, s, lane, kblk * NB, bad);
    __syncthreads();
    // ---- (b) row panel of sub-block row s: P <- W16 * P   (7 tiles: U columns right, W columns left)
    for (int t = wave; t < NSB - 1; t += 4) {
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
    int idx = 0;
    for (int t = s + 1; t < NSB; ++t) {
      for (int u = t; u < NSB; ++u, ++idx) {
        if ((idx & 3) != wave) continue;
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
      for (int c = 0; c <= s; ++c, ++idx) {
        if ((idx & 3) != wave) continue;
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
    __syncthreads();
  }

  // ---- write back: U_kk (upper part), Vd = W_kk^T (full block, zeros below), optional W_kk (lower)
  T *vd = Vd + (int64_t)lat * strideV + (int64_t)kblk * NB * NB;
  T *wo = Wout ? Wout + (int64_t)lat * strideW : nullptr;
  for (int e = tid; e < NB * NB; e += NTHREADS) {
    int i = e >> 7, j = e & 127;
    if (j >= i) {
      blk[(int64_t)i * lda + j] = sU[rowU(i) + j];
      vd[e] = sW[rowL(j) + i];                     // V[i][j] = W[j][i]
    } else {
      vd[e] = T(0);
    }
    if (wo) wo[(int64_t)i * ldw + j] = (j <= i) ? sW[rowL(i) + j] : T(0);
  }
  // log det of the block = sum of log(pivot): one log per thread, then a fixed-order reduction
  double lg = tid < NB ? log((double)sPiv[tid]) : 0.0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) lg += __shfl_down(lg, off, 64);
  __syncthreads();
  double *red = reinterpret_cast<double *>(sU);
  if (lane == 0) red[wave] = lg;
  __syncthreads();
  if (tid == 0) {
    const double lacc = red[0] + red[1];
    if (kblk == 0) { logdet[lat] = lacc; info[lat] = bad; }
    else { logdet[lat] += lacc; if (bad && info[lat] == 0) info[lat] = bad; }
  }
}

}  // namespace plmc
