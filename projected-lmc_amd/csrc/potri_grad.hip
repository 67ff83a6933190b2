// potri_grad.hip -- K^-1 = W^T W on MFMA with the analytic MLL gradient fused into the epilogue.
//
// Replaces the autograd backward of cholesky + kernel chain that `loss.backward()`
// (experiments.py:270) runs through gpytorch:
//     d logp / d theta = 1/2 tr((alpha alpha^T - Khat^-1) dKhat/dtheta),   d logp / d y = -alpha.
// For each upper tile (ib <= jb) of K^-1 the workgroup accumulates the tile on MFMA, then,
// without writing it to HBM, re-derives dK/dtheta for the same (i,j) from X staged in LDS and
// reduces (alpha_i alpha_j - Kinv_ij) * dK_ij to d+2 partial sums.  A second tiny kernel sums the
// per-tile partials in a fixed order (deterministic, no float atomics).
#include <stdlib.h>
#include "api_common.hpp"
#include "covariance.hpp"
#include "bf3_engine.hpp"
#include "../../include/plmc.h"

namespace plmc {

constexpr int GP = MAX_DIM + 2;      // partial-sum slots per tile: d lengthscales, noise, outputscale

// ---- gradient epilogue for d <= 8 (DCAP in {4, 8}): the metric shape and every BASELINE config but the SARCOS one.
// The 64 accumulator registers stay live through the whole epilogue, so at 4 waves per SIMD (128 registers) the
// epilogue itself has to fit ~60: round 1's version kept 16 packed column pairs + 16 packed sums + 8 row values and
// spilled 80 bytes per lane (WRITE_SIZE 3.4 GB per launch for 9 MB of output).  Here the packed pairs are two
// DIMENSIONS of one element instead of two elements: 4 packed sums, 4 packed column values, 4 packed differences
// (kept for the second use: no recomputation), one scalar transcendental per element -- the same number of vector
// instructions per element, half the registers, no scratch (tools/resource_usage.py, profiles/r02_resource_usage.md).
// The loops are real loops: the 16 accumulator registers of one sub-tile row mt are parked in LDS -- each thread reads
// back only what it wrote, no barrier -- so that (nt, r) can be runtime indices.
// LDS plan (T elements), all inside tile_smem_elems():
//   ui [128][DCAP + 4], uj [128][DCAP + 4]   scaled inputs u = x / ell of the tile's rows / columns (0 beyond n and d)
//   ai [128], aj [128]                       alpha of rows / columns
//   accs[4 nt][256 threads][4 r]             the parked accumulator slice
template <int DCAP> struct GradLds {
  static constexpr int LDI = DCAP + 4;
  static constexpr int UI = 0, UJ = UI + NB * LDI, AI = UJ + NB * LDI, AJ = AI + NB, ACCS = AJ + NB, END = ACCS + 4 * NTHREADS * 4;
};

// INTERIOR: tile strictly above the diagonal, no padded rows or columns, nothing stored -- every element is live and
// counts twice, no per-element predicate.  Otherwise (diagonal tiles, ragged edge, K^-1 or its diagonal wanted):
//   weight 2 above the diagonal, 1 on it (where df = 0, so it adds nothing to the lengthscale sums), 0 below / outside n.
//     g[k] += os sum_ij wt w base df_k^2,   g_os += sum wt w val,   g_noise += sum_ii w,   w = alpha_i alpha_j - Kinv_ij.
template <typename T, int DCAP, int KIND, bool INTERIOR>
__device__ __forceinline__ void grad_tile_small(const Acc<T> &acc, T *smem, const int tid, T os, int ib, int jb, int n, int lat, int64_t n_pad,
                                                T *Kinv, int64_t ldk, int64_t strideK, T *kinv_diag, T (&g)[DCAP],
                                                T &g_noise, T &g_os) {
  typedef Pair<T> T2;
  typedef GradLds<DCAP> L;
  constexpr int NP = DCAP / 2;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  T2 g2[NP];
#pragma unroll
  for (int j = 0; j < NP; ++j) g2[j] = T2{T(0), T(0)};
  T gos = T(0), gnz = T(0);
  T *accs = smem + L::ACCS + tid * 4;
#pragma unroll 1
  for (int mt = 0; mt < 4; ++mt) {
    // park acc.v[mt][0..3] (static register indices per case)
#define PLMC_PARK(M)                                                                                    \
  _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) _Pragma("unroll") for (int r = 0; r < 4; ++r)         \
      accs[nt * NTHREADS * 4 + r] = acc.v[M][nt][r];
    if (mt == 0) { PLMC_PARK(0) } else if (mt == 1) { PLMC_PARK(1) } else if (mt == 2) { PLMC_PARK(2) } else { PLMC_PARK(3) }
#undef PLMC_PARK
#pragma unroll 1
    for (int nt = 0; nt < 4; ++nt) {
      const int col = tile_col(wn, nt, lane);
      const T *ujc = smem + L::UJ + col * L::LDI;
      T2 u2[NP];
#pragma unroll
      for (int j = 0; j < NP; ++j) u2[j] = *reinterpret_cast<const T2 *>(ujc + 2 * j);
      const T a_j = smem[L::AJ + col];
      const int gj = jb * NB + col;
#pragma unroll 1
      for (int r = 0; r < 4; ++r) {
        const int row = tile_row<T>(wm, mt, lane, r);
        const T kin = accs[nt * NTHREADS * 4 + r];
        const T *uir = smem + L::UI + row * L::LDI;
        T2 df[NP];
        T2 r2p = {T(0), T(0)};
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          df[j] = *reinterpret_cast<const T2 *>(uir + 2 * j) - u2[j];
          r2p += df[j] * df[j];
        }
        T val, base;
        if constexpr (KIND == K_SPLINE) {                // no lengthscale: the value only feeds the outputscale sum
          val = T(1);
#pragma unroll
          for (int j = 0; j < NP; ++j) {
            const T2 xr = *reinterpret_cast<const T2 *>(uir + 2 * j);
            val *= spline_factor(xr.x, u2[j].x) * spline_factor(xr.y, u2[j].y);
          }
          base = T(0);
        } else {
          kern_value_base_fast(KIND, r2p.x + r2p.y, val, base);
        }
        T w = smem[L::AI + row] * a_j - kin;
        if (!INTERIOR) {
          const int gi = ib * NB + row;
          if (Kinv && gj >= gi) Kinv[(int64_t)lat * strideK + (int64_t)gi * ldk + gj] = kin;
          if (kinv_diag && gi == gj) kinv_diag[(int64_t)lat * n_pad + gi] = kin;
          const bool live = gi < n && gj < n && gj >= gi;
          if (live && gi == gj) gnz += w;
          w = live ? (gj > gi ? T(2) * w : w) : T(0);
        }
        gos += w * val;
        const T c = w * base;
#pragma unroll
        for (int j = 0; j < NP; ++j) g2[j] += (c * df[j]) * df[j];
      }
    }
  }
  const T sc = INTERIOR ? T(2) * os : os;
#pragma unroll
  for (int j = 0; j < NP; ++j) {
    g[2 * j] += sc * g2[j].x;
    g[2 * j + 1] += sc * g2[j].y;
  }
  g_os += INTERIOR ? T(2) * gos : gos;
  g_noise += gnz;
}

// occupancy floor: fp32 with up to 8 input dimensions fits 128 registers (4 waves per SIMD, as the update kernels)
template <typename T, int DCAP> constexpr int KG_MIN_WAVES = sizeof(T) == 8 ? 2 : (DCAP <= 8 ? 4 : (DCAP <= 16 ? 2 : 1));

// ---- gradient epilogue of one 128 x 128 tile (ib, jb) of K^-1 held in `acc` (the accumulator layout of the tile engines),
// as a function: for the halves of the 512-thread macro-tile kernel (tid = threadIdx.x & 255; each half on its own tile with
// its own `smem` of tile_smem_elems<T>() elements).  Both halves execute the same barriers; a half without a tile
// (live = false) stages zeros and writes nothing.  Body: kinv_epilogue.inc.
template <typename T, int DCAP, bool SPLINE>
__device__ __forceinline__ void kinv_tile_epilogue(const Acc<T> &acc, T *smem, const int tid, const bool live, int kind, int ib, int jb, int lat,
                                                   int m, int64_t n_pad, const T *__restrict__ alpha, const T *__restrict__ X, int n, int d,
                                                   const T *__restrict__ ell, const T *__restrict__ oscale, T *Kinv, int64_t ldk,
                                                   int64_t strideK, T *kinv_diag, double *__restrict__ partials, int plain) {
  if (!live) { n = 0; Kinv = nullptr; kinv_diag = nullptr; }       // every element predicate below is then false
  // 512 threads = two waves per SIMD = 256 registers: with 32 dimensions the one-pass form does not fit beside the accumulators
#define PLMC_KINV_KEEP (DCAP <= 16)
#include "kinv_epilogue.inc"
#undef PLMC_KINV_KEEP
}

// SPLINE (general epilogue only): the product-form spline kernel gets its own instantiation, so that its extra live
// values do not raise the register pressure (and the scratch) of the stationary kernels' code.
template <typename T, int DCAP, bool SPLINE = false>
__global__ __launch_bounds__(NTHREADS, (KG_MIN_WAVES<T, DCAP>)) void k_kinv_grad(int kind, const T *__restrict__ W, int64_t n_pad, int64_t ldw,
                                                         int64_t strideW, const T *__restrict__ alpha,
                                                         const T *__restrict__ X, int n, int d,
                                                         const T *__restrict__ ell, const T *__restrict__ oscale,
                                                         T *Kinv, int64_t ldk, int64_t strideK, T *kinv_diag,
                                                         double *__restrict__ partials, int nlat, int plain) {
  const int m = (int)(n_pad / NB);
  int lat, ib, jb;
  if (plain >= 4) {                    // longest tiles first: jb ascending outermost, latent fastest
    const int w = blockIdx.x;
    lat = w % nlat;
    const int t = w / nlat;
    jb = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
    while ((jb + 1) * (jb + 2) / 2 <= t) ++jb;
    while (jb * (jb + 1) / 2 > t) --jb;
    ib = t - jb * (jb + 1) / 2;
  } else if (plain) {                  // (jb, ib, lat) grid
    jb = blockIdx.x; ib = blockIdx.y; lat = blockIdx.z;
    if (jb < ib) return;
  } else if (!xcd_tri_decode(blockIdx.x, m, nlat, lat, ib, jb)) return;   // XCD-dealt 8 x 8 super-tiles
  __shared__ __align__(16) T smem[tile_smem_elems<T>()];
  const T *Wl = W + (int64_t)lat * strideW + (int64_t)jb * NB * ldw;
  Acc<T> acc;
  acc.zero();
  tile_mainloop<T, false, true>(acc, Wl + (int64_t)ib * NB, ldw, Wl + (int64_t)jb * NB, ldw, (int)(n_pad - (int64_t)jb * NB), smem);
  // the epilogue body is textually included, not called: as an (always inlined) function it cost k_kinv_grad<float, 8>, which
  // sits at exactly 128 registers, 24 bytes of scratch per lane
  const int tid = threadIdx.x;
  constexpr bool live = true;
#define PLMC_KINV_KEEP (sizeof(T) == 4)
#include "kinv_epilogue.inc"
#undef PLMC_KINV_KEEP
}

// The same on the 16-bit matrix cores (fp32 only; bf3_engine.hpp): a workgroup of 512 threads takes the macro tile
// (ib, ib + 1) x jb of K^-1 = W^T W from the k8-ordered planes of W (`Wp`, n_pad columns per plane row, `wp_lat_stride` elements
// per latent: the planes the sweep left in its Vd scratch, or the ones k_split_w writes behind the partials), then each half of
// 256 threads runs the gradient epilogue on its own tile.
// `wscale`: per latent the power-of-two scale the planes of W were written with (SplitH2; SplitB3: ones).
template <class S, int DCAP, bool SPLINE = false>
__global__ __launch_bounds__(B3_NT, 2) void k_kinv_grad_bf3(int kind, int64_t n_pad, const float *__restrict__ alpha, const float *__restrict__ X, int n,
                                                            int d, const float *__restrict__ ell, const float *__restrict__ oscale, float *Kinv,
                                                            int64_t ldk, int64_t strideK, float *kinv_diag, double *__restrict__ partials, int nlat,
                                                            int plain, const unsigned short *__restrict__ Wp, const float *__restrict__ wscale,
                                                            int64_t wp_lat_stride, int64_t ws_stride) {
  constexpr int LDS_BYTES = b3_lds_bytes<S>() > 2 * tile_smem_elems<float>() * (int)sizeof(float) ? b3_lds_bytes<S>() : 2 * tile_smem_elems<float>() * (int)sizeof(float);
  __shared__ __align__(16) unsigned char lds[LDS_BYTES];
  const int m = (int)(n_pad / NB);
  // latent-major: the ~256 resident workgroups are consecutive macro tiles of ONE matrix (a few block columns jb, all their
  // ibm): 16 + 16 operand strips instead of one B and 32 A strips per XCD and latent -- the strips are shared across the XCDs
  // through the Infinity Cache (step 18.5 -> 18.1 ms at q = 8; PLMC_KINV_ORDER=7: latent fastest, the fp32 kernel's order)
  // Tile order: XCD-dealt super-blocks of 4 macro rows x 8 block columns (= the 32 workgroups an XCD holds; workgroup w lands on
  // XCD w % 8), longest K range first, latent by latent.  With the K range walked from its END (every range ends at row n) the
  // 32 tiles of a block read their 4 A strips and 8 B strips in lockstep through one L2, and the blocks in flight on the eight
  // XCDs belong to one or two matrices (Infinity Cache).  PMC, q = 8: 12.7 GB fetched per launch; the plain orders (latent
  // fastest / latent by latent, forward walk) 14.7 / 26.4 GB at 5.1 / 4.7 ms against 4.7 ms here.
  const int SJ = (m + 7) / 8, NSB = SJ * (SJ + 1) / 2;
  const int w = blockIdx.x, xcd = w & 7, slot = w >> 3;
  const int gb = xcd + 8 * (slot >> 5), in = slot & 31;
  if (gb >= nlat * NSB) return;
  const int lat = gb / NSB, kb = gb - lat * NSB;
  int sj = (int)((sqrtf(8.0f * (float)kb + 1.0f) - 1.0f) * 0.5f);
  while ((sj + 1) * (sj + 2) / 2 <= kb) ++sj;
  while (sj * (sj + 1) / 2 > kb) --sj;
  const int sa = kb - sj * (sj + 1) / 2;
  const int ibm = 2 * (4 * sa + (in >> 3)), jb = 8 * sj + (in & 7);
  if (jb >= m || ibm > jb) return;
  Acc<float> acc0, acc1;
  acc0.zero();
  acc1.zero();
  // planes taken over from a sweep (ws_stride > 1): the per-latent stride of that scratch is the one the sweep recorded behind
  // its scheme tag -- plmc_vd_blocks_for blocks, or plmc_vd_blocks_keep when the sweep kept its planes (with_inverse | 4)
  if (ws_stride > 1) {
    ws_stride = (int64_t)wscale[VD_W_TAG + 1] * NB * NB;
    wp_lat_stride = 2 * ws_stride;
  }
  const unsigned short *Pl = Wp + (int64_t)lat * wp_lat_stride + b3_index<S>((int64_t)jb * NB, 0, 0, n_pad);
  b3_mainloop<S, 2, 0, B3NoPre, true>(acc0, acc1, Pl + (int64_t)ibm * NB * 8, n_pad, Pl + (int64_t)jb * NB * 8, n_pad, (int)(n_pad - (int64_t)jb * NB), lds);
  float ws = wscale[(int64_t)lat * ws_stride];
  // planes taken over from a sweep (ws_stride > 1): they must be of THIS scheme -- a sweep without eig_lo followed by a K^-1 call
  // with it (or a changed PLMC_SPLIT in between) would read three-plane rows as two-plane rows; poison the result instead
  if (ws_stride > 1 && wscale[(int64_t)lat * ws_stride + VD_W_TAG] != (float)S::NPL) ws = __builtin_nanf("");
  b3_combine<S>(acc0, acc1, 1.0f / (ws * ws));
  const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
  const int ib = ibm + half;
  kinv_tile_epilogue<float, DCAP, SPLINE>(acc0, reinterpret_cast<float *>(lds) + half * tile_smem_elems<float>(), (int)threadIdx.x & 255, ib <= jb, kind,
                                          ib, jb, lat, m, n_pad, alpha, X, n, d, ell, oscale, Kinv, ldk, strideK, kinv_diag, partials, plain);
}


// grad[lat][k] = 1/2 * sum over upper tiles of partials, with the 1/ell_k factor for lengthscales.
// grid (q), 1024 threads = 30 groups of GP = 34 slots; every group walks its tiles with 4 independent
// accumulators (the loads are latency-bound); fixed summation order throughout.
constexpr int RED_NT = 1024;
template <typename T>
__global__ __launch_bounds__(RED_NT) void k_reduce_grad(const double *__restrict__ partials, int m, int d,
                                                        const T *__restrict__ ell, double *__restrict__ grad) {
  __shared__ double red[RED_NT];
  const int lat = blockIdx.x;
  const int ntile = m * m;
  const int slot = threadIdx.x % GP;
  const int grp = threadIdx.x / GP;
  constexpr int NG = RED_NT / GP;
  double s[4] = {0.0, 0.0, 0.0, 0.0};
  if (grp < NG) {
    const double *base = partials + (int64_t)lat * ntile * GP + slot;
    for (int t0 = grp; t0 < ntile; t0 += 4 * NG) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = t0 + u * NG;
        if (t < ntile) {
          const int ib = t / m, jb = t - ib * m;
          if (jb >= ib) s[u] += base[(int64_t)t * GP];
        }
      }
    }
  }
  red[threadIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
  __syncthreads();
  if (threadIdx.x < GP) {
    double tot = 0.0;
    for (int gq = 0; gq < NG; ++gq) tot += red[gq * GP + threadIdx.x];
    const int k = threadIdx.x;
    if (k < d) grad[(int64_t)lat * (d + 2) + k] = 0.5 * tot / (double)ell[(int64_t)lat * d + k];
    else if (k == MAX_DIM) grad[(int64_t)lat * (d + 2) + d] = 0.5 * tot;
    else if (k == MAX_DIM + 1) grad[(int64_t)lat * (d + 2) + d + 1] = 0.5 * tot;
  }
}

// ---- gradient pass over a K^-1 that the sweep accumulated (plmc_potrf_* with_inverse = 2; storage: potrf.hip, k_kacc).
// HBM-bound: one read of the q n^2 / 2 stored elements; per upper tile the same sums as the fused epilogue above,
//     g[k] += wt os w base df_k^2,  g_os += wt w val,  g_noise += w on the diagonal,   w = alpha_i alpha_j - Kinv_ij,
// but with the tile coming from memory there are no accumulators to keep alive: thread = one column of the tile (its
// scaled inputs in registers) x 64 rows (row inputs are wave-uniform LDS broadcasts); loads are whole 512-byte rows.
// Same partials layout and fixed-order reduction (k_reduce_grad) as k_kinv_grad.  grid (m (m + 1) / 2 * q).
template <typename T, int DCAP>
__global__ __launch_bounds__(NTHREADS) void k_grad_tiles(int kind, const T *__restrict__ A, int64_t n_pad, int64_t lda, int64_t strideA,
                                                         const T *__restrict__ Kd, int64_t strideKd, const T *__restrict__ alpha,
                                                         const T *__restrict__ X, int n, int d, const T *__restrict__ ell,
                                                         const T *__restrict__ oscale, T *kinv_diag, double *__restrict__ partials,
                                                         int nlat) {
  const int m = (int)(n_pad / NB);
  const int w = blockIdx.x, lat = w % nlat, t = w / nlat;
  int jb = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while ((jb + 1) * (jb + 2) / 2 <= t) ++jb;
  while (jb * (jb + 1) / 2 > t) --jb;
  const int ib = t - jb * (jb + 1) / 2;
  __shared__ T ui[NB][DCAP + 1];
  __shared__ T ai[NB];
  __shared__ double red[4][GP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const T *el = ell + (int64_t)lat * d;
  for (int e = tid; e < NB * DCAP; e += NTHREADS) {
    const int r = e / DCAP, k = e % DCAP, gi = ib * NB + r;
    ui[r][k] = (k < d && gi < n) ? X[(int64_t)gi * d + k] / el[k] : T(0);
  }
  if (tid < NB) ai[tid] = alpha[(int64_t)lat * n_pad + ib * NB + tid];
  const int col = tid & (NB - 1), rh = tid >> 7;                    // column of the tile, row parity
  const int gj = jb * NB + col;
  T uj[DCAP];
#pragma unroll
  for (int k = 0; k < DCAP; ++k) uj[k] = (k < d && gj < n) ? X[(int64_t)gj * d + k] / el[k] : T(0);
  const T a_j = alpha[(int64_t)lat * n_pad + gj];
  const T os = oscale ? oscale[lat] : T(1);
  const T *Kt;
  int64_t ldk;
  if (ib == jb) { Kt = Kd + (int64_t)lat * strideKd + (int64_t)ib * NB * NB; ldk = NB; }
  else { Kt = A + (int64_t)lat * strideA + (int64_t)jb * NB * lda + (int64_t)ib * NB; ldk = lda; }
  __syncthreads();
  T g[DCAP];
#pragma unroll
  for (int k = 0; k < DCAP; ++k) g[k] = T(0);
  T g_noise = T(0), g_os = T(0);
#pragma unroll 4
  for (int rr = 0; rr < NB / 2; ++rr) {
    const int row = 2 * rr + rh, gi = ib * NB + row;
    const T kin = Kt[(int64_t)row * ldk + col];
    if (kinv_diag && gi == gj) kinv_diag[(int64_t)lat * n_pad + gi] = kin;
    const bool live = gi < n && gj < n && gj >= gi;
    T wv = ai[row] * a_j - kin;
    if (live && gi == gj) g_noise += wv;
    wv = live ? (gj > gi ? T(2) * wv : wv) : T(0);
    T df[DCAP];
    T r2 = T(0), sp = T(1);
#pragma unroll
    for (int k = 0; k < DCAP; ++k) {
      df[k] = ui[row][k] - uj[k];
      r2 += df[k] * df[k];
      if (kind == K_SPLINE) sp *= spline_factor(ui[row][k], uj[k]);
    }
    T val, base;
    if (kind == K_SPLINE) { val = sp; base = T(0); }
    else kern_value_base_fast(kind, r2, val, base);
    g_os += wv * val;
    const T c = wv * os * base;
#pragma unroll
    for (int k = 0; k < DCAP; ++k) g[k] += c * df[k] * df[k];
  }
  auto wave_sum = [&](double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
  };
#pragma unroll
  for (int k = 0; k < DCAP; ++k) {
    double s = wave_sum((double)g[k]);
    if (lane == 0) red[wave][k] = s;
  }
  {
    double s = wave_sum((double)g_noise);
    if (lane == 0) red[wave][MAX_DIM] = s;
    s = wave_sum((double)g_os);
    if (lane == 0) red[wave][MAX_DIM + 1] = s;
  }
  __syncthreads();
  double *out = partials + (((int64_t)lat * m + ib) * m + jb) * GP;
  if (tid < GP) {
    const bool lv = tid < DCAP || tid >= MAX_DIM;
    out[tid] = lv ? red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid] : 0.0;
  }
}

template <typename T>
int grad_tiles_impl(int kind, const T *A, int64_t n_pad, int64_t lda, int64_t strideA, const T *Vd, const T *alpha, const T *X, int n, int d,
                    const T *ell, const T *oscale, double *grad, T *kinv_diag, void *partials, int q, void *stream) {
  PLMC_REQUIRE(kind >= 0 && kind <= 4, "unknown kernel kind");
  PLMC_REQUIRE(A && Vd && alpha && X && ell && grad && partials, "null pointer");
  PLMC_REQUIRE(n_pad > 0 && n_pad % NB == 0 && lda % NB == 0 && n <= n_pad && n > n_pad - NB, "n_pad must be plmc_pad(n)");
  PLMC_REQUIRE(d > 0 && d <= MAX_DIM && q > 0, "need 0<d<=plmc_max_dim(), q>0");
  hipStream_t st = (hipStream_t)stream;
  const int m = (int)(n_pad / NB);
  const int64_t strideV = plmc_vd_blocks_for(n_pad, lda, (int)sizeof(T)) * (int64_t)NB * NB;
  const T *Kd = Vd + strideV - (int64_t)m * NB * NB;                  // last m blocks of the scratch: diagonal K^-1 tiles
  double *part = reinterpret_cast<double *>(partials);
  const dim3 grid(q * (m * (m + 1) / 2)), block(NTHREADS);
#define PLMC_LAUNCH_GT(DC) \
  hipLaunchKernelGGL((k_grad_tiles<T, DC>), grid, block, 0, st, kind, A, n_pad, lda, strideA, Kd, strideV, alpha, X, n, d, ell, oscale, kinv_diag, part, q)
  {
    const double np = (double)n_pad;
    ProfScope ps(PK_GRAD_TILES, st, 0.0, q * (np * np / 2) * sizeof(T));
    if (d <= 4) PLMC_LAUNCH_GT(4);
    else if (d <= 8) PLMC_LAUNCH_GT(8);
    else if (d <= 16) PLMC_LAUNCH_GT(16);
    else PLMC_LAUNCH_GT(32);
  }
#undef PLMC_LAUNCH_GT
  {
    ProfScope ps(PK_REDUCE, st, 0.0, (double)m * m * q * GP * sizeof(double) / 2);
    hipLaunchKernelGGL(k_reduce_grad<T>, dim3(q), dim3(RED_NT), 0, st, part, m, d, ell, grad);
  }
  return launch_status(__func__);
}

// Split of the inverse factor for the split-engine gradient kernel: W (fp32, lower block triangle: block (lb, cb) with
// cb <= lb) -> k8-ordered planes Wp[latent][k / 8][plane][n_pad columns][k % 8] (bf3_engine.hpp), every value times the
// latent's scale `wscale` (SplitH2: 2^13 / bound of |W|, written by k_w_scale; SplitB3: 1).
// grid (m, m, q), one 128 x 128 block per workgroup; HBM-bound (4 bytes read, 2 NPL written per element).
template <class S>
__global__ __launch_bounds__(NTHREADS) void k_split_w(const float *__restrict__ W, int64_t n_pad, int64_t ldw, int64_t strideW,
                                                      unsigned short *__restrict__ Wp, const float *__restrict__ wscale) {
  const int cb = blockIdx.x, lb = blockIdx.y, lat = blockIdx.z;
  if (cb > lb) return;
  b3_split_block<S, false>(W + (int64_t)lat * strideW + (int64_t)lb * NB * ldw + (int64_t)cb * NB, ldw,
                           Wp + (int64_t)lat * b3_elems<S>(n_pad, n_pad) + b3_index<S>((int64_t)lb * NB, 0, (int64_t)cb * NB, n_pad), n_pad, wscale[lat],
                           nullptr, 0, threadIdx.x);
}
// wscale[lat] = scale for |W_ij| <= 1 / sqrt(lambda_min(Khat)) <= 1 / sqrt(eig_lo[lat])  (SplitB3 / no bound: 1); the identity
// padding of the rows beyond n has eigenvalue 1 whatever the noise (`padded`).  1 x q threads.
template <class S>
__global__ void k_w_scale(const float *__restrict__ eig_lo, float *__restrict__ wscale, int q, int padded) {
  const int lat = threadIdx.x;
  if (lat >= q) return;
  if constexpr (S::NPL == 3) wscale[lat] = 1.0f;
  else {
    float lam = fmaxf(eig_lo[lat], 1e-30f);
    if (padded) lam = fminf(lam, 1.0f);
    wscale[lat] = b3_scale_for(1.0f / sqrtf(lam));
  }
}

// S: split scheme of the W^T W products (void: MFMA of the element type); eig_lo: see potrf_impl.
template <typename T, class S>
int kinv_grad_impl(int kind, const T *W, int64_t n_pad, int64_t ldw, int64_t strideW, const T *alpha, const T *X, int n,
                   int d, const T *ell, const T *oscale, double *grad, T *Kinv, int64_t ldk, int64_t strideK,
                   T *kinv_diag, void *partials, int q, const float *eig_lo, void *stream, const float *Vd = nullptr, int64_t lda_vd = 0) {
  PLMC_REQUIRE(kind >= 0 && kind <= 4, "unknown kernel kind");
  PLMC_REQUIRE(W && alpha && X && ell && grad && partials, "null pointer");
  PLMC_REQUIRE(n_pad > 0 && n_pad % NB == 0 && ldw % NB == 0 && n <= n_pad && n > n_pad - NB, "n_pad must be plmc_pad(n)");
  PLMC_REQUIRE(d > 0 && d <= MAX_DIM && q > 0, "need 0<d<=plmc_max_dim(), q>0");
  PLMC_REQUIRE(!Kinv || (ldk >= n_pad), "ldk too small");
  PLMC_REQUIRE(aligned16(W), "unaligned W");
  hipStream_t st = (hipStream_t)stream;
  const int m = (int)(n_pad / NB);
  // tile order: longest K range first (jb ascending outermost, latent fastest).  The (jb, ib, lat) grid ran the
  // long tiles of the last latent at the end of the launch: 112 -> 120 TF at n = 8192, q = 8, 97 -> 118 TF at q = 1,
  // 77 -> 104 TF at n = 4096 (the same launch with every tile reading one panel pair ran no faster, so operand
  // locality is not what limits it; an XCD-dealt super-tile order was level with the grid).  Dev knob
  // PLMC_KINV_ORDER: 0 = XCD-dealt 8 x 8 super-tiles, 1 = (jb, ib, lat) grid, 5 = default order, general epilogue only.
  const int plain = knobs().kinv_order;
  const dim3 grid = plain >= 4 ? dim3(q * (m * (m + 1) / 2)) : plain ? dim3(m, m, q) : dim3(xcd_tri_grid(m, q)), block(NTHREADS);
  double *part = reinterpret_cast<double *>(partials);
  const double np = (double)n_pad;
  // split engine (bf3_engine.hpp): split W into k8-ordered planes behind the partials (and the q scales behind the
  // planes), then the macro-tile kernel
  bool done = false;
  if constexpr (!std::is_void<S>::value) {
    PLMC_REQUIRE(q <= 1024, "too many latents for one scale launch");
    // the planes of W and the scale of that operand family: the ones the sweep left in its Vd scratch (plmc_kinv_grad_vd_*:
    // same scheme, since both calls see the same knob and the same eig_lo) -- or split W now, behind the partials
    const unsigned short *wp = nullptr;
    const float *wsc = nullptr;
    int64_t wp_lat = b3_elems<S>(n_pad, n_pad), ws_lat = 1;
    if (!(Vd && vd_w_planes(Vd, n_pad, lda_vd, &wp, &wp_lat, &wsc, &ws_lat))) {
      char *pb = reinterpret_cast<char *>(partials) + (int64_t)m * m * q * GP * (int64_t)sizeof(double);
      unsigned short *wpo = reinterpret_cast<unsigned short *>(pb);
      float *wsco = reinterpret_cast<float *>(pb + (int64_t)q * b3_elems<SplitB3>(n_pad, n_pad) * 2);
      hipLaunchKernelGGL((k_w_scale<S>), dim3(1), dim3(q < 64 ? 64 : ((q + 63) / 64) * 64), 0, st, eig_lo, wsco, q, (int)(n < n_pad));
      ProfScope ps(PK_SPLIT, st, 0.0, (double)q * np * np / 2 * (4 + 2 * S::NPL));
      hipLaunchKernelGGL((k_split_w<S>), dim3(m, m, q), dim3(NTHREADS), 0, st, (const float *)W, n_pad, ldw, strideW, wpo, (const float *)wsco);
      wp = wpo;
      wsc = wsco;
      wp_lat = b3_elems<S>(n_pad, n_pad);
      ws_lat = 1;
    }
    const int SJ = (m + 7) / 8, NSB = SJ * (SJ + 1) / 2;
    const dim3 gridb(8 * ((q * NSB + 7) / 8) * 32);                       // XCD-dealt super-blocks of 32 macro tiles (see the kernel)
#define PLMC_LAUNCH_KB(DC, SP) \
  hipLaunchKernelGGL((k_kinv_grad_bf3<S, DC, SP>), gridb, dim3(B3_NT), 0, st, kind, n_pad, alpha, X, n, d, ell, oscale, Kinv, ldk, strideK, kinv_diag, \
                     part, q, plain, wp, wsc, wp_lat, ws_lat)
    ProfScope ps(PK_KINV_GRAD, st, q * np * np * np / 3.0, q * (np * np / 2) * sizeof(T));
    if (d <= 4) PLMC_LAUNCH_KB(4, false);
    else if (d <= 8) PLMC_LAUNCH_KB(8, false);
    else if (d <= 16) { if (kind == K_SPLINE) PLMC_LAUNCH_KB(16, true); else PLMC_LAUNCH_KB(16, false); }
    else { if (kind == K_SPLINE) PLMC_LAUNCH_KB(32, true); else PLMC_LAUNCH_KB(32, false); }
#undef PLMC_LAUNCH_KB
    done = true;
  }
#define PLMC_LAUNCH_KG(DC, SP)                                                                                       \
  hipLaunchKernelGGL((k_kinv_grad<T, DC, SP>), grid, block, 0, st, kind, W, n_pad, ldw, strideW, alpha, X, n, d, ell, \
                     oscale, Kinv, ldk, strideK, kinv_diag, part, q, plain)
  if (!done) {
    ProfScope ps(PK_KINV_GRAD, st, q * np * np * np / 3.0, q * (np * np / 2) * sizeof(T));
    if (d <= 4) PLMC_LAUNCH_KG(4, false);
    else if (d <= 8) PLMC_LAUNCH_KG(8, false);
    else if (d <= 16) { if (kind == K_SPLINE) PLMC_LAUNCH_KG(16, true); else PLMC_LAUNCH_KG(16, false); }
    else { if (kind == K_SPLINE) PLMC_LAUNCH_KG(32, true); else PLMC_LAUNCH_KG(32, false); }
  }
#undef PLMC_LAUNCH_KG
  {
    ProfScope ps(PK_REDUCE, st, 0.0, (double)m * m * q * GP * sizeof(double) / 2);
    hipLaunchKernelGGL(k_reduce_grad<T>, dim3(q), dim3(RED_NT), 0, st, part, m, d, ell, grad);
  }
  return launch_status(__func__);
}

}  // namespace plmc

extern "C" {
int plmc_grad_tiles_f32(int kind, const float *A, int64_t n_pad, int64_t lda, int64_t strideA, const float *Vd, const float *alpha,
                        const float *X, int n, int d, const float *ell, const float *oscale, double *grad, float *kinv_diag,
                        void *partials, int q, void *stream) {
  return plmc::grad_tiles_impl<float>(kind, A, n_pad, lda, strideA, Vd, alpha, X, n, d, ell, oscale, grad, kinv_diag, partials, q, stream);
}
int plmc_grad_tiles_f64(int kind, const double *A, int64_t n_pad, int64_t lda, int64_t strideA, const double *Vd, const double *alpha,
                        const double *X, int n, int d, const double *ell, const double *oscale, double *grad, double *kinv_diag,
                        void *partials, int q, void *stream) {
  return plmc::grad_tiles_impl<double>(kind, A, n_pad, lda, strideA, Vd, alpha, X, n, d, ell, oscale, grad, kinv_diag, partials, q, stream);
}
// per-tile partial sums + (4-byte elements) the planes of W for the split engine (the entry points without Vd); independent of the knobs
int64_t plmc_grad_scratch_bytes_for(int64_t n_pad, int q, int elem_bytes) {
  int64_t m = n_pad / plmc::NB;
  const int64_t planes = elem_bytes == 4 ? (int64_t)q * plmc::b3_elems<plmc::SplitB3>(n_pad, n_pad) * 2 + 4096 : 0;   // + the q scales
  return m * m * (int64_t)q * plmc::GP * (int64_t)sizeof(double) + planes;
}
int64_t plmc_grad_scratch_bytes(int64_t n_pad, int q) { return plmc_grad_scratch_bytes_for(n_pad, q, 4); }
// what plmc_kinv_grad_vd_* needs when the sweep's scratch supplies the planes of W: the per-tile partial sums only
int64_t plmc_grad_partials_bytes(int64_t n_pad, int q) {
  const int64_t m = n_pad / plmc::NB;
  return m * m * (int64_t)q * plmc::GP * (int64_t)sizeof(double);
}
static int kinv_grad_f32_any(int kind, const float *W, int64_t n_pad, int64_t ldw, int64_t strideW, const float *alpha,
                             const float *X, int n, int d, const float *ell, const float *oscale, double *grad,
                             float *Kinv, int64_t ldk, int64_t strideK, float *kinv_diag, void *partials, int q, const float *eig_lo,
                             void *stream, const float *Vd = nullptr) {
  const int split = plmc::knobs().split;
  if (split == 0)
    return plmc::kinv_grad_impl<float, void>(kind, W, n_pad, ldw, strideW, alpha, X, n, d, ell, oscale, grad, Kinv, ldk, strideK, kinv_diag, partials, q,
                                             nullptr, stream);
  // Vd: the scratch of the sweep that produced W, whose leading dimension is ldw (W lives in the factor buffer's columns)
  if (split == 2 && eig_lo)
    return plmc::kinv_grad_impl<float, plmc::SplitH2>(kind, W, n_pad, ldw, strideW, alpha, X, n, d, ell, oscale, grad, Kinv, ldk, strideK, kinv_diag,
                                                      partials, q, eig_lo, stream, Vd, ldw);
  return plmc::kinv_grad_impl<float, plmc::SplitB3>(kind, W, n_pad, ldw, strideW, alpha, X, n, d, ell, oscale, grad, Kinv, ldk, strideK, kinv_diag,
                                                    partials, q, nullptr, stream, Vd, ldw);
}
int plmc_kinv_grad_f32(int kind, const float *W, int64_t n_pad, int64_t ldw, int64_t strideW, const float *alpha,
                       const float *X, int n, int d, const float *ell, const float *oscale, double *grad,
                       float *Kinv, int64_t ldk, int64_t strideK, float *kinv_diag, void *partials, int q,
                       void *stream) {
  return kinv_grad_f32_any(kind, W, n_pad, ldw, strideW, alpha, X, n, d, ell, oscale, grad, Kinv, ldk, strideK, kinv_diag, partials, q, nullptr, stream);
}
int plmc_kinv_grad_ex_f32(int kind, const float *W, int64_t n_pad, int64_t ldw, int64_t strideW, const float *alpha,
                          const float *X, int n, int d, const float *ell, const float *oscale, double *grad,
                          float *Kinv, int64_t ldk, int64_t strideK, float *kinv_diag, void *partials, int q,
                          const float *eig_lo, void *stream) {
  return kinv_grad_f32_any(kind, W, n_pad, ldw, strideW, alpha, X, n, d, ell, oscale, grad, Kinv, ldk, strideK, kinv_diag, partials, q, eig_lo, stream);
}
int plmc_kinv_grad_vd_f32(int kind, const float *W, int64_t n_pad, int64_t ldw, int64_t strideW, const float *alpha,
                          const float *X, int n, int d, const float *ell, const float *oscale, double *grad,
                          float *Kinv, int64_t ldk, int64_t strideK, float *kinv_diag, void *partials, int q,
                          const float *eig_lo, const float *Vd, void *stream) {
  return kinv_grad_f32_any(kind, W, n_pad, ldw, strideW, alpha, X, n, d, ell, oscale, grad, Kinv, ldk, strideK, kinv_diag, partials, q, eig_lo, stream, Vd);
}
int plmc_kinv_grad_vd_f64(int kind, const double *W, int64_t n_pad, int64_t ldw, int64_t strideW, const double *alpha,
                          const double *X, int n, int d, const double *ell, const double *oscale, double *grad,
                          double *Kinv, int64_t ldk, int64_t strideK, double *kinv_diag, void *partials, int q,
                          const double *eig_lo, const double *Vd, void *stream) {
  (void)eig_lo;
  (void)Vd;
  return plmc::kinv_grad_impl<double, void>(kind, W, n_pad, ldw, strideW, alpha, X, n, d, ell, oscale, grad, Kinv, ldk,
                                            strideK, kinv_diag, partials, q, nullptr, stream);
}
int plmc_kinv_grad_f64(int kind, const double *W, int64_t n_pad, int64_t ldw, int64_t strideW, const double *alpha,
                       const double *X, int n, int d, const double *ell, const double *oscale, double *grad,
                       double *Kinv, int64_t ldk, int64_t strideK, double *kinv_diag, void *partials, int q,
                       void *stream) {
  return plmc::kinv_grad_impl<double, void>(kind, W, n_pad, ldw, strideW, alpha, X, n, d, ell, oscale, grad, Kinv, ldk,
                                            strideK, kinv_diag, partials, q, nullptr, stream);
}
int plmc_kinv_grad_ex_f64(int kind, const double *W, int64_t n_pad, int64_t ldw, int64_t strideW, const double *alpha,
                          const double *X, int n, int d, const double *ell, const double *oscale, double *grad,
                          double *Kinv, int64_t ldk, int64_t strideK, double *kinv_diag, void *partials, int q,
                          const double *eig_lo, void *stream) {
  (void)eig_lo;
  return plmc::kinv_grad_impl<double, void>(kind, W, n_pad, ldw, strideW, alpha, X, n, d, ell, oscale, grad, Kinv, ldk,
                                            strideK, kinv_diag, partials, q, nullptr, stream);
}
}
