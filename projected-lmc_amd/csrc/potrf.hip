// potrf.hip -- blocked right-looking Cholesky Khat = U^T U on the augmented factor buffer, with the
// inverse factor W = U^-T produced in the same sweep, and the two small vector kernels around them.
//
// Replaces torch.linalg.cholesky_ex / triangular solves that gpytorch runs behind
// `latent_output.log_prob(proj_target)` (projected_lmc.py:1201) and
// gp.mlls.ExactMarginalLogLikelihood (experiments.py:233), plus the first half of the
// cholesky-inverse its autograd backward needs (experiments.py:270); SURVEY.md 8a rows a3/a4.
//
// The buffer is [ Khat | rhs | I ]: everything right of the square part is "just more columns" of the
// same elimination, so  U^-T y  (forward solve), U^-T K*^T (prediction) and W = U^-T I (inverse factor)
// all fall out of one sweep built from two MFMA tile kernels:
//     k_panel  : row panel  P <- V_rr^T P                      (in place, K = 128)
//     k_update : C[i][j]  -= sum_k P[k][i] P[k][j]              (upper tiles of U, aug, live tiles of W)
// Block rows are processed in GROUPS of g (two-level blocking): inside a group each row is brought up to
// date with the rows of the group done so far (depth <= 128 (g - 1)); the large trailing update then runs
// once per group with depth 128 g, which divides the read-modify-write traffic of the trailing matrix by
// g.  The latency-bound chain of the next group runs beside that update on a helper stream (potrf_impl).
// Tiles of W are written (not accumulated) the first time they are touched, so W needs no memset.
#include <stdlib.h>
#include <vector>
#include "api_common.hpp"
#include "covariance.hpp"
#include "diag_block.hpp"
#include "../../include/plmc.h"

namespace plmc {

// Column-tile decoding shared by k_panel / k_update.  Tiles along grid.x are laid out as
//   [ U block columns u0 .. m-1 | aug tiles (Taug) | W block columns 0 .. nW-1 ].
struct ColMap {
  int u0, nU, Taug, nW;
  int64_t n_pad, wcol0;
};

// Row panel solve P <- V_rr^T P for block row r.  grid (nU + Taug + nW, q, 4 / NT): NT = 2 splits every 128 x 128 tile
// into two 64-COLUMN halves -- the solve is in place and every output row needs all 128 input rows of its column,
// so only a column split keeps workgroups independent -- for launches that would not fill the CUs.
template <typename T, int NT>
__global__ __launch_bounds__(NTHREADS, TILE_MIN_WAVES<T>) void k_panel(T *A, int64_t lda, int64_t strideA, int r, ColMap cm,
                                                     const T *__restrict__ Vd, int64_t strideV) {
  __builtin_amdgcn_s_setprio(3);       // chain kernel: ahead of the concurrently running trailing update
  __shared__ __align__(16) T smem[tile_smem_elems<T>()];
  const int lat = blockIdx.y, t = blockIdx.x;
  int64_t col0;
  if (t < cm.nU) col0 = (int64_t)(cm.u0 + t) * NB;
  else if (t < cm.nU + cm.Taug) col0 = cm.n_pad + (int64_t)(t - cm.nU) * NB;
  else col0 = cm.wcol0 + (int64_t)(t - cm.nU - cm.Taug) * NB;
  T *P = A + (int64_t)lat * strideA + (int64_t)r * NB * lda + col0 + (int)blockIdx.z * (32 * NT);
  const T *V = Vd + (int64_t)lat * strideV + (int64_t)r * NB * NB;
  Acc<T, 4, NT> acc;
  acc.zero();
  tile_mainloop<T, false, false, 4, NT>(acc, V, NB, P, lda, NB, smem);
  tile_store<T, 4, NT>(acc, P, lda);
}

// Rank-(128 g) update of block rows [ib0, ib0 + nrows) with the panel rows of block rows
// r_lo..r_hi:  C[i][j] -= sum_{k in panel} P[k][i] P[k][j].   grid (nU + Taug + nW, nrows, q).
//   U columns : tiles with jb >= ib (upper), read-modify-write.
//   aug       : read-modify-write.
//   W column cb < r_lo : read-modify-write, full panel depth;
//            cb == r_lo : first touch -> plain store, full depth;
//            cb == r_hi (> r_lo): first touch, only the rows of block r_hi contribute (W[r_lo][r_hi] = 0).
// ROLE 0 = the big trailing ("tail") update, 1 = the single-row launches inside a group (latency-critical:
// raised wave priority), 2 = the "head" rows the next group needs.  Separate symbols keep the three launch
// shapes apart in kernel traces and counter passes.
// MT = 2 splits every 128 x 128 tile into two 64-row halves (grid.y doubled): twice the workgroups for the chain's
// launches when full tiles would not fill the CUs (single-latent shards).  (k_panel cannot be split this way: it
// works in place and every output row needs all 128 input rows of its column.)
template <typename T, int ROLE, int MT = 4>
__global__ __launch_bounds__(NTHREADS, TILE_MIN_WAVES<T>) void k_update(T *A, int64_t lda, int64_t strideA, int ib0, int r_lo, int r_hi,
                                                      ColMap cm) {
  if (ROLE == 1) __builtin_amdgcn_s_setprio(3);
  if (ROLE == 2) __builtin_amdgcn_s_setprio(2);       // head rows: the next chain waits for them
  // plain row-major tile order: an XCD-dealt super-block order (xcd_tri_decode, gemm_core.hpp) was 3 % faster for a
  // launch that has the GPU to itself and 25 % slower in the sweep, where launches from three streams
  // interleave and "workgroup w lands on XCD w % 8" no longer holds
  constexpr int SPLIT = 4 / MT;                                      // half tiles per tile
  const int bx = blockIdx.x, ib = ib0 + (int)blockIdx.y / SPLIT, lat = blockIdx.z;
  const int h0 = ((int)blockIdx.y % SPLIT) * (32 * MT);             // first row of this half inside the block row
  int64_t col0;
  int kr0 = r_lo * NB, depth = (r_hi - r_lo + 1) * NB;
  bool first = false;
  if (bx < cm.nU) {
    const int jb = cm.u0 + bx;
    if (jb < ib) return;
    col0 = (int64_t)jb * NB;
  } else if (bx < cm.nU + cm.Taug) {
    col0 = cm.n_pad + (int64_t)(bx - cm.nU) * NB;
  } else {
    const int cb = bx - cm.nU - cm.Taug;
    col0 = cm.wcol0 + (int64_t)cb * NB;
    // columns that start inside the current group: W[r][cb] = 0 for r < cb, so only panel rows
    // cb..r_hi contribute, and this is the first time the tile is touched -> plain store
    if (cb >= r_lo) { first = true; kr0 = cb * NB; depth = (r_hi - cb + 1) * NB; }
  }
  __shared__ __align__(16) T smem[tile_smem_elems<T>()];
  T *Al = A + (int64_t)lat * strideA;
  const T *Prow = Al + (int64_t)kr0 * lda;
  Acc<T, MT> acc;
  acc.zero();
  tile_mainloop<T, false, false, MT>(acc, Prow + (int64_t)ib * NB + h0, lda, Prow + col0, lda, depth, smem);
  T *C = Al + ((int64_t)ib * NB + h0) * lda + col0;
  if (first) tile_writeback<T, WB_STORE_NEG, MT>(acc, C, lda, smem);   // C = -P^T P (first touch of a W tile)
  else tile_writeback<T, WB_SUB, MT>(acc, C, lda, smem);               // C -= P^T P
}

// ----------------------------------------------------------------------------------------------
// z[lat][i] = A[i][n_pad + c];  quad[lat] = sum z^2 (double).  grid (q).
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_extract_col(const T *__restrict__ A, int64_t n_pad, int64_t lda,
                                                           int64_t strideA, int c, T *__restrict__ z,
                                                           double *__restrict__ quad) {
  __shared__ double red[NTHREADS];
  const int lat = blockIdx.x;
  const T *Al = A + (int64_t)lat * strideA + n_pad + c;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n_pad; i += NTHREADS) {
    T v = Al[i * lda];
    z[(int64_t)lat * n_pad + i] = v;
    s += (double)v * (double)v;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = NTHREADS / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) quad[lat] = red[0];
}

// alpha[i] = sum_{l >= block(i)} W[l][i] z[l].  grid (n_pad / 128, q): one workgroup of 1024 threads per block column;
// 32 row groups (16 fp64) x 32 lanes x 16-byte loads (one full 512-byte row segment per row group and step,
// 4 rows in flight per thread).  HBM-bound: reads the lower triangle of W once.  With 256 threads a workgroup
// kept too few bytes in flight, and a single-latent shard has only n_pad / 128 workgroups (0.5 TB/s).
constexpr int WTMV_NT = 1024;
template <typename T>
__global__ __launch_bounds__(WTMV_NT) void k_wt_matvec(const T *__restrict__ W, int64_t n_pad, int64_t ldw,
                                                       int64_t strideW, const T *__restrict__ z,
                                                       T *__restrict__ alpha) {
  using vec_t = typename Traits<T>::vec_t;
  constexpr int EPV = Traits<T>::EPV;
  constexpr int LPR = 128 / EPV;                 // lanes per row segment (32 fp32 / 64 fp64)
  constexpr int NRG = WTMV_NT / LPR;             // row groups (32 / 16)
  __shared__ double red[NRG][NB];
  const int lat = blockIdx.y;
  const int cl = (threadIdx.x % LPR) * EPV, rg = threadIdx.x / LPR;
  const int64_t col0 = (int64_t)blockIdx.x * NB;
  const T *Wl = W + (int64_t)lat * strideW + col0 + cl;
  const T *zl = z + (int64_t)lat * n_pad;
  double s[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) s[e] = 0.0;
  for (int64_t l = col0 + rg; l < n_pad; l += 4 * NRG) {
    vec_t v[4];
    T zz[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t ll = l + u * NRG;
      const bool ok = ll < n_pad;
      v[u] = ok ? *reinterpret_cast<const vec_t *>(Wl + ll * ldw) : vec_t{};
      zz[u] = ok ? zl[ll] : T(0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < EPV; ++e) s[e] += (double)v[u][e] * (double)zz[u];
  }
#pragma unroll
  for (int e = 0; e < EPV; ++e) red[rg][cl + e] = s[e];
  __syncthreads();
  if (threadIdx.x < NB) {
    double t = 0.0;
#pragma unroll
    for (int g = 0; g < NRG; ++g) t += red[g][threadIdx.x];
    alpha[(int64_t)lat * n_pad + col0 + threadIdx.x] = (T)t;
  }
}

// ----------------------------------------------------------------------------------------------
template <typename T>
int potrf_impl(T *A, int64_t n_pad, int64_t lda, int naug, int64_t strideA, T *Vd, double *logdet, int *info,
               int with_inverse, int q, void *stream) {
  PLMC_REQUIRE(A && Vd && logdet && info, "null pointer");
  PLMC_REQUIRE(n_pad > 0 && n_pad % NB == 0 && lda % NB == 0 && lda >= n_pad, "n_pad/lda must be multiples of NB");
  const int64_t naug_pad = plmc_pad(naug);
  PLMC_REQUIRE(naug >= 0 && n_pad + naug_pad + (with_inverse ? n_pad : 0) <= lda,
               "lda too small for naug (+ the n_pad columns of the inverse factor)");
  PLMC_REQUIRE(q > 0 && aligned16(A) && aligned16(Vd), "bad q or unaligned buffer");
  const hipStream_t st = (hipStream_t)stream;
  const int m = (int)(n_pad / NB);
  const int Taug = (int)(naug_pad / NB);
  const int64_t strideV = (int64_t)m * NB * NB;
  const int64_t wcol0 = n_pad + naug_pad;
  const double nb = (double)NB, nb3 = nb * nb * nb, esz = sizeof(T);

  auto diag = [&](int r, hipStream_t st) {
    ProfScope ps(PK_DIAG, st, q * (2.0 / 3.0) * nb3, q * 3.0 * nb * nb * esz);
    T *wout = with_inverse ? A + (int64_t)r * NB * lda + wcol0 + (int64_t)r * NB : (T *)nullptr;
    hipLaunchKernelGGL((k_diag<T>), dim3(q), dim3(DIAG_NT), 0, st, A, lda, strideA, r, Vd, strideV, wout, lda, strideA);
  };
  // launches with few tiles run on half tiles (dev knob PLMC_HALF_TILES: 0 = never, 1 = always, N > 1 = tile-count
  // threshold); read once per sweep, not per launch
  const char *henv = getenv("PLMC_HALF_TILES");
  const double hthr = henv ? (atoi(henv) == 1 ? 1e30 : (double)atoi(henv)) : 640.0;
  // part: 0 = every column of the row, 1 = U + augmented columns only, 2 = inverse-factor (W) columns only
  auto panel = [&](int r, hipStream_t st, int part = 0) {
    ColMap cm{r + 1, part == 2 ? 0 : m - 1 - r, part == 2 ? 0 : Taug, (with_inverse && part != 1) ? r : 0, n_pad, wcol0};
    const int nt = cm.nU + cm.Taug + cm.nW;
    if (nt == 0) return;
    // algorithmic: triangular solve of nt*NB columns with a 128 x 128 factor = nb^2 flops per column
    ProfScope ps(PK_PANEL, st, q * (double)nt * nb3, q * 2.0 * nt * nb * nb * esz);
    if ((double)nt * q <= hthr)
      hipLaunchKernelGGL((k_panel<T, 2>), dim3(nt, q, 2), dim3(NTHREADS), 0, st, A, lda, strideA, r, cm, Vd, strideV);
    else
      hipLaunchKernelGGL((k_panel<T, 4>), dim3(nt, q, 1), dim3(NTHREADS), 0, st, A, lda, strideA, r, cm, Vd, strideV);
  };
  // cls: profiler class of the launch -- PK_TRAIL (the big trailing update), PK_TRAIL_HEAD (the rows the next
  // group needs, on the chain stream), PK_TRAIL_ROW (single row inside a group)
  auto update = [&](int ib0, int nrows, int r_lo, int r_hi, hipStream_t st, int cls, int part = 0) {
    if (nrows <= 0) return;
    ColMap cm{ib0, part == 2 ? 0 : m - ib0, part == 2 ? 0 : Taug, (with_inverse && part != 1) ? r_hi + 1 : 0, n_pad, wcol0};
    if (cm.nU + cm.Taug + cm.nW == 0) return;
    const double depth = (r_hi - r_lo + 1) * nb;
    // algorithmic flops: symmetric rank-k update of the nrows block rows (upper tiles only) + rectangular parts
    const double nr = (double)nrows;
    const double tilesU = part == 2 ? 0.0 : nr * (cm.nU) - nr * (nr - 1) / 2.0;  // tiles jb >= ib
    const double flopsU = part == 2 ? 0.0 : 2.0 * nb * nb * depth * (tilesU - nr / 2.0);   // diagonal tiles count half
    const double tilesA = nr * cm.Taug;
    double depthW = 0.0;                                                         // summed panel depth over W columns
    for (int cb = 0; cb < cm.nW; ++cb) depthW += (cb >= r_lo ? (r_hi - cb + 1) : (r_hi - r_lo + 1)) * nb;
    const double flopsR = 2.0 * nb * nb * nr * (depth * cm.Taug + depthW);
    const int nfirst = cm.nW > 0 ? r_hi - r_lo + 1 : 0;                          // first-touch W columns: no read
    const double bytes = (2.0 * (tilesU + tilesA + nr * cm.nW) - nr * nfirst) * nb * nb * esz;
    ProfScope ps(cls, st, q * (flopsU + flopsR), q * bytes);
    const int Cn = cm.nU + cm.Taug + cm.nW;
    // chain launches with few tiles (single-latent shards) run on 64-row half tiles: twice the workgroups
    const bool half = cls != PK_TRAIL && (double)Cn * nrows * q <= hthr;
    const dim3 grid(Cn, half ? 2 * nrows : nrows, q);
    if (cls == PK_TRAIL_ROW) {
      if (half) hipLaunchKernelGGL((k_update<T, 1, 2>), grid, dim3(NTHREADS), 0, st, A, lda, strideA, ib0, r_lo, r_hi, cm);
      else hipLaunchKernelGGL((k_update<T, 1, 4>), grid, dim3(NTHREADS), 0, st, A, lda, strideA, ib0, r_lo, r_hi, cm);
    } else if (cls == PK_TRAIL_HEAD) {
      if (half) hipLaunchKernelGGL((k_update<T, 2, 2>), grid, dim3(NTHREADS), 0, st, A, lda, strideA, ib0, r_lo, r_hi, cm);
      else hipLaunchKernelGGL((k_update<T, 2, 4>), grid, dim3(NTHREADS), 0, st, A, lda, strideA, ib0, r_lo, r_hi, cm);
    } else {
      hipLaunchKernelGGL((k_update<T, 0, 4>), grid, dim3(NTHREADS), 0, st, A, lda, strideA, ib0, r_lo, r_hi, cm);
    }
  };

  // whole-sweep bracket on the caller's stream (the per-kernel records of overlapped kernels add up to
  // more than the wall time once the look-ahead runs the chain beside the trailing update)
  const double npd = (double)n_pad;
  ProfScope whole(PK_SWEEP, st, q * (with_inverse ? 2.0 : 1.0) * npd * npd * npd / 3.0, 0.0);
  // tiles of the diagonal-block outputs that k_diag leaves alone (they are read as parts of full 128 x 128 operands)
  hipLaunchKernelGGL(k_zero_diag_out<T>, dim3(m, q), dim3(NTHREADS), 0, st, Vd, strideV,
                     with_inverse ? A + wcol0 : (T *)nullptr, lda, strideA, (int64_t)NB * lda + NB);
  auto finish = [&]() {
    hipLaunchKernelGGL(k_logdet<T>, dim3(q), dim3(NTHREADS), 0, st, (const T *)A, n_pad, lda, strideA, logdet, info);
    return launch_status("potrf_impl");
  };
  // Block rows are processed in GROUPS: within a group each diagonal block is factored/inverted, its row
  // panel solved, and the next row of the group brought up to date (rank-(128 j) update with the rows of
  // the group done so far); the rest of the matrix then gets ONE update of depth 128 * (group size).
  // Deeper updates = less read-modify-write traffic on the trailing matrix, but the within-group work runs
  // one block row at a time.  Measured on MI355X the schedule matters little (27.7 .. 28.3 ms for the sweep
  // of the benchmark shape over a dozen schedules); large groups, smaller ones at the end, is the default.
  std::vector<int> gb;                                    // group boundaries: gb[i] .. gb[i+1]
  {
    const char *genv = getenv("PLMC_GRP");                // dev knob: fixed group size
    const int fixed = genv ? atoi(genv) : 0;
    const int big = fixed > 0 ? fixed : ((q >= 8 && m >= 32) ? 8 : (q == 1 ? 3 : 4));   // measured per q on MI355X
    int r = 0;
    gb.push_back(0);
    const char *senv = getenv("PLMC_GRP_SCHED");          // dev knob: explicit comma-separated group sizes
    if (senv) {
      const char *p = senv;
      while (*p && r < m) {
        int g = atoi(p);
        if (g <= 0) break;
        if (r + g > m) g = m - r;
        r += g;
        gb.push_back(r);
        while (*p && *p != ',') ++p;
        if (*p == ',') ++p;
      }
    }
    while (r < m) {
      int g = big;
      if (fixed <= 0 && big > 4 && m - r <= 16) g = 4;    // ramp down: the chain is the bottleneck at the end
      if (r + g > m) g = m - r;
      r += g;
      gb.push_back(r);
    }
  }
  const int ng = (int)gb.size() - 1;
  auto chain = [&](int gi, hipStream_t s) {
    const int g0 = gb[gi], g1 = gb[gi + 1];
    for (int r = g0; r < g1; ++r) {
      diag(r, s);
      panel(r, s);
      if (r + 1 < g1) update(r + 1, 1, g0, r, s, PK_TRAIL_ROW);
    }
  };
  // The same chain with the inverse-factor columns taken off the critical path: the next diagonal block depends
  // only on the U columns, so stream s carries diag / U panel / U row update and stream w follows one row behind
  // with the W panel and W row update (their A operand is the U panel of the same rows: event e_row).
  auto chain_split = [&](int gi, hipStream_t s, hipStream_t w, hipEvent_t e_row) {
    const int g0 = gb[gi], g1 = gb[gi + 1];
    for (int r = g0; r < g1; ++r) {
      diag(r, s);
      panel(r, s, 1);
      (void)hipEventRecord(e_row, s);
      (void)hipStreamWaitEvent(w, e_row, 0);
      panel(r, w, 2);
      if (r + 1 < g1) {
        update(r + 1, 1, g0, r, s, PK_TRAIL_ROW, 1);
        update(r + 1, 1, g0, r, w, PK_TRAIL_ROW, 2);
      }
    }
  };
  // Look-ahead on two streams.  C (helper, high priority) carries the latency-bound work: the chain of
  // each group and the "head" update (the block rows the NEXT chain needs); T (the caller's stream) carries the
  // "tail" update of all other rows.
  //   head(g) needs chain(g) [same stream] and tail(g-1) [event, normally long complete];
  //   tail(g) needs chain(g) [event] and tail(g-1) [same stream]; head(g) and tail(g) touch disjoint rows.
  // C never waits on an event that is still pending when the chain is the bottleneck (few latents), and T
  // runs its updates back to back when the updates are (many latents).  Falls back to one stream.
  hipStream_t C = side_stream();
  // dev knob PLMC_CUMASK=1: tail on a CU-masked stream (one CU per XCD kept free for the chain).  It paid when a
  // diagonal-block workgroup needed 66 KB of LDS + 16 wave slots at once (2 % then); with the 19 KB / 8-wave
  // kernel the tail is better off with all CUs (42.9 vs 43.5 ms/step), so the default is the caller's stream.
  hipStream_t s2 = getenv("PLMC_CUMASK") ? tail_stream(8) : nullptr;
  hipEvent_t e_chain = sync_event(0), e_tail = sync_event(1), e_entry = sync_event(2), e_done = sync_event(3);
  const bool la = C && e_chain && e_tail && e_entry && e_done && ng > 2 && !getenv("PLMC_SERIAL");   // dev knob: one stream
  if (!la) {
    chain(0, st);
    for (int gi = 0; gi + 1 < ng; ++gi) {
      update(gb[gi + 1], m - gb[gi + 1], gb[gi], gb[gi + 1] - 1, st, PK_TRAIL);
      chain(gi + 1, st);
    }
    return finish();
  }
  // Three streams when the inverse factor is wanted and the launches are wide (dev knob PLMC_WSTREAM=0/1): the W
  // columns of the chain and of the head rows run on a second helper stream Wc.
  //   Wc: whead(g) [needs chain_U(g): e_chain, wchain(g): same stream, tail(g-1): e_tail] -> wchain(g+1)
  //   T : tail(g)  [needs e_chain, e_wchain]
  hipStream_t Wc = with_inverse ? side_stream(1) : nullptr;
  hipEvent_t e_row = sync_event(4), e_wchain = sync_event(5), e_wdone = sync_event(6);
  const char *wenv = getenv("PLMC_WSTREAM");
  const bool wsplit = Wc && e_row && e_wchain && e_wdone && (wenv ? atoi(wenv) != 0 : q >= 4);
  if (wsplit) {
    (void)hipEventRecord(e_entry, st);
    (void)hipStreamWaitEvent(C, e_entry, 0);
    (void)hipStreamWaitEvent(Wc, e_entry, 0);
    chain_split(0, C, Wc, e_row);
    bool tail_pending = false;
    for (int gi = 0; gi + 1 < ng; ++gi) {
      const int g0 = gb[gi], g1 = gb[gi + 1], first = g1, nrest = m - first;
      const int nhead = gb[gi + 2] - gb[gi + 1];
      (void)hipEventRecord(e_chain, C);                     // U panels of group gi complete
      (void)hipEventRecord(e_wchain, Wc);                   // W panels of group gi complete
      if (tail_pending) {
        (void)hipStreamWaitEvent(C, e_tail, 0);
        (void)hipStreamWaitEvent(Wc, e_tail, 0);
      }
      update(first, nhead, g0, g1 - 1, C, PK_TRAIL_HEAD, 1);
      (void)hipStreamWaitEvent(Wc, e_chain, 0);
      update(first, nhead, g0, g1 - 1, Wc, PK_TRAIL_HEAD, 2);
      tail_pending = nrest > nhead;
      if (tail_pending) {
        (void)hipStreamWaitEvent(st, e_chain, 0);
        (void)hipStreamWaitEvent(st, e_wchain, 0);
        update(first + nhead, nrest - nhead, g0, g1 - 1, st, PK_TRAIL);
        (void)hipEventRecord(e_tail, st);
      }
      chain_split(gi + 1, C, Wc, e_row);
    }
    (void)hipEventRecord(e_done, C);
    (void)hipEventRecord(e_wdone, Wc);
    (void)hipStreamWaitEvent(st, e_done, 0);
    (void)hipStreamWaitEvent(st, e_wdone, 0);
    return finish();
  }
  hipStream_t Tq = s2 ? s2 : st;
  (void)hipEventRecord(e_entry, st);
  (void)hipStreamWaitEvent(C, e_entry, 0);
  if (Tq != st) (void)hipStreamWaitEvent(Tq, e_entry, 0);
  chain(0, C);
  bool tail_pending = false, any_tail = false;
  for (int gi = 0; gi + 1 < ng; ++gi) {
    const int g0 = gb[gi], g1 = gb[gi + 1], first = g1, nrest = m - first;
    const int nhead = gb[gi + 2] - gb[gi + 1];              // rows of the next group
    (void)hipEventRecord(e_chain, C);                       // panels of group gi complete
    if (tail_pending) (void)hipStreamWaitEvent(C, e_tail, 0);
    update(first, nhead, g0, g1 - 1, C, PK_TRAIL_HEAD);
    tail_pending = nrest > nhead;
    if (tail_pending) {
      (void)hipStreamWaitEvent(Tq, e_chain, 0);
      update(first + nhead, nrest - nhead, g0, g1 - 1, Tq, PK_TRAIL);
      (void)hipEventRecord(e_tail, Tq);
      any_tail = true;
    }
    chain(gi + 1, C);
  }
  (void)hipEventRecord(e_done, C);
  (void)hipStreamWaitEvent(st, e_done, 0);
  if (any_tail && Tq != st) (void)hipStreamWaitEvent(st, e_tail, 0);
  return finish();
}

template <typename T>
int extract_col_impl(const T *A, int64_t n_pad, int64_t lda, int64_t strideA, int c, T *z, double *quad, int q,
                     void *stream) {
  PLMC_REQUIRE(A && z && quad, "null pointer");
  PLMC_REQUIRE(c >= 0 && n_pad + c < lda, "column outside the augmented block");
  ProfScope ps(PK_EXTRACT, (hipStream_t)stream, 0.0, q * 2.0 * n_pad * sizeof(T));
  hipLaunchKernelGGL(k_extract_col<T>, dim3(q), dim3(NTHREADS), 0, (hipStream_t)stream, A, n_pad, lda, strideA, c, z,
                     quad);
  return launch_status(__func__);
}

template <typename T>
int wt_matvec_impl(const T *W, int64_t n_pad, int64_t ldw, int64_t strideW, const T *z, T *alpha, int q,
                   void *stream) {
  PLMC_REQUIRE(W && z && alpha, "null pointer");
  PLMC_REQUIRE(n_pad > 0 && n_pad % NB == 0, "n_pad must be a multiple of NB");
  ProfScope ps(PK_WTMV, (hipStream_t)stream, q * (double)n_pad * n_pad, q * ((double)n_pad * n_pad / 2) * sizeof(T));
  hipLaunchKernelGGL(k_wt_matvec<T>, dim3((unsigned)(n_pad / NB), q), dim3(WTMV_NT), 0, (hipStream_t)stream, W,
                     n_pad, ldw, strideW, z, alpha);
  return launch_status(__func__);
}

}  // namespace plmc

extern "C" {
int plmc_potrf_f32(float *A, int64_t n_pad, int64_t lda, int naug, int64_t strideA, float *Vd, double *logdet,
                   int *info, int with_inverse, int q, void *stream) {
  return plmc::potrf_impl<float>(A, n_pad, lda, naug, strideA, Vd, logdet, info, with_inverse, q, stream);
}
int plmc_potrf_f64(double *A, int64_t n_pad, int64_t lda, int naug, int64_t strideA, double *Vd, double *logdet,
                   int *info, int with_inverse, int q, void *stream) {
  return plmc::potrf_impl<double>(A, n_pad, lda, naug, strideA, Vd, logdet, info, with_inverse, q, stream);
}
int plmc_extract_col_f32(const float *A, int64_t n_pad, int64_t lda, int64_t strideA, int c, float *z, double *quad,
                         int q, void *stream) {
  return plmc::extract_col_impl<float>(A, n_pad, lda, strideA, c, z, quad, q, stream);
}
int plmc_extract_col_f64(const double *A, int64_t n_pad, int64_t lda, int64_t strideA, int c, double *z,
                         double *quad, int q, void *stream) {
  return plmc::extract_col_impl<double>(A, n_pad, lda, strideA, c, z, quad, q, stream);
}
int plmc_wt_matvec_f32(const float *W, int64_t n_pad, int64_t ldw, int64_t strideW, const float *z, float *alpha,
                       int q, void *stream) {
  return plmc::wt_matvec_impl<float>(W, n_pad, ldw, strideW, z, alpha, q, stream);
}
int plmc_wt_matvec_f64(const double *W, int64_t n_pad, int64_t ldw, int64_t strideW, const double *z, double *alpha,
                       int q, void *stream) {
  return plmc::wt_matvec_impl<double>(W, n_pad, ldw, strideW, z, alpha, q, stream);
}
}
