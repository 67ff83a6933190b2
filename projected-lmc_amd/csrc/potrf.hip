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
// all fall out of one sweep.  Block rows are processed in GROUPS of G <= 8 (two-level blocking):
//   1. the latency-bound CHAIN only touches the group's G x G diagonal triangle of 128-blocks: factor + invert the
//      diagonal block (k_diag), solve / update the <= G - 1 tiles right of it (k_panel, k_update), and build the
//      group's inverse triangle Wgg = Ugg^-T the same way in a scratch buffer (Wg);
//   2. k_vtrans turns Wgg into its transpose Vgg = Ugg^-1 (K-major, what the TN tile engine wants);
//   3. the GROUP PANEL (k_gpanel) then solves the whole block row of the group against the triangle as ONE product
//      P = Vgg^T A over all columns right of the group, the augmented columns and the inverse-factor columns left
//      of the group -- the same flops as G row-by-row panel solves and within-group row updates, in one MFMA-bound
//      launch instead of 2 G latency-bound ones;
//   4. the trailing update C -= P^T P of depth 128 G (k_update) runs once per group.
// Tiles of W are written (not accumulated) the first time they are touched, so W needs no memset.
// The chain of the NEXT group runs beside the group panel / trailing update of the current one (potrf_impl).
#include <stdlib.h>
#include <vector>
#include "api_common.hpp"
#include "covariance.hpp"
#include "diag_block.hpp"
#include "bf3_engine.hpp"
#include "chain_engine.hpp"
#include "../../include/plmc.h"

namespace plmc {

constexpr int GMAX = 8;                       // largest group (block rows; 16 was measured: 4 groups at n = 8192 pipeline too coarsely, 36.8 -> 38.8 ms at q = 8)
// slabs of 16 rows the chain kernels keep in flight (tile_mainloop_burst): fp32 all 8 of K = 128, fp64 4 (registers)
template <typename T> constexpr int CHAIN_BURST = sizeof(T) == 8 ? 4 : 8;
constexpr int LDG = (GMAX + 1) * NB;          // leading dimension of the group scratch matrices (Wg, Vg, Ph): an odd number of
                                              // 128-blocks, like lda -- a power-of-two row stride camps on a few L2 channels
// per-latent scratch behind the m inverse diagonal blocks of Vd: Wg + two Vg (ping-pong), GMAX^2 blocks each
// + the head panel buffer (GMAX^2 blocks) + the bulk panel buffer (GMAX block rows of lda / NB blocks)
constexpr int VD_FIXED_BLOCKS = 4 * GMAX * (GMAX + 1);

// Column-tile decoding shared by k_panel / k_update / k_gpanel.  Tiles along grid.x are laid out as
//   [ U block columns u0 .. u0+nU-1 | aug tiles (Taug) | W block columns w0 .. w0+nW-1 ].
// The W part has its own base / leading dimension / batch stride, indexed by ABSOLUTE (block row, block column):
// either the inverse-factor columns of the factor buffer (W = A + wcol0, ldw = lda) or the group scratch Wg
// (pointer pre-shifted by the group's first block so that absolute indices work).
template <typename T> struct ColMap {
  int u0, nU, Taug, w0, nW;
  int64_t n_pad;
  T *W;
  int64_t ldw, strideW;
};

// Row panel solve P <- V_rr^T P for block row r.  grid (nU + Taug + nW, q, 4 / NT): NT = 2 splits every 128 x 128 tile
// into two 64-COLUMN halves -- the solve is in place and every output row needs all 128 input rows of its column,
// so only a column split keeps workgroups independent -- for launches that would not fill the CUs.
template <typename T, int NT, bool BURSTED = false>
__global__ __launch_bounds__(NTHREADS, TILE_MIN_WAVES<T>) void k_panel(T *A, int64_t lda, int64_t strideA, int r, ColMap<T> cm,
                                                     const T *__restrict__ Vd, int64_t strideV) {
  __builtin_amdgcn_s_setprio(3);       // chain kernel: ahead of the concurrently running trailing update
  __shared__ __align__(16) T smem[tile_smem_elems<T>()];
  const int lat = blockIdx.y, t = blockIdx.x;
  T *P;
  int64_t ldp = lda;
  if (t < cm.nU) P = A + (int64_t)lat * strideA + (int64_t)r * NB * lda + (int64_t)(cm.u0 + t) * NB;
  else if (t < cm.nU + cm.Taug) P = A + (int64_t)lat * strideA + (int64_t)r * NB * lda + cm.n_pad + (int64_t)(t - cm.nU) * NB;
  else {
    ldp = cm.ldw;
    P = cm.W + (int64_t)lat * cm.strideW + (int64_t)r * NB * ldp + (int64_t)(cm.w0 + t - cm.nU - cm.Taug) * NB;
  }
  P += (int)blockIdx.z * (32 * NT);
  const T *V = Vd + (int64_t)lat * strideV + (int64_t)r * NB * NB;
  Acc<T, 4, NT> acc;
  acc.zero();
  // BURSTED (few latents: the chain is the critical path and its launches find free slots at once): all of K = 128
  // (fp64: half of it) in flight at once.  With many latents the launches queue for CU slots behind bulk tiles and the
  // extra staging registers only make a workgroup harder to place (q = 8: 48 -> 55 us in situ), so the plain loop stays.
  if constexpr (BURSTED) tile_mainloop_burst<T, 4, NT, CHAIN_BURST<T>>(acc, V, NB, P, ldp, NB, smem);
  else tile_mainloop<T, false, false, 4, NT>(acc, V, NB, P, ldp, NB, smem);
  tile_store<T, 4, NT>(acc, P, ldp);
}

// Rank-(128 g) update of block rows [ib0, ib0 + nrows) with the panel rows of block rows
// r_lo..r_hi:  C[i][j] -= sum_{k in panel} P[k][i] P[k][j].   grid (nU + Taug + nW, nrows, q).
//   U columns : tiles with jb >= ib (upper), read-modify-write; tiles with ib < skip_ib && jb < skip_jb are left to
//               another launch (the look-ahead updates the next group's triangle on the chain stream).
//   aug       : read-modify-write.
//   W column cb < r_lo : read-modify-write, full panel depth;
//            cb >= r_lo : first touch -> plain store; only the panel rows cb..r_hi contribute (W[r][cb] = 0 for r < cb).
// ROLE 0 = the big trailing ("tail") update, 1 = the rank-128 launches inside a group's triangle (latency-critical:
// raised wave priority), 2 = the look-ahead update of the next group's triangle (few tiles, critical path: burst
// loads), 3 = the "head" rows of the next group on the second helper stream (bulk-sized), 4 = role 1 with burst loads
// (few latents).  Separate symbols keep the launch shapes apart in kernel traces and counter passes.
// MT = 2 splits every 128 x 128 tile into two 64-row halves (grid.y doubled): twice the workgroups for the chain's
// launches when full tiles would not fill the CUs.  (k_panel cannot be split this way: it works in place and every
// output row needs all 128 input rows of its column.)
template <typename T, int ROLE, int MT = 4>
__global__ __launch_bounds__(NTHREADS, TILE_MIN_WAVES<T>) void k_update(T *A, int64_t lda, int64_t strideA, int ib0, int r_lo, int r_hi,
                                                      ColMap<T> cm, int skip_ib, int skip_jb) {
  if (ROLE == 1 || ROLE == 4) __builtin_amdgcn_s_setprio(3);
  if (ROLE == 2 || ROLE == 3) __builtin_amdgcn_s_setprio(2);       // look-ahead rows: the next chain waits for them
  // plain row-major tile order: an XCD-dealt super-block order (xcd_tri_decode, gemm_core.hpp) was 3 % faster for a
  // launch that has the GPU to itself and 25 % slower in the sweep, where launches from several streams
  // interleave and "workgroup w lands on XCD w % 8" no longer holds
  constexpr int SPLIT = 4 / MT;                                      // half tiles per tile
  const int bx = blockIdx.x, ib = ib0 + (int)blockIdx.y / SPLIT, lat = blockIdx.z;
  const int h0 = ((int)blockIdx.y % SPLIT) * (32 * MT);             // first row of this half inside the block row
  int kr0 = r_lo * NB, depth = (r_hi - r_lo + 1) * NB;
  bool first = false;
  T *Al = A + (int64_t)lat * strideA;
  T *Cb = Al;                          // base / leading dimension of the C tile and of the B operand
  int64_t ldc = lda, col0;
  if (bx < cm.nU) {
    const int jb = cm.u0 + bx;
    if (jb < ib || (ib < skip_ib && jb < skip_jb)) return;
    col0 = (int64_t)jb * NB;
  } else if (bx < cm.nU + cm.Taug) {
    col0 = cm.n_pad + (int64_t)(bx - cm.nU) * NB;
  } else {
    const int cb = cm.w0 + bx - cm.nU - cm.Taug;
    Cb = cm.W + (int64_t)lat * cm.strideW;
    ldc = cm.ldw;
    col0 = (int64_t)cb * NB;
    // columns that start inside the current group: W[r][cb] = 0 for r < cb, so only panel rows
    // cb..r_hi contribute, and this is the first time the tile is touched -> plain store
    if (cb >= r_lo) { first = true; kr0 = cb * NB; depth = (r_hi - cb + 1) * NB; }
  }
  __shared__ __align__(16) T smem[tile_smem_elems<T>()];
  Acc<T, MT> acc;
  acc.zero();
  const T *Ap = Al + (int64_t)kr0 * lda + (int64_t)ib * NB + h0, *Bp = Cb + (int64_t)kr0 * ldc + col0;
  // the rank-128 updates inside a group's triangle (ROLE 1, half tiles) are latency-bound: burst loads (gemm_core.hpp)
  if constexpr (ROLE == 4) {                           // rank-128 update of the triangle, few latents (see k_panel)
    tile_mainloop_burst<T, MT, 4, CHAIN_BURST<T>>(acc, Ap, lda, Bp, ldc, depth, smem);
  } else if constexpr (ROLE == 2 && MT == 2) {         // look-ahead update of the next triangle: few tiles, critical path
    tile_mainloop_burst<T, MT, 4, (sizeof(T) == 8 ? 4 : 8)>(acc, Ap, lda, Bp, ldc, depth, smem);
  } else {
    tile_mainloop<T, false, false, MT>(acc, Ap, lda, Bp, ldc, depth, smem);
  }
  T *C = Cb + ((int64_t)ib * NB + h0) * ldc + col0;
  if (first) tile_writeback<T, WB_STORE_NEG, MT>(acc, C, ldc, smem);   // C = -P^T P (first touch of a W tile)
  else tile_writeback<T, WB_SUB, MT>(acc, C, ldc, smem);               // C -= P^T P
}

// ---- The chain of one group as ONE resident launch (chain_engine.hpp).  1-D grid of q + NP workgroups:
//  * workgroup l < q is the CRITICAL workgroup of latent l: it factors / inverts the diagonal blocks (diag_body) and carries the two
//    products between two of them itself -- the panel tile right of the diagonal block and the update of the next diagonal block --
//    so the critical path of a block row never crosses to another workgroup;
//  * the NP others form a POOL that takes every other tile operation of the q triangles and inverse triangles by TICKET: one
//    atomic counter hands out the positions of one list -- the operations in the order of the launches this replaces (per block
//    row r: panel tiles P <- V_r^T P of row r, U tiles right of the diagonal and W tiles left of it; then the rank-128 updates
//    C -= P_i^T P_j of every tile below, W tiles of column r being first touches C = -P^T W_rr), latent fastest.
// That list is a topological order of the dependency graph and a ticket is only ever held by a resident workgroup, so the
// lowest unfinished operation can always run -- deadlock-free for ANY number of resident pool workgroups, with no assumption on
// placement or dispatch order -- PROVIDED the q critical workgroups of the launch are resident: they are dispatched first
// (lowest block indices) and the host keeps q <= CHAIN_QB per launch (potrf_impl) so that they always fit.  Every wait is
// bounded besides (chain_wait).  Dependencies = version counters per tile (`cnt`, per latent: 8 x 8 U tiles, 8 x 8 W tiles): tile
// U(i,j) is final at version i + 1 (i updates + its panel solve / factorisation), W(i,c) at i - c + 1.  Every tile receives its
// operations in the order of the launches, each the same K = 128 product: the results are bit-identical to the launch-per-step chain.
// Counters live in the pad column of each latent's group scratch Wg (row 0: the 128 counters; row 1 of latent 0: finished
// workgroups, abort word, ticket counter); zero at entry (k_zero_diag_out at the start of the sweep), zeroed again by the last
// workgroup to finish.
// dev: -DPLMC_CHAIN_TRACE stamps the 100 MHz wall clock of the critical workgroup's phases into the pad column of its latent's Wg
// (rows 2..; tools/chain_trace.py)
#ifdef PLMC_CHAIN_TRACE
#define PLMC_CH_STAMP()                                                                                                          \
  do {                                                                                                                           \
    if (crit && threadIdx.x == 0) {                                                                                              \
      reinterpret_cast<long long *>(Wg + (int64_t)blockIdx.x * strideW + (int64_t)(2 + tr_n / 32) * LDG + (int64_t)GMAX * NB)[tr_n % 32] = wall_clock64(); \
      ++tr_n;                                                                                                                    \
    }                                                                                                                            \
  } while (0)
#else
#define PLMC_CH_STAMP() do { } while (0)
#endif
template <typename T>
__global__ __launch_bounds__(CH_NT, 2) void k_chain(T *A, int64_t lda, int64_t strideA, int g0, int G, T *__restrict__ Vd, int64_t strideV, T *Wg,
                                                     int64_t strideW, int q) {
  constexpr int SMEM = tile_smem_elems<T>() > DIAG_LDS ? tile_smem_elems<T>() : DIAG_LDS;
  __shared__ __align__(16) T smem[SMEM + 8];
  int *lds_ctl = reinterpret_cast<int *>(smem + SMEM);     // [0] hand-over flag of diag_body, [1] result of a wait / the ticket drawn
  const bool crit = (int)blockIdx.x < q;
  int *ctl = reinterpret_cast<int *>(Wg + (int64_t)LDG + (int64_t)GMAX * NB);        // latent 0: [0] finished workgroups, [1] abort, [2] tickets
  // helper operations per latent: row r has (G - 1) panel tiles and sum_i ((G - i) + (r + 1)) update tiles, two of them (one of
  // each) the critical workgroup's while a next block row exists
  auto row_ops = [&](int r) { const int below = G - 1 - r; return (G - 1) + below * (G - r) / 2 + below * (r + 1) - (below > 0 ? 2 : 0); };
  int nops = 0;
  for (int r = 0; r < G; ++r) nops += row_ops(r);
  if (crit) __builtin_amdgcn_s_setprio(3);
  else __builtin_amdgcn_s_setprio(2);
  int step = 0;                                             // critical workgroup: position in its own sequence (2 per block row)
#ifdef PLMC_CHAIN_TRACE
  int tr_n = 0;
#endif
  PLMC_CH_STAMP();
#pragma unroll 1
  while (true) {
    int lat, r, ph, i, t;
    if (crit) {
      lat = blockIdx.x;
      r = step >> 1;
      const int kind = step & 1;
      ++step;
      if (r >= G) break;
      T *Al = A + (int64_t)lat * strideA + (int64_t)g0 * NB * lda + (int64_t)g0 * NB;
      T *Wl = Wg + (int64_t)lat * strideW;
      T *Vr = Vd + (int64_t)lat * strideV + (int64_t)(g0 + r) * NB * NB;
      int *cnt = reinterpret_cast<int *>(Wl + (int64_t)GMAX * NB);
      if (kind == 0) {
        // ---- D(r): factor + invert the diagonal block (its last update was this workgroup's own: nothing to wait for beyond the
        // entry state).  Its loads bypass the L1, V_r and W(r,r) leave write-through (SC1 body)
        if (r == 0 && !chain_wait(cnt, 0, nullptr, 0, nullptr, 0, ctl + 1, lds_ctl + 1)) return;
        PLMC_CH_STAMP();
        diag_body<T, 0, true>(Al + (int64_t)r * NB * lda + (int64_t)r * NB, (unsigned)lda, Vr, Wl + (int64_t)r * NB * LDG + (int64_t)r * NB, (unsigned)LDG,
                              smem, lds_ctl);
        PLMC_CH_STAMP();
        chain_post(cnt + r * GMAX + r, r + 1);
        PLMC_CH_STAMP();
        continue;
      }
      if (r + 1 >= G) continue;
      // ---- the two products between D(r) and D(r + 1), fused (chain_panel_then_update): P = V_r^T U(r, r+1), then
      // U(r+1, r+1) -= P^T P with P taken from LDS.  Inputs from the pool: both tiles at version r
      T *Pg = Al + (int64_t)r * NB * lda + (int64_t)(r + 1) * NB, *Cg = Al + (int64_t)(r + 1) * NB * lda + (int64_t)(r + 1) * NB;
      int *cP = cnt + r * GMAX + r + 1, *cC = cnt + (r + 1) * GMAX + r + 1;
      if (!chain_wait(cP, r, cC, r, nullptr, 0, ctl + 1, lds_ctl + 1)) return;
      PLMC_CH_STAMP();
      Acc<T, 2, 4> accp, accu;
      accp.zero();
      accu.zero();
      ChainC<T> vcc = {};
      constexpr bool CPRE2 = sizeof(T) == 4;
      auto pre2 = [&]() { if (CPRE2) chain_cload<T>(vcc, Cg, lda); };
      chain_mainloop<T>(accp, Vr, NB, Pg, lda, smem, pre2, true);
      PLMC_CH_STAMP();
      chain_panel_then_update<T>(accp, accu, Pg, lda, smem);
      PLMC_CH_STAMP();
      chain_post(cP, r + 1);
      PLMC_CH_STAMP();
      if (!CPRE2) chain_cload<T>(vcc, Cg, lda);
      chain_writeback<T>(accu, Cg, lda, smem, WB_SUB, vcc);
      PLMC_CH_STAMP();
      chain_post(cC, r + 1);
      PLMC_CH_STAMP();
      continue;
    } else {
      if (threadIdx.x == 0) lds_ctl[1] = __hip_atomic_fetch_add(ctl + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const int tk = lds_ctl[1];
      __syncthreads();
      if (tk >= nops * q || tk < 0) break;
      lat = tk % q;
      int e = tk / q;
      r = 0;
      while (e >= row_ops(r)) { e -= row_ops(r); ++r; }
      const int skip = r + 1 < G ? 1 : 0;                   // tile 0 of the panel and of update row r + 1 is the critical workgroup's
      if (e < (G - 1) - skip) { ph = 0; i = r; t = e + skip; }
      else {
        e -= (G - 1) - skip;
        ph = 1;
        i = r + 1;
        while (true) {
          const int n_i = (G - i) + (r + 1) - (i == r + 1 ? 1 : 0);
          if (e < n_i) break;
          e -= n_i;
          ++i;
        }
        t = e + (i == r + 1 ? 1 : 0);
      }
    }
    // ---- one tile operation: phase 0 = panel of row r (i == r), phase 1 = rank-128 update of row i > r with the panel row r
    T *Al = A + (int64_t)lat * strideA + (int64_t)g0 * NB * lda + (int64_t)g0 * NB;     // U(0,0) of the group
    T *Wl = Wg + (int64_t)lat * strideW;                                                 // W(0,0)
    int *cnt = reinterpret_cast<int *>(Wl + (int64_t)GMAX * NB);
    auto U = [&](int a, int b) { return Al + (int64_t)a * NB * lda + (int64_t)b * NB; };
    auto W = [&](int a, int b) { return Wl + (int64_t)a * NB * LDG + (int64_t)b * NB; };
    auto cU = [&](int a, int b) { return cnt + a * GMAX + b; };
    auto cW = [&](int a, int b) { return cnt + GMAX * GMAX + a * GMAX + b; };
    const int nUt = ph ? G - i : G - 1 - r;
    const bool isU = t < nUt;
    const int j = isU ? (ph ? i + t : r + 1 + t) : t - nUt;                         // U column (>= i / > r) or W column (<= r / < r)
    T *C = isU ? U(i, j) : W(i, j);                                                 // the tile written (phase 0: in place)
    const int64_t ldc = isU ? lda : (int64_t)LDG;
    int *c = isU ? cU(i, j) : cW(i, j);
    const int v = isU ? r : r - j;                                                  // version the earlier rows left the target in
    const T *Ap = ph ? U(r, i) : Vd + (int64_t)lat * strideV + (int64_t)(g0 + r) * NB * NB;
    const int64_t lda_ = ph ? lda : (int64_t)NB;
    const T *Bp = ph ? (isU ? U(r, j) : W(r, j)) : C;
    // inputs: phase 0: D(r); phase 1: the panel tiles U(r,i) and U(r,j) / W(r,j) (W(r,r) comes out of D(r))
    const int *pa = ph ? cU(r, i) : cU(r, r);
    const int *pb = ph ? (isU ? cU(r, j) : (j == r ? cU(r, r) : cW(r, j))) : nullptr;
    const int vb = isU ? r + 1 : (j == r ? r + 1 : r - j + 1);
    if (!chain_wait(c, v, pa, r + 1, pb, vb, ctl + 1, lds_ctl + 1)) return;
    PLMC_CH_STAMP();
    Acc<T, 2, 4> acc;
    acc.zero();
    const int mode = ph == 0 ? WB_STORE : ((!isU && j == r) ? WB_STORE_NEG : WB_SUB);
    ChainC<T> vc = {};
    // fp32: the C tile of a read-modify-write is requested right behind the operand loads (32 more registers); fp64 (whose
    // operand bursts already fill the registers) requests it after the product
    constexpr bool CPRE = sizeof(T) == 4;
    auto pre = [&]() { if (CPRE && mode == WB_SUB) chain_cload<T>(vc, C, ldc); };
    chain_mainloop<T>(acc, Ap, lda_, Bp, ldc, smem, pre, ph == 0);                // (panel products: V_r is upper triangular)
    PLMC_CH_STAMP();
    if (!CPRE && mode == WB_SUB) chain_cload<T>(vc, C, ldc);
    chain_writeback<T>(acc, C, ldc, smem, mode, vc);
    PLMC_CH_STAMP();
    chain_post(c, v + 1);
    PLMC_CH_STAMP();
  }
  // the last workgroup to get here leaves counters and control words zeroed for the next group's launch
  __syncthreads();
  if (threadIdx.x == 0) {
    const int prev = __hip_atomic_fetch_add(ctl, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    lds_ctl[1] = prev == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (lds_ctl[1]) {
    for (int l = 0; l < q; ++l)
      if (threadIdx.x < 2 * GMAX * GMAX) chain_st(reinterpret_cast<int *>(Wg + (int64_t)l * strideW + (int64_t)GMAX * NB) + threadIdx.x, 0);
    if (threadIdx.x < 3) chain_st(ctl + threadIdx.x, 0);
  }
}

// ---- scales of the split engine's operand families (bf3_engine.hpp).  Per latent eight floats in the Vd scratch:
enum { SC_SU = 0, SC_SW = 1, SC_RU = 2, SC_RW = 3, SC_SA = 4, SC_RA = 5, SC_N = 8, SC_TAG = 8 + 3 * 32 };   // SC_TAG: behind the scan partials (3 SCAN_PARTS)
//   SU solved rows, U columns:      |U_kj| <= sqrt(A_jj) <= sqrt(D),  D = largest diagonal entry of the input
//   SW solved rows, W columns, and the inverse triangle Vgg:   |W_ij| <= ||U^-1||_2 = 1 / sqrt(lambda_min)
//   RU raw rows (before their panel solve), U columns: entries of Schur complements, <= D
//   RW raw rows, W columns:  -U[<g, R]^T W[<g, c],  <= ||U[:, j]||_2 ||W||_2 <= sqrt(D / lambda_min)
//   SA solved rows, augmented columns:  z = U^-T r,  |z_i| <= ||W||_2 ||r||_2 <= R / sqrt(lambda_min),  R = sqrt(n) max |r_ij| >= the
//      2-norm of every augmented column of the input
//   RA raw rows, augmented columns:  r[R] - U[<g, R]^T z[<g],  <= R (1 + sqrt(D / lambda_min))
// lambda_min is bounded below by the caller's `eig_lo` (the noise variance of a GP covariance) -- and by the smallest
// diagonal entry, which brings in the identity padding of the rows beyond n (eigenvalue 1, whatever the noise).  SplitB3
// needs no scales (all 1).  grid (q).
// Two launches: k_scale_scan (grid (SCAN_PARTS, q): largest / smallest diagonal entry and largest |augmented entry| of a slice
// of the rows, into 3 SCAN_PARTS floats behind the scales) and k_split_scales (grid (q): reduction of the partials + the scales).
constexpr int SCAN_PARTS = 32;
__global__ __launch_bounds__(NTHREADS) void k_scale_scan(const float *__restrict__ A, int64_t n_pad, int64_t lda, int64_t strideA, int naug_pad,
                                                         float *__restrict__ sc, int64_t sc_stride) {
  __shared__ float r0[NTHREADS], r1[NTHREADS], r2[NTHREADS];
  const int part = blockIdx.x, lat = blockIdx.y;
  const int64_t rows = (n_pad + SCAN_PARTS - 1) / SCAN_PARTS, i0 = part * rows, i1 = i0 + rows < n_pad ? i0 + rows : n_pad;
  float d = 0.0f, dm = 3.0e38f, am = 0.0f;
  for (int64_t i = i0 + threadIdx.x; i < i1; i += NTHREADS) {
    const float v = A[(int64_t)lat * strideA + i * lda + i];
    d = fmaxf(d, v);
    dm = fminf(dm, v);
  }
  // largest |entry| of the augmented columns: a maximum, unlike a sum, does not depend on the order -- the scales are the
  // same in every run; sqrt(n_pad) times it bounds the 2-norm of every augmented column
  const int cq = naug_pad / 4;                                        // 16-byte pieces per row
  const int64_t items = (i1 > i0 ? i1 - i0 : 0) * cq;
  const float *Aa = A + (int64_t)lat * strideA + i0 * lda + n_pad;
#pragma unroll 8
  for (int64_t w = threadIdx.x; w < items; w += NTHREADS) {
    const float4 v = *reinterpret_cast<const float4 *>(Aa + (w / cq) * lda + (w % cq) * 4);
    am = fmaxf(fmaxf(am, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  r0[threadIdx.x] = d; r1[threadIdx.x] = dm; r2[threadIdx.x] = am;
  __syncthreads();
  for (int k = NTHREADS / 2; k > 0; k >>= 1) {
    if ((int)threadIdx.x < k) {
      r0[threadIdx.x] = fmaxf(r0[threadIdx.x], r0[threadIdx.x + k]);
      r1[threadIdx.x] = fminf(r1[threadIdx.x], r1[threadIdx.x + k]);
      r2[threadIdx.x] = fmaxf(r2[threadIdx.x], r2[threadIdx.x + k]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float *o = sc + (int64_t)lat * sc_stride + SC_N + 3 * part;
    o[0] = r0[0]; o[1] = r1[0]; o[2] = r2[0];
  }
}
template <class S>
__global__ __launch_bounds__(64) void k_split_scales(int64_t n_pad, const float *__restrict__ eig_lo, float *__restrict__ sc, int64_t sc_stride) {
  const int lat = blockIdx.x;
  float *o = sc + (int64_t)lat * sc_stride;
  if (threadIdx.x == 0) {
    o[SC_TAG] = (float)S::NPL;               // which scheme wrote the planes of this scratch (checked by the K^-1 kernel)
    // ... and the per-latent stride of this scratch in 128 x 128 blocks (exact in a float: < 2^24 blocks): a sweep that keeps its
    // planes (with_inverse | 4) has a larger one than vd_w_planes assumes, and plmc_kinv_grad_vd_* may be handed either scratch
    o[SC_TAG + 1] = (float)(sc_stride / ((int64_t)NB * NB));
  }
  if constexpr (S::NPL == 3) {
    if (threadIdx.x < SC_N) o[threadIdx.x] = 1.0f;
  } else {
    if (threadIdx.x == 0) {
      float dmax = 0.0f, dmin = 3.0e38f, amax = 0.0f;
      for (int p = 0; p < SCAN_PARTS; ++p) {
        dmax = fmaxf(dmax, o[SC_N + 3 * p]);
        dmin = fminf(dmin, o[SC_N + 3 * p + 1]);
        amax = fmaxf(amax, o[SC_N + 3 * p + 2]);
      }
      const float D = dmax > 0.0f ? dmax : 1.0f;
      const float Rn = sqrtf((float)n_pad) * amax + 1e-30f;     // >= the 2-norm of every augmented column (zero right-hand side: any scale will do)
      float lam = fminf(eig_lo[lat], dmin);                      // (lambda_min <= every diagonal entry: still a lower bound)
      if (!(lam > 1e-12f * D)) lam = 1e-12f * D;                // no usable bound: assume a condition number of 1e12
      o[SC_SU] = b3_scale_for(sqrtf(D));
      o[SC_SW] = b3_scale_for(1.0f / sqrtf(lam));
      o[SC_RU] = b3_scale_for(D);
      o[SC_RW] = b3_scale_for(sqrtf(D / lam));
      o[SC_SA] = b3_scale_for(Rn / sqrtf(lam));
      o[SC_RA] = b3_scale_for(Rn * (1.0f + sqrtf(D / lam)));
      o[6] = D; o[7] = lam;
    }
  }
}

// fp32, split engine (PLMC_SPLIT != 0, the fp32 default): the bulk trailing updates (tail, head, look-ahead) with their depth-(128 G)
// products on the 16-bit matrix cores (bf3_engine.hpp).  A workgroup of 512 threads takes the macro tile (block rows ib, ib + 1) x (column tile
// bx).  The operands are the panel rows of the current group, which k_gpanel_bf3 / k_wtri_planes also write as k8-ordered
// bf16 planes into a rolling two-group buffer `Pl` (per buffer b3_elems(128 GMAX, lda) elements, same column coordinates as
// the factor buffer); tile decoding, skip and depth rules and the write-back are those of k_update, per half.
// Block rows ib < raw_end are the NEXT group's rows and this is their last update: their final values also go, as planes,
// into `Praw` (row 128 (ib - ib0) .., same columns) -- the right-hand side of the next group panel (k_gpanel_bf3).
// grid (nU + Taug + nW, (nrows + 1) / 2, q).
template <class S, int ROLE>
__global__ __launch_bounds__(B3_NT, 2) void k_update_bf3(float *A, int64_t lda, int64_t strideA, int ib0, int nrows, int r_lo, int r_hi,
                                                         ColMap<float> cm, int skip_ib, int skip_jb, const unsigned short *__restrict__ Pl,
                                                         int64_t pl_lat_stride, int64_t wcol0, unsigned short *__restrict__ Praw,
                                                         int64_t praw_lat_stride, int raw_end, const float *__restrict__ sc, int64_t sc_stride,
                                                         const unsigned short *__restrict__ Wk, int64_t wk_lat_stride) {
  if (ROLE == 2 || ROLE == 3) __builtin_amdgcn_s_setprio(2);
  __shared__ __align__(16) unsigned char lds[b3_lds_bytes<S>()];
  const int bx = blockIdx.x, ibm = ib0 + 2 * (int)blockIdx.y, lat = blockIdx.z;
  int kr0 = r_lo * NB, depth = (r_hi - r_lo + 1) * NB;
  bool first = false;
  float *Al = A + (int64_t)lat * strideA;
  float *Cb = Al;
  int64_t ldc = lda, col0, colp;                                       // colp: column of the B operand in the plane buffer
  bool v0 = true, v1 = ibm + 1 < ib0 + nrows;                          // which halves have a tile
  const float *scl = sc + (int64_t)lat * sc_stride;
  float sB = scl[SC_SU], sRaw = scl[SC_RU];                             // scale of the B operand's family / of this tile's raw planes
  if (bx < cm.nU) {
    const int jb = cm.u0 + bx;
    v0 = jb >= ibm && !(ibm < skip_ib && jb < skip_jb);
    v1 = v1 && jb >= ibm + 1 && !(ibm + 1 < skip_ib && jb < skip_jb);
    col0 = colp = (int64_t)jb * NB;
  } else if (bx < cm.nU + cm.Taug) {
    col0 = colp = cm.n_pad + (int64_t)(bx - cm.nU) * NB;
    sB = scl[SC_SA];
    sRaw = scl[SC_RA];
  } else {
    const int cb = cm.w0 + bx - cm.nU - cm.Taug;
    Cb = cm.W + (int64_t)lat * cm.strideW;
    ldc = cm.ldw;
    col0 = (int64_t)cb * NB;
    colp = wcol0 + col0;
    sB = scl[SC_SW];
    sRaw = scl[SC_RW];
    if (cb >= r_lo) { first = true; kr0 = cb * NB; depth = (r_hi - cb + 1) * NB; }
  }
  if (!v0 && !v1) return;
  Acc<float> acc0, acc1;
  acc0.zero();
  acc1.zero();
  const unsigned short *Pr = Pl + (int64_t)lat * pl_lat_stride + b3_index<S>(kr0 - r_lo * NB, 0, 0, lda);
  const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8);
  float *C = Cb + (int64_t)(ibm + half) * NB * ldc + col0;
  float *stg = reinterpret_cast<float *>(lds + half * B3_WB_BYTES);
  const bool live = half ? v1 : v0;
  const int tid = (int)threadIdx.x & 255;
  // the C rows of the first write-back pass are requested two stages before the main loop ends (nothing for a first touch)
  f32x4 vc0[B3_WB_NCH];
  auto pre = [&]() { b3_preload(vc0, C, ldc, tid, live && !first); };
  // B operand: the column's planes of the group's rows -- U / augmented columns in the rolling buffer, inverse-factor columns in
  // the full-height planes of W (global row index, n_pad columns per plane row)
  const unsigned short *Bp = Pr + colp * 8;
  int64_t ldb = lda;
  if (bx >= cm.nU + cm.Taug) { Bp = Wk + (int64_t)lat * wk_lat_stride + b3_index<S>(kr0, 0, col0, cm.n_pad); ldb = cm.n_pad; }
  b3_mainloop<S, 2, B3_WB_NCH>(acc0, acc1, Pr + (int64_t)ibm * NB * 8, lda, Bp, ldb, depth, lds, pre);
  b3_combine<S>(acc0, acc1, 1.0f / (scl[SC_SU] * sB));
  if (Praw && ibm < raw_end) {              // uniform per workgroup; a half at or beyond raw_end writes no planes
    unsigned short *Pp = Praw + (int64_t)lat * praw_lat_stride + b3_index<S>((int64_t)(ibm + half - ib0) * NB, 0, colp, lda);
    const bool pl = ibm + half < raw_end;
    if (first) b3_writeback<S, WB_STORE_NEG, true>(acc0, C, ldc, stg, tid, live, Pp, lda, pl, sRaw);
    else b3_writeback<S, WB_SUB, true, true>(acc0, C, ldc, stg, tid, live, Pp, lda, pl, sRaw, vc0);
    return;
  }
  if (first) b3_writeback<S, WB_STORE_NEG, false>(acc0, C, ldc, stg, tid, live);   // first touch of a W tile
  else b3_writeback<S, WB_SUB, false, true>(acc0, C, ldc, stg, tid, live, nullptr, 0, true, 1.0f, vc0);
}

// Group panel: U^-T applied to the whole block row of the group as products with Vgg = Ugg^-1 (upper, K-major, leading
// dimension ldv).  One workgroup per (128-column strip t of the column map, block row i of the group):
//     Pb[i][t] = sum_{k <= i} Vgg[k][i]^T A[k][strip t]         (depth 128 (i + 1))
// written OUT OF PLACE into the panel buffer Pb (G x tiles tiles, leading dimension ldp) -- in place, the workgroup
// of row i would overwrite what the workgroups of rows > i still read -- and copied back by k_gpanel_copy.  (A strip
// form, one workgroup walking i = G-1 .. 0 in place, needs no buffer but has G times fewer workgroups: 61 TF at q = 8,
// 15 TF at q = 1 on the benchmark shape.)  Heavy rows first.  grid (tiles, G, q).
template <typename T, int HEAD>
__global__ __launch_bounds__(NTHREADS, (sizeof(T) == 8 || HEAD ? 2 : 4)) void k_gpanel_rows(const T *A, int64_t lda, int64_t strideA, int g0,
                                                           int G, ColMap<T> cm, const T *__restrict__ Vg, int64_t ldv,
                                                           int64_t strideVg, T *__restrict__ Pb, int64_t ldp, int64_t strideP) {
  constexpr int head = HEAD;
  if (head) __builtin_amdgcn_s_setprio(2);
  __shared__ __align__(16) T smem[tile_smem_elems<T>()];
  const int lat = blockIdx.z, t = blockIdx.x;
  const T *S;
  int64_t lds = lda;
  if (t < cm.nU) S = A + (int64_t)lat * strideA + (int64_t)g0 * NB * lda + (int64_t)(cm.u0 + t) * NB;
  else if (t < cm.nU + cm.Taug) S = A + (int64_t)lat * strideA + (int64_t)g0 * NB * lda + cm.n_pad + (int64_t)(t - cm.nU) * NB;
  else {
    lds = cm.ldw;
    S = cm.W + (int64_t)lat * cm.strideW + (int64_t)g0 * NB * lds + (int64_t)(cm.w0 + t - cm.nU - cm.Taug) * NB;
  }
  if constexpr (HEAD) {
    // head launch (latency-critical, few strips; grid.y = 2 G): one 64-row half of one block row per workgroup, heavy
    // rows first.  What bounds a workgroup here is the number of dependent global round trips (5-10 us each beside the
    // bulk kernels), so the half tile keeps HEAD_BURST slabs in flight (the registers a full tile would need for that
    // do not fit) and the launch has twice the workgroups.
    constexpr int HEAD_BURST = sizeof(T) == 8 ? 4 : 8;
    const int i = G - 1 - (int)blockIdx.y / 2, h0 = ((int)blockIdx.y & 1) * 64;
    Acc<T, 2> acc;
    acc.zero();
    tile_mainloop_burst<T, 2, 4, HEAD_BURST>(acc, Vg + (int64_t)lat * strideVg + (int64_t)i * NB + h0, ldv, S, lds, (i + 1) * NB, smem);
    tile_writeback<T, WB_STORE, 2>(acc, Pb + (int64_t)lat * strideP + ((int64_t)i * NB + h0) * ldp + (int64_t)t * NB, ldp, smem);
  } else {
    // bulk launch: rows y and G-1-y share a workgroup -- every workgroup then carries depth 128 (G + 1) instead of
    // 128 .. 128 G, and the short products no longer dominate the launch (77 -> ~100 TF at q = 8)
    const int y = blockIdx.y;
    const int i0 = G - 1 - y, i1 = y < G - 1 - y ? y : -1;
#pragma unroll 1
    for (int pass = 0; pass < 2; ++pass) {
      const int i = pass == 0 ? i0 : i1;
      if (i < 0) break;
      Acc<T> acc;
      acc.zero();
      tile_mainloop<T, false, false>(acc, Vg + (int64_t)lat * strideVg + (int64_t)i * NB, ldv, S, lds, (i + 1) * NB, smem);
      tile_writeback<T, WB_STORE>(acc, Pb + (int64_t)lat * strideP + (int64_t)i * NB * ldp + (int64_t)t * NB, ldp, smem);
      __syncthreads();                   // staging buffer free before the second product refills it
    }
  }
}

// Panel buffer -> factor buffer (block row g0 + i, column strip t of the column map).  grid (tiles, G, q); HBM-bound.
// (fp32 / fp64 engine only: the split engine's group panel works in place, k_gpanel_bf3.)
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_gpanel_copy(T *A, int64_t lda, int64_t strideA, int g0, ColMap<T> cm, const T *__restrict__ Pb,
                                                          int64_t ldp, int64_t strideP) {
  using vec_t = typename Traits<T>::vec_t;
  constexpr int EPV = Traits<T>::EPV, CPR = NB / EPV;
  const int lat = blockIdx.z, t = blockIdx.x, i = blockIdx.y;
  T *D;
  int64_t ldd = lda;
  if (t < cm.nU) D = A + (int64_t)lat * strideA + (int64_t)(g0 + i) * NB * lda + (int64_t)(cm.u0 + t) * NB;
  else if (t < cm.nU + cm.Taug) D = A + (int64_t)lat * strideA + (int64_t)(g0 + i) * NB * lda + cm.n_pad + (int64_t)(t - cm.nU) * NB;
  else {
    ldd = cm.ldw;
    D = cm.W + (int64_t)lat * cm.strideW + (int64_t)(g0 + i) * NB * ldd + (int64_t)(cm.w0 + t - cm.nU - cm.Taug) * NB;
  }
  const T *Src = Pb + (int64_t)lat * strideP + (int64_t)i * NB * ldp + (int64_t)t * NB;
  for (int c = threadIdx.x; c < NB * CPR; c += NTHREADS) {
    const int r = c / CPR, col = (c % CPR) * EPV;
    *reinterpret_cast<vec_t *>(D + (int64_t)r * ldd + col) = *reinterpret_cast<const vec_t *>(Src + (int64_t)r * ldp + col);
  }
}

// Transpose of the group's inverse triangle: Vg[k][i] = Wg[i][k]^T for block pairs k <= i < G (Vg = Ugg^-1, upper,
// K-major for the tile engine); when the inverse factor is wanted, the copy of Wg[i][k] into the factor buffer's W columns
// (block row g0 + i, block column g0 + k); and, for the split engine (S != void), Vg as k8-ordered planes with 128 GMAX columns
// -- the A operand of k_gpanel_bf3 -- with zeros in the blocks below the diagonal (a macro row of the panel product runs both
// of its block rows over the depth of the second).  One 128 x 128 block per workgroup, through an LDS tile: coalesced row
// loads, coalesced transposed row stores, and the planes from 8 consecutive SOURCE columns per plane column.
// (`Wg` with leading dimension lds_ and batch stride strideS: the group scratch in the sweep; the W columns of a finished
// factor buffer in plmc_potrs_aug.)  grid (planes ? GMAX * GMAX : G (G + 1) / 2, q).
template <typename T, class S>
__global__ __launch_bounds__(NTHREADS) void k_vtrans(const T *__restrict__ Wg, int64_t lds_, int64_t strideS, int64_t strideG, T *__restrict__ Vg, int G,
                                                     T *Wout, int64_t ldw, int64_t strideW, unsigned short *__restrict__ VgP, int64_t vgp_lat_stride,
                                                     const float *__restrict__ sc, int64_t sc_stride) {
  __builtin_amdgcn_s_setprio(3);
  constexpr bool PLANES = !std::is_void<S>::value;
  using SS = typename std::conditional<PLANES, S, SplitB3>::type;
  using vec_t = typename Traits<T>::vec_t;
  constexpr int EPV = Traits<T>::EPV, LT = NB + 1;
  __shared__ T tile[NB * LT];
  const int lat = blockIdx.y;
  int i, k;
  if (PLANES) { k = (int)blockIdx.x / GMAX; i = (int)blockIdx.x % GMAX; }
  else { i = 0; k = (int)blockIdx.x; while (k > i) { k -= i + 1; ++i; } }   // blockIdx.x enumerates (i, k <= i) row by row
  if constexpr (PLANES) {
    if (k > i || i >= G) {                                          // below the diagonal / beyond a short group: zero planes
      unsigned short *P = VgP + (int64_t)lat * vgp_lat_stride + b3_index<SS>((int64_t)k * NB, 0, (int64_t)i * NB, GMAX * NB);
      const b3_s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      constexpr int PER = SS::NPL * 128;
      for (int w = threadIdx.x; w < 16 * PER; w += NTHREADS) {
        const int k8 = w / PER, r = w % PER;
        *reinterpret_cast<b3_s16x8 *>(P + ((int64_t)k8 * SS::NPL * GMAX * NB + (int64_t)(r / 128) * GMAX * NB + (r % 128)) * 8) = z;
      }
      return;
    }
  }
  const T *src = Wg + (int64_t)lat * strideS + (int64_t)i * NB * lds_ + (int64_t)k * NB;
  T *dst = Vg + (int64_t)lat * strideG + (int64_t)k * NB * LDG + (int64_t)i * NB;
  T *wo = Wout ? Wout + (int64_t)lat * strideW + (int64_t)i * NB * ldw + (int64_t)k * NB : nullptr;
  constexpr int CPR = NB / EPV;                                       // 16-byte chunks per row
  for (int c = threadIdx.x; c < NB * CPR; c += NTHREADS) {            // rows of Wg[i][k]: tile[row][col]
    const int r = c / CPR, col = (c % CPR) * EPV;
    const vec_t v = *reinterpret_cast<const vec_t *>(src + (int64_t)r * lds_ + col);
    if (wo) *reinterpret_cast<vec_t *>(wo + (int64_t)r * ldw + col) = v;
#pragma unroll
    for (int e = 0; e < EPV; ++e) tile[r * LT + col + e] = v[e];
  }
  __syncthreads();
  for (int c = threadIdx.x; c < NB * CPR; c += NTHREADS) {            // rows of Vg[k][i]: dst[kk][ii] = tile[ii][kk]
    const int kk = c / CPR, ii = (c % CPR) * EPV;
    vec_t v;
#pragma unroll
    for (int e = 0; e < EPV; ++e) v[e] = tile[(ii + e) * LT + kk];
    *reinterpret_cast<vec_t *>(dst + (int64_t)kk * LDG + ii) = v;
  }
  if constexpr (PLANES && sizeof(T) == 4) {
    // plane element (kk, column ii) = tile[ii][kk]: 8 consecutive kk of one source row; thread = (k8 group, ii): 16 x 128 items
    unsigned short *P = VgP + (int64_t)lat * vgp_lat_stride + b3_index<SS>((int64_t)k * NB, 0, (int64_t)i * NB, GMAX * NB);
    const float scale = sc[(int64_t)lat * sc_stride + SC_SW];
    for (int w = threadIdx.x; w < 16 * NB; w += NTHREADS) {
      const int k8 = w / NB, ii = w % NB;
      b3_s16x8 pl[SS::NPL];
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        short o[SS::NPL];
        SS::split((float)tile[ii * LT + k8 * 8 + r] * scale, o);
#pragma unroll
        for (int p = 0; p < SS::NPL; ++p) pl[p][r] = o[p];
      }
#pragma unroll
      for (int p = 0; p < SS::NPL; ++p)
        *reinterpret_cast<b3_s16x8 *>(P + ((int64_t)(k8 * SS::NPL + p) * GMAX * NB + ii) * 8) = pl[p];
    }
  }
}

// Split engine (fp32): the group's inverse triangle W[g0 + i][g0 + k] (k <= i < G, already in the factor buffer's W columns:
// k_vtrans) as k8-ordered planes of the full-height plane buffer of W (row 128 (g0 + i) .., column 128 (g0 + k)): the operands
// of the first-touch W tiles of the tail / head updates and, later, of the K^-1 kernel.  Off the chain's stream.
// grid (G (G + 1) / 2, q).
template <class S>
__global__ __launch_bounds__(NTHREADS) void k_wtri_planes(const float *__restrict__ WA, int64_t lda, int64_t strideA, int g0,
                                                          const float *__restrict__ sc, int64_t sc_stride, unsigned short *__restrict__ Wk,
                                                          int64_t wk_lat_stride, int64_t n_pad) {
  const int lat = blockIdx.y;
  int i = 0, k = (int)blockIdx.x;
  while (k > i) { k -= i + 1; ++i; }
  b3_split_block<S, false>(WA + (int64_t)lat * strideA + (int64_t)(g0 + i) * NB * lda + (int64_t)(g0 + k) * NB, lda,
                           Wk + (int64_t)lat * wk_lat_stride + b3_index<S>((int64_t)(g0 + i) * NB, 0, (int64_t)(g0 + k) * NB, n_pad), n_pad,
                           sc[(int64_t)lat * sc_stride + SC_SW], nullptr, 0, threadIdx.x);
}

// Split engine (fp32): rows of the factor buffer as planes of `Praw` -- the raw (not yet solved) rows of the FIRST group, which
// no update kernel has written (every later group's raw rows come out of k_update_bf3).  grid (tiles of the column map, G, q).
template <class S>
__global__ __launch_bounds__(NTHREADS) void k_raw_planes(const float *__restrict__ A, int64_t lda, int64_t strideA, int g0, ColMap<float> cm,
                                                         unsigned short *__restrict__ Praw, int64_t praw_lat_stride, int64_t wcol0,
                                                         const float *__restrict__ sc, int64_t sc_stride) {
  const int lat = blockIdx.z, t = blockIdx.x, i = blockIdx.y;
  const float *Src;
  int64_t lds_ = lda, colp;
  if (t < cm.nU) { colp = (int64_t)(cm.u0 + t) * NB; Src = A + (int64_t)lat * strideA + (int64_t)(g0 + i) * NB * lda + colp; }
  else if (t < cm.nU + cm.Taug) { colp = cm.n_pad + (int64_t)(t - cm.nU) * NB; Src = A + (int64_t)lat * strideA + (int64_t)(g0 + i) * NB * lda + colp; }
  else {
    lds_ = cm.ldw;
    colp = wcol0 + (int64_t)(cm.w0 + t - cm.nU - cm.Taug) * NB;
    Src = cm.W + (int64_t)lat * cm.strideW + (int64_t)(g0 + i) * NB * lds_ + (int64_t)(cm.w0 + t - cm.nU - cm.Taug) * NB;
  }
  const float scale = sc[(int64_t)lat * sc_stride + (t < cm.nU ? SC_RU : (t < cm.nU + cm.Taug ? SC_RA : SC_RW))];
  b3_split_block<S, false>(Src, lds_, Praw + (int64_t)lat * praw_lat_stride + b3_index<S>((int64_t)i * NB, 0, colp, lda), lda, scale, nullptr, 0,
                           threadIdx.x);
}

// Split engine (fp32): the group panel on the 16-bit matrix cores.  P[i] = sum_{k <= i} Vgg[k][i]^T A[k] for the block rows
// i of the group and one 128-column strip t of the column map, from the planes of Vgg (VgP, k_vtrans) and of the raw
// rows (Praw: k_update_bf3 of the previous group / k_raw_planes).  The operands are read from plane buffers only, so the
// result goes IN PLACE into the factor buffer -- no panel buffer, no copy kernel -- and, while the tile is in LDS, as planes
// into the rolling buffer `Pl` for the trailing updates.  A workgroup takes the macro rows (2 a, 2 a + 1) at the depth of
// the second (the block Vgg[2 a + 1][2 a] is zero in VgP), heavy and light macro rows paired: y and nm - 1 - y (`paired`; the
// latency-critical launch over the few columns of the next group takes one macro row per workgroup instead).
// Accuracy: the product with the inverse triangle cancels; tools/split_numerics_probe.hip holds that case (0.37 x the error
// of the fp32 MFMA chain).  grid (tiles, paired ? (nm + 1) / 2 : nm, q), nm = (G + 1) / 2.
template <class S>
__global__ __launch_bounds__(B3_NT, 2) void k_gpanel_bf3(float *A, int64_t lda, int64_t strideA, int g0, int G, ColMap<float> cm,
                                                         const unsigned short *__restrict__ VgP, int64_t vgp_lat_stride,
                                                         const unsigned short *__restrict__ Praw, int64_t praw_lat_stride,
                                                         unsigned short *__restrict__ Pl, int64_t pl_lat_stride, int64_t wcol0,
                                                         const float *__restrict__ sc, int64_t sc_stride, int paired,
                                                         unsigned short *__restrict__ Wk, int64_t wk_lat_stride) {
  if (!paired) __builtin_amdgcn_s_setprio(2);                         // the head columns: the next chain waits for them
  __shared__ __align__(16) unsigned char lds[b3_lds_bytes<S>()];
  const int lat = blockIdx.z, t = blockIdx.x, y = blockIdx.y;
  const float *scl = sc + (int64_t)lat * sc_stride;
  const int fam = t < cm.nU ? 0 : (t < cm.nU + cm.Taug ? 2 : 1);       // U / W / augmented columns: raw scale RU / RW / RA, solved SU / SW / SA
  const float unscale = 1.0f / (scl[SC_SW] * scl[fam == 0 ? SC_RU : (fam == 1 ? SC_RW : SC_RA)]);
  const float pscale = scl[fam == 0 ? SC_SU : (fam == 1 ? SC_SW : SC_SA)];
  float *D;
  int64_t ldd = lda, colp;
  if (t < cm.nU) { colp = (int64_t)(cm.u0 + t) * NB; D = A + (int64_t)lat * strideA + (int64_t)g0 * NB * lda + colp; }
  else if (t < cm.nU + cm.Taug) { colp = cm.n_pad + (int64_t)(t - cm.nU) * NB; D = A + (int64_t)lat * strideA + (int64_t)g0 * NB * lda + colp; }
  else {
    ldd = cm.ldw;
    colp = wcol0 + (int64_t)(cm.w0 + t - cm.nU - cm.Taug) * NB;
    D = cm.W + (int64_t)lat * cm.strideW + (int64_t)g0 * NB * ldd + (int64_t)(cm.w0 + t - cm.nU - cm.Taug) * NB;
  }
  const int nm = (G + 1) / 2;
  const unsigned short *Vp = VgP + (int64_t)lat * vgp_lat_stride, *Rp = Praw + (int64_t)lat * praw_lat_stride + colp * 8;
  // planes of the result: U and augmented columns into the rolling buffer of this group's rows (lda columns per plane row); the
  // inverse-factor columns -- final here -- into the FULL-HEIGHT planes of W (n_pad columns per plane row), which the trailing
  // updates and, after the sweep, the K^-1 kernel read
  unsigned short *Pp = Pl + (int64_t)lat * pl_lat_stride + colp * 8;
  int64_t pld = lda;
  if (fam == 1) { Pp = Wk + (int64_t)lat * wk_lat_stride + b3_index<S>((int64_t)g0 * NB, 0, colp - wcol0, cm.n_pad); pld = cm.n_pad; }
  const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8), tid = (int)threadIdx.x & 255;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    const int a = pass == 0 ? nm - 1 - y : ((paired && y < nm - 1 - y) ? y : -1);
    if (a < 0) break;
    const int i0 = 2 * a, rows = G - i0 < 2 ? G - i0 : 2;
    Acc<float> acc0, acc1;
    acc0.zero();
    acc1.zero();
    b3_mainloop<S>(acc0, acc1, Vp + (int64_t)i0 * NB * 8, (int64_t)GMAX * NB, Rp, lda, (i0 + rows) * NB, lds);
    b3_combine<S>(acc0, acc1, unscale);
    const int i = i0 + half;
    b3_writeback<S, WB_STORE, true>(acc0, D + (int64_t)i * NB * ldd, ldd, reinterpret_cast<float *>(lds + half * B3_WB_BYTES), tid, half < rows,
                                    Pp + (int64_t)i * 16 * S::NPL * pld * 8, pld, true, pscale);
    __syncthreads();                     // staging areas free before the next product's DMA lands
  }
}

// K^-1 accumulation inside the sweep (with_inverse = 2).  Khat^-1 = W^T W = sum over groups of W[R]^T W[R] (R = the block
// rows of one group): as soon as the rows R of the inverse factor are final, their rank-(128 G) contribution goes into
// the K^-1 tiles (ib <= jb < g1) -- bulk work that is largest for the LAST groups, where the trailing updates of the
// sweep have shrunk and the GPU would otherwise idle behind the chain.  Only rows l >= jb contribute (W[l][jb] = 0 for
// l < jb), so tiles with jb inside the group take a shorter range and are written for the first time (plain store).
// Storage (no extra buffer): tile (ib < jb) lives in the strictly lower triangle of the factor buffer's square part,
// at block (jb, ib) -- the lower triangle is never read by anything else -- and the diagonal tiles (ib == jb) in a
// strip of the Vd scratch (Kd, leading dimension NB).  grid (g1 (g1 + 1) / 2, q).
template <typename T>
__global__ __launch_bounds__(NTHREADS, TILE_MIN_WAVES<T>) void k_kacc(T *A, int64_t lda, int64_t strideA, const T *__restrict__ W,
                                                    int64_t ldw, int64_t strideW, T *Kd, int64_t strideKd, int g0, int g1) {
  __shared__ __align__(16) T smem[tile_smem_elems<T>()];
  const int lat = blockIdx.y, t = blockIdx.x;
  int jb = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
  while ((jb + 1) * (jb + 2) / 2 <= t) ++jb;
  while (jb * (jb + 1) / 2 > t) --jb;
  const int ib = t - jb * (jb + 1) / 2;
  const bool first = jb >= g0;
  const int r0 = first ? jb : g0;
  const T *Wl = W + (int64_t)lat * strideW + (int64_t)r0 * NB * ldw;
  Acc<T> acc;
  acc.zero();
  tile_mainloop<T, false, false>(acc, Wl + (int64_t)ib * NB, ldw, Wl + (int64_t)jb * NB, ldw, (g1 - r0) * NB, smem);
  T *C;
  int64_t ldc;
  if (ib == jb) { C = Kd + (int64_t)lat * strideKd + (int64_t)ib * NB * NB; ldc = NB; }
  else { C = A + (int64_t)lat * strideA + (int64_t)jb * NB * lda + (int64_t)ib * NB; ldc = lda; }
  if (first) tile_writeback<T, WB_STORE>(acc, C, ldc, smem);
  else tile_writeback<T, WB_ADD>(acc, C, ldc, smem);
}

// The same on the split engine (fp32): the operands are the planes of the group's rows of W that the group panel and
// k_wtri_planes already wrote into the full-height plane buffer `Wk` (family SC_SW), so the accumulation needs no pass over W
// of its own.  Macro tile = K^-1 tiles (ibm, jb) and (ibm + 1, jb): both live at block row
// jb of the lower triangle, side by side.  grid (g1, (g1 + 1) / 2, q); workgroups above the diagonal leave at once.
template <class S>
__global__ __launch_bounds__(B3_NT, 2) void k_kacc_bf3(float *A, int64_t lda, int64_t strideA, float *Kd, int64_t strideKd,
                                                       const unsigned short *__restrict__ Wk, int64_t wk_lat_stride, int64_t n_pad, int g0, int g1,
                                                       const float *__restrict__ sc, int64_t sc_stride) {
  __shared__ __align__(16) unsigned char lds[b3_lds_bytes<S>()];
  const int jb = blockIdx.x, ibm = 2 * (int)blockIdx.y, lat = blockIdx.z;
  if (ibm > jb) return;
  const bool first = jb >= g0;
  const int r0 = first ? jb : g0;
  const float sW = sc[(int64_t)lat * sc_stride + SC_SW];
  const unsigned short *Pr = Wk + (int64_t)lat * wk_lat_stride + b3_index<S>((int64_t)r0 * NB, 0, 0, n_pad);
  const int half = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8), ib = ibm + half;
  const bool live = ib <= jb;
  float *C;
  int64_t ldc;
  if (ib == jb) { C = Kd + (int64_t)lat * strideKd + (int64_t)ib * NB * NB; ldc = NB; }
  else { C = A + (int64_t)lat * strideA + (int64_t)jb * NB * lda + (int64_t)ib * NB; ldc = lda; }
  float *stg = reinterpret_cast<float *>(lds + half * B3_WB_BYTES);
  const int tid = (int)threadIdx.x & 255;
  Acc<float> acc0, acc1;
  acc0.zero();
  acc1.zero();
  f32x4 vc0[B3_WB_NCH];
  auto pre = [&]() { b3_preload(vc0, C, ldc, tid, live && !first); };
  b3_mainloop<S, 2, B3_WB_NCH>(acc0, acc1, Pr + (int64_t)ibm * NB * 8, n_pad, Pr + (int64_t)jb * NB * 8, n_pad, (g1 - r0) * NB, lds, pre);
  b3_combine<S>(acc0, acc1, 1.0f / (sW * sW));
  if (first) b3_writeback<S, WB_STORE, false>(acc0, C, ldc, stg, tid, live);
  else b3_writeback<S, WB_ADD, false, true>(acc0, C, ldc, stg, tid, live, nullptr, 0, true, 1.0f, vc0);
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
// CW = the columns one workgroup takes (128, or 32 of a block column for shards of a few latents: four times the workgroups of a
// quarter of the threads each -- at one latent 64 workgroups put 64 of the 256 CUs to work, 114 us for 134 MB).  A column's sum is
// built from the same row groups in the same order either way: the results are bit-identical.
constexpr int WTMV_NT = 1024;
template <typename T, int CW = NB>
__global__ __launch_bounds__(WTMV_NT * CW / NB) void k_wt_matvec(const T *__restrict__ W, int64_t n_pad, int64_t ldw,
                                                                 int64_t strideW, const T *__restrict__ z,
                                                                 T *__restrict__ alpha) {
  using vec_t = typename Traits<T>::vec_t;
  constexpr int EPV = Traits<T>::EPV;
  constexpr int LPR = CW / EPV;                  // lanes per row segment (CW = 128: 32 fp32 / 64 fp64)
  constexpr int NRG = WTMV_NT / (NB / EPV);      // row groups (32 / 16), whatever CW
  __shared__ double red[NRG][CW];
  const int lat = blockIdx.y;
  const int cl = (threadIdx.x % LPR) * EPV, rg = threadIdx.x / LPR;
  const int64_t colw = (int64_t)blockIdx.x * CW;                   // first column of this workgroup
  const int64_t col0 = colw / NB * NB;                             // first row of its block column
  const T *Wl = W + (int64_t)lat * strideW + colw + cl;
  const T *zl = z + (int64_t)lat * n_pad;
  double s[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) s[e] = 0.0;
  // rows in flight per thread: what a workgroup keeps in flight bounds its rate (128 rows x 128 bytes = 16 KB per 256-thread workgroup
  // at 4: 8 GB/s over a 2 us round trip, 125 us for the longest column quarter); a 1024-thread workgroup at 4 is at the CU's fill rate
  constexpr int U = CW == NB ? 4 : 8;
  for (int64_t l = col0 + rg; l < n_pad; l += U * NRG) {
    vec_t v[U];
    T zz[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // unconditional loads from a clamped row, the multiplier zeroed instead: a predicated load is a branch with a full wait behind
      // it, and the U loads of a thread went out one round trip after the other (round 4, one latent: 114 -> 41 us with both changes)
      const int64_t ll = l + u * NRG;
      const int64_t lc = ll < n_pad ? ll : n_pad - 1;
      v[u] = *reinterpret_cast<const vec_t *>(Wl + lc * ldw);
      const T zv = zl[lc];
      zz[u] = ll < n_pad ? zv : T(0);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int e = 0; e < EPV; ++e) s[e] += (double)v[u][e] * (double)zz[u];
  }
#pragma unroll
  for (int e = 0; e < EPV; ++e) red[rg][cl + e] = s[e];
  __syncthreads();
  if (threadIdx.x < CW) {
    double t = 0.0;
#pragma unroll
    for (int g = 0; g < NRG; ++g) t += red[g][threadIdx.x];
    alpha[(int64_t)lat * n_pad + colw + threadIdx.x] = (T)t;
  }
}

// kd[i] = sum_{l >= block(i)} W[l][i]^2 = [Khat^-1]_ii (Khat^-1 = W^T W).  Same walk as k_wt_matvec (HBM-bound: reads the
// lower triangle of W once); fp64 accumulation.  grid (n_pad / 128, q).
template <typename T>
__global__ __launch_bounds__(WTMV_NT) void k_w_diag(const T *__restrict__ W, int64_t n_pad, int64_t ldw, int64_t strideW,
                                                    T *__restrict__ kd) {
  using vec_t = typename Traits<T>::vec_t;
  constexpr int EPV = Traits<T>::EPV;
  constexpr int LPR = 128 / EPV;
  constexpr int NRG = WTMV_NT / LPR;
  __shared__ double red[NRG][NB];
  const int lat = blockIdx.y;
  const int cl = (threadIdx.x % LPR) * EPV, rg = threadIdx.x / LPR;
  const int64_t col0 = (int64_t)blockIdx.x * NB;
  const T *Wl = W + (int64_t)lat * strideW + col0 + cl;
  double s[EPV];
#pragma unroll
  for (int e = 0; e < EPV; ++e) s[e] = 0.0;
  for (int64_t l = col0 + rg; l < n_pad; l += 4 * NRG) {         // four rows in flight per thread, as k_wt_matvec (same summation order)
    vec_t v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t ll = l + u * NRG;
      v[u] = *reinterpret_cast<const vec_t *>(Wl + (ll < n_pad ? ll : n_pad - 1) * ldw);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (l + u * NRG < n_pad) {
#pragma unroll
        for (int e = 0; e < EPV; ++e) s[e] += (double)v[u][e] * (double)v[u][e];
      }
  }
#pragma unroll
  for (int e = 0; e < EPV; ++e) red[rg][cl + e] = s[e];
  __syncthreads();
  if (threadIdx.x < NB) {
    double t = 0.0;
#pragma unroll
    for (int g = 0; g < NRG; ++g) t += red[g][threadIdx.x];
    kd[(int64_t)lat * n_pad + col0 + threadIdx.x] = (T)t;
  }
}

// ----------------------------------------------------------------------------------------------
// S: split scheme of the bulk fp32 products (bf3_engine.hpp); void = none (fp64, or PLMC_SPLIT=0: MFMA of the element type
// everywhere).  eig_lo: q lower bounds of the smallest eigenvalue (device), needed by SplitH2 only.
template <typename T, class S>
int potrf_impl(T *A, int64_t n_pad, int64_t lda, int naug, int64_t strideA, T *Vd, double *logdet, int *info,
               int with_inverse_arg, int q, const float *eig_lo, void *stream, const AssembleJob *job = nullptr) {
  constexpr bool bf3 = !std::is_void<S>::value;
  // with_inverse & 4 (split engine, with the inverse factor): KEEP the planes of the solved rows of every group instead of
  // rolling over two buffers -- Vd then has plmc_vd_blocks_keep blocks per latent -- so that plmc_potrs_aug_kept_* can later run
  // its updates on the split engine too (the eval-mode factorisation cache)
  const bool keep = bf3 && (with_inverse_arg & 4) != 0 && (with_inverse_arg & 3) != 0;
  const int with_inverse = with_inverse_arg & 3;
  using SS = typename std::conditional<bf3, S, SplitB3>::type;       // a valid scheme type for the (dead) template arguments when bf3 is off
  constexpr bool aug_fp32 = false;                                     // (the augmented columns have bounds too: k_split_scales)
  PLMC_REQUIRE(A && Vd && logdet && info, "null pointer");
  PLMC_REQUIRE(n_pad > 0 && n_pad % NB == 0 && lda % NB == 0 && lda >= n_pad, "n_pad/lda must be multiples of NB");
  const int64_t naug_pad = plmc_pad(naug);
  PLMC_REQUIRE(naug >= 0 && n_pad + naug_pad + (with_inverse ? n_pad : 0) <= lda,
               "lda too small for naug (+ the n_pad columns of the inverse factor)");
  PLMC_REQUIRE(q > 0 && aligned16(A) && aligned16(Vd), "bad q or unaligned buffer");
  const hipStream_t st = (hipStream_t)stream;
  const int m = (int)(n_pad / NB);
  const int Taug = (int)(naug_pad / NB);
  const int64_t strideV = (keep ? plmc_vd_blocks_keep(n_pad, lda) : plmc_vd_blocks_for(n_pad, lda, (int)sizeof(T))) * (int64_t)NB * NB; // per latent
  const int64_t wcol0 = n_pad + naug_pad;
  const double nb = (double)NB, nb3 = nb * nb * nb, esz = sizeof(T);
  // group scratch of latent 0 (batch stride strideV): the inverse triangle Wg and two transposed copies (ping-pong:
  // the group panel of one group may still read its copy while the chain of the next group writes the other)
  T *const Wg = Vd + (int64_t)m * NB * NB;
  T *const Vg2[2] = {Wg + (int64_t)GMAX * NB * LDG, Wg + 2 * (int64_t)GMAX * NB * LDG};
  T *const Ph = Wg + 3 * (int64_t)GMAX * NB * LDG;                      // panel buffer of the head columns (ld LDG)
  T *const Pbulk = Ph + (int64_t)GMAX * NB * LDG;                       // panel buffer of the other columns (ld lda)
  T *const Kd = Vd + strideV - (int64_t)m * NB * NB;                    // diagonal tiles of the accumulated K^-1: last m blocks
  // split engine (fp32): plane buffers behind the bulk panel buffer -- the rolling two-group buffer of the solved panel rows,
  // the raw rows of the next group, Vgg, and the scales of the operand families.  The offsets are those of the three-plane
  // scheme whatever the scheme (plmc_vd_blocks_for sizes the scratch for it).
  static_assert(!bf3 || sizeof(T) == 4, "split schemes stand in for fp32 products");
  const int64_t pl_buf = b3_elems<SplitB3>((int64_t)GMAX * NB, lda);    // 16-bit elements reserved per group of rows
  unsigned short *const Pl0 = bf3 ? reinterpret_cast<unsigned short *>(Pbulk + (int64_t)GMAX * NB * lda) : nullptr;
  unsigned short *const Praw = bf3 ? Pl0 + 2 * pl_buf : nullptr;         // raw rows of the group whose panel comes next (one group)
  // planes of Vgg (128 GMAX x 128 GMAX), ping-pong like Vg2: the rest panel of group gi (helper stream H) may still read its
  // copy while the chain stream transposes the triangle of group gi + 1 (nothing orders those two; with one copy a delayed
  // rest panel -- other sweeps in flight on the device -- read half-overwritten planes)
  const int64_t vgp_elems = b3_elems<SplitB3>((int64_t)GMAX * NB, (int64_t)GMAX * NB);
  unsigned short *const VgP2[2] = {bf3 ? Praw + pl_buf : nullptr, bf3 ? Praw + pl_buf + vgp_elems : nullptr};
  float *const scl = bf3 ? reinterpret_cast<float *>(VgP2[1] + vgp_elems) : nullptr;   // SC_N floats
  // ... and, with the inverse factor, the full-height planes of W (n_pad columns per plane row) that plmc_kinv_grad_vd_* reads:
  // written by the same epilogues that write the rolling buffer, so the K^-1 kernel needs no split pass over W
  unsigned short *const Wk = (bf3 && with_inverse && vd_wk_blocks(n_pad, lda, (int)sizeof(T)) > 0)
                                 ? reinterpret_cast<unsigned short *>(reinterpret_cast<float *>(scl) + (int64_t)NB * NB) : nullptr;
  const int64_t pl_lat = strideV * (int64_t)(sizeof(T) / 2);            // latent stride in 16-bit elements
  const int64_t sc_lat = strideV * (int64_t)sizeof(T) / 4;              // ... in floats
  // block rows per group (the dev knob PLMC_GRP; as below).  A sweep that KEEPS its planes always works in groups of GMAX:
  // plmc_vd_blocks_keep reserves one buffer per GMAX block rows and plmc_potrs_aug_kept walks them in groups of GMAX (a
  // smaller PLMC_GRP would write more buffers than were reserved and hand the cached substitution planes of the wrong rows)
  const int grp_rows = (!keep && knobs().grp > 0 && knobs().grp < GMAX) ? knobs().grp : GMAX;
  // (keep: one buffer per group, behind the planes of W)
  unsigned short *const Uk0 = keep ? reinterpret_cast<unsigned short *>(reinterpret_cast<float *>(Wk) + vd_wk_blocks(n_pad, lda, 4) * (int64_t)NB * NB) : nullptr;
  auto planes = [&](int g0) -> unsigned short * {                       // buffer of the group that starts at block row g0
    if (keep) return Uk0 + (int64_t)(g0 / grp_rows) * pl_buf;
    return bf3 ? Pl0 + (int64_t)((g0 / grp_rows) & 1) * pl_buf : nullptr;
  };
  const bool kacc_on = with_inverse == 2;
  T *const WA = with_inverse ? A + wcol0 : (T *)nullptr;                // inverse-factor columns of the factor buffer

  const Knobs &kn = knobs();                            // dev knobs: read once per process (api.hip)
  const double hthr = kn.half_tiles;
  const bool serial = kn.serial;
  // With one or two latents the chain is the critical path: the bulk kernels then ask for LDS they do not use, which caps
  // their occupancy -- 16 KB (three workgroups per CU) at q = 2, 20 KB (two per CU: 46 KB of LDS and half the register file
  // stay free, so a chain workgroup is placed at once instead of waiting for a bulk workgroup to retire) at q = 1.
  // q = 2: 11.4 -> 11.0 ms/step; q = 1: sweep 5.55 -> 5.30 (16 KB) -> 5.03 ms (20 KB); 26 KB and more lose again (5.36);
  // with many latents the bulk is the bound and a cap costs 1-2 %.
  const unsigned bulk_lds = (unsigned)(kn.bulk_lds >= 0 ? kn.bulk_lds : (q == 1 ? 20000 : (q == 2 ? 16000 : 0)));

  auto diag = [&](int r, int g0, hipStream_t s) {
    ProfScope ps(PK_DIAG, s, q * (2.0 / 3.0) * nb3, q * 3.0 * nb * nb * esz);
    T *wout = Wg + (int64_t)(r - g0) * NB * LDG + (int64_t)(r - g0) * NB;
    hipLaunchKernelGGL((k_diag<T>), dim3(q), dim3(DIAG_NT), 0, s, A, lda, strideA, r, Vd, strideV, wout, (int64_t)LDG, strideV);
  };
  // W part of a column map: the group scratch, shifted so that absolute block indices address it
  auto cm_tri = [&](int g0, int u0, int nU, int w0, int nW) {
    return ColMap<T>{u0, nU, 0, w0, nW, n_pad, Wg - (int64_t)g0 * NB * LDG - (int64_t)g0 * NB, (int64_t)LDG, strideV};
  };
  // ... or the factor buffer's own columns (aug + inverse factor)
  auto cm_buf = [&](int u0, int nU, int taug, int w0, int nW) {
    return ColMap<T>{u0, nU, taug, w0, WA ? nW : 0, n_pad, WA, lda, strideA};
  };
  auto panel = [&](int r, const ColMap<T> &cm, hipStream_t s) {
    const int nt = cm.nU + cm.Taug + cm.nW;
    if (nt == 0) return;
    // algorithmic: triangular solve of nt*NB columns with a 128 x 128 factor = nb^2 flops per column
    ProfScope ps(PK_PANEL, s, q * (double)nt * nb3, q * 2.0 * nt * nb * nb * esz);
    if ((double)nt * q <= hthr && q <= 2)
      hipLaunchKernelGGL((k_panel<T, 2, true>), dim3(nt, q, 2), dim3(NTHREADS), 0, s, A, lda, strideA, r, cm, Vd, strideV);
    else if ((double)nt * q <= hthr)
      hipLaunchKernelGGL((k_panel<T, 2>), dim3(nt, q, 2), dim3(NTHREADS), 0, s, A, lda, strideA, r, cm, Vd, strideV);
    else
      hipLaunchKernelGGL((k_panel<T, 4>), dim3(nt, q, 1), dim3(NTHREADS), 0, s, A, lda, strideA, r, cm, Vd, strideV);
  };
  // cls: profiler class of the launch -- PK_TRAIL (the big trailing update), PK_TRAIL_HEAD (look-ahead updates of the next
  // group's rows: `crit` = the triangle on the chain stream, else the other columns on the second helper stream),
  // PK_TRAIL_ROW (rank-128 update inside a group's triangle)
  // raw_end (split engine): block rows below it are the next group's -- this launch is their last update and also writes
  // their planes (Praw) for the next group panel
  auto update = [&](int ib0, int nrows, int r_lo, int r_hi, const ColMap<T> &cm, hipStream_t s, int cls, int skip_ib = 0,
                    int skip_jb = 0, bool crit = false, int raw_end = 0) {
    const int Cn = cm.nU + cm.Taug + cm.nW;
    if (nrows <= 0 || Cn == 0) return;
    const double depth = (r_hi - r_lo + 1) * nb, nr = (double)nrows;
    // algorithmic flops: upper tiles of the U part (diagonal tiles count half), rectangular aug / W parts
    double tilesU = 0.0, halfU = 0.0;
    for (int i = 0; i < nrows; ++i)
      for (int t = 0; t < cm.nU; ++t) {
        const int ib = ib0 + i, jb = cm.u0 + t;
        if (jb < ib || (ib < skip_ib && jb < skip_jb)) continue;
        tilesU += 1.0;
        if (jb == ib) halfU += 0.5;
      }
    double depthW = 0.0;
    int nfirst = 0;
    for (int c = 0; c < cm.nW; ++c) {
      const int cb = cm.w0 + c;
      depthW += (cb >= r_lo ? (r_hi - cb + 1) : (r_hi - r_lo + 1)) * nb;
      nfirst += cb >= r_lo;
    }
    const double flops = 2.0 * nb * nb * (depth * (tilesU - halfU) + nr * (depth * cm.Taug + depthW));
    const double bytes = (2.0 * (tilesU + nr * (cm.Taug + cm.nW)) - nr * nfirst) * nb * nb * esz;
    ProfScope ps(cls, s, q * flops, q * bytes);
    // launches with few tiles run on 64-row half tiles: twice the workgroups
    // Split engine: which engine a tile goes through must not depend on the schedule (bit-identical results with and
    // without the look-ahead): every tail / head tile takes the split engine on macro tiles, the next group's triangle
    // (U1; `crit`) and the chain stay on the fp32 MFMAs -- and so do the augmented columns under SplitH2 (aug_fp32)
    const bool use_bf3 = bf3 && (cls == PK_TRAIL || cls == PK_TRAIL_HEAD);
    const unsigned dyn = (cls == PK_TRAIL || (cls == PK_TRAIL_HEAD && !crit)) ? bulk_lds : 0u;
    auto fp32_launch = [&](const ColMap<T> &c) {
      const int cn = c.nU + c.Taug + c.nW;
      if (cn == 0) return;
      const bool half = !use_bf3 && cls != PK_TRAIL && (double)cn * nrows * q <= hthr;
      const dim3 grid(cn, half ? 2 * nrows : nrows, q);
#define PLMC_UPD(ROLE, MT) \
  hipLaunchKernelGGL((k_update<T, ROLE, MT>), grid, dim3(NTHREADS), dyn, s, A, lda, strideA, ib0, r_lo, r_hi, c, skip_ib, skip_jb)
      if (cls == PK_TRAIL_ROW) { if (half && q <= 2 && r_hi == r_lo) PLMC_UPD(4, 2); else if (half) PLMC_UPD(1, 2); else PLMC_UPD(1, 4); }
      else if (cls == PK_TRAIL_HEAD && crit) { if (half) PLMC_UPD(2, 2); else PLMC_UPD(2, 4); }
      else if (cls == PK_TRAIL_HEAD) { if (half) PLMC_UPD(3, 2); else PLMC_UPD(3, 4); }
      else PLMC_UPD(0, 4);
#undef PLMC_UPD
    };
    if constexpr (bf3) {
      if (use_bf3) {
        ColMap<T> cb = cm, ca = cm;
        if (aug_fp32) { cb.Taug = 0; ca.nU = 0; ca.nW = 0; }
        const int cnb = cb.nU + cb.Taug + cb.nW;
        const unsigned short *pl = planes(r_lo);
        const dim3 gridb(cnb, (nrows + 1) / 2, q);
        if (cnb > 0) {
          if (cls == PK_TRAIL)
            hipLaunchKernelGGL((k_update_bf3<SS, 0>), gridb, dim3(B3_NT), 0, s, A, lda, strideA, ib0, nrows, r_lo, r_hi, cb, skip_ib, skip_jb, pl, pl_lat,
                               wcol0, Praw, pl_lat, raw_end, (const float *)scl, sc_lat, Wk, pl_lat);
          else if (crit)
            hipLaunchKernelGGL((k_update_bf3<SS, 2>), gridb, dim3(B3_NT), 0, s, A, lda, strideA, ib0, nrows, r_lo, r_hi, cb, skip_ib, skip_jb, pl, pl_lat,
                               wcol0, Praw, pl_lat, raw_end, (const float *)scl, sc_lat, Wk, pl_lat);
          else
            hipLaunchKernelGGL((k_update_bf3<SS, 3>), gridb, dim3(B3_NT), 0, s, A, lda, strideA, ib0, nrows, r_lo, r_hi, cb, skip_ib, skip_jb, pl, pl_lat,
                               wcol0, Praw, pl_lat, raw_end, (const float *)scl, sc_lat, Wk, pl_lat);
        }
        if (aug_fp32) fp32_launch(ca);
        return;
      }
    }
    fp32_launch(cm);
  };
  // head != 0: the few columns of the next group (panel buffer Ph), on the chain stream
  auto gpanel = [&](int g0, int G, const ColMap<T> &cm, const T *Vg, hipStream_t s, int head) {
    const int nt = cm.nU + cm.Taug + cm.nW;
    if (nt == 0) return;
    const double prods = G * (G + 1) / 2.0;               // 128-deep tile products per column strip
    ProfScope ps(PK_GPANEL, s, q * (double)nt * prods * 2.0 * nb3, q * (double)nt * (prods + G) * nb * nb * esz);
    // fp32 / fp64 engine: out of place into a panel buffer + copy
    auto fp32_panel = [&](const ColMap<T> &c) {
      const int ct = c.nU + c.Taug + c.nW;
      if (ct == 0) return;
      T *Pb = head ? Ph : Pbulk;
      const int64_t ldp = head ? (int64_t)LDG : lda;
      if (head)
        hipLaunchKernelGGL((k_gpanel_rows<T, 1>), dim3(ct, 2 * G, q), dim3(NTHREADS), 0, s, (const T *)A, lda, strideA, g0, G, c, Vg, (int64_t)LDG,
                           strideV, Pb, ldp, strideV);
      else
        hipLaunchKernelGGL((k_gpanel_rows<T, 0>), dim3(ct, (G + 1) / 2, q), dim3(NTHREADS), bulk_lds, s, (const T *)A, lda, strideA, g0, G, c, Vg,
                           (int64_t)LDG, strideV, Pb, ldp, strideV);
      hipLaunchKernelGGL((k_gpanel_copy<T>), dim3(ct, G, q), dim3(NTHREADS), 0, s, A, lda, strideA, g0, c, (const T *)Pb, ldp, strideV);
    };
    if constexpr (bf3) {                                   // split engine, in place, planes from the epilogue (k_gpanel_bf3)
      ColMap<T> cb = cm, ca = cm;
      if (aug_fp32) { cb.Taug = 0; ca.nU = 0; ca.nW = 0; }
      const int nb_ = cb.nU + cb.Taug + cb.nW, nm = (G + 1) / 2;
      // (round 4: one macro row per workgroup, heaviest first, instead of heavy + light pairs for the rest panel too: 17.86-17.89
      // against 17.82-17.84 ms/step at q = 8, level at q = 4 -- the pairing stays)
      if (nb_ > 0)
        hipLaunchKernelGGL((k_gpanel_bf3<SS>), dim3(nb_, head ? nm : (nm + 1) / 2, q), dim3(B3_NT), 0, s, A, lda, strideA, g0, G, cb,
                           (const unsigned short *)VgP2[Vg == Vg2[1]], pl_lat, (const unsigned short *)Praw, pl_lat, planes(g0), pl_lat, wcol0,
                           (const float *)scl, sc_lat, head ? 0 : 1, Wk, pl_lat);
      if (aug_fp32) fp32_panel(ca);
      return;
    }
    fp32_panel(cm);
  };

  // whole-sweep bracket on the caller's stream (the per-kernel records of overlapped kernels add up to
  // more than the wall time once the look-ahead runs the chain beside the trailing update)
  const double npd = (double)n_pad;
  ProfScope whole(PK_SWEEP, st, q * (with_inverse == 2 ? 3.0 : (with_inverse ? 2.0 : 1.0)) * npd * npd * npd / 3.0, 0.0);
  // tiles of the diagonal-block outputs that k_diag leaves alone (they are read as parts of full 128 x 128 operands)
  hipLaunchKernelGGL(k_zero_diag_out<T>, dim3(m > GMAX ? m : GMAX, q), dim3(NTHREADS), 0, st, Vd, strideV, m, Wg, (int64_t)LDG,
                     strideV, (int64_t)NB * LDG + NB, GMAX, (int64_t)GMAX * NB);    // (+ the chain kernel's counters in the pad column of Wg)
  auto scales = [&](hipStream_t s) {                      // scales of the operand families (SplitB3: ones), before anything splits
    if constexpr (bf3) {
      if (SS::NPL == 2)
        hipLaunchKernelGGL(k_scale_scan, dim3(SCAN_PARTS, q), dim3(NTHREADS), 0, s, (const float *)A, n_pad, lda, strideA, (int)naug_pad, scl, sc_lat);
      hipLaunchKernelGGL((k_split_scales<SS>), dim3(q), dim3(64), 0, s, n_pad, eig_lo, scl, sc_lat);
    }
  };
  // A sweep that also assembles (plmc_factorize_ex_*, `job`) with the look-ahead on: only the first group's own block triangle is written
  // in front of the chain, everything else -- and the scan of the diagonal for the scales, which needs it -- rides on the helper stream H
  // beside the first group's chain (the first consumer of either, the transpose + head panel of group 0, waits for them: e_sc).
  // `fused_la` is settled below, once the look-ahead is known to run.
  auto finish = [&](hipStream_t s) {
    hipLaunchKernelGGL(k_logdet<T>, dim3(q), dim3(NTHREADS), 0, s, (const T *)A, n_pad, lda, strideA, logdet, info,
                       (const T *)(Wg + (int64_t)LDG + (int64_t)GMAX * NB), strideV, q);    // (the chain kernel's abort words: one per launch, at its first latent)
    return launch_status("potrf_impl");
  };
  // Group boundaries.  Large groups divide the read-modify-write traffic of the trailing matrix by G and put
  // G (G + 1) / 2 tile products into every group-panel strip; the chain of a group costs ~3 G small launches.
  std::vector<int> gb;                                    // group boundaries: gb[i] .. gb[i+1]
  {
    const int big = grp_rows;
    gb.push_back(0);
    for (int r = 0; r < m;) {
      int g = big;
      if (r + g > m) g = m - r;
      r += g;
      gb.push_back(r);
    }
  }
  const int ng = (int)gb.size() - 1;
  auto G0 = [&](int gi) { return gb[gi < ng ? gi : ng]; };   // first block row of group gi (m beyond the last group)

  // ---- the pieces of one group gi (rows g0 .. g1-1; next group g1 .. g2-1; the one after g2 .. g3-1)
  // chain: the group's diagonal triangle + its inverse triangle in Wg
  // resident form (k_chain): one launch per group, q critical workgroups + a pool for the ~220 other tile operations per latent
  // and group of 8 (12-16 us each).  The pool is sized to keep up with the critical workgroups' ~40 us per block row without
  // holding more CUs than that (a chain workgroup does not fit on a CU beside a 228-register bulk workgroup); PLMC_CHAIN_NW
  // overrides, PLMC_CHAIN=0 brings the launches back.
  constexpr int CHAIN_QB = 32;
  const int chain_pool = kn.chain_nw > 0 ? (kn.chain_nw < 200 ? kn.chain_nw : 200) : (q >= 8 ? 80 : (q >= 4 ? 44 : (q >= 2 ? 36 : 31)));   // (PLMC_CHAIN_NW > 80: dev only)
  auto chain = [&](int gi, hipStream_t s) {
    const int g0 = G0(gi), g1 = G0(gi + 1);
    if (kn.chain) {
      const int G = g1 - g0;
      ProfScope ps(PK_DIAG, s, q * (2.0 / 3.0) * nb3 * G, q * 3.0 * nb * nb * esz * G);
      // (dev knob PLMC_CHAIN_EDGE: another pool size for the first group -- the device to itself -- and the last two -- the drain,
      // where the chain is the critical path.  Measured at q = 8 / 4 with 144 and 200: 18.26-18.43 / 9.93-10.0 ms/step against
      // 18.22 / 9.82 with one size for all groups: the extra workgroups mostly wait on the per-latent dependency fronts.)
      const bool edge = gi == 0 || gi + 2 >= ng;
      const int pool = (edge && kn.chain_edge > 0) ? kn.chain_edge : chain_pool;
      // RESIDENCY: the critical workgroups of a launch must all be resident (the pool's tickets wait for them), and a chain workgroup
      // needs a CU of its own.  A launch therefore carries at most CHAIN_QB latents: CHAIN_QB + pool <= 112 workgroups, so that even
      // the two sweeps that may be in flight at once (two caller streams, api.hip) fit the 256 CUs together; more latents go as
      // further launches behind it on the same stream (their control words: those of the launch's first latent).
      for (int l0 = 0; l0 < q; l0 += CHAIN_QB) {
        const int ql = q - l0 < CHAIN_QB ? q - l0 : CHAIN_QB;
        hipLaunchKernelGGL((k_chain<T>), dim3(ql + (G > 1 ? pool : 0)), dim3(CH_NT), 0, s, A + (int64_t)l0 * strideA, lda, strideA, g0, G,
                           Vd + (int64_t)l0 * strideV, strideV, Wg + (int64_t)l0 * strideV, strideV, ql);
      }
      return;
    }
    for (int r = g0; r < g1; ++r) {
      diag(r, g0, s);
      panel(r, cm_tri(g0, r + 1, g1 - 1 - r, g0, r - g0), s);
      // right-looking inside the triangle: every remaining row of the group gets the rank-128 update of row r at once
      // (depth 128 per launch; a left-looking row update grows to depth 128 (G - 1) on ONE workgroup's critical path:
      // 10 -> 43 us per launch at G = 8).  W columns g0 .. r: column r is touched for the first time (plain store).
      // Two ways of taking this launch off the chain's critical path were built and measured, and dropped:
      //  - the updates on a second helper stream, ordered row by row with events: ~10 us per cross-stream hop on the GPU
      //    and ~17 us per event call on the host (q = 1: sweep 5.3 -> 6.5 ms, step 7.3 -> 11.9 ms);
      //  - one launch carrying this update AND the next diagonal block, whose workgroup applied the last rank-128
      //    update of its block itself (two launches per block row instead of three): the extra round trips of that
      //    prologue beside the bulk kernels made the fused launch as long as the two it replaced (51 us vs 30 + 17 at
      //    q = 1; step 7.35 vs 7.35 ms at q = 1, 11.4 vs 10.9 at q = 2, 37.8 vs 36.8 at q = 8).
      if (r + 1 < g1) update(r + 1, g1 - r - 1, r, r, cm_tri(g0, r + 1, g1 - r - 1, g0, r + 1 - g0), s, PK_TRAIL_ROW);
    }
  };
  auto vtrans = [&](int gi, hipStream_t s) {
    const int g0 = G0(gi), G = G0(gi + 1) - g0;
    T *wo = WA ? WA + (int64_t)g0 * NB * lda + (int64_t)g0 * NB : (T *)nullptr;
    // (split engine: also the planes of Vgg, zero blocks included -- one workgroup per block of the GMAX x GMAX grid)
    hipLaunchKernelGGL((k_vtrans<T, S>), dim3(bf3 ? GMAX * GMAX : G * (G + 1) / 2, q), dim3(NTHREADS), 0, s, (const T *)Wg, (int64_t)LDG, strideV,
                       strideV, Vg2[gi & 1], G, wo, lda, strideA, VgP2[gi & 1], pl_lat, (const float *)scl, sc_lat);
  };
  // split engine: planes of the group's inverse triangle (read back from the W columns k_vtrans wrote; any stream behind it)
  auto wtri_planes = [&](int gi, hipStream_t s) {
    if constexpr (bf3) {
      if (!WA) return;
      const int g0 = G0(gi), G = G0(gi + 1) - g0;
      hipLaunchKernelGGL((k_wtri_planes<SS>), dim3(G * (G + 1) / 2, q), dim3(NTHREADS), 0, s, (const float *)WA, lda, strideA, g0, (const float *)scl, sc_lat, Wk, pl_lat,
                         n_pad);
    }
  };
  // split engine: planes of the first group's raw rows (columns of the bulk panel: everything right of the second group + aug)
  auto raw_planes0 = [&](hipStream_t s, int u0) {
    if constexpr (bf3) {
      const int g1 = G0(1);
      const ColMap<float> cm = cm_buf(u0, m - u0, aug_fp32 ? 0 : Taug, 0, 0);
      const int nt = cm.nU + cm.Taug + cm.nW;
      if (nt > 0)
        hipLaunchKernelGGL((k_raw_planes<SS>), dim3(nt, g1, q), dim3(NTHREADS), 0, s, (const float *)A, lda, strideA, 0, cm, Praw, pl_lat, wcol0,
                           (const float *)scl, sc_lat);
    }
  };

  auto kacc = [&](int gi, hipStream_t s) {
    if (!kacc_on) return;
    const int g0 = G0(gi), g1 = G0(gi + 1);
    const int nt = g1 * (g1 + 1) / 2;
    double fl = 0.0;
    for (int jb = 0; jb < g1; ++jb) fl += (jb + 0.5) * 2.0 * nb * nb * (double)(g1 - (jb >= g0 ? jb : g0)) * nb;   // diagonal tiles: half
    ProfScope ps(PK_KACC, s, q * fl, q * (2.0 * nt - (double)(g1 - g0) * (g0 + g1 + 1) / 2.0) * nb * nb * esz);
    if constexpr (bf3) {
      hipLaunchKernelGGL((k_kacc_bf3<SS>), dim3(g1, (g1 + 1) / 2, q), dim3(B3_NT), 0, s, (float *)A, lda, strideA, (float *)Kd, strideV,
                         (const unsigned short *)Wk, pl_lat, n_pad, g0, g1, (const float *)scl, sc_lat);
    } else {
      hipLaunchKernelGGL((k_kacc<T>), dim3(nt, q), dim3(NTHREADS), bulk_lds, s, A, lda, strideA, (const T *)WA, lda, strideA, Kd, strideV, g0, g1);
    }
  };

  // the helper streams and ordering events of THIS caller stream (api.hip: sweeps from different streams may overlap)
  if (!serial) bind_sweep_ctx(st, 8);
  // PLMC_BULK_STREAMS=1: the group panel of the other columns and the head rows ride on the caller's stream, in front of the tail
  hipStream_t C = serial ? nullptr : side_stream(), H = serial ? nullptr : (kn.bulk_streams == 1 ? st : side_stream(1));
  // the K^-1 accumulation rides on the caller's stream behind the tail (e_tail is recorded before it, so nothing on the
  // critical path waits for it).  A stream of its own shared a hardware queue with one of the others (HIP maps streams
  // onto four hardware queues; with the gradient stream of the Python layer this library already uses four) and
  // serialised the chain behind bulk launches: sweep + accumulation took exactly the sum of the two.
  hipEvent_t e_entry = sync_event(0), e_v = sync_event(1), e_gh = sync_event(2), e_p = sync_event(3), e_hd = sync_event(4),
             e_tail = sync_event(5), e_doneC = sync_event(6), e_doneH = sync_event(7), e_prev = sync_event(8);
  // split engine: the accumulation reads the full-height planes of W (rows of group gi: final behind e_p, never rewritten) and
  // is pure filler, so it gets a (low-priority) stream of its own
  hipStream_t K = (serial || !kacc_on || !bf3) ? nullptr : side_stream(2);
  hipEvent_t e_doneK = sync_event(11);
  if (!e_doneK) K = nullptr;
  hipEvent_t e_sc = sync_event(12);
  const bool la = C && H && e_entry && e_v && e_gh && e_p && e_hd && e_tail && e_doneC && e_doneH && e_prev && ng > 2;
  const bool fused_la = la && job && e_sc;
  if (job && !fused_la) {                                // no overlap to be had: the whole matrix first, as the separate call would
    const int rc = assemble_rows(*job, (int)sizeof(T), A, lda, strideA, q, 0, m, st);
    if (rc != 0) return rc;
  }
  if (!fused_la) scales(st);
  if (!la) {
    // one stream: chain -> transpose -> group panel over every column -> trailing update of every row below
    // split engine: which engine a tile goes through must not depend on the schedule -- the same launches as under the look-ahead
    // (head panel, rest panel, tail, U1), one after the other
    raw_planes0(st, G0(1));
    for (int gi = 0; gi < ng; ++gi) {
      const int g0 = G0(gi), g1 = G0(gi + 1);
      chain(gi, st);
      vtrans(gi, st);
      wtri_planes(gi, st);
      if (bf3) {
        const int g2 = G0(gi + 2);
        gpanel(g0, g1 - g0, cm_buf(g1, g2 - g1, 0, 0, 0), Vg2[gi & 1], st, 1);
        gpanel(g0, g1 - g0, cm_buf(g2, m - g2, Taug, 0, g0), Vg2[gi & 1], st, 0);
      } else {
        gpanel(g0, g1 - g0, cm_buf(g1, m - g1, Taug, 0, g0), Vg2[gi & 1], st, 0);
      }
      if (bf3) {
        const int g2 = G0(gi + 2);
        update(g1, m - g1, g0, g1 - 1, cm_buf(g1, m - g1, Taug, 0, g1), st, PK_TRAIL, g2, g2, false, g2);
        update(g1, g2 - g1, g0, g1 - 1, cm_buf(g1, g2 - g1, 0, 0, 0), st, PK_TRAIL_HEAD, 0, 0, true);
      } else {
        update(g1, m - g1, g0, g1 - 1, cm_buf(g1, m - g1, Taug, 0, g1), st, PK_TRAIL);
      }
      kacc(gi, st);
    }
    return finish(st);
  }
  // Look-ahead on three streams.  With R0 = the rows of group gi, R1 = the next group, R2 = the one after:
  //   C (helper, high priority; latency-bound): chain(gi) -> vtrans(gi) -> [e_v] -> (wait e_hd(gi-1)) gpanel_head(gi):
  //       panel columns R1 -> [e_gh] -> (wait e_tail(gi-1)) U1(gi): update of the next triangle R1 x R1 -> chain(gi+1) ...
  //   H (helper): (wait e_v) gpanel_rest(gi): every other panel column (U right of R1, augmented, inverse-factor columns
  //       left of the group) -> [e_p] -> (wait e_gh, e_tail(gi-1)) head(gi): rows R1, every column but R1 -> [e_hd]
  //   T (the caller's stream; bulk): (wait e_p, e_gh) tail(gi): rows below R1, every column -> [e_tail]
  // chain(gi+1) needs U1(gi) only and runs beside tail(gi); gpanel_rest(gi+1) needs head(gi) (same stream) and runs
  // beside tail(gi) as well, so the bulk stream never waits for a panel.  gpanel_head(gi+1) reads tiles head(gi) wrote
  // (e_hd); U1(gi+1) and head(gi+1) rewrite rows tail(gi) wrote (e_tail).  vtrans(gi+2) reuses the Vg copy
  // gpanel_rest(gi) read: by then C has waited for e_hd(gi), recorded behind it.  U1 / head / tail touch disjoint
  // tiles; every tile receives its updates in the same order as on one stream, so the result is bit-identical to the
  // serial schedule (tests/test_gpu_edges.py).
  if (fused_la) {                                                               // the first group's own triangle: all its chain reads
    const int rc = assemble_rows(*job, (int)sizeof(T), A, lda, strideA, q, 0, G0(1), st, G0(1));
    if (rc != 0) return rc;
  }
  (void)hipEventRecord(e_entry, st);
  (void)hipStreamWaitEvent(C, e_entry, 0);
  (void)hipStreamWaitEvent(H, e_entry, 0);
  if (fused_la) {
    const int rc = assemble_rows(*job, (int)sizeof(T), A, lda, strideA, q, 0, m, H, -1, G0(1));            // everything else, beside chain(0)
    if (rc != 0) return rc;
    scales(H);
    (void)hipEventRecord(e_sc, H);
  }
  raw_planes0(H, G0(1));                                                        // (the head panel of the first group waits for it: e_hd)
  (void)hipEventRecord(e_hd, H);
  for (int gi = 0; gi < ng; ++gi) {
    const int g0 = G0(gi), g1 = G0(gi + 1), g2 = G0(gi + 2), G = g1 - g0;
    const T *Vg = Vg2[gi & 1];
    chain(gi, C);
    if (fused_la && gi == 0) (void)hipStreamWaitEvent(C, e_sc, 0);             // rows below the first group and the scales (planes of V, head panel)
    vtrans(gi, C);
    (void)hipEventRecord(e_v, C);
    if (gi > 0 || bf3) (void)hipStreamWaitEvent(C, e_hd, 0);                   // head(gi - 1): rows R0 final (split engine: and their raw planes)
    gpanel(g0, G, cm_buf(g1, g2 - g1, 0, 0, 0), Vg, C, 1);                      // head columns R1
    (void)hipEventRecord(e_gh, C);
    if (gi > 0) (void)hipStreamWaitEvent(C, e_tail, 0);                        // tail(gi - 1): rows R1 up to date
    update(g1, g2 - g1, g0, g1 - 1, cm_buf(g1, g2 - g1, 0, 0, 0), C, PK_TRAIL_HEAD, 0, 0, true);   // U1: next triangle

    (void)hipStreamWaitEvent(H, e_v, 0);
    wtri_planes(gi, H);
    gpanel(g0, G, cm_buf(g2, m - g2, Taug, 0, g0), Vg, H, 0);                   // rest of the panel columns
    (void)hipEventRecord(e_p, H);
    if (K) {                                                                   // rows R0 of W are final and in planes: filler work
      (void)hipStreamWaitEvent(K, e_p, 0);
      kacc(gi, K);
    }
    (void)hipStreamWaitEvent(H, e_gh, 0);
    if (gi > 0) (void)hipStreamWaitEvent(H, e_tail, 0);
    update(g1, g2 - g1, g0, g1 - 1, cm_buf(g2, m - g2, Taug, 0, g1), H, PK_TRAIL_HEAD, 0, 0, false, g2);   // head: rows R1, columns right of R1
    (void)hipEventRecord(e_hd, H);

    (void)hipStreamWaitEvent(st, e_p, 0);
    (void)hipStreamWaitEvent(st, e_gh, 0);
    update(g2, m - g2, g0, g1 - 1, cm_buf(g2, m - g2, Taug, 0, g1), st, PK_TRAIL);         // tail: rows below R1
    (void)hipEventRecord(e_tail, st);
    if (!K) kacc(gi, st);    // rows R0 of the inverse factor are final (e_p: panel copy; e_gh is behind vtrans): filler work
  }
  if (K) {
    (void)hipEventRecord(e_doneK, K);
    (void)hipStreamWaitEvent(st, e_doneK, 0);
  }
  // log det and the pivot / abort check read the diagonal of U and the chain kernels' control words: final behind the last chain,
  // so they ride on the chain stream beside the last group's panel instead of behind the whole sweep (25 us + a launch gap)
  const int rc_fin = finish(C);
  (void)hipEventRecord(e_doneC, C);
  (void)hipEventRecord(e_doneH, H);
  (void)hipStreamWaitEvent(st, e_doneC, 0);
  (void)hipStreamWaitEvent(st, e_doneH, 0);
  (void)hipEventRecord(e_prev, st);                      // everything of this sweep is behind this point of the caller's stream

  return rc_fin;
}

// Where a sweep of the split engine left the full-height planes of W and the scale of that operand family inside its `Vd`
// scratch (fp32, layout with inverse-factor columns): for kinv_grad_impl (potri_grad.hip).  Strides per latent: 16-bit
// elements / floats.  Returns false when the layout has no such planes.
bool vd_w_planes(const float *Vd, int64_t n_pad, int64_t lda, const unsigned short **wk, int64_t *wk_lat_stride, const float **w_scale,
                 int64_t *w_scale_lat_stride) {
  if (vd_wk_blocks(n_pad, lda, 4) <= 0) return false;
  const int64_t m = n_pad / NB, strideV = plmc_vd_blocks_for(n_pad, lda, 4) * (int64_t)NB * NB;
  const float *Wg = Vd + m * NB * NB;
  const float *Pbulk = Wg + 4 * (int64_t)GMAX * NB * LDG;
  const int64_t pl_buf = b3_elems<SplitB3>((int64_t)GMAX * NB, lda);
  const unsigned short *Pl0 = reinterpret_cast<const unsigned short *>(Pbulk + (int64_t)GMAX * NB * lda);
  const unsigned short *VgP = Pl0 + 3 * pl_buf;
  const float *scl = reinterpret_cast<const float *>(VgP + 2 * b3_elems<SplitB3>((int64_t)GMAX * NB, (int64_t)GMAX * NB));
  *wk = reinterpret_cast<const unsigned short *>(scl + (int64_t)NB * NB);
  *wk_lat_stride = strideV * 2;
  *w_scale = scl + SC_SW;                                        // (the scheme tag sits SC_TAG - SC_SW floats behind it)
  *w_scale_lat_stride = strideV;
  return true;
}

// Forward substitution of NEW augmented columns against a factor buffer that plmc_potrf_* already factorised WITH the inverse
// factor: aug <- U^-T aug for the first naug augmented columns.  What an eval-mode model does on its second and later calls
// (projected_lmc.py:1133-1134: gpytorch's prediction strategy keeps the factorisation): n^2 naug flops instead of a new sweep.
// Per group of block rows: the group's inverse triangle W[R][R] (in the W columns) is transposed into the Vg scratch, the
// group panel solves the rows R of the augmented columns, one depth-(128 G) update takes them out of the rows below.
// MFMA of the element type (no planes of the factor exist any more); one stream.  wcol0 = first inverse-factor column.
template <typename T>
int potrs_aug_at(T *A, int64_t n_pad, int64_t lda, int naug, int64_t wcol0, int64_t strideA, T *Vd, int q, void *stream) {
  PLMC_REQUIRE(A && Vd, "null pointer");
  PLMC_REQUIRE(n_pad > 0 && n_pad % NB == 0 && lda % NB == 0 && wcol0 % NB == 0, "n_pad / lda / wcol0 must be multiples of NB");
  const int64_t naug_pad = plmc_pad(naug);
  PLMC_REQUIRE(naug > 0 && q > 0 && n_pad + naug_pad <= wcol0 && wcol0 + n_pad <= lda, "augmented columns must fit between the square part and the W columns");
  PLMC_REQUIRE(aligned16(A) && aligned16(Vd), "unaligned buffer");
  const hipStream_t st = (hipStream_t)stream;
  const int m = (int)(n_pad / NB), Tu = (int)(naug_pad / NB);
  const int64_t strideV = plmc_vd_blocks_for(n_pad, lda, (int)sizeof(T)) * (int64_t)NB * NB;
  const double nb = (double)NB, nb3 = nb * nb * nb, esz = sizeof(T);
  T *const Wg = Vd + (int64_t)m * NB * NB;
  T *const Vg = Wg + (int64_t)GMAX * NB * LDG;
  T *const Pbulk = Wg + 4 * (int64_t)GMAX * NB * LDG;
  T *const WA = A + wcol0;
  const ColMap<T> cm{0, 0, Tu, 0, 0, n_pad, (T *)nullptr, lda, strideA};
  for (int g0 = 0; g0 < m; g0 += GMAX) {
    const int g1 = g0 + GMAX < m ? g0 + GMAX : m, G = g1 - g0;
    hipLaunchKernelGGL((k_vtrans<T, void>), dim3(G * (G + 1) / 2, q), dim3(NTHREADS), 0, st, (const T *)(WA + (int64_t)g0 * NB * lda + (int64_t)g0 * NB), lda,
                       strideA, strideV, Vg, G, (T *)nullptr, lda, strideA, (unsigned short *)nullptr, (int64_t)0, (const float *)nullptr, (int64_t)0);
    {
      const double prods = G * (G + 1) / 2.0;
      ProfScope ps(PK_GPANEL, st, q * (double)Tu * prods * 2.0 * nb3, q * (double)Tu * (prods + G) * nb * nb * esz);
      hipLaunchKernelGGL((k_gpanel_rows<T, 0>), dim3(Tu, (G + 1) / 2, q), dim3(NTHREADS), 0, st, (const T *)A, lda, strideA, g0, G, cm, (const T *)Vg,
                         (int64_t)LDG, strideV, Pbulk, lda, strideV);
      hipLaunchKernelGGL((k_gpanel_copy<T>), dim3(Tu, G, q), dim3(NTHREADS), 0, st, A, lda, strideA, g0, cm, (const T *)Pbulk, lda, strideV);
    }
    if (g1 < m) {
      const double tiles = (double)Tu * (m - g1);
      ProfScope ps(PK_TRAIL, st, q * tiles * 2.0 * nb * nb * (G * nb), q * 2.0 * tiles * nb * nb * esz);
      hipLaunchKernelGGL((k_update<T, 0, 4>), dim3(Tu, m - g1, q), dim3(NTHREADS), 0, st, A, lda, strideA, g1, g0, g1 - 1, cm, 0, 0);
    }
  }
  return launch_status("plmc_potrs_aug");
}

// The scales of the augmented operand families (SC_SA, SC_RA) for NEW augmented columns of a KEPT factorisation: k_scale_scan has
// just left the largest |augmented entry| of each row slice behind the scales (its diagonal figures are those of U now and are
// ignored); D and lambda of the factorised matrix are still in the scale block (k_split_scales).  grid (q), 64 threads.
template <class S>
__global__ __launch_bounds__(64) void k_aug_scales(int64_t n_pad, float *__restrict__ sc, int64_t sc_stride) {
  if constexpr (S::NPL == 3) return;                                  // SplitB3: all scales are 1
  if (threadIdx.x != 0) return;
  float *o = sc + (int64_t)blockIdx.x * sc_stride;
  float amax = 0.0f;
  for (int p = 0; p < SCAN_PARTS; ++p) amax = fmaxf(amax, o[SC_N + 3 * p + 2]);
  const float D = o[6], lam = o[7];
  const float Rn = sqrtf((float)n_pad) * amax + 1e-30f;
  o[SC_SA] = b3_scale_for(Rn / sqrtf(lam));
  o[SC_RA] = b3_scale_for(Rn * (1.0f + sqrtf(D / lam)));
}

// The same forward substitution on the split engine, against a factorisation that KEPT its planes (plmc_potrf_ex_f32 with
// with_inverse | 4; Vd sized by plmc_vd_blocks_keep): per group the planes of Vgg from the W columns (k_vtrans), the group panel
// of the augmented columns (k_gpanel_bf3: in place + planes), one depth-(128 G) macro-tile update of the rows below whose A
// operand is the kept planes of the group's U rows; its epilogue writes the raw planes of the next group's rows.  One stream.
template <class S>
int potrs_aug_kept(float *A, int64_t n_pad, int64_t lda, int naug, int64_t wcol0, int64_t strideA, float *Vd, int q, void *stream) {
  PLMC_REQUIRE(A && Vd, "null pointer");
  PLMC_REQUIRE(n_pad > 0 && n_pad % NB == 0 && lda % NB == 0 && wcol0 % NB == 0, "n_pad / lda / wcol0 must be multiples of NB");
  const int64_t naug_pad = plmc_pad(naug);
  PLMC_REQUIRE(naug > 0 && q > 0 && n_pad + naug_pad <= wcol0 && wcol0 + n_pad <= lda, "augmented columns must fit between the square part and the W columns");
  PLMC_REQUIRE(aligned16(A) && aligned16(Vd), "unaligned buffer");
  const hipStream_t st = (hipStream_t)stream;
  const int m = (int)(n_pad / NB), Tu = (int)(naug_pad / NB);
  const int64_t strideV = plmc_vd_blocks_keep(n_pad, lda) * (int64_t)NB * NB;
  const double nb = (double)NB, nb3 = nb * nb * nb;
  // the scratch layout of potrf_impl (fp32)
  float *const Wg = Vd + (int64_t)m * NB * NB;
  float *const Vg2[2] = {Wg + (int64_t)GMAX * NB * LDG, Wg + 2 * (int64_t)GMAX * NB * LDG};
  float *const Pbulk = Wg + 4 * (int64_t)GMAX * NB * LDG;
  const int64_t pl_buf = b3_elems<SplitB3>((int64_t)GMAX * NB, lda), vgp_elems = b3_elems<SplitB3>((int64_t)GMAX * NB, (int64_t)GMAX * NB);
  unsigned short *const Pl0 = reinterpret_cast<unsigned short *>(Pbulk + (int64_t)GMAX * NB * lda);
  unsigned short *const Praw = Pl0 + 2 * pl_buf;
  unsigned short *const VgP2[2] = {Praw + pl_buf, Praw + pl_buf + vgp_elems};
  float *const scl = reinterpret_cast<float *>(VgP2[1] + vgp_elems);
  unsigned short *const Wk = reinterpret_cast<unsigned short *>(scl + (int64_t)NB * NB);
  unsigned short *const Uk0 = reinterpret_cast<unsigned short *>(reinterpret_cast<float *>(Wk) + vd_wk_blocks(n_pad, lda, 4) * (int64_t)NB * NB);
  const int64_t pl_lat = strideV * 2, sc_lat = strideV;
  float *const WA = A + wcol0;
  const ColMap<float> cm{0, 0, Tu, 0, 0, n_pad, (float *)nullptr, lda, strideA};
  hipLaunchKernelGGL(k_scale_scan, dim3(SCAN_PARTS, q), dim3(NTHREADS), 0, st, (const float *)A, n_pad, lda, strideA, (int)naug_pad, scl, sc_lat);
  hipLaunchKernelGGL((k_aug_scales<S>), dim3(q), dim3(64), 0, st, n_pad, scl, sc_lat);
  const int G0 = GMAX < m ? GMAX : m;
  hipLaunchKernelGGL((k_raw_planes<S>), dim3(Tu, G0, q), dim3(NTHREADS), 0, st, (const float *)A, lda, strideA, 0, cm, Praw, pl_lat, wcol0, (const float *)scl,
                     sc_lat);
  for (int g0 = 0, gi = 0; g0 < m; g0 += GMAX, ++gi) {
    const int g1 = g0 + GMAX < m ? g0 + GMAX : m, g2 = g1 + GMAX < m ? g1 + GMAX : m, G = g1 - g0, nm = (G + 1) / 2;
    unsigned short *const Pg = Uk0 + (int64_t)gi * pl_buf;
    hipLaunchKernelGGL((k_vtrans<float, S>), dim3(GMAX * GMAX, q), dim3(NTHREADS), 0, st, (const float *)(WA + (int64_t)g0 * NB * lda + (int64_t)g0 * NB), lda,
                       strideA, strideV, Vg2[gi & 1], G, (float *)nullptr, lda, strideA, VgP2[gi & 1], pl_lat, (const float *)scl, sc_lat);
    {
      const double prods = G * (G + 1) / 2.0;
      ProfScope ps(PK_GPANEL, st, q * (double)Tu * prods * 2.0 * nb3, q * (double)Tu * (prods + G) * nb * nb * 4.0);
      hipLaunchKernelGGL((k_gpanel_bf3<S>), dim3(Tu, (nm + 1) / 2, q), dim3(B3_NT), 0, st, A, lda, strideA, g0, G, cm, (const unsigned short *)VgP2[gi & 1], pl_lat,
                         (const unsigned short *)Praw, pl_lat, Pg, pl_lat, wcol0, (const float *)scl, sc_lat, 1, Wk, pl_lat);
    }
    if (g1 < m) {
      const int nrows = m - g1;
      const double tiles = (double)Tu * nrows;
      ProfScope ps(PK_TRAIL, st, q * tiles * 2.0 * nb * nb * (G * nb), q * 2.0 * tiles * nb * nb * 4.0);
      hipLaunchKernelGGL((k_update_bf3<S, 0>), dim3(Tu, (nrows + 1) / 2, q), dim3(B3_NT), 0, st, A, lda, strideA, g1, nrows, g0, g1 - 1, cm, 0, 0,
                         (const unsigned short *)Pg, pl_lat, wcol0, Praw, pl_lat, g2, (const float *)scl, sc_lat, (const unsigned short *)Wk, pl_lat);
    }
  }
  return launch_status("plmc_potrs_aug_kept");
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
  if ((int64_t)q * (n_pad / NB) < 128)                  // a shard of one latent: 32-column workgroups (same sums, four times the workgroups)
    hipLaunchKernelGGL((k_wt_matvec<T, 32>), dim3((unsigned)(n_pad / 32), q), dim3(WTMV_NT / 4), 0, (hipStream_t)stream, W, n_pad, ldw, strideW, z,
                       alpha);
  else
    hipLaunchKernelGGL((k_wt_matvec<T, NB>), dim3((unsigned)(n_pad / NB), q), dim3(WTMV_NT), 0, (hipStream_t)stream, W, n_pad, ldw, strideW, z,
                       alpha);
  return launch_status(__func__);
}

template <typename T>
int w_diag_impl(const T *W, int64_t n_pad, int64_t ldw, int64_t strideW, T *kinv_diag, int q, void *stream) {
  PLMC_REQUIRE(W && kinv_diag, "null pointer");
  PLMC_REQUIRE(n_pad > 0 && n_pad % NB == 0 && q > 0 && aligned16(W), "n_pad must be a multiple of NB");
  ProfScope ps(PK_WTMV, (hipStream_t)stream, q * (double)n_pad * n_pad, q * ((double)n_pad * n_pad / 2) * sizeof(T));
  hipLaunchKernelGGL(k_w_diag<T>, dim3((unsigned)(n_pad / NB), q), dim3(WTMV_NT), 0, (hipStream_t)stream, W, n_pad, ldw, strideW,
                     kinv_diag);
  return launch_status(__func__);
}

}  // namespace plmc

extern "C" {
// Vd blocks per latent: m diagonal inverses + the fixed group scratch + the bulk panel buffer (GMAX block rows of lda) + m
// diagonal K^-1 tiles; for 4-byte elements also the plane buffers of the split engine -- whether or not PLMC_SPLIT is
// on, so that a workspace never depends on a knob.
int64_t plmc_vd_blocks_for(int64_t n_pad, int64_t lda, int elem_bytes) {
  const int64_t ldb = (lda + plmc::NB - 1) / plmc::NB;
  // (version 4) + the full-height planes of W for plmc_kinv_grad_vd_*, when the layout has inverse-factor columns
  // planes (4-byte elements): rolling buffer of the solved panel rows (2 groups) + raw rows of the next group (1 group), each
  // 128 GMAX rows x 3 planes x lda x 2 bytes = 12 ldb blocks at GMAX = 8, + two copies of Vgg (3 x (128 GMAX)^2 x 2 bytes = 96 blocks each)
  // + 1 block holding the scales of the operand families
  const int64_t planes = elem_bytes == 4 ? 3 * (3 * plmc::GMAX * ldb / 2) + 2 * (6 * plmc::GMAX * plmc::GMAX / 4) + 1 : 0;
  return 2 * (n_pad / plmc::NB) + plmc::VD_FIXED_BLOCKS + plmc::GMAX * ldb + planes + plmc::vd_wk_blocks(n_pad, lda, elem_bytes);
}
int64_t plmc_vd_blocks(int64_t n_pad, int64_t lda) { return plmc_vd_blocks_for(n_pad, lda, 4); }
// ... when the sweep keeps the planes of every group's solved rows (with_inverse | 4, fp32): one 128 GMAX-row plane buffer per
// group instead of the two rolling ones
int64_t plmc_vd_blocks_keep(int64_t n_pad, int64_t lda) {
  const int64_t ldb = (lda + plmc::NB - 1) / plmc::NB, m = n_pad / plmc::NB;
  return plmc_vd_blocks_for(n_pad, lda, 4) + ((m + plmc::GMAX - 1) / plmc::GMAX) * (3 * plmc::GMAX * ldb / 2);
}
// PLMC_SPLIT picks the arithmetic of the bulk fp32 products: 0 = fp32 MFMA everywhere, 3 = SplitB3, 2 (default) = SplitH2
// where the caller supplies eigenvalue bounds (plmc_potrf_ex_f32), SplitB3 otherwise.
static int potrf_f32_any(float *A, int64_t n_pad, int64_t lda, int naug, int64_t strideA, float *Vd, double *logdet, int *info,
                         int with_inverse, int q, const float *eig_lo, void *stream, const plmc::AssembleJob *job = nullptr) {
  const int split = plmc::knobs().split;
  if (split == 0) return plmc::potrf_impl<float, void>(A, n_pad, lda, naug, strideA, Vd, logdet, info, with_inverse, q, nullptr, stream, job);
  if (split == 2 && eig_lo)
    return plmc::potrf_impl<float, plmc::SplitH2>(A, n_pad, lda, naug, strideA, Vd, logdet, info, with_inverse, q, eig_lo, stream, job);
  return plmc::potrf_impl<float, plmc::SplitB3>(A, n_pad, lda, naug, strideA, Vd, logdet, info, with_inverse, q, nullptr, stream, job);
}
// plmc_assemble_* + plmc_potrf_ex_* in one call (the augmented columns are written by the caller BEFORE it: plmc_write_rhs_*,
// plmc_assemble_cross_*): the sweep writes the rows of its first group of block rows itself and queues the others beside that
// group's chain.  Same kernels on the same data as the two calls: bit-identical results.
int plmc_factorize_ex_f32(int kind, const float *X, int n, int d, const float *ell, const float *oscale, const float *noise, float *A,
                          int64_t n_pad, int64_t lda, int naug, int64_t strideA, float *Vd, double *logdet, int *info, int with_inverse, int q,
                          const float *eig_lo, void *stream) {
  PLMC_REQUIRE(n_pad == plmc_pad(n), "n_pad must be plmc_pad(n)");
  const plmc::AssembleJob job{kind, n, d, X, ell, oscale, noise};
  return potrf_f32_any(A, n_pad, lda, naug, strideA, Vd, logdet, info, with_inverse, q, eig_lo, stream, &job);
}
int plmc_factorize_ex_f64(int kind, const double *X, int n, int d, const double *ell, const double *oscale, const double *noise, double *A,
                          int64_t n_pad, int64_t lda, int naug, int64_t strideA, double *Vd, double *logdet, int *info, int with_inverse, int q,
                          const double *eig_lo, void *stream) {
  (void)eig_lo;
  PLMC_REQUIRE(n_pad == plmc_pad(n), "n_pad must be plmc_pad(n)");
  const plmc::AssembleJob job{kind, n, d, X, ell, oscale, noise};
  return plmc::potrf_impl<double, void>(A, n_pad, lda, naug, strideA, Vd, logdet, info, with_inverse, q, nullptr, stream, &job);
}
int plmc_potrf_f32(float *A, int64_t n_pad, int64_t lda, int naug, int64_t strideA, float *Vd, double *logdet,
                   int *info, int with_inverse, int q, void *stream) {
  return potrf_f32_any(A, n_pad, lda, naug, strideA, Vd, logdet, info, with_inverse, q, nullptr, stream);
}
int plmc_potrf_f64(double *A, int64_t n_pad, int64_t lda, int naug, int64_t strideA, double *Vd, double *logdet,
                   int *info, int with_inverse, int q, void *stream) {
  return plmc::potrf_impl<double, void>(A, n_pad, lda, naug, strideA, Vd, logdet, info, with_inverse, q, nullptr, stream);
}
int plmc_potrf_ex_f32(float *A, int64_t n_pad, int64_t lda, int naug, int64_t strideA, float *Vd, double *logdet,
                      int *info, int with_inverse, int q, const float *eig_lo, void *stream) {
  return potrf_f32_any(A, n_pad, lda, naug, strideA, Vd, logdet, info, with_inverse, q, eig_lo, stream);
}
int plmc_potrf_ex_f64(double *A, int64_t n_pad, int64_t lda, int naug, int64_t strideA, double *Vd, double *logdet,
                      int *info, int with_inverse, int q, const double *eig_lo, void *stream) {
  (void)eig_lo;
  return plmc::potrf_impl<double, void>(A, n_pad, lda, naug, strideA, Vd, logdet, info, with_inverse, q, nullptr, stream);
}
int plmc_potrs_aug_f32(float *A, int64_t n_pad, int64_t lda, int naug, int64_t wcol0, int64_t strideA, float *Vd, int q, void *stream) {
  return plmc::potrs_aug_at<float>(A, n_pad, lda, naug, wcol0, strideA, Vd, q, stream);
}
int plmc_potrs_aug_f64(double *A, int64_t n_pad, int64_t lda, int naug, int64_t wcol0, int64_t strideA, double *Vd, int q, void *stream) {
  return plmc::potrs_aug_at<double>(A, n_pad, lda, naug, wcol0, strideA, Vd, q, stream);
}
// ... against a factorisation that kept its planes (plmc_potrf_ex_f32 with with_inverse | 4, Vd of plmc_vd_blocks_keep blocks per
// latent): the updates run on the split engine of that factorisation (same PLMC_SPLIT, eig_lo given or not as there)
int plmc_potrs_aug_kept_f32(float *A, int64_t n_pad, int64_t lda, int naug, int64_t wcol0, int64_t strideA, float *Vd, int q, const float *eig_lo,
                            void *stream) {
  const int split = plmc::knobs().split;
  if (split == 0) return plmc::potrs_aug_at<float>(A, n_pad, lda, naug, wcol0, strideA, Vd, q, stream);   // (that sweep ignored the flag)
  if (split == 2 && eig_lo) return plmc::potrs_aug_kept<plmc::SplitH2>(A, n_pad, lda, naug, wcol0, strideA, Vd, q, stream);
  return plmc::potrs_aug_kept<plmc::SplitB3>(A, n_pad, lda, naug, wcol0, strideA, Vd, q, stream);
}
int plmc_potrs_aug_kept_f64(double *A, int64_t n_pad, int64_t lda, int naug, int64_t wcol0, int64_t strideA, double *Vd, int q, const double *eig_lo,
                            void *stream) {
  (void)eig_lo;
  return plmc::potrs_aug_at<double>(A, n_pad, lda, naug, wcol0, strideA, Vd, q, stream);
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
int plmc_w_diag_f32(const float *W, int64_t n_pad, int64_t ldw, int64_t strideW, float *kinv_diag, int q, void *stream) {
  return plmc::w_diag_impl<float>(W, n_pad, ldw, strideW, kinv_diag, q, stream);
}
int plmc_w_diag_f64(const double *W, int64_t n_pad, int64_t ldw, int64_t strideW, double *kinv_diag, int q, void *stream) {
  return plmc::w_diag_impl<double>(W, n_pad, ldw, strideW, kinv_diag, q, stream);
}
}
