// lmc.hip -- the exact (dense) LMC / ICM path: Kronecker-structured coregionalisation
//     K_full = sum_i os_i K_i(X,X) (x) B_i + I_n (x) Sigma          ((n p) x (n p), data-major interleaved:
//                                                                     flat index = i_point * p + i_task)
// Replaces `MultitaskGPModel.forward` with gpytorch's LCMKernel / MultitaskKernel
// (projected_lmc.py:462-466, 586-589) + MultitaskGaussianLikelihood (experiments.py:184) and the
// autograd backward through them.  The (np x np) matrix goes through the same
// blocked sweep (potrf.hip); this file adds
//   k_lmc_assemble  : X, {ell_i, B_i}, Sigma -> upper tiles of K_full        (HBM-write bound)
//   k_lmc_cross     : K_full(X, X*) into augmented columns (prediction)
//   k_lmc_kinv_grad : K_full^-1 = W^T W tile on MFMA with the gradient w.r.t. every ell_ik, os_i,
//                     B_i[s][t] and Sigma[s][t] reduced in the epilogue (LDS accumulators, fp64),
//                     per-tile partials summed in fixed order by k_lmc_reduce.
#include "api_common.hpp"
#include "covariance.hpp"
#include "../../include/plmc.h"

namespace plmc {

// Per-tile staging in LDS for the element-wise LMC evaluation: data point and task of each of the
// 128 rows / columns of the tile, plus the small parameter tables.
template <typename T> struct LmcTables {
  T *xi, *xj;          // [128][d+1] raw inputs of the row / column data points (0 beyond N)
  T *invl;             // [q][d]  1 / ell
  T *os;               // [q]
  T *B;                // [q][p][p]
  T *Sg;               // [p][p]
};

template <typename T>
__device__ __forceinline__ T *lmc_stage(T *base, LmcTables<T> &t, const T *X, int n, int d, int p, int q,
                                         const T *ell, const T *oscale, const T *B, const T *Sigma, int row0, int col0) {
  const int ldu = d + 1;
  t.xi = base; t.xj = t.xi + NB * ldu; t.invl = t.xj + NB * ldu; t.os = t.invl + q * d;
  t.B = t.os + q; t.Sg = t.B + q * p * p;
  T *end = t.Sg + p * p;
  const int tid = threadIdx.x;
  for (int e = tid; e < NB * d; e += NTHREADS) {
    int r = e / d, k = e % d;
    int a = (row0 + r) / p, b = (col0 + r) / p;
    t.xi[r * ldu + k] = a < n ? X[(int64_t)a * d + k] : T(0);
    t.xj[r * ldu + k] = b < n ? X[(int64_t)b * d + k] : T(0);
  }
  for (int e = tid; e < q * d; e += NTHREADS) t.invl[e] = T(1) / ell[e];
  for (int e = tid; e < q; e += NTHREADS) t.os[e] = oscale ? oscale[e] : T(1);
  for (int e = tid; e < q * p * p; e += NTHREADS) t.B[e] = B[e];
  if (Sigma) for (int e = tid; e < p * p; e += NTHREADS) t.Sg[e] = Sigma[e];
  return end;
}

// grid (m, m): upper tiles of the N_pad x N_pad matrix (N = n p).
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_lmc_assemble(int kind, const T *__restrict__ X, int n, int d, int p, int q,
                                                            const T *__restrict__ ell, const T *__restrict__ oscale,
                                                            const T *__restrict__ B, const T *__restrict__ Sigma,
                                                            T *__restrict__ A, int64_t lda) {
  const int jb = blockIdx.x, ib = blockIdx.y;
  if (jb < ib) return;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  LmcTables<T> t;
  lmc_stage<T>(reinterpret_cast<T *>(smem_raw), t, X, n, d, p, q, ell, oscale, B, Sigma, ib * NB, jb * NB);
  __syncthreads();
  const int64_t N = (int64_t)n * p;
  const int ldu = d + 1, tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  for (int rr = 0; rr < 8; ++rr) {
    const int r = ty + 16 * rr;
    const int64_t I = (int64_t)ib * NB + r;
    const int a = (int)(I / p), s = (int)(I % p);
    for (int h = 0; h < 2; ++h) {
      const int c0 = tx * 4 + 64 * h;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int64_t J = (int64_t)jb * NB + c0 + c;
        T val;
        if (I < N && J < N) {
          const int b = (int)(J / p), tt = (int)(J % p);
          val = (a == b) ? t.Sg[s * p + tt] : T(0);
          for (int i = 0; i < q; ++i) {
            T r2 = T(0);
            for (int k = 0; k < d; ++k) {
              T df = (t.xi[r * ldu + k] - t.xj[(c0 + c) * ldu + k]) * t.invl[i * d + k];
              r2 += df * df;
            }
            val += t.os[i] * kern_value<T>(kind, r2) * t.B[(i * p + s) * p + tt];
          }
        } else {
          val = (I == J) ? T(1) : T(0);
        }
        A[I * lda + J] = val;
      }
    }
  }
}

// Out[(a,s)][col0 + b*p + t] = sum_i os_i k_i(x_a, xs_b) B_i[s][t]; rows >= N zero.  block 64 x 4.
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_lmc_cross(int kind, const T *__restrict__ X, int n, const T *__restrict__ Xs,
                                                         int ns, int d, int p, int q, const T *__restrict__ ell,
                                                         const T *__restrict__ oscale, const T *__restrict__ B,
                                                         T *__restrict__ Out, int64_t ldo, int64_t col0, int64_t n_rows) {
  const int64_t J = (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
  const int64_t I = (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
  if (J >= (int64_t)ns * p || I >= n_rows) return;
  T val = T(0);
  if (I < (int64_t)n * p) {
    const int a = (int)(I / p), s = (int)(I % p), b = (int)(J / p), tt = (int)(J % p);
    for (int i = 0; i < q; ++i) {
      T r2 = T(0);
      for (int k = 0; k < d; ++k) {
        T df = (X[(int64_t)a * d + k] - Xs[(int64_t)b * d + k]) / ell[i * d + k];
        r2 += df * df;
      }
      val += (oscale ? oscale[i] : T(1)) * kern_value<T>(kind, r2) * B[(i * p + s) * p + tt];
    }
  }
  Out[I * ldo + col0 + J] = val;
}

// Number of fp64 accumulators per tile: [q*p*p dB | q*d d ell | q d os | p*p dSigma].
__host__ __device__ inline int lmc_nacc(int p, int q, int d) { return q * p * p + q * d + q + p * p; }

// The epilogue walks the tile once per latent: the lengthscale / outputscale sums of that latent (the same d + 1
// addresses for every element) stay in per-lane registers and are flushed to the LDS accumulators once per latent;
// with every lane adding to the same LDS words per element the fp64 LDS atomics serialised 64-fold.  The
// (task, task) scatter of dB and dSigma stays on LDS atomics (different lanes hit different words).
// The walk is three real loops (mt, nt, r): the 16 accumulator registers of one sub-tile row mt are parked in LDS
// (each thread reads back only what it wrote, no barrier) so that (nt, r) are runtime indices, and the (data point,
// task) of every row / column comes from LDS tables instead of integer divisions per element.  Round 1 unrolled
// mt x nt around a select chain and spilled 148 / 288 / 492 bytes per lane in fp64 (the C2 kernel); this form needs
// no scratch (profiles/r02_resource_usage.md).
constexpr int LMC_PARK_ELEMS = 4 * NTHREADS * 4;      // parked accumulator slice, T elements
constexpr int LMC_TABLE_INTS = 4 * NB;                // rowA, rowS, colB, colT
template <typename T, int DCAP>
__global__ __launch_bounds__(NTHREADS, TILE_MIN_WAVES<T>) void k_lmc_kinv_grad(int kind, const T *__restrict__ W, int64_t N_pad, int64_t ldw,
                                                             const T *__restrict__ alpha, const T *__restrict__ X,
                                                             int n, int d, int p, int q, const T *__restrict__ ell,
                                                             const T *__restrict__ oscale, const T *__restrict__ B,
                                                             double *__restrict__ partials) {
  // XCD-dealt 8 x 8 super-tiles (gemm_core.hpp, the order of k_kinv_grad): the tiles of a super-tile stream their strips of W through
  // one L2 (round 4; a plain (jb, ib) grid before)
  const int m = (int)(N_pad / NB);
  int lat_, ib, jb;
  if (!xcd_tri_decode(blockIdx.x, m, 1, lat_, ib, jb)) return;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);
  const T *Wl = W + (int64_t)jb * NB * ldw;
  Acc<T> acc;
  acc.zero();
  tile_mainloop<T, false, true>(acc, Wl + (int64_t)ib * NB, ldw, Wl + (int64_t)jb * NB, ldw,
                                (int)(N_pad - (int64_t)jb * NB), smem);
  // ---- epilogue
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
  LmcTables<T> t;
  T *end = lmc_stage<T>(smem, t, X, n, d, p, q, ell, oscale, B, (const T *)nullptr, ib * NB, jb * NB);
  T *ai = end, *aj = ai + NB;
  const int nacc = lmc_nacc(p, q, d);
  uintptr_t ap = (reinterpret_cast<uintptr_t>(aj + NB) + 7) & ~(uintptr_t)7;
  double *accs = reinterpret_cast<double *>(ap);
  double *gB = accs, *gL = gB + q * p * p, *gO = gL + q * d, *gS = gO + q;
  T *park = reinterpret_cast<T *>(accs + nacc) + tid * 4;             // [4 nt][256 threads][4 r]
  int *rowA = reinterpret_cast<int *>(reinterpret_cast<T *>(accs + nacc) + LMC_PARK_ELEMS);
  int *rowS = rowA + NB, *colB = rowS + NB, *colT = colB + NB;
  const int64_t N = (int64_t)n * p;
  if (tid < NB) {
    ai[tid] = alpha[ib * NB + tid];
    aj[tid] = alpha[jb * NB + tid];
    const int64_t I = (int64_t)ib * NB + tid, J = (int64_t)jb * NB + tid;
    rowA[tid] = I < N ? (int)(I / p) : -1;                            // data point (-1: padding) and task of the row
    rowS[tid] = (int)(I % p);
    colB[tid] = J < N ? (int)(J / p) : -1;
    colT[tid] = (int)(J % p);
  }
  for (int e = tid; e < nacc; e += NTHREADS) accs[e] = 0.0;
  __syncthreads();
  const int ldu = d + 1;
  const bool diag_tile = jb == ib;
  // fp64 with more than 8 input dimensions: the lengthscale sums are taken GH = 8 dimensions per walk of the tile (the
  // 128 accumulator registers leave no room for 16 or 32 more doubles at 2 waves per SIMD); the scatter terms and the
  // outputscale sum are added in the first walk only.
  constexpr int GH = (sizeof(T) == 8 && DCAP > 8) ? 8 : DCAP;
#pragma unroll 1
  for (int i = 0; i < q; ++i) {
    const T os_i = t.os[i];
    const T *invl = t.invl + i * d;
    const T *Bi = t.B + i * p * p;
#pragma unroll 1
    for (int k0 = 0; k0 < DCAP && (k0 == 0 || k0 < d); k0 += GH) {
    const bool first = k0 == 0;
    T gl[GH], go = T(0);
#pragma unroll
    for (int k = 0; k < GH; ++k) gl[k] = T(0);
#pragma unroll 1
    for (int mt = 0; mt < 4; ++mt) {
#define PLMC_PARK(M)                                                                                    \
  _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) _Pragma("unroll") for (int r = 0; r < 4; ++r)         \
      park[nt * NTHREADS * 4 + r] = acc.v[M][nt][r];
      if (mt == 0) { PLMC_PARK(0) } else if (mt == 1) { PLMC_PARK(1) } else if (mt == 2) { PLMC_PARK(2) } else { PLMC_PARK(3) }
#undef PLMC_PARK
#pragma unroll 1
      for (int nt = 0; nt < 4; ++nt) {
        const int col = tile_col(wn, nt, lane);
        const int b = colB[col], tt = colT[col];
        const T a_j = aj[col];
        const T *xjc = t.xj + col * ldu;
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
          const int row = tile_row<T>(wm, mt, lane, r);
          const int a = rowA[row], s = rowS[row];
          if (a >= 0 && b >= 0 && (!diag_tile || col >= row)) {
            const T kin = park[nt * NTHREADS * 4 + r];
            const T wij = ((diag_tile && col == row) ? T(1) : T(2)) * (ai[row] * a_j - kin);
            if (i == 0 && first && a == b) atomicAdd(&gS[s * p + tt], (double)wij);
            const T *xir = t.xi + row * ldu;
            T r2 = T(0);
            if constexpr (GH == DCAP) {
#pragma unroll
              for (int k = 0; k < DCAP; ++k) {
                const T df = k < d ? (xir[k] - xjc[k]) * invl[k] : T(0);
                r2 += df * df;
              }
            } else {
#pragma unroll 4
              for (int k = 0; k < d; ++k) { const T df = (xir[k] - xjc[k]) * invl[k]; r2 += df * df; }
            }
            T val, base;
            kern_value_base<T>(kind, r2, val, base);
            const T bst = Bi[s * p + tt];
            if (first) {
              atomicAdd(&gB[(i * p + s) * p + tt], (double)(wij * os_i * val));
              go += wij * val * bst;
            }
            const T c = a != b ? wij * os_i * bst * base : T(0);
            // the differences are formed a second time instead of keeping DCAP squares alive across the kernel value
#pragma unroll
            for (int k = 0; k < GH; ++k) {
              const T df = k0 + k < d ? (xir[k0 + k] - xjc[k0 + k]) * invl[k0 + k] : T(0);
              gl[k] += c * (df * df);
            }
          }
        }
      }
    }
    if (first) atomicAdd(&gO[i], (double)go);
#pragma unroll
    for (int k = 0; k < GH; ++k)
      if (k0 + k < d) atomicAdd(&gL[i * d + k0 + k], (double)gl[k]);
    }
  }
  __syncthreads();
  double *out = partials + ((int64_t)ib * m + jb) * nacc;
  for (int e = tid; e < nacc; e += NTHREADS) out[e] = accs[e];
}

// grad[e] = 1/2 sum over upper tiles (fixed order); lengthscale entries additionally / ell.
// grid (nacc), 256 threads striding over the m^2 tile slots, fixed-order tree reduction.
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_lmc_reduce(const double *__restrict__ partials, int m, int nacc, int p, int q,
                                                          int d, const T *__restrict__ ell, double *__restrict__ grad) {
  __shared__ double red[NTHREADS];
  const int e = blockIdx.x;
  double s = 0.0;
  for (int t = threadIdx.x; t < m * m; t += NTHREADS) {
    const int ib = t / m, jb = t - ib * m;
    if (jb >= ib) s += partials[(int64_t)t * nacc + e];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = NTHREADS / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double v = 0.5 * red[0];
    const int l0 = q * p * p;
    if (e >= l0 && e < l0 + q * d) v /= (double)ell[e - l0];
    grad[e] = v;
  }
}

template <typename T> size_t lmc_stage_elems(int p, int q, int d) {
  return (size_t)2 * NB * (d + 1) + (size_t)q * d + q + (size_t)q * p * p + (size_t)p * p;
}

template <typename T>
int lmc_assemble_impl(int kind, const T *X, int n, int d, int p, int q, const T *ell, const T *oscale, const T *B,
                      const T *Sigma, T *A, int64_t lda, void *stream) {
  PLMC_REQUIRE(kind >= 0 && kind <= 3, "unknown kernel kind");
  PLMC_REQUIRE(X && ell && B && Sigma && A, "null pointer");
  PLMC_REQUIRE(n > 0 && p > 0 && q > 0 && d > 0 && d <= MAX_DIM, "bad sizes");
  const int64_t N_pad = plmc_pad((int64_t)n * p);
  PLMC_REQUIRE(lda >= N_pad && lda % NB == 0, "lda must be a multiple of NB and >= plmc_pad(n*p)");
  const size_t smem = lmc_stage_elems<T>(p, q, d) * sizeof(T);
  PLMC_REQUIRE(smem <= 150 * 1024, "q*p*p too large for the LDS parameter tables");
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lmc_assemble<T>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  const int m = (int)(N_pad / NB);
  ProfScope ps(PK_ASSEMBLE, (hipStream_t)stream, 0.0, ((double)N_pad * N_pad / 2) * sizeof(T));
  hipLaunchKernelGGL(k_lmc_assemble<T>, dim3(m, m), dim3(NTHREADS), smem, (hipStream_t)stream, kind, X, n, d, p, q, ell,
                     oscale, B, Sigma, A, lda);
  return launch_status(__func__);
}

template <typename T>
int lmc_cross_impl(int kind, const T *X, int n, const T *Xs, int ns, int d, int p, int q, const T *ell, const T *oscale,
                   const T *B, T *Out, int64_t ldo, int64_t col0, int64_t n_rows, void *stream) {
  PLMC_REQUIRE(kind >= 0 && kind <= 3, "unknown kernel kind");
  PLMC_REQUIRE(X && Xs && ell && B && Out, "null pointer");
  PLMC_REQUIRE(n > 0 && ns > 0 && p > 0 && q > 0 && d > 0 && d <= MAX_DIM, "bad sizes");
  PLMC_REQUIRE(n_rows >= (int64_t)n * p && col0 >= 0 && col0 + (int64_t)ns * p <= ldo, "cross block exceeds the buffer");
  const int64_t nc = (int64_t)ns * p;
  ProfScope ps(PK_CROSS, (hipStream_t)stream, 0.0, (double)n_rows * nc * sizeof(T));
  hipLaunchKernelGGL(k_lmc_cross<T>, dim3((unsigned)((nc + 63) / 64), (unsigned)((n_rows + 3) / 4)), dim3(NTHREADS), 0,
                     (hipStream_t)stream, kind, X, n, Xs, ns, d, p, q, ell, oscale, B, Out, ldo, col0, n_rows);
  return launch_status(__func__);
}

template <typename T>
int lmc_kinv_grad_impl(int kind, const T *W, int64_t N_pad, int64_t ldw, const T *alpha, const T *X, int n, int d, int p,
                       int q, const T *ell, const T *oscale, const T *B, double *grad, void *partials, void *stream) {
  PLMC_REQUIRE(kind >= 0 && kind <= 3, "unknown kernel kind");
  PLMC_REQUIRE(W && alpha && X && ell && B && grad && partials, "null pointer");
  PLMC_REQUIRE(n > 0 && p > 0 && q > 0 && d > 0 && d <= MAX_DIM, "bad sizes");
  PLMC_REQUIRE(N_pad == plmc_pad((int64_t)n * p) && ldw % NB == 0 && aligned16(W), "N_pad must be plmc_pad(n*p)");
  const int nacc = lmc_nacc(p, q, d);
  size_t epi = (lmc_stage_elems<T>(p, q, d) + 2 * NB) * sizeof(T) + 8 + (size_t)nacc * sizeof(double) +
               (size_t)LMC_PARK_ELEMS * sizeof(T) + (size_t)LMC_TABLE_INTS * sizeof(int);
  size_t smem = (size_t)tile_smem_elems<T>() * sizeof(T);
  if (epi > smem) smem = epi;
  PLMC_REQUIRE(smem <= 150 * 1024, "q*p*p too large for the LDS accumulators");
  hipStream_t st = (hipStream_t)stream;
  const int m = (int)(N_pad / NB);
  double *part = reinterpret_cast<double *>(partials);
  {
    const double np = (double)N_pad;
    ProfScope ps(PK_KINV_GRAD, st, np * np * np / 3.0, (np * np / 2) * sizeof(T));
#define PLMC_LAUNCH_LKG(DC)                                                                                              \
  do {                                                                                                                   \
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_lmc_kinv_grad<T, DC>),                                    \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);                                     \
    hipLaunchKernelGGL((k_lmc_kinv_grad<T, DC>), dim3(xcd_tri_grid(m, 1)), dim3(NTHREADS), smem, st, kind, W, N_pad, ldw, alpha, X, n,  \
                       d, p, q, ell, oscale, B, part);                                                                    \
  } while (0)
    if (d <= 8) PLMC_LAUNCH_LKG(8);
    else if (d <= 16) PLMC_LAUNCH_LKG(16);
    else PLMC_LAUNCH_LKG(32);
#undef PLMC_LAUNCH_LKG
  }
  {
    ProfScope ps(PK_REDUCE, st, 0.0, (double)m * m / 2 * nacc * 8);
    hipLaunchKernelGGL(k_lmc_reduce<T>, dim3(nacc), dim3(NTHREADS), 0, st, part, m, nacc, p, q, d, ell, grad);
  }
  return launch_status(__func__);
}

}  // namespace plmc

extern "C" {
int64_t plmc_lmc_grad_len(int p, int q, int d) { return plmc::lmc_nacc(p, q, d); }
int64_t plmc_lmc_grad_scratch_bytes(int64_t N_pad, int p, int q, int d) {
  int64_t m = N_pad / plmc::NB;
  return m * m * (int64_t)plmc::lmc_nacc(p, q, d) * (int64_t)sizeof(double);
}
int plmc_lmc_assemble_f32(int kind, const float *X, int n, int d, int p, int q, const float *ell, const float *oscale,
                          const float *B, const float *Sigma, float *A, int64_t lda, void *stream) {
  return plmc::lmc_assemble_impl<float>(kind, X, n, d, p, q, ell, oscale, B, Sigma, A, lda, stream);
}
int plmc_lmc_assemble_f64(int kind, const double *X, int n, int d, int p, int q, const double *ell, const double *oscale,
                          const double *B, const double *Sigma, double *A, int64_t lda, void *stream) {
  return plmc::lmc_assemble_impl<double>(kind, X, n, d, p, q, ell, oscale, B, Sigma, A, lda, stream);
}
int plmc_lmc_cross_f32(int kind, const float *X, int n, const float *Xs, int ns, int d, int p, int q, const float *ell,
                       const float *oscale, const float *B, float *Out, int64_t ldo, int64_t col0, int64_t n_rows,
                       void *stream) {
  return plmc::lmc_cross_impl<float>(kind, X, n, Xs, ns, d, p, q, ell, oscale, B, Out, ldo, col0, n_rows, stream);
}
int plmc_lmc_cross_f64(int kind, const double *X, int n, const double *Xs, int ns, int d, int p, int q, const double *ell,
                       const double *oscale, const double *B, double *Out, int64_t ldo, int64_t col0, int64_t n_rows,
                       void *stream) {
  return plmc::lmc_cross_impl<double>(kind, X, n, Xs, ns, d, p, q, ell, oscale, B, Out, ldo, col0, n_rows, stream);
}
int plmc_lmc_kinv_grad_f32(int kind, const float *W, int64_t N_pad, int64_t ldw, const float *alpha, const float *X, int n,
                           int d, int p, int q, const float *ell, const float *oscale, const float *B, double *grad,
                           void *partials, void *stream) {
  return plmc::lmc_kinv_grad_impl<float>(kind, W, N_pad, ldw, alpha, X, n, d, p, q, ell, oscale, B, grad, partials,
                                         stream);
}
int plmc_lmc_kinv_grad_f64(int kind, const double *W, int64_t N_pad, int64_t ldw, const double *alpha, const double *X,
                           int n, int d, int p, int q, const double *ell, const double *oscale, const double *B,
                           double *grad, void *partials, void *stream) {
  return plmc::lmc_kinv_grad_impl<double>(kind, W, N_pad, ldw, alpha, X, n, d, p, q, ell, oscale, B, grad, partials,
                                          stream);
}
}
