// engine_rate_probe.hip -- where does a macro tile of the split engine (bf3_engine.hpp) spend its time?
//
//   hipcc -O3 --offload-arch=gfx950 -o tools/bin/engine_rate_probe tools/engine_rate_probe.hip && tools/bin/engine_rate_probe
//
// The SHIPPED main loop (b3_mainloop<SplitH2>) on grids that are whole multiples of the 256 CUs (no quantisation), at depths
// K = 256 .. 8192 and with three epilogues (none: one store per lane; WB_STORE; WB_SUB = the read-modify-write of the tail
// update): time per macro tile = T0 (launch + prologue + epilogue) + (K / 32) t_stage.  A straight-line fit over K gives
// t_stage (the steady-state rate of the loop) and T0 (what one workgroup per CU cannot overlap with anything).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <random>
#include <vector>
#include "../projected-lmc_amd/csrc/bf3_engine.hpp"

using namespace plmc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class S> __global__ __launch_bounds__(NTHREADS) void k_fill(unsigned short *P, int64_t ld, int K, unsigned seed) {
  // random planes: split of uniform values in (-1, 1) times 2^10 (fp16 range); one thread per (k8 group, column)
  const int64_t col = (int64_t)blockIdx.x * NTHREADS + threadIdx.x;
  const int k8 = blockIdx.y;
  if (col >= ld) return;
  unsigned h = seed ^ (unsigned)(col * 2654435761u) ^ (unsigned)(k8 * 40503u);
  b3_s16x8 pl[S::NPL];
  for (int r = 0; r < 8; ++r) {
    h = h * 1664525u + 1013904223u;
    const float v = ((float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f) * (S::NPL == 2 ? 1024.0f : 1.0f);
    short o[S::NPL];
    S::split(v, o);
    for (int p = 0; p < S::NPL; ++p) pl[p][r] = o[p];
  }
  for (int p = 0; p < S::NPL; ++p) *reinterpret_cast<b3_s16x8 *>(P + b3_index<S>(k8 * 8, p, col, ld)) = pl[p];
}

// EPI 0: one store per lane (keeps the loop alive); 1: WB_STORE; 2: WB_SUB (C read-modify-write)
template <class S, int EPI>
__global__ __launch_bounds__(B3_NT, 2) void k_rate(const unsigned short *__restrict__ P, float *C, int64_t ld, int K, int mrows, int tcols) {
  __shared__ __align__(16) unsigned char lds[b3_lds_bytes<S>()];
  const int t = blockIdx.x, mb = t % mrows, jb = (t / mrows) % tcols;
  Acc<float> acc0, acc1;
  acc0.zero();
  acc1.zero();
  b3_mainloop<S>(acc0, acc1, P + (int64_t)mb * 256 * 8, ld, P + (int64_t)(2 * mrows + jb) * NB * 8, ld, K, lds);
  b3_combine<S>(acc0, acc1, 1.0f / 1048576.0f);
  const int tid = threadIdx.x & 255, half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  float *Cg = C + ((int64_t)mb * 256 + half * 128) * ld + (int64_t)jb * NB;
  if (EPI == 0) {
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) s += acc0.v[a][b][0] + acc0.v[a][b][1] + acc0.v[a][b][2] + acc0.v[a][b][3];
    Cg[(int64_t)(tid >> 5) * ld + (tid & 31) * 4] = s;
  } else {
    tile_writeback<float, EPI == 1 ? WB_STORE : WB_SUB>(acc0, Cg, ld, reinterpret_cast<float *>(lds + half * B3_WB_BYTES), tid);
  }
}


// ---- the tail update's real shape: per latent a 1024-row panel of `ld` columns (planes), 24 macro rows x (48 triangular
// + 17 full) tile columns, C read-modify-write.  ORDER 0: the shipped grid (column fastest, then macro row, then latent);
// ORDER 1 / 2: one-dimensional grid per latent, workgroup t -> XCD t % 8 gets whole blocks of BR x BC macro tiles (its 32
// resident workgroups share BR A strips and BC B strips), blocks dealt round-robin to the XCDs along the rows.
template <class S, int NSTG, int ORDER, int BR, int BC, int EPI = 2>
__global__ __launch_bounds__(B3_NT, 2) void k_tail(const unsigned short *__restrict__ P, int64_t pstride, float *C, int64_t cstride, int64_t ld, int K,
                                                    int mrows, int nU, int nF, int tri = 1) {
  __shared__ __align__(16) unsigned char lds[NSTG * b3_stage_bytes<S>()];
  const int tcols = nU + nF;
  int mb, jb;
  if (ORDER == 0) { jb = blockIdx.x; mb = blockIdx.y; }
  else {
    const int t = blockIdx.x, xcd = t & 7, slot = t >> 3;
    const int nbx = (tcols + BC - 1) / BC, nby = (mrows + BR - 1) / BR;
    const int b = xcd + 8 * (slot / (BR * BC)), in = slot % (BR * BC);
    if (b >= nbx * nby) return;
    mb = (b / nbx) * BR + in / BC;
    jb = (b % nbx) * BC + in % BC;
    if (mb >= mrows || jb >= tcols) return;
  }
  const int lat = blockIdx.z;
  bool v0 = true, v1 = true;
  if (tri && jb < nU) { v0 = jb >= 2 * mb; v1 = jb >= 2 * mb + 1; }
  if (!v0 && !v1) return;
  const unsigned short *Pl = P + (int64_t)lat * pstride;
  Acc<float> acc0, acc1;
  acc0.zero();
  acc1.zero();
  b3_mainloop<S, NSTG>(acc0, acc1, Pl + (int64_t)mb * 256 * 8, ld, Pl + (int64_t)jb * NB * 8, ld, K, lds);
  b3_combine<S>(acc0, acc1, 1.0f / 1048576.0f);
  const int tid = threadIdx.x & 255, half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
  float *Cg = C + (int64_t)lat * cstride + ((int64_t)mb * 256 + half * 128) * ld + (int64_t)jb * NB;
  if (EPI == 0) {
    float sm = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) sm += acc0.v[a][b][0] + acc0.v[a][b][1] + acc0.v[a][b][2] + acc0.v[a][b][3];
    Cg[(int64_t)(tid >> 5) * ld + (tid & 31) * 4] = sm;
  } else if (EPI == 1) {
    tile_writeback<float, WB_STORE>(acc0, Cg, ld, reinterpret_cast<float *>(lds + half * B3_WB_BYTES), tid, half ? v1 : v0);
  } else {
    tile_writeback<float, WB_SUB>(acc0, Cg, ld, reinterpret_cast<float *>(lds + half * B3_WB_BYTES), tid, half ? v1 : v0);
  }
}

template <class S, int NSTG, int ORDER, int BR, int BC, int EPI = 2>
int run_tail_one(const char *name, const unsigned short *P, int64_t pstride, float *C, int64_t cstride, int64_t ld, int K, int mrows, int nU, int nF, int q) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int tcols = nU + nF;
  dim3 grid;
  if (ORDER == 0) grid = dim3(tcols, mrows, q);
  else {
    const int nb = ((tcols + BC - 1) / BC) * ((mrows + BR - 1) / BR);
    grid = dim3(8 * ((nb + 7) / 8) * BR * BC, 1, q);
  }
  auto launch = [&]() { hipLaunchKernelGGL((k_tail<S, NSTG, ORDER, BR, BC, EPI>), grid, dim3(B3_NT), 0, 0, P, pstride, C, cstride, ld, K, mrows, nU, nF); };
  for (int w = 0; w < 2; ++w) launch();
  float ms = 0.f;
  const int reps = 5;
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  double tiles = 0;                                           // 128 x 128 tiles with work
  for (int mb = 0; mb < mrows; ++mb)
    for (int jb = 0; jb < tcols; ++jb) tiles += jb < nU ? (jb >= 2 * mb) + (jb >= 2 * mb + 1) : 2;
  const double us = 1e3 * ms / reps;
  printf("  %-58s %9.1f us per launch  %7.1f TF fp32-equivalent\n", name, us, 2.0 * 128 * 128 * (double)K * tiles * q / (us * 1e-6) / 1e12);
  return 0;
}

// What separates the loop's 457 TF on a small, L2-resident working set from ~370 TF on the tail's shape?  The same kernel (no
// epilogue, shipped order) over shapes between the two: rows x columns, latents, with / without the triangle.
int run_shapes() {
  typedef SplitH2 S;
  const int K = 1024;
  struct Shape { int mr, nu, nf, q, tri; const char *what; };
  const Shape shapes[] = {{16, 32, 0, 1, 0, "16 x 32, 1 latent, all tiles (the K-sweep probe's shape)"}, {16, 32, 0, 8, 0, "16 x 32, 8 latents"},
                          {24, 48, 17, 1, 0, "24 x 65, 1 latent, all tiles"}, {24, 48, 17, 8, 0, "24 x 65, 8 latents, all tiles"},
                          {24, 48, 17, 8, 1, "24 x 65, 8 latents, triangle (the tail's shape)"}, {8, 48, 17, 8, 0, "8 x 65, 8 latents"},
                          {24, 16, 0, 8, 0, "24 x 16, 8 latents"}, {24, 130, 0, 2, 0, "24 x 130, 2 latents"}};
  for (const Shape &sh : shapes) {
    const int64_t ld = (int64_t)(sh.nu + sh.nf > 2 * sh.mr ? sh.nu + sh.nf : 2 * sh.mr) * NB + 128;
    const int64_t pstride = b3_elems<S>(K, ld), cstride = (int64_t)sh.mr * 256 * ld;
    unsigned short *P = nullptr;
    float *C = nullptr;
    CK(hipMalloc(&P, (size_t)pstride * sh.q * 2));
    CK(hipMalloc(&C, (size_t)cstride * sh.q * 4));
    for (int l = 0; l < sh.q; ++l)
      hipLaunchKernelGGL((k_fill<S>), dim3((unsigned)((ld + NTHREADS - 1) / NTHREADS), K / 8), dim3(NTHREADS), 0, 0, P + (int64_t)l * pstride, ld, K, 99u + l);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const dim3 grid(sh.nu + sh.nf, sh.mr, sh.q);
    auto launch = [&]() { hipLaunchKernelGGL((k_tail<S, 2, 0, 1, 1, 0>), grid, dim3(B3_NT), 0, 0, P, pstride, C, cstride, ld, K, sh.mr, sh.nu, sh.nf, sh.tri); };
    for (int w = 0; w < 3; ++w) launch();
    float ms = 0.f;
    CK(hipEventRecord(e0));
    for (int r = 0; r < 8; ++r) launch();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    double tiles = 0;
    for (int mb = 0; mb < sh.mr; ++mb)
      for (int jb = 0; jb < sh.nu + sh.nf; ++jb) tiles += (sh.tri && jb < sh.nu) ? (jb >= 2 * mb) + (jb >= 2 * mb + 1) : 2;
    const double us = 1e3 * ms / 8;
    printf("  %-62s planes %4.0f MB  %8.1f us  %6.1f TF\n", sh.what, pstride * 2.0 * sh.q / 1e6, us, 2.0 * 128 * 128 * (double)K * tiles * sh.q / (us * 1e-6) / 1e12);
    CK(hipFree(P));
    CK(hipFree(C));
  }
  return 0;
}

int run_tail() {
  typedef SplitH2 S;
  const int K = 1024, MR = 24, NU = 48, NF = 17, Q = 8;
  const int64_t ld = (int64_t)(NU + NF) * NB + 128;
  const int64_t pstride = b3_elems<S>(K, ld), cstride = (int64_t)MR * 256 * ld;
  unsigned short *P = nullptr;
  float *C = nullptr;
  CK(hipMalloc(&P, (size_t)pstride * Q * 2));
  CK(hipMalloc(&C, (size_t)cstride * Q * 4));
  CK(hipMemset(C, 0, (size_t)cstride * Q * 4));
  for (int l = 0; l < Q; ++l)
    hipLaunchKernelGGL((k_fill<S>), dim3((unsigned)((ld + NTHREADS - 1) / NTHREADS), K / 8), dim3(NTHREADS), 0, 0, P + (int64_t)l * pstride, ld, K, 777u + l);
  CK(hipDeviceSynchronize());
  printf("tail shape: %d latents x (%d macro rows x (%d triangular + %d full) tile columns), depth %d, planes %.0f MB + C %.0f MB per latent\n", Q, MR, NU, NF,
         K, pstride * 2 / 1e6, cstride * 4 / 1e6);
  for (int rep = 0; rep < 2; ++rep) {
    if (run_tail_one<S, 2, 0, 1, 1, 0>("shipped order, 2 stages, NO epilogue", P, pstride, C, cstride, ld, K, MR, NU, NF, Q)) return 1;
    if (run_tail_one<S, 2, 0, 1, 1, 1>("shipped order, 2 stages, store-only epilogue", P, pstride, C, cstride, ld, K, MR, NU, NF, Q)) return 1;
    if (run_tail_one<S, 3, 1, 4, 8, 0>("XCD blocks 4 x 8, 3 stages, NO epilogue", P, pstride, C, cstride, ld, K, MR, NU, NF, Q)) return 1;
    if (run_tail_one<S, 2, 0, 1, 1>("shipped order, 2 stages", P, pstride, C, cstride, ld, K, MR, NU, NF, Q)) return 1;
    if (run_tail_one<S, 3, 0, 1, 1>("shipped order, 3 stages", P, pstride, C, cstride, ld, K, MR, NU, NF, Q)) return 1;
    if (run_tail_one<S, 2, 1, 4, 8>("XCD blocks 4 x 8, 2 stages", P, pstride, C, cstride, ld, K, MR, NU, NF, Q)) return 1;
    if (run_tail_one<S, 3, 1, 4, 8>("XCD blocks 4 x 8, 3 stages", P, pstride, C, cstride, ld, K, MR, NU, NF, Q)) return 1;
    if (run_tail_one<S, 2, 1, 2, 16>("XCD blocks 2 x 16, 2 stages", P, pstride, C, cstride, ld, K, MR, NU, NF, Q)) return 1;
    if (run_tail_one<S, 3, 1, 2, 16>("XCD blocks 2 x 16, 3 stages", P, pstride, C, cstride, ld, K, MR, NU, NF, Q)) return 1;
    if (run_tail_one<S, 2, 1, 8, 4>("XCD blocks 8 x 4, 2 stages", P, pstride, C, cstride, ld, K, MR, NU, NF, Q)) return 1;
    if (run_tail_one<S, 3, 1, 8, 4>("XCD blocks 8 x 4, 3 stages", P, pstride, C, cstride, ld, K, MR, NU, NF, Q)) return 1;
  }
  CK(hipFree(P));
  CK(hipFree(C));
  return 0;
}

template <class S> int run(const char *name, int products) {
  const int MR = 16, TC = 32;                                // 16 macro rows x 32 tile columns = 512 workgroups = 2 per CU
  const int KMAX = 8192;
  const int64_t ld = (int64_t)(2 * MR + TC) * NB + 128;
  unsigned short *P = nullptr;
  float *C = nullptr;
  CK(hipMalloc(&P, (size_t)b3_elems<S>(KMAX, ld) * 2));
  CK(hipMalloc(&C, (size_t)MR * 256 * ld * 4));
  CK(hipMemset(C, 0, (size_t)MR * 256 * ld * 4));
  hipLaunchKernelGGL((k_fill<S>), dim3((unsigned)((ld + NTHREADS - 1) / NTHREADS), KMAX / 8), dim3(NTHREADS), 0, 0, P, ld, KMAX, 12345u);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("%s: %d plane products; LDS %d bytes per workgroup\n", name, products, b3_lds_bytes<S>());
  for (int rounds = 1; rounds <= 4; rounds *= 2) {
    const int grid = 256 * rounds;
    for (int epi = 0; epi < 3; ++epi) {
      double xs[8], ys[8];
      int np = 0;
      for (int K = 256; K <= KMAX; K *= 2) {
        auto launch = [&]() {
          if (epi == 0) hipLaunchKernelGGL((k_rate<S, 0>), dim3(grid), dim3(B3_NT), 0, 0, P, C, ld, K, MR, TC);
          else if (epi == 1) hipLaunchKernelGGL((k_rate<S, 1>), dim3(grid), dim3(B3_NT), 0, 0, P, C, ld, K, MR, TC);
          else hipLaunchKernelGGL((k_rate<S, 2>), dim3(grid), dim3(B3_NT), 0, 0, P, C, ld, K, MR, TC);
        };
        for (int w = 0; w < 3; ++w) launch();
        float ms = 0.f;
        const int reps = 10;
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / reps;
        const double tf = 2.0 * 256 * 128 * (double)K * grid / (us * 1e-6) / 1e12;
        printf("  grid %4d  epilogue %d  K %5d  %9.2f us per launch  %7.1f TF fp32-equivalent\n", grid, epi, K, us, tf);
        xs[np] = K / 32.0; ys[np] = us / rounds; ++np;
      }
      // least squares over the points with K >= 1024 (np - 2 .. ): us per round = T0 + stages * t
      double sx = 0, sy = 0, sxx = 0, sxy = 0; int m = 0;
      for (int i = 2; i < np; ++i) { sx += xs[i]; sy += ys[i]; sxx += xs[i] * xs[i]; sxy += xs[i] * ys[i]; ++m; }
      const double t = (m * sxy - sx * sy) / (m * sxx - sx * sx), T0 = (sy - t * sx) / m;
      printf("  => grid %4d epilogue %d: T0 = %.2f us per round of tiles, t_stage = %.4f us (= %.0f TF steady state); depth 1024 spends %.0f %% in T0\n",
             grid, epi, T0, t, 2.0 * 256 * 128 * 32 * 256 / (t * 1e-6) / 1e12, 100.0 * T0 / (T0 + 32 * t));
    }
  }
  CK(hipFree(P));
  CK(hipFree(C));
  return 0;
}

int main(int argc, char **argv) {
  if (argc > 1 && argv[1][0] == 's') return run_shapes();
  if (run_tail()) return 1;
  if (argc < 2) return 0;                                    // any argument: also the K sweep of the bare loop
  if (run<SplitH2>("SplitH2 (two fp16 planes)", 3)) return 1;
  if (run<SplitB3>("SplitB3 (three bf16 planes)", 6)) return 1;
  return 0;
}
