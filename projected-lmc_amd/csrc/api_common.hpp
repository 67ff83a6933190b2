// api_common.hpp -- argument checking and error reporting shared by the extern "C" entry points.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "gemm_core.hpp"

namespace plmc {

constexpr int MAX_DIM = 32;          // largest input dimension handled by the fused kernels

char *err_buf();                     // thread-local, defined in api.hip

inline int fail(const char *fn, const char *msg) {
  snprintf(err_buf(), 256, "%s: %s", fn, msg);
  return -1;
}

inline int launch_status(const char *fn) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(err_buf(), 256, "%s: HIP launch failed: %s", fn, hipGetErrorString(e));
    return -2;
  }
  return 0;
}

#define PLMC_REQUIRE(cond, msg) \
  do { if (!(cond)) return plmc::fail(__func__, msg); } while (0)

// kernel classes known to the optional profiler (api.hip)
enum ProfKernel { PK_ASSEMBLE, PK_WRITE_RHS, PK_CROSS, PK_DIAG, PK_PANEL, PK_TRAIL, PK_WDIAG, PK_TRTRI, PK_EXTRACT,
                  PK_WTMV, PK_KINV_GRAD, PK_REDUCE, PK_VJP, PK_SWEEP, PK_TRAIL_ROW, PK_TRAIL_HEAD, PK_GPANEL, PK_KACC, PK_GRAD_TILES, PK_SPLIT, PK_POST, PK_COUNT };

// Brackets the launches made while it is alive with two hipEvents (no-op unless its class is enabled, plmc_prof_enable).
// flops / bytes = ALGORITHMIC work of the bracketed launch (DESIGN.md gives the formulas).
struct ProfScope {
  ProfScope(int id, hipStream_t st, double flops, double bytes);
  ~ProfScope();
  long idx_;
  hipStream_t st_;
};

// Dev knobs (environment variables), read ONCE per process on first use; plmc_dev_reload_knobs() re-reads them (the
// tests and bench.py change a knob and reload).  They change schedules; PLMC_GRP also changes the depth of the updates (and
// with it the rounding), PLMC_SPLIT selects the arithmetic of the bulk fp32 products.
struct Knobs {
  double half_tiles;   // PLMC_HALF_TILES: 0 never, 1 always, N > 1 = tile-count threshold (default 640)
  int grp;             // PLMC_GRP: fixed group size of the sweep (default 0 = 8)
  bool serial;         // PLMC_SERIAL: one stream, no look-ahead
  int bulk_lds;        // PLMC_BULK_LDS: extra dynamic LDS bytes per bulk workgroup (caps bulk occupancy); -1 = default by q
  int split;           // PLMC_SPLIT: arithmetic of the bulk fp32 products (tail / head updates, group panel, K^-1): 0 = v_mfma_f32_16x16x4_f32
                       // everywhere; 3 = three bf16 planes, six products (bf3_engine.hpp SplitB3); 2 (default) = two fp16 planes, three
                       // products (SplitH2) where the caller gives eigenvalue bounds (plmc_*_ex_f32), SplitB3 otherwise
  int bulk_streams;    // PLMC_BULK_STREAMS: 2 = group panel + head rows on their own helper stream beside the tail, 1 = in front of the tail on the caller's stream
  int chain;           // PLMC_CHAIN: 1 (default) = the chain of a group as one resident launch (k_chain), 0 = three launches per block row
  int chain_edge;      // PLMC_CHAIN_EDGE: pool size for the first and the last two groups of a sweep (<= 0: as the others)
  int chain_nw;        // PLMC_CHAIN_NW: pool workgroups of the resident chain beside the q critical ones (0 = by the number of latents)
  int kinv_order;      // PLMC_KINV_ORDER: tile order of the gradient kernel (0 XCD-dealt, 1 grid, 4 longest first, 5 = 4 + general epilogue)
};
const Knobs &knobs();

// Covariance assembly handed to the sweep (plmc_factorize_ex_*): the sweep queues the rows of its first group on the caller's stream and
// the rest on a helper stream beside the first group's chain, instead of the caller assembling the whole matrix in front of the sweep
struct AssembleJob {
  int kind, n, d;
  const void *X, *ell, *oscale, *noise;
};
// block rows ib0 .. ib0 + nrows - 1 of the covariance matrices (assemble.hip), the first ncols block columns (< 0: all) without the
// leading skip x skip block triangle; elem_bytes 4 / 8
int assemble_rows(const AssembleJob &job, int elem_bytes, void *A, int64_t lda, int64_t strideA, int q, int ib0, int nrows, void *stream,
                  int ncols = -1, int skip = 0);

hipStream_t side_stream(int which = 0);   // per-device helper streams (api.hip), which in {0, 1, 2}; nullptr on failure
hipEvent_t sync_event(int idx);     // per-device ordering events, idx in [0,16)
void bind_sweep_ctx(hipStream_t caller, int e_prev_idx);   // select the stream / event set of this caller stream (api.hip)

// NB x NB blocks (4-byte elements) of the full-height W planes inside Vd: 3 planes x 2 bytes x n_pad^2 (the three-plane
// scheme's size, whatever the scheme), only for 4-byte elements and a layout with inverse-factor columns (lda >= 2 n_pad)
inline int64_t vd_wk_blocks(int64_t n_pad, int64_t lda, int elem_bytes) {
  const int64_t m = n_pad / 128;
  return (elem_bytes == 4 && lda >= 2 * n_pad) ? (3 * m * m + 1) / 2 : 0;
}
constexpr int VD_W_TAG = 8 + 3 * 32 - 1;   // floats from the W-family scale (vd_w_planes: *w_scale) to the scheme tag (planes per element) of that scratch
bool vd_w_planes(const float *Vd, int64_t n_pad, int64_t lda, const unsigned short **wk, int64_t *wk_lat_stride, const float **w_scale,
                 int64_t *w_scale_lat_stride);                              // potrf.hip

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace plmc
