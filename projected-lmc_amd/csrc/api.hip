// api.hip -- library identification, error text, and the optional per-kernel HIP-event profiler.
#include <stdlib.h>
#include <vector>

#include "api_common.hpp"
#include "../../include/plmc.h"

namespace plmc {
char *err_buf() {
  static thread_local char buf[256] = {0};
  return buf;
}

// ---- profiler: when enabled, every kernel launch is bracketed by two hipEvents recorded on the
// launch stream; plmc_prof_collect() synchronises them and accumulates time and algorithmic work
// per kernel class.  Off by default (no events, no state).
static const char *kProfNames[PK_COUNT] = {"k_assemble", "k_write_rhs", "k_assemble_cross", "k_diag",   "k_panel",
                                           "k_trail",    "k_wdiag",     "k_trtri_row",      "k_extract_col",
                                           "k_wt_matvec", "k_kinv_grad", "k_reduce_grad", "k_kernel_vjp",
                                           "sweep_total", "k_trail_row", "k_trail_head", "k_gpanel"};
struct ProfRec { int id; hipEvent_t a, b; double flops, bytes; };
static unsigned g_prof_mask = 0;          // bit i: bracket kernel class i
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_free;

static hipEvent_t get_event() {
  if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}

// ---- helper stream + ordering events for the look-ahead of the blocked sweep (one per device,
// created on first use; together with the profiler record this is all the process-global state).
static hipStream_t g_side[64] = {nullptr};
static hipStream_t g_side2[64] = {nullptr};
static hipEvent_t g_sync[64][8] = {{nullptr}};
hipStream_t side_stream(int which) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  if (which == 1) {                    // second helper: the inverse-factor chain of the sweep
    if (!g_side2[dev]) {
      int lo = 0, hi = 0;
      (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
      if (hipStreamCreateWithPriority(&g_side2[dev], hipStreamNonBlocking, hi) != hipSuccess) g_side2[dev] = nullptr;
    }
    return g_side2[dev];
  }
  if (!g_side[dev]) {
    // highest priority: the latency-bound chain must get CU slots ahead of the queued tail tiles
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&g_side[dev], hipStreamNonBlocking, hi) != hipSuccess) g_side[dev] = nullptr;
  }
  return g_side[dev];
}
hipEvent_t sync_event(int idx) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  hipEvent_t &e = g_sync[dev][idx & 7];
  if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
  return e;
}

ProfScope::ProfScope(int id, hipStream_t st, double flops, double bytes) : idx_(-1), st_(st) {
  if (!((g_prof_mask >> id) & 1u) || g_recs.size() >= (1u << 20)) return;
  ProfRec r{id, get_event(), get_event(), flops, bytes};
  if (!r.a || !r.b) return;
  (void)hipEventRecord(r.a, st);
  g_recs.push_back(r);
  idx_ = (long)g_recs.size() - 1;
}
ProfScope::~ProfScope() {
  if (idx_ >= 0) (void)hipEventRecord(g_recs[idx_].b, st_);
}
}  // namespace plmc

extern "C" {
int plmc_version(void) { return 1; }
int plmc_block(void) { return plmc::NB; }
int64_t plmc_pad(int64_t n) { return (n + plmc::NB - 1) / plmc::NB * plmc::NB; }
int plmc_max_dim(void) { return plmc::MAX_DIM; }
const char *plmc_last_error(void) { return plmc::err_buf(); }

int plmc_prof_enable(int on) {
  const unsigned prev = plmc::g_prof_mask, all = (1u << plmc::PK_COUNT) - 1u;
  plmc::g_prof_mask = on == 0 ? 0u : (on == 1 ? all : ((unsigned)on >> 1) & all);
  return prev == 0 ? 0 : (prev == all ? 1 : (int)(prev << 1));
}
int plmc_prof_kernels(void) { return plmc::PK_COUNT; }
const char *plmc_prof_name(int id) { return (id >= 0 && id < plmc::PK_COUNT) ? plmc::kProfNames[id] : ""; }
int plmc_prof_collect(double *ms, int64_t *launches, double *flops, double *bytes) {
  using namespace plmc;
  for (int i = 0; i < PK_COUNT; ++i) { ms[i] = 0; launches[i] = 0; flops[i] = 0; bytes[i] = 0; }
  for (auto &r : g_recs) {
    float t = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
      ms[r.id] += t; launches[r.id] += 1; flops[r.id] += r.flops; bytes[r.id] += r.bytes;
    }
    g_free.push_back(r.a);
    g_free.push_back(r.b);
  }
  g_recs.clear();
  return 0;
}
}
