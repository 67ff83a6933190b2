// api.hip -- library identification, error text, and the optional per-kernel HIP-event profiler.
#include <stdlib.h>
#include <mutex>
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
                                           "sweep_total", "k_trail_row", "k_trail_head", "k_gpanel", "k_kacc", "k_grad_tiles", "k_split_w", "k_posterior_moments"};
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

// One lock for the process-global bookkeeping (profiler record, sweep contexts, knobs): every entry point that launches
// profiled kernels holds it while it ENQUEUES (a ProfScope takes it for its lifetime; recursive: scopes nest), so calls from
// several host threads serialise their enqueueing instead of racing -- the kernels themselves still overlap on the device as the
// streams allow ("documented, not enforced").
static std::recursive_mutex g_api_mutex;
void api_lock() { g_api_mutex.lock(); }
void api_unlock() { g_api_mutex.unlock(); }

static Knobs g_knobs;
static bool g_knobs_loaded = false;
static void load_knobs() {
  const char *h = getenv("PLMC_HALF_TILES"), *g = getenv("PLMC_GRP"), *o = getenv("PLMC_KINV_ORDER");
  g_knobs.half_tiles = h ? (atoi(h) == 1 ? 1e30 : (double)atoi(h)) : 640.0;
  g_knobs.grp = g ? atoi(g) : 0;
  g_knobs.serial = getenv("PLMC_SERIAL") && atoi(getenv("PLMC_SERIAL")) != 0;
  g_knobs.kinv_order = o ? atoi(o) : 4;
  g_knobs.bulk_lds = getenv("PLMC_BULK_LDS") ? atoi(getenv("PLMC_BULK_LDS")) : -1;
  g_knobs.split = getenv("PLMC_SPLIT") ? atoi(getenv("PLMC_SPLIT")) : 2;     // fp32 entry points only
  if (g_knobs.split != 0 && g_knobs.split != 3) g_knobs.split = 2;
  g_knobs.bulk_streams = getenv("PLMC_BULK_STREAMS") ? atoi(getenv("PLMC_BULK_STREAMS")) : 2;
  g_knobs.chain = getenv("PLMC_CHAIN") ? atoi(getenv("PLMC_CHAIN")) : 1;
  g_knobs.chain_nw = getenv("PLMC_CHAIN_NW") ? atoi(getenv("PLMC_CHAIN_NW")) : 0;
  g_knobs.chain_edge = getenv("PLMC_CHAIN_EDGE") ? atoi(getenv("PLMC_CHAIN_EDGE")) : 0;
  g_knobs_loaded = true;
}
const Knobs &knobs() {
  if (!g_knobs_loaded) {
    std::lock_guard<std::recursive_mutex> lk(g_api_mutex);
    if (!g_knobs_loaded) load_knobs();
  }
  return g_knobs;
}

// ---- helper streams + ordering events for the look-ahead of the blocked sweep: SWEEP_CTX sets per device, created on
// first use (together with the profiler record this is all the process-global state).  A sweep uses the set bound to its
// CALLER stream (bind_sweep_ctx): sweeps queued on one stream are ordered by the stream and share a set; sweeps from two
// different streams get a set each and may overlap on the device (what _engine.py does with the two halves of the latents);
// a third stream takes over the least recently used set and first waits for that set's last sweep (its e_prev event).
constexpr int SWEEP_CTX = 2;
struct SweepCtx {
  hipStream_t caller = nullptr;
  bool used = false;
  unsigned long long stamp = 0;
  hipStream_t side[3] = {nullptr, nullptr, nullptr};
  hipEvent_t ev[16] = {nullptr};
};
static SweepCtx g_ctx[64][SWEEP_CTX];
static unsigned long long g_stamp = 0;
static thread_local int t_ctx = 0;
static SweepCtx *cur_ctx() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  return &g_ctx[dev][t_ctx];
}
// which: 0 = the chain (highest priority: its small launches must get CU slots ahead of queued bulk tiles), 1 = the group
// panel + head rows (high), 2 = the K^-1 accumulation (lowest priority: filler work)
hipStream_t side_stream(int which) {
  SweepCtx *c = cur_ctx();
  if (!c || which < 0 || which > 2) return nullptr;
  if (!c->side[which]) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&c->side[which], hipStreamNonBlocking, which == 2 ? lo : hi) != hipSuccess) c->side[which] = nullptr;
  }
  return c->side[which];
}
hipEvent_t sync_event(int idx) {
  SweepCtx *c = cur_ctx();
  if (!c) return nullptr;
  hipEvent_t &e = c->ev[idx & 15];
  if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
  return e;
}
// Select the set of `caller` for the following side_stream / sync_event calls of this thread.  `e_prev_idx`: index of the
// event a sweep records on its caller stream when everything it queued is behind that point.
void bind_sweep_ctx(hipStream_t caller, int e_prev_idx) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { t_ctx = 0; return; }
  SweepCtx *set = g_ctx[dev];
  int pick = -1;
  for (int i = 0; i < SWEEP_CTX; ++i)
    if (set[i].used && set[i].caller == caller) pick = i;
  if (pick < 0) {
    for (int i = 0; i < SWEEP_CTX && pick < 0; ++i)
      if (!set[i].used) pick = i;
    if (pick < 0) {
      pick = 0;
      for (int i = 1; i < SWEEP_CTX; ++i)
        if (set[i].stamp < set[pick].stamp) pick = i;
      if (set[pick].ev[e_prev_idx & 15]) (void)hipStreamWaitEvent(caller, set[pick].ev[e_prev_idx & 15], 0);   // that set's last sweep
    }
    set[pick].caller = caller;
    set[pick].used = true;
  }
  set[pick].stamp = ++g_stamp;
  t_ctx = pick;
}

// Bare MFMA stream (no memory traffic): 16 independent accumulators per wave, 4 waves per SIMD -- the rate the matrix
// cores sustain on THIS device under its current clocks; bench.py reports it beside the nominal peak.
template <typename T>
__global__ __launch_bounds__(NTHREADS) void k_mfma_rate(T *sink, int iters) {
  using Tr = Traits<T>;
  typename Tr::acc_t acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[i][r] = T(0);
  T a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { a[i] = T(1) + T(threadIdx.x & 7) * T(1e-3) + T(i); b[i] = T(0.5) - T(i) * T(1e-3); }
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = Tr::mfma(a[i >> 2], b[i & 3], acc[i]);
  }
  T s = T(0);
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  sink[(size_t)blockIdx.x * NTHREADS + threadIdx.x] = s;
}

// the same for v_mfma_f32_16x16x32_bf16 (the split engine's instruction), fp32 accumulators
typedef __bf16 rate_bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(NTHREADS) void k_mfma_rate_bf16(float *sink, int iters) {
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  rate_bf16x8 a[4], b[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)(1.0f + 0.01f * (float)((threadIdx.x + j) & 7) + (float)i); b[i][j] = (__bf16)(0.5f - 0.01f * (float)(i + j)); }
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i >> 2], b[i & 3], acc[i], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  sink[(size_t)blockIdx.x * NTHREADS + threadIdx.x] = s;
}

template <typename T, bool BF16 = false> static int mfma_rate_impl(void *sink, int64_t sink_bytes, double *tflops) {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail("plmc_prof_mfma_rate", "no device");
  const int grid = 4 * prop.multiProcessorCount, iters = 4000;
  if (!sink || !tflops || sink_bytes < (int64_t)grid * NTHREADS * (int64_t)sizeof(T)) return fail("plmc_prof_mfma_rate", "sink too small");
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return fail("plmc_prof_mfma_rate", "event");
  double best = 0.0;
  for (int rep = 0; rep < 3; ++rep) {                 // first repetition warms the clocks up
    (void)hipEventRecord(e0, nullptr);
    if (BF16) hipLaunchKernelGGL(k_mfma_rate_bf16, dim3(grid), dim3(NTHREADS), 0, nullptr, (float *)sink, iters);
    else hipLaunchKernelGGL(k_mfma_rate<T>, dim3(grid), dim3(NTHREADS), 0, nullptr, (T *)sink, iters);
    (void)hipEventRecord(e1, nullptr);
    float ms = 0.f;
    if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess) break;
    const double fl = (double)grid * (NTHREADS / 64) * (double)iters * 16.0 * 2.0 * 16 * 16 * (BF16 ? 32 : 4);
    if (ms > 0.f && fl / (ms * 1e-3) / 1e12 > best) best = fl / (ms * 1e-3) / 1e12;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *tflops = best;
  return launch_status("plmc_prof_mfma_rate");
}

ProfScope::ProfScope(int id, hipStream_t st, double flops, double bytes) : idx_(-1), st_(st) {
  api_lock();
  if (!((g_prof_mask >> id) & 1u) || g_recs.size() >= (1u << 20)) return;
  ProfRec r{id, get_event(), get_event(), flops, bytes};
  if (!r.a || !r.b) return;
  (void)hipEventRecord(r.a, st);
  g_recs.push_back(r);
  idx_ = (long)g_recs.size() - 1;
}
ProfScope::~ProfScope() {
  if (idx_ >= 0) (void)hipEventRecord(g_recs[idx_].b, st_);
  api_unlock();
}
}  // namespace plmc

extern "C" {
int plmc_version(void) { return 4; }   // 4: Vd carries the full-height planes of W (plmc_kinv_grad_vd_*, plmc_grad_partials_bytes); 3: Vd / gradient scratch sizes independent of the knobs (+ plmc_*_for), k8-ordered bf16 planes
int plmc_block(void) { return plmc::NB; }
int64_t plmc_pad(int64_t n) { return (n + plmc::NB - 1) / plmc::NB * plmc::NB; }
int plmc_max_dim(void) { return plmc::MAX_DIM; }
const char *plmc_last_error(void) { return plmc::err_buf(); }

int plmc_prof_enable(int on) {
  std::lock_guard<std::recursive_mutex> lk(plmc::g_api_mutex);
  const unsigned prev = plmc::g_prof_mask, all = (1u << plmc::PK_COUNT) - 1u;
  plmc::g_prof_mask = on == 0 ? 0u : (on == 1 ? all : ((unsigned)on >> 1) & all);
  return prev == 0 ? 0 : (prev == all ? 1 : (int)(prev << 1));
}
int plmc_prof_mfma_rate(int kind, void *sink, int64_t sink_bytes, double *tflops) {      // kind: 0 f32, 1 f64, 2 bf16
  if (kind == 2) return plmc::mfma_rate_impl<float, true>(sink, sink_bytes, tflops);
  return kind ? plmc::mfma_rate_impl<double>(sink, sink_bytes, tflops) : plmc::mfma_rate_impl<float>(sink, sink_bytes, tflops);
}
int plmc_dev_reload_knobs(void) {
  std::lock_guard<std::recursive_mutex> lk(plmc::g_api_mutex);
  plmc::load_knobs();
  return 0;
}
int plmc_prof_kernels(void) { return plmc::PK_COUNT; }
const char *plmc_prof_name(int id) { return (id >= 0 && id < plmc::PK_COUNT) ? plmc::kProfNames[id] : ""; }
int plmc_prof_collect(double *ms, int64_t *launches, double *flops, double *bytes) {
  using namespace plmc;
  std::lock_guard<std::recursive_mutex> lk(g_api_mutex);
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
