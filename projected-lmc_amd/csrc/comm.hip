// comm.hip -- the one exchange of the path, for a host that does not bring torch.distributed: a direct RCCL
// all-reduce (sum) of the fused [loss share | parameter gradients] buffer, O(q d + p^2) numbers once per step, and of the
// (2, n*, p) partial prediction sums.  RCCL is opened at run time (dlopen: the library has no link-time dependency on it and
// single-GPU users never load it); one communicator per process, on the device that was current at plmc_comm_init.
// The host's own launcher carries the 128-byte unique id from rank 0 to the others (an environment variable, a file, MPI,
// or -- projectedlmc/parallel.py -- one torch.distributed broadcast used as bootstrap only).
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "api_common.hpp"

namespace {
struct UniqueId { char internal[128]; };                    // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void *Comm;                                         // ncclComm_t
typedef int (*GetUniqueIdFn)(UniqueId *);
typedef int (*CommInitRankFn)(Comm *, int, UniqueId, int);
typedef int (*AllReduceFn)(const void *, void *, size_t, int, int, Comm, hipStream_t);
typedef int (*CommDestroyFn)(Comm);
typedef const char *(*GetErrorStringFn)(int);
constexpr int kFloat32 = 7, kFloat64 = 8, kSum = 0;         // ncclFloat32, ncclFloat64, ncclSum (rccl.h)

void *g_lib = nullptr;
GetUniqueIdFn p_unique = nullptr;
CommInitRankFn p_init = nullptr;
AllReduceFn p_allreduce = nullptr;
CommDestroyFn p_destroy = nullptr;
GetErrorStringFn p_errstr = nullptr;
Comm g_comm = nullptr;
int g_world = 0, g_rank = -1;

bool load_rccl() {
  if (g_lib) return true;
  for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    g_lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (g_lib) break;
  }
  if (!g_lib) return false;
  p_unique = (GetUniqueIdFn)dlsym(g_lib, "ncclGetUniqueId");
  p_init = (CommInitRankFn)dlsym(g_lib, "ncclCommInitRank");
  p_allreduce = (AllReduceFn)dlsym(g_lib, "ncclAllReduce");
  p_destroy = (CommDestroyFn)dlsym(g_lib, "ncclCommDestroy");
  p_errstr = (GetErrorStringFn)dlsym(g_lib, "ncclGetErrorString");
  if (!(p_unique && p_init && p_allreduce && p_destroy)) { dlclose(g_lib); g_lib = nullptr; return false; }
  return true;
}
int rccl_status(int rc, const char *what) {
  if (rc == 0) return 0;
  snprintf(plmc::err_buf(), 256, "%s: %s", what, p_errstr ? p_errstr(rc) : "RCCL error");
  return -4;
}
int allreduce(void *buf, int64_t count, int dtype, void *stream) {
  PLMC_REQUIRE(g_comm, "plmc_comm_init has not been called");
  PLMC_REQUIRE(buf && count >= 0, "bad buffer");
  if (count == 0) return 0;
  return rccl_status(p_allreduce(buf, buf, (size_t)count, dtype, kSum, g_comm, (hipStream_t)stream), "ncclAllReduce");
}
}  // namespace

extern "C" {
int plmc_comm_unique_id(void *id128) {
  PLMC_REQUIRE(id128, "null pointer");
  PLMC_REQUIRE(load_rccl(), "librccl.so could not be opened");
  UniqueId id;
  const int rc = rccl_status(p_unique(&id), "ncclGetUniqueId");
  if (rc == 0) memcpy(id128, &id, sizeof(id));
  return rc;
}
int plmc_comm_init(const void *id128, int rank, int world) {
  PLMC_REQUIRE(id128 && world >= 1 && rank >= 0 && rank < world, "bad rank / world size");
  PLMC_REQUIRE(!g_comm, "a communicator already exists (plmc_comm_destroy first)");
  PLMC_REQUIRE(load_rccl(), "librccl.so could not be opened");
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  const int rc = rccl_status(p_init(&g_comm, world, id, rank), "ncclCommInitRank");
  if (rc != 0) { g_comm = nullptr; return rc; }
  g_world = world;
  g_rank = rank;
  return 0;
}
int plmc_comm_world(void) { return g_comm ? g_world : 0; }
int plmc_comm_rank(void) { return g_comm ? g_rank : -1; }
int plmc_comm_allreduce_sum_f32(float *buf, int64_t count, void *stream) { return allreduce(buf, count, kFloat32, stream); }
int plmc_comm_allreduce_sum_f64(double *buf, int64_t count, void *stream) { return allreduce(buf, count, kFloat64, stream); }
int plmc_comm_destroy(void) {
  if (!g_comm) return 0;
  const int rc = rccl_status(p_destroy(g_comm), "ncclCommDestroy");
  g_comm = nullptr;
  g_world = 0;
  g_rank = -1;
  return rc;
}
}
