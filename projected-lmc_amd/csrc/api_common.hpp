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

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace plmc
