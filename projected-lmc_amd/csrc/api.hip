// api.hip -- library identification and error text.
#include "api_common.hpp"
#include "../../include/plmc.h"

namespace plmc {
char *err_buf() {
  static thread_local char buf[256] = {0};
  return buf;
}
}  // namespace plmc

extern "C" {
int plmc_version(void) { return 1; }
int plmc_block(void) { return plmc::NB; }
int64_t plmc_pad(int64_t n) { return (n + plmc::NB - 1) / plmc::NB * plmc::NB; }
int plmc_max_dim(void) { return plmc::MAX_DIM; }
const char *plmc_last_error(void) { return plmc::err_buf(); }
}
