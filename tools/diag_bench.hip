// diag_bench.hip -- dev microbenchmark: where do the cycles of k_diag go (phase-skip variants)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include "../projected-lmc_amd/csrc/diag_block.hpp"
using namespace plmc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k_empty(int *p) { if (p == nullptr) p[0] = 1; }
template <int DBG> void run(const char *name, float *A, float *Vd, double *ld, int *info, int q) {
  size_t sm = 0;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_diag<float, DBG>), dim3(q), dim3(DIAG_NT), sm, 0, A, (int64_t)128, (int64_t)128 * 128, 0, Vd, (int64_t)128 * 128, (float *)nullptr, (int64_t)0, (int64_t)0);
  CK(hipEventRecord(e0));
  const int R = 20;
  for (int r = 0; r < R; ++r) hipLaunchKernelGGL((k_diag<float, DBG>), dim3(q), dim3(DIAG_NT), sm, 0, A, (int64_t)128, (int64_t)128 * 128, 0, Vd, (int64_t)128 * 128, (float *)nullptr, (int64_t)0, (int64_t)0);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-34s %7.1f us per launch\n", name, 1e3 * ms / R);
}
int main() {
  const int q = 8;
  std::vector<float> h(q * 128 * 128);
  for (int l = 0; l < q; ++l) for (int i = 0; i < 128; ++i) for (int j = 0; j < 128; ++j)
    h[(l * 128 + i) * 128 + j] = (i == j ? 130.f : 0.f) + 0.5f * ((i * 37 + j * 11) % 17) / 17.f;   // SPD-ish (diag dominant), identity-like after 1st pass
  float *A, *Vd; double *ld; int *info;
  CK(hipMalloc(&A, h.size() * 4)); CK(hipMalloc(&Vd, h.size() * 4)); CK(hipMalloc(&ld, q * 8)); CK(hipMalloc(&info, q * 4 + 8192)); CK(hipMemset(info, 0, q * 4 + 8192));
  CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  run<0>("full", A, Vd, ld, info, q);
  run<1>("no factor16", A, Vd, ld, info, q);
  run<2>("no panel", A, Vd, ld, info, q);
  run<4>("no trailing", A, Vd, ld, info, q);
  run<7>("load/store + barriers only", A, Vd, ld, info, q);
  {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_empty, dim3(q), dim3(512), 0, 0, info);
    CK(hipEventRecord(e0));
    for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(k_empty, dim3(q), dim3(512), 0, 0, info);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("empty kernel back to back          %7.1f us per launch\n", 1e3 * ms / 20);
  }
  return 0;
}
