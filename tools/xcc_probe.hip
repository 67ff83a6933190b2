// xcc_probe.hip -- dev probe: which XCD does workgroup w land on (HW_REG_XCC_ID), alone and behind another kernel?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k(int *out, int spin) {
  const int x = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
  if (threadIdx.x == 0) out[blockIdx.x] = x;
  long long t0 = clock64();
  while (clock64() - t0 < spin) {}
}
int main() {
  const int n = 4096;
  int *d; CK(hipMalloc(&d, n * 4));
  std::vector<int> h(n);
  hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d, 20000);
  CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
  int match = 0, hist[16] = {0};
  for (int w = 0; w < n; ++w) { match += (h[w] == (w & 7)); hist[h[w] & 15]++; }
  printf("alone: %d of %d workgroups on XCD w%%8; histogram:", match, n);
  for (int i = 0; i < 8; ++i) printf(" %d", hist[i]);
  printf("\n");
  // two kernels racing from two streams
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  int *d2; CK(hipMalloc(&d2, n * 4));
  hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, s1, d, 20000);
  hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, s2, d2, 20000);
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemcpy(h.data(), rep ? d2 : d, n * 4, hipMemcpyDeviceToHost));
    match = 0;
    for (int w = 0; w < n; ++w) match += (h[w] == (w & 7));
    printf("racing kernel %d: %d of %d workgroups on XCD w%%8\n", rep, match, n);
  }
  return 0;
}
