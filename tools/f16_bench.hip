// f16_bench.hip -- dev microbenchmark: cycles of factor16 alone (1 wave) and beside busy MFMA waves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include "../projected-lmc_amd/csrc/diag_block.hpp"
using namespace plmc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k(long long *out, float *sink, int busy_iters) {
  __shared__ float S[512], ub[272], wb[272];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  for (int e = threadIdx.x; e < 512; e += blockDim.x) { int r = e >> 5, c = e & 31; S[e] = c < 16 ? ((r == c ? 20.f : 0.f) + 0.01f * (((r < c ? r * 16 + c : c * 16 + r) * 7) % 13)) : (c - 16 == r ? 1.f : 0.f); }
  __syncthreads();
  if (wave == 0) {
    __builtin_amdgcn_s_setprio(3);
    long long t0 = 0, t1 = 0;
#pragma unroll 1
    for (int it = 0; it < 9; ++it) {
      if (it == 1) t0 = (long long)__builtin_readcyclecounter();
      int ls = lane;
      asm volatile("" : "+v"(ls));
      factor16<float>(S, ub, wb, ls);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    t1 = (long long)__builtin_readcyclecounter();
    if (lane == 0) out[0] = (t1 - t0) / 8;
  } else {
    // busy MFMA waves (like the trailing update next to the factor wave)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + lane, b = 0.5f;
#pragma unroll 1
    for (int it = 0; it < busy_iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    sink[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  }
}
int main() {
  long long *out; float *sink;
  CK(hipMalloc(&out, 64)); CK(hipMalloc(&sink, 4 * 1024));
  for (int nt : {64, 128, 320, 512}) for (int busy : {0, 2000}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(nt), 0, 0, out, sink, busy);
    hipLaunchKernelGGL(k, dim3(1), dim3(nt), 0, 0, out, sink, busy);
    long long h; CK(hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost));
    printf("threads %4d busy_iters %5d: factor16 = %lld clk\n", nt, busy, h);
  }
  return 0;
}
