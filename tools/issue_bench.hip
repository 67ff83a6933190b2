// issue_bench.hip -- dev microbenchmark: single-wave instruction issue / dependency latencies on gfx950.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void k(long long *out, float *sink, float seed) {
  const int lane = threadIdx.x & 63;
  float a = seed + lane, b = seed * 0.5f, c0 = 1.f, c1 = 2.f, c2 = 3.f, c3 = 4.f, c4 = 5.f, c5 = 6.f, c6 = 7.f, c7 = 8.f;
  long long t[8];
#define STAMP(i, v0, v1, v2, v3) asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t[i]), "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) :: "memory")
  STAMP(0, a, b, c0, c1);
#pragma unroll
  for (int i = 0; i < 256; ++i) a = __builtin_fmaf(a, b, b);            // dependent chain
  STAMP(1, a, b, c0, c1);
#pragma unroll
  for (int i = 0; i < 32; ++i) {                                          // 8 independent chains
    c0 = __builtin_fmaf(c0, b, b); c1 = __builtin_fmaf(c1, b, b); c2 = __builtin_fmaf(c2, b, b); c3 = __builtin_fmaf(c3, b, b);
    c4 = __builtin_fmaf(c4, b, b); c5 = __builtin_fmaf(c5, b, b); c6 = __builtin_fmaf(c6, b, b); c7 = __builtin_fmaf(c7, b, b);
  }
  STAMP(2, c0, c1, c2, c3); STAMP(6, c4, c5, c6, c7); t[2] = t[6];
  float r = a;
#pragma unroll
  for (int i = 0; i < 128; ++i) {                                         // readlane -> fma dependent chain
    float m = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, r), i & 63));
    r = __builtin_fmaf(r, m, b);
  }
  float q0 = a, q1 = c0, q2 = c1, q3 = c2;
  STAMP(3, r, q0, q1, q2);
#pragma unroll
  for (int i = 0; i < 32; ++i) {                                          // 4 readlanes of one value, then 4 independent fmas
    float m0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q0), (4 * i) & 63));
    float m1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q0), (4 * i + 1) & 63));
    float m2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q0), (4 * i + 2) & 63));
    float m3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, q0), (4 * i + 3) & 63));
    q1 = __builtin_fmaf(q1, m1, b); q2 = __builtin_fmaf(q2, m2, b); q3 = __builtin_fmaf(q3, m3, b); q0 = __builtin_fmaf(q0, m0, b);
  }
  STAMP(4, q0, q1, q2, q3);
  float z = r;
  STAMP(7, z, q0, q1, q2); t[4] = t[7];
#pragma unroll
  for (int i = 0; i < 64; ++i) z = __builtin_amdgcn_rsqf(z) + b;          // dependent rsq + add
  STAMP(5, z, q0, q1, q2);
  if (lane == 0 && blockIdx.x == 0 && threadIdx.x < 64) for (int i = 0; i < 6; ++i) out[i] = t[i];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = a + c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + r + q0 + q1 + q2 + q3 + z;
}
int main() {
  long long *out; float *sink;
  CK(hipMalloc(&out, 64)); CK(hipMalloc(&sink, 4 * 1024 * 16));
  for (int nt : {64, 256, 1024}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(nt), 0, 0, out, sink, 1.0001f);
    hipLaunchKernelGGL(k, dim3(1), dim3(nt), 0, 0, out, sink, 1.0001f);
    long long h[6]; CK(hipMemcpy(h, out, 48, hipMemcpyDeviceToHost));
    printf("threads %4d: dep fma %.1f clk/instr | 8 indep fma %.1f | readlane+fma dep pair %.1f | 4 readlane + 4 fma group %.1f (per 8 instr) | rsq+add dep pair %.1f\n", nt,
           (h[1] - h[0]) / 256.0, (h[2] - h[1]) / 256.0, (h[3] - h[2]) / 128.0, (h[4] - h[3]) / 32.0, (h[5] - h[4]) / 64.0);
  }
  return 0;
}
