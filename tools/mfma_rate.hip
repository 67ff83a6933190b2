// mfma_rate.hip -- dev microbenchmark: cycles per v_mfma_f32_16x16x4_f32 for bare streams (no memory), by waves per SIMD
// and by how operands are shared between consecutive instructions.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC, int MODE>
__global__ __launch_bounds__(1024) void k(long long *out, float *sink, int iters, float seed) {
  f32x4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = seed + threadIdx.x + i; b[i] = seed * 0.5f - i; }
  __syncthreads();
  long long t0 = (long long)__builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      const int ai = MODE == 0 ? (i / 8) % 8 : (MODE == 1 ? i % 8 : 0), bi = MODE == 0 ? i % 8 : (MODE == 1 ? (i / 8) % 8 : 0);
      acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ai], b[bi], acc[i], 0, 0, 0);
    }
  }
  long long t1 = (long long)__builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC, int MODE> void run(const char *name, int nt, long long *out, float *sink) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<NACC, MODE>), dim3(256), dim3(nt), 0, 0, out, sink, iters, 1.0f);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<NACC, MODE>), dim3(256), dim3(nt), 0, 0, out, sink, iters, 1.0f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("   wall %.3f ms -> %.1f TFLOP/s   ", ms, 256.0 * (nt / 64) * iters * NACC * 2048.0 / (ms * 1e-3) / 1e12);
  long long h; CK(hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost));
  const double per = (double)h / ((double)iters * NACC);
  printf("%-36s threads/WG %4d (waves/SIMD %d): %.2f clk per MFMA per wave -> %.2f clk per MFMA per SIMD\n", name, nt, nt / 256, per, per / (nt / 256));
}
int main() {
  long long *out; float *sink;
  CK(hipMalloc(&out, 64)); CK(hipMalloc(&sink, 4 * 1024 * 256));
  for (int nt : {256, 512, 1024}) {
    run<16, 0>("16 acc, A shared by 8 in a row", nt, out, sink);
    run<16, 1>("16 acc, B shared stride", nt, out, sink);
    run<16, 2>("16 acc, same A and B", nt, out, sink);
    if (nt == 256) run<64, 0>("64 acc (256 regs), A shared by 8", nt, out, sink);
    run<4, 2>("4 acc (dependent after 4)", nt, out, sink);
  }
  return 0;
}
