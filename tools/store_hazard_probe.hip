// store_hazard_probe.hip -- hardware probe behind DESIGN.md "write-back hazard" (gfx950).
//
// Question: after `buffer_store_dwordx4 v[a:a+3], voff, rsrc, SOFF offen`, how many wait states does the next
// VALU write to v[a] need before it no longer lands in the stored data?  hipcc's hazard recognizer inserts
// s_nop only when SOFF is not an SGPR (llvm GCNHazardRecognizer::createsVALUHazard); the tile write-back uses an
// SGPR soffset and a compiler-scheduled `v_and_b32 v16, ...` right behind `buffer_store_dwordx4 v[16:19]`
// produced wrong tiles (tools/wb_race_probe.py).  Each kernel stores a known pattern with hand-written asm,
// overwrites the first data register with 0xDEADBEEF after NOPS wait states, and the host counts how many
// stored dwords carry the marker.   hipcc --offload-arch=gfx950 -O2 -o store_hazard_probe store_hazard_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef int i32x4 __attribute__((ext_vector_type(4)));

// ISSGPR = 1: row offset in an SGPR soffset (the form LLVM exempts); 0: soffset = 0, row offset folded into voffset.
// Per iteration a wave stores four 8-row chunks back to back (three to back the pipeline up, the fourth probed).
#define PROBE_KERNEL(NAME, NOPSTR, ISSGPR)                                                                        \
  __global__ __launch_bounds__(256) void NAME(float *out, const float *bg, float *sink, int rows_per_wg, int ld) { \
    const int tid = threadIdx.x;                                                                                   \
    float *base = out + (size_t)blockIdx.x * rows_per_wg * ld;                                                     \
    const uint64_t ba = (uint64_t)base;                                                                            \
    i32x4 rs;                                                                                                      \
    rs.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)ba);                                                      \
    rs.y = __builtin_amdgcn_readfirstlane((int)((uint32_t)(ba >> 32) & 0xffffu));                                  \
    rs.z = 0x7fffffff;                                                                                             \
    rs.w = 0x00020000;                                                                                             \
    const __amdgpu_buffer_rsrc_t rb =                                                                              \
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(bg) + (size_t)blockIdx.x * rows_per_wg * ld, 0, 0x7fffffff, 0x00020000); \
    const unsigned voff = (unsigned)(((tid / 32) * ld + (tid % 32) * 4) * 4);                                      \
    const unsigned rstep = (unsigned)(8 * ld * 4);                                                                 \
    i32x4 acc = {0, 0, 0, 0};                                                                                      \
    for (int it = 0; it < rows_per_wg / 32; ++it) {                                                                \
      const unsigned s0 = __builtin_amdgcn_readfirstlane((4 * it + 0) * rstep), s1 = s0 + rstep, s2 = s1 + rstep, s3 = s2 + rstep; \
      i32x4 l0 = __builtin_amdgcn_raw_buffer_load_b128(rb, voff, s0, 0);                                           \
      i32x4 l1 = __builtin_amdgcn_raw_buffer_load_b128(rb, voff, s1, 0);                                           \
      const unsigned key = (blockIdx.x * rows_per_wg + it * 32 + tid / 32) * 128u + (tid % 32) * 4u;               \
      if (ISSGPR) {                                                                                                \
        asm volatile(                                                                                              \
            "v_add_u32 v40, 3072, %0\n v_add_u32 v41, 3073, %0\n v_add_u32 v42, 3074, %0\n v_add_u32 v43, 3075, %0\n" \
            "v_add_u32 v44, 0, %0\n v_add_u32 v45, 1, %0\n v_add_u32 v46, 2, %0\n v_add_u32 v47, 3, %0\n"         \
            "v_add_u32 v48, 1024, %0\n v_add_u32 v49, 1025, %0\n v_add_u32 v50, 1026, %0\n v_add_u32 v51, 1027, %0\n" \
            "v_add_u32 v52, 2048, %0\n v_add_u32 v53, 2049, %0\n v_add_u32 v54, 2050, %0\n v_add_u32 v55, 2051, %0\n" \
            "s_nop 4\n"                                                                                            \
            "buffer_store_dwordx4 v[44:47], %1, %2, %3 offen\n"                                                    \
            "buffer_store_dwordx4 v[48:51], %1, %2, %4 offen\n"                                                    \
            "buffer_store_dwordx4 v[52:55], %1, %2, %5 offen\n"                                                    \
            "buffer_store_dwordx4 v[40:43], %1, %2, %6 offen\n" NOPSTR                                             \
            "v_mov_b32 v40, 0xdeadbeef\n"                                                                          \
            "s_nop 4\n"                                                                                            \
            :                                                                                                      \
            : "v"(key), "v"(voff), "s"(rs), "s"(s0), "s"(s1), "s"(s2), "s"(s3)                                     \
            : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53",   \
              "v54", "v55", "memory");                                                                             \
      } else {                                                                                                     \
        asm volatile(                                                                                              \
            "v_add_u32 v40, 3072, %0\n v_add_u32 v41, 3073, %0\n v_add_u32 v42, 3074, %0\n v_add_u32 v43, 3075, %0\n" \
            "v_add_u32 v44, 0, %0\n v_add_u32 v45, 1, %0\n v_add_u32 v46, 2, %0\n v_add_u32 v47, 3, %0\n"         \
            "v_add_u32 v48, 1024, %0\n v_add_u32 v49, 1025, %0\n v_add_u32 v50, 1026, %0\n v_add_u32 v51, 1027, %0\n" \
            "v_add_u32 v52, 2048, %0\n v_add_u32 v53, 2049, %0\n v_add_u32 v54, 2050, %0\n v_add_u32 v55, 2051, %0\n" \
            "s_nop 4\n"                                                                                            \
            "buffer_store_dwordx4 v[44:47], %3, %2, 0 offen\n"                                                     \
            "buffer_store_dwordx4 v[48:51], %4, %2, 0 offen\n"                                                     \
            "buffer_store_dwordx4 v[52:55], %5, %2, 0 offen\n"                                                     \
            "buffer_store_dwordx4 v[40:43], %6, %2, 0 offen\n" NOPSTR                                              \
            "v_mov_b32 v40, 0xdeadbeef\n"                                                                          \
            "s_nop 4\n"                                                                                            \
            :                                                                                                      \
            : "v"(key), "v"(voff), "s"(rs), "v"(voff + s0), "v"(voff + s1), "v"(voff + s2), "v"(voff + s3)         \
            : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53",   \
              "v54", "v55", "memory");                                                                             \
      }                                                                                                            \
      acc += l0 + l1;                                                                                              \
    }                                                                                                              \
    if (acc.x == 0x12345) sink[tid] = 1.f;                                                                         \
  }

PROBE_KERNEL(k_sgpr_nop0, "", 1)
PROBE_KERNEL(k_sgpr_nop1, "s_nop 0\n", 1)
PROBE_KERNEL(k_sgpr_nop2, "s_nop 1\n", 1)
PROBE_KERNEL(k_imm_nop0, "", 0)
PROBE_KERNEL(k_imm_nop1, "s_nop 0\n", 0)
PROBE_KERNEL(k_imm_nop2, "s_nop 1\n", 0)

typedef void (*kern_t)(float *, const float *, float *, int, int);

int main() {
  const int nwg = 4096, rows = 64, ld = 128;
  const size_t n = (size_t)nwg * rows * ld;
  float *out, *bg, *sink;
  CK(hipMalloc(&out, n * 4));
  CK(hipMalloc(&bg, n * 4));
  CK(hipMalloc(&sink, 4096));
  CK(hipMemset(bg, 0, n * 4));
  std::vector<uint32_t> h(n);
  struct { const char *name; kern_t k; } ks[] = {{"sgpr_soffset,0_wait_states", k_sgpr_nop0}, {"sgpr_soffset,1_wait_state", k_sgpr_nop1},
                                                 {"sgpr_soffset,2_wait_states", k_sgpr_nop2}, {"imm_soffset,0_wait_states", k_imm_nop0},
                                                 {"imm_soffset,1_wait_state", k_imm_nop1},  {"imm_soffset,2_wait_states", k_imm_nop2}};
  printf("{");
  for (int v = 0; v < 6; ++v) {
    long bad = 0, wrong = 0, lanes[16] = {0};
    for (int rep = 0; rep < 4; ++rep) {
      CK(hipMemset(out, 0xff, n * 4));
      hipLaunchKernelGGL(ks[v].k, dim3(nwg), dim3(256), 0, 0, out, bg, sink, rows, ld);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < n; ++i) {
        if (h[i] == 0xdeadbeefu) { ++bad; lanes[(i / 4) % 16]++; }
        else if (h[i] != (uint32_t)i) ++wrong;
      }
    }
    printf("%s\"%s\": {\"marker_dwords\": %ld, \"other_wrong\": %ld, \"of\": %zu, \"by_lane_mod16\": [", v ? ", " : "", ks[v].name, bad, wrong, 4 * n);
    for (int l = 0; l < 16; ++l) printf("%s%ld", l ? "," : "", lanes[l]);
    printf("]}");
  }
  printf("}\n");
  return 0;
}
