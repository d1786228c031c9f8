// Issue cost of the instructions of the bandwidth fit's pair loop on gfx950, relative to v_fma_f64: a chip-filling launch
// (8 waves per SIMD, 16 independent chains per wave) of ONE instruction kind; and the LDS cost of the loop's three accesses
// (consecutive 8-byte reads, random reads of a 2 KB table, ds_add_f64 without return) per CU.
// hipcc --offload-arch=gfx950 -O3 valu_mix.hip -o valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#define C 16
#define BODY_D(NAME, ASM)                                                                              \
  __global__ void __launch_bounds__(512) NAME(double *out, int iters, double a, double b) {            \
    double x[C];                                                                                       \
    for (int c = 0; c < C; c++) x[c] = threadIdx.x * 1e-3 + c;                                         \
    double sa = a, sb = b;                                                                             \
    asm volatile("" : "+s"(sa), "+s"(sb));                                                             \
    for (int i = 0; i < iters; i++) {                                                                  \
      _Pragma("unroll") for (int u = 0; u < 4; u++) _Pragma("unroll") for (int c = 0; c < C; c++) ASM; \
    }                                                                                                  \
    double s = 0;                                                                                      \
    for (int c = 0; c < C; c++) s += x[c];                                                             \
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xfffff] = s;                                        \
  }
BODY_D(k_fma_vvv, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b)))
BODY_D(k_fma_vsv, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "s"(sa), "v"(b)))
BODY_D(k_mul, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[c]) : "v"(a)))
BODY_D(k_add, asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b)))
BODY_D(k_min, asm volatile("v_min_f64 %0, %0, %1" : "+v"(x[c]) : "v"(b)))
BODY_D(k_fmac, asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b)))
BODY_D(k_mov64, asm volatile("v_mov_b64 %0, %1" : "+v"(x[c]) : "v"(b)))
BODY_D(k_ldexp, asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(x[c])))
#define BODY_I(NAME, ASM)                                                                              \
  __global__ void __launch_bounds__(512) NAME(double *out, int iters, double a, double b) {            \
    int x[C];                                                                                          \
    for (int c = 0; c < C; c++) x[c] = threadIdx.x + c;                                                \
    int k3 = 3, m = (int)a + 255;                                                                      \
    asm volatile("" : "+s"(k3));                                                                       \
    for (int i = 0; i < iters; i++) {                                                                  \
      _Pragma("unroll") for (int u = 0; u < 4; u++) _Pragma("unroll") for (int c = 0; c < C; c++) ASM; \
    }                                                                                                  \
    int s = 0;                                                                                         \
    for (int c = 0; c < C; c++) s += x[c];                                                             \
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xfffff] = s;                                        \
  }
BODY_I(k_and, asm volatile("v_and_b32 %0, %1, %0" : "+v"(x[c]) : "v"(m)))
BODY_I(k_lshladd, asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x[c]) : "v"(m)))
BODY_I(k_sdwa, asm volatile("v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "+v"(x[c]) : "s"(k3)))
BODY_I(k_mov32, asm volatile("v_mov_b32 %0, %1" : "+v"(x[c]) : "v"(m)))
BODY_I(k_dpp, asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(x[c])))
BODY_I(k_dppw, asm volatile("v_mov_b32_dpp %0, %0 wave_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(x[c])))
BODY_I(k_addu, asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[c]) : "v"(m)))

// LDS: one kind of access, 8 per iteration and lane, 16 waves per CU
template <int KIND>
__global__ void __launch_bounds__(1024) k_lds(double *out, int iters, int stride) {
  extern __shared__ double sm[];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) sm[i] = i;
  __syncthreads();
  double s = 0;
  unsigned r = threadIdx.x * 2654435761u;
  const int base = (threadIdx.x & 63) + (threadIdx.x >> 6) * 512;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      if (KIND == 0) s += sm[base + u + (i & 7) * 16];                          // consecutive lanes, consecutive doubles
      if (KIND == 1) { r = r * 1664525u + 1013904223u; s += sm[(r >> 24)]; }   // random entries of a 2 KB table
      if (KIND == 2) (void)__builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double *)(sm + base + u + (i & 7) * 16), 1.0);
      if (KIND == 3) { r = r * 1664525u + 1013904223u; s += sm[((r >> 24) << 3) + (threadIdx.x & 7)]; }  // random entries, 8 copies interleaved
    }
  }
  out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xfffff] = s + sm[threadIdx.x];
}
// the random-index arithmetic alone (to subtract)
__global__ void __launch_bounds__(1024) k_rng(double *out, int iters) {
  unsigned r = threadIdx.x * 2654435761u;
  unsigned s = 0;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) { r = r * 1664525u + 1013904223u; s += r >> 24; }
  }
  out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xfffff] = s;
}

int main() {
  double *o; hipMalloc(&o, 8 << 20);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms, ref = 0;
  const int blocks = 256 * 4 * 4, it = 500;  // 512-lane workgroups, 4 per CU resident (8 waves per SIMD), 4 rounds
#define RUN(K)                                                                                                         \
  hipLaunchKernelGGL(K, dim3(blocks), dim3(512), 0, 0, o, it, 0.999, 1e-3);                                            \
  hipEventRecord(e0); hipLaunchKernelGGL(K, dim3(blocks), dim3(512), 0, 0, o, it, 0.999, 1e-3); hipEventRecord(e1);    \
  hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);                                                           \
  if (ref == 0) ref = ms;                                                                                              \
  printf("%-12s %8.3f ms  %.2f x v_fma_f64  (%.2f cycles per wave-instruction and SIMD if v_fma_f64 = 4)\n", #K, ms, ms / ref, 4.0 * ms / ref);
  RUN(k_fma_vvv) RUN(k_fma_vsv) RUN(k_mul) RUN(k_add) RUN(k_min) RUN(k_fmac) RUN(k_mov64) RUN(k_ldexp)
  RUN(k_and) RUN(k_lshladd) RUN(k_sdwa) RUN(k_mov32) RUN(k_dpp) RUN(k_dppw) RUN(k_addu)
  {
    const double wave_instr = (double)blocks * 8 * it * 4 * C;  // per kernel
    printf("reference: %.3f ms for %.3g wave-instructions on 1024 SIMDs = %.2f ns per instruction and SIMD\n", ref, wave_instr, ref * 1e6 / (wave_instr / 1024));
  }
  const int lb = 256 * 4, lit = 2000;
#define RUNL(KIND, NAME)                                                                                               \
  hipLaunchKernelGGL(k_lds<KIND>, dim3(lb), dim3(1024), 65536, 0, o, lit, 1);                                          \
  hipEventRecord(e0); hipLaunchKernelGGL(k_lds<KIND>, dim3(lb), dim3(1024), 65536, 0, o, lit, 1); hipEventRecord(e1);  \
  hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);                                                           \
  printf("lds %-28s %8.3f ms: %.2f ns per wave-access and CU\n", NAME, ms, ms * 1e6 / ((double)lb / 256 * 16 * lit * 8));
  RUNL(0, "read b64 consecutive") RUNL(1, "read b64 random 2 KB") RUNL(2, "ds_add_f64 consecutive") RUNL(3, "read b64 random, 8 copies")
  hipLaunchKernelGGL(k_rng, dim3(lb), dim3(1024), 0, 0, o, lit);
  hipEventRecord(e0); hipLaunchKernelGGL(k_rng, dim3(lb), dim3(1024), 0, 0, o, lit); hipEventRecord(e1);
  hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  printf("index arithmetic of the random reads alone: %8.3f ms\n", ms);
  return 0;
}
