// FP64 VALU issue rate and dependent-issue latency on gfx950: cycles per v_fma_f64 for C independent chains per wave,
// W waves per SIMD (one workgroup of 256*W threads on one CU).  hipcc --offload-arch=gfx950 -O3 dp_rate.hip -o dp_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int C>
__global__ void k(double *out, long long *cyc, int iters, double a, double b) {
  double x[C];
  for (int c = 0; c < C; c++) x[c] = threadIdx.x * 1e-3 + c;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < C; c++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
  for (int c = 0; c < C; c++) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int C>
__global__ void ki(int *out, long long *cyc, int iters, int a) {
  int x[C];
  for (int c = 0; c < C; c++) x[c] = threadIdx.x + c;
  __syncthreads();
  long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int c = 0; c < C; c++) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(x[c]) : "v"(a));
  }
  long long t1 = __builtin_readcyclecounter();
  int s = 0;
  for (int c = 0; c < C; c++) s += x[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int C>
__global__ void kfull(double *out, int iters, double a, double b) {
  double x[C];
  for (int c = 0; c < C; c++) x[c] = threadIdx.x * 1e-3 + c;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int c = 0; c < C; c++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
  }
  double s = 0;
  for (int c = 0; c < C; c++) s += x[c];
  out[(blockIdx.x * blockDim.x + threadIdx.x) & 0xfffff] = s;
}
int main() {
  {
    double *o; hipMalloc(&o, 8 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpb = 4; wpb <= 16; wpb *= 2) {
      const int blocks = 256 * 32 / wpb * 4, it = 2000;
      hipLaunchKernelGGL(kfull<8>, dim3(blocks), dim3(64 * wpb), 0, 0, o, it, 0.999, 1e-3);
      hipEventRecord(e0);
      hipLaunchKernelGGL(kfull<8>, dim3(blocks), dim3(64 * wpb), 0, 0, o, it, 0.999, 1e-3);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)blocks * 64 * wpb * it * 32 * 2;
      printf("chip-filling f64 fma: %d blocks x %d waves: %.3f ms, %.1f TFLOP/s\n", blocks, wpb, ms, flops / ms / 1e9);
    }
  }
  double *out; long long *cyc, h;
  hipMalloc(&out, 8 << 20); hipMalloc(&cyc, 1024);
  const int it = 4000;
  for (int W = 1; W <= 4; W *= 2) {
#define RUN(C) hipLaunchKernelGGL(k<C>, dim3(1), dim3(256 * W), 0, 0, out, cyc, it, 0.999, 1e-3); hipDeviceSynchronize(); hipLaunchKernelGGL(k<C>, dim3(1), dim3(256 * W), 0, 0, out, cyc, it, 0.999, 1e-3); hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("f64 fma: waves/SIMD %d chains %d: %.2f cycles per wave-instruction (wave 0 view), %.2f per instruction per SIMD\n", W, C, (double)h / (it * C), (double)h / (it * C * W));
    RUN(1) RUN(2) RUN(4) RUN(8)
#define RUNI(C) hipLaunchKernelGGL(ki<C>, dim3(1), dim3(256 * W), 0, 0, (int *)out, cyc, it, 3); hipDeviceSynchronize(); hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost); printf("u32 lshl_add: waves/SIMD %d chains %d: %.2f cycles per wave-instruction, %.2f per instruction per SIMD\n", W, C, (double)h / (it * C), (double)h / (it * C * W));
    RUNI(1) RUNI(4)
  }
  return 0;
}
