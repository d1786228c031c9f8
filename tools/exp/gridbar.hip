// microbenchmark: latency of a device-scope barrier between K workgroups (atomic counter + fences), the
// building block a multi-CU bandwidth fit would need once per likelihood evaluation.
// Measured on MI355X (round 1): 2.1 us per round for K = 2, 3.1 us for K = 4, 6.1 us for K = 8 with one
// group on the chip, more with several -- against 5.8 us for a whole evaluation on one CU, so splitting
// a fit over several CUs does not pay.  (Latency only: the payload buffer is reused without a second
// barrier, its contents are not meaningful.)
// build: hipcc --offload-arch=gfx950 -O3 -o gridbar gridbar.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(1024) bar_kernel(unsigned *cnt, double *buf, int K, int iters, long long *cycles) {
  const int g = blockIdx.x / K, r = blockIdx.x % K;  // group g, member r
  unsigned *c = cnt + g * 32;
  double *b = buf + (size_t)g * K * 256;
  long long t0 = wall_clock64();
  double acc = 0;
  for (int e = 0; e < iters; e++) {
    if (threadIdx.x < 256) b[r * 256 + threadIdx.x] = (double)(e + r + threadIdx.x);
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)K * (unsigned)(e + 1);
      while (__hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
      __threadfence();
    }
    __syncthreads();
    if (threadIdx.x < 256)
      for (int q = 0; q < K; q++) acc += __builtin_nontemporal_load(&b[q * 256 + threadIdx.x]);
    __syncthreads();
  }
  long long t1 = wall_clock64();
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
  if (acc == -1.0) buf[0] = acc;
  // check: last round's sum
  if (threadIdx.x < 256 && blockIdx.x == 0) buf[(size_t)gridDim.x * 256 + threadIdx.x] = acc;
}
int main() {
  for (int K : {2, 4, 8}) for (int groups : {1, 8, 24}) {
    const int iters = 2000;
    unsigned *cnt; double *buf; long long *cyc;
    hipMalloc(&cnt, groups * 32 * 4); hipMemset(cnt, 0, groups * 32 * 4);
    hipMalloc(&buf, ((size_t)groups * K + 1) * 256 * 8 + 4096); hipMalloc(&cyc, groups * K * 8);
    hipLaunchKernelGGL(bar_kernel, dim3(groups * K), dim3(1024), 0, 0, cnt, buf, K, iters, cyc);
    hipDeviceSynchronize();
    std::vector<long long> h(groups * K);
    hipMemcpy(h.data(), cyc, groups * K * 8, hipMemcpyDeviceToHost);
    long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
    printf("K=%d groups=%d: %.3f us per barrier round (100 MHz clock)\n", K, groups, mx / 100.0 / iters);
    hipFree(cnt); hipFree(buf); hipFree(cyc);
  }
  return 0;
}
