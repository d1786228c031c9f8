// Prototype of the bandwidth fit's likelihood evaluation in single precision (the "bracketing" evaluations of the golden-section
// search): what one evaluation costs per ordered pair on gfx950 for several forms of the pair loop.  One workgroup of 256 lanes
// per fit, N = 200 points, lane i owns point i, EV evaluations in a row at different bandwidths, chip-filling launch.
//   hipcc --offload-arch=gfx950 -O3 lcv_f32_proto.hip -o lcv_f32_proto && ./lcv_f32_proto
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define N 200
#define NP 208  // N rounded up to a multiple of 8, the padding at a distance that underflows

__device__ __forceinline__ float exp2_fast(float x) { return __builtin_amdgcn_exp2f(x); }

// V0: broadcast reads (ds_read_b128: four partners per read), self term included, removed afterwards (s - 1)
// V1: the same with the self term masked per pair (v_cndmask)
// V2: packed arithmetic (v_pk_add_f32 / v_pk_mul_f32), self term included
// V3: rotation: lane i reads x[i + t] (x stored twice), no self term by construction
template <int V>
__global__ void __launch_bounds__(256) k_eval(const double *pts, float *out, int EV) {
  __shared__ __attribute__((aligned(16))) float xs[2 * NP];
  const int i = threadIdx.x;
  const double *x = pts + (size_t)blockIdx.x * N;
  float acc_out = 0.f;
  for (int ev = 0; ev < EV; ev++) {
    const double h = 0.2 + 0.01 * ev;
    const double sc = sqrt(1.4426950408889634 / (2.0 * h * h));
    __syncthreads();
    for (int j = i; j < 2 * NP; j += 256) {
      const int jj = j < NP ? j : j - NP;
      xs[j] = (jj < N) ? (float)(x[jj] * sc) : 1e18f;
    }
    __syncthreads();
    const float xi = (i < N) ? xs[i] : 0.f;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (V == 0) {
      const float4 *x4 = (const float4 *)xs;
#pragma unroll 2
      for (int g = 0; g < NP / 4; g++) {
        const float4 y = x4[g];
        const float d0 = xi - y.x, d1 = xi - y.y, d2 = xi - y.z, d3 = xi - y.w;
        s0 += exp2_fast(-d0 * d0);
        s1 += exp2_fast(-d1 * d1);
        s2 += exp2_fast(-d2 * d2);
        s3 += exp2_fast(-d3 * d3);
      }
      s0 -= 1.0f;
    } else if (V == 1) {
      const float4 *x4 = (const float4 *)xs;
#pragma unroll 2
      for (int g = 0; g < NP / 4; g++) {
        const float4 y = x4[g];
        const float d0 = xi - y.x, d1 = xi - y.y, d2 = xi - y.z, d3 = xi - y.w;
        const int j = 4 * g;
        s0 += (j == i) ? 0.f : exp2_fast(-d0 * d0);
        s1 += (j + 1 == i) ? 0.f : exp2_fast(-d1 * d1);
        s2 += (j + 2 == i) ? 0.f : exp2_fast(-d2 * d2);
        s3 += (j + 3 == i) ? 0.f : exp2_fast(-d3 * d3);
      }
    } else if (V == 2) {
      typedef float v2f __attribute__((ext_vector_type(2)));
      const float4 *x4 = (const float4 *)xs;
      v2f sa = {0.f, 0.f}, sb = {0.f, 0.f};
      const v2f xi2 = {xi, xi};
#pragma unroll 2
      for (int g = 0; g < NP / 4; g++) {
        const float4 y = x4[g];
        const v2f ya = {y.x, y.y}, yb = {y.z, y.w};
        const v2f da = xi2 - ya, db = xi2 - yb;
        const v2f qa = -da * da, qb = -db * db;
        v2f ea, eb;
        ea.x = exp2_fast(qa.x); ea.y = exp2_fast(qa.y);
        eb.x = exp2_fast(qb.x); eb.y = exp2_fast(qb.y);
        sa += ea;
        sb += eb;
      }
      s0 = sa.x - 1.0f; s1 = sa.y; s2 = sb.x; s3 = sb.y;
    } else if (V == 3) {
      if (i < N) {
        // x stored twice at stride N (not NP) for the rotation: rebuild a private view through the first copy + wrap
        for (int t = 1; t + 3 < N; t += 4) {
          int j0 = i + t, j1 = j0 + 1, j2 = j0 + 2, j3 = j0 + 3;
          j0 -= (j0 >= N) ? N : 0; j1 -= (j1 >= N) ? N : 0; j2 -= (j2 >= N) ? N : 0; j3 -= (j3 >= N) ? N : 0;
          const float d0 = xi - xs[j0], d1 = xi - xs[j1], d2 = xi - xs[j2], d3 = xi - xs[j3];
          s0 += exp2_fast(-d0 * d0);
          s1 += exp2_fast(-d1 * d1);
          s2 += exp2_fast(-d2 * d2);
          s3 += exp2_fast(-d3 * d3);
        }
        for (int t = 1 + ((N - 1) & ~3); t < N; t++) {
          int j0 = i + t;
          j0 -= (j0 >= N) ? N : 0;
          const float d0 = xi - xs[j0];
          s0 += exp2_fast(-d0 * d0);
        }
      }
    }
    const float s = (s0 + s1) + (s2 + s3);
    acc_out += __logf(fmaxf(s, 1e-30f));
  }
  if (i < N) out[(size_t)blockIdx.x * N + i] = acc_out;
}

template <int V>
static void run(const char *name, const double *dp, float *dout, int fits, int EV, std::vector<float> *res) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  k_eval<V><<<fits, 256>>>(dp, dout, 2);
  hipDeviceSynchronize();
  hipEventRecord(a);
  k_eval<V><<<fits, 256>>>(dp, dout, EV);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  const double pairs = (double)fits * EV * N * (N - 1);  // ordered pairs
  res->resize((size_t)fits * N);
  hipMemcpy(res->data(), dout, res->size() * 4, hipMemcpyDeviceToHost);
  double cs = 0;
  for (size_t q = 0; q < res->size(); q++) cs += (*res)[q];
  printf("%-44s %8.3f ms  %7.3f ps per ordered pair  = %7.3f ps per symmetric pair   checksum %.6e\n", name, ms, ms * 1e9 / pairs, 2 * ms * 1e9 / pairs, cs);
}

int main(int argc, char **argv) {
  const int fits = argc > 1 ? atoi(argv[1]) : 8192, EV = argc > 2 ? atoi(argv[2]) : 16;
  std::vector<double> p((size_t)fits * N);
  srand(1);
  for (auto &v : p) {
    double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
    v = sqrt(-2 * log(u1)) * cos(6.283185307179586 * u2);
  }
  double *dp;
  float *dout;
  hipMalloc(&dp, p.size() * 8);
  hipMalloc(&dout, p.size() * 4);
  hipMemcpy(dp, p.data(), p.size() * 8, hipMemcpyHostToDevice);
  std::vector<float> r;
  printf("fits %d, N %d, %d evaluations each (the double-precision loop of the library: 0.765 ps per symmetric pair)\n", fits, N, EV);
  run<0>("V0 broadcast b128, self term subtracted", dp, dout, fits, EV, &r);
  run<1>("V1 broadcast b128, self term masked", dp, dout, fits, EV, &r);
  run<2>("V2 packed f32, self term subtracted", dp, dout, fits, EV, &r);
  run<3>("V3 rotation (per-lane reads)", dp, dout, fits, EV, &r);
  return 0;
}
