// What pass 1 of a product draw (the weights of ALL nodes of a tree level, summed on a running maximum: product_body's chunk())
// costs in double precision (the library's own node weight: rsqrt_pos, exp_nonpos) and would cost in single precision
// (v_rsq_f32, v_exp_f32, node statistics as floats in LDS).  Euclid(2), the general form (non-leaf, not on the point):
//   w_z = nw_z / sqrt(v0 v1) * exp(-0.5 (t0^2 / v0 + t1^2 / v1)),  t_k = m_zk - mn_k,  v_k = var_zk + vn_k
// One workgroup of 512 lanes per "product", two lanes per sample (each half of the nodes), NS nodes, DRAWS draws per lane.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -w prod_pass1_proto.hip -o prod_pass1_proto && ./prod_pass1_proto
#include "../../incrementalinference.jl_amd/csrc/nbp_device.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define NS 455

template <int PREC>  // 64: double; 32: single
__global__ void __launch_bounds__(512) k_pass1(const double *stats, double *out, int draws) {
  __shared__ double lm[2 * NS], lv[2 * NS], nw[NS], tab[NBP_EXPTAB];
  __shared__ float fm[2 * NS], fv[2 * NS], fw[NS];
  nbp_exp_tab_init(tab);
  for (int i = threadIdx.x; i < NS; i += blockDim.x) {
    lm[i] = stats[i]; lm[NS + i] = stats[NS + i]; lv[i] = stats[2 * NS + i]; lv[NS + i] = stats[3 * NS + i]; nw[i] = stats[4 * NS + i];
    fm[i] = (float)lm[i]; fm[NS + i] = (float)lm[NS + i]; fv[i] = (float)lv[i]; fv[NS + i] = (float)lv[NS + i]; fw[i] = (float)nw[i];
  }
  __syncthreads();
  const int h = threadIdx.x & 1, z0 = h ? NS / 2 : 0, z1 = h ? NS : NS / 2;
  double acc = 0;
  for (int dr = 0; dr < draws; dr++) {
    const double mn0 = 0.01 * (threadIdx.x >> 1) + 0.001 * dr, mn1 = -0.02 * (threadIdx.x >> 1), vn0 = 0.05 + 1e-4 * dr, vn1 = 0.07;
    if (PREC == 64) {
      double m = -INFINITY, tot = 0;
      for (int z = z0; z + 3 < z1; z += 4) {
        double a[4], g[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const double t0 = lm[z + q] - mn0, t1 = lm[NS + z + q] - mn1, v0 = lv[z + q] + vn0, v1 = lv[NS + z + q] + vn1;
          const double pv = v0 * v1, num = t0 * t0 * v1 + t1 * t1 * v0, r = rsqrt_pos(pv);
          a[q] = -0.5 * num * (r * r);
          g[q] = r * nw[z + q];
        }
        const double am = fmax(fmax(a[0], a[1]), fmax(a[2], a[3]));
        if (am > m) { tot *= exp_nonpos(m - am, tab); m = am; }
        tot += (exp_nonpos(a[0] - m, tab) * g[0] + exp_nonpos(a[1] - m, tab) * g[1]) + (exp_nonpos(a[2] - m, tab) * g[2] + exp_nonpos(a[3] - m, tab) * g[3]);
      }
      acc += tot + m;
    } else {
      const float fmn0 = (float)mn0, fmn1 = (float)mn1, fvn0 = (float)vn0, fvn1 = (float)vn1;
      float m = -INFINITY, tot = 0;
      for (int z = z0; z + 3 < z1; z += 4) {
        float a[4], g[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const float t0 = fm[z + q] - fmn0, t1 = fm[NS + z + q] - fmn1, v0 = fv[z + q] + fvn0, v1 = fv[NS + z + q] + fvn1;
          const float rs = __builtin_amdgcn_rsqf(v0 * v1);  // one transcendental for both the quadratic form and the normalisation
          a[q] = -0.72134752f * (t0 * t0 * v1 + t1 * t1 * v0) * (rs * rs);  // in log2 units
          g[q] = rs * fw[z + q];
        }
        const float am = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
        if (am > m) { tot *= __builtin_amdgcn_exp2f(m - am); m = am; }
        tot += (__builtin_amdgcn_exp2f(a[0] - m) * g[0] + __builtin_amdgcn_exp2f(a[1] - m) * g[1]) + (__builtin_amdgcn_exp2f(a[2] - m) * g[2] + __builtin_amdgcn_exp2f(a[3] - m) * g[3]);
      }
      acc += (double)tot + (double)m;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main(int argc, char **argv) {
  const int nprod = argc > 1 ? atoi(argv[1]) : 975, draws = argc > 2 ? atoi(argv[2]) : 36;
  std::vector<double> st(5 * NS);
  srand(3);
  for (int i = 0; i < NS; i++) { st[i] = rand() / (double)RAND_MAX * 4 - 2; st[NS + i] = rand() / (double)RAND_MAX * 4 - 2; st[2 * NS + i] = 0.01 + rand() / (double)RAND_MAX * 0.2; st[3 * NS + i] = 0.01 + rand() / (double)RAND_MAX * 0.2; st[4 * NS + i] = 1.0 / NS; }
  double *ds, *dout;
  hipMalloc(&ds, st.size() * 8); hipMalloc(&dout, (size_t)nprod * 512 * 8);
  hipMemcpy(ds, st.data(), st.size() * 8, hipMemcpyHostToDevice);
  for (int prec = 64; prec >= 32; prec -= 32) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(a);
      if (prec == 64) k_pass1<64><<<nprod, 512>>>(ds, dout, draws); else k_pass1<32><<<nprod, 512>>>(ds, dout, draws);
      hipEventRecord(b); hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<double> o((size_t)nprod * 512);
    hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost);
    double cs = 0; for (double v : o) cs += v;
    printf("%d products x 256 samples x %d draws x %d nodes, %s precision: %8.3f ms  = %.3f ps per node weight   checksum %.9e\n", nprod, draws, NS, prec == 64 ? "double" : "single",
           ms, ms * 1e9 / ((double)nprod * 256 * draws * NS), cs);
  }
  return 0;
}
