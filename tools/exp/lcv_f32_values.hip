// The single-precision likelihood evaluation against the double-precision one, value by value: one workgroup, a belief of
// `cnt` points in a row of Npad lanes (P rows), a ladder of bandwidths.  Prints f64, f32, their difference and the bound.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on lcv_f32_values.hip -o lcv_f32_values && ./lcv_f32_values
#include "../../incrementalinference.jl_amd/csrc/nbp_device.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

__global__ void __launch_bounds__(1024) k_vals(const double *pts, int N, int Npad, int circ, const double *hs, int nh, double *out) {
  extern __shared__ double smem[];
  const int P = blockDim.x / Npad, n = threadIdx.x;
  double *tab = smem, *X = smem + NBP_FITTAB, *part = X + 2 * N, *red = part + P * Npad + (blockDim.x >> 6) * 2 * N;
  nbp_fit_tab_init(tab);
  if (n < N) X[n] = X[n + N] = pts[n];
  double *acc = part + P * Npad;
  for (int q = n; q < (int)(blockDim.x >> 6) * 2 * N; q += blockDim.x) acc[q] = 0.0;
  for (int q = n; q < (int)blockDim.x; q += blockDim.x) part[q] = 0.0;
  __syncthreads();
  double lo = INFINITY, hi = -INFINITY;
  if (n < N) lo = hi = X[n];
  lo = block_min(lo, red);
  hi = block_max(hi, red);
  const double cen = circ ? 0.0 : 0.5 * (lo + hi), xmax = circ ? 0.0125 : 0.5 * (hi - lo);
  const double ln0 = 0.5 * log(NBP_TWO_PI) + log((double)(N - 1));
  for (int k = 0; k < nh; k++) {
    const double a = neg_loo_ll(X, N, Npad, circ != 0, hs[k], ln0, part, red, tab);
    const double b = neg_loo_ll_f32(X, N, Npad, circ != 0, hs[k], ln0, cen, part, red);
    const double c = neg_loo_ll(X, N, Npad, circ != 0, hs[k], ln0, part, red, tab);  // (the double evaluation after a single one)
    if (n == 0) { out[4 * k] = a; out[4 * k + 1] = b; out[4 * k + 2] = c; out[4 * k + 3] = lcv_f32_bound(xmax, hs[k], N); }
  }
}

int main(int argc, char **argv) {
  const int Ncap = 200, Npad = 256;
  const int cnts[] = {200, 150, 65, 64, 33, 9, 3, 129, 97};
  for (int P = 1; P <= 4; P *= 2)
    for (int circ = 0; circ < 2; circ++)
      for (int ci = 0; ci < 9; ci++) {
        const int N = cnts[ci];
        std::vector<double> p(N);
        srand(7 + N);
        for (auto &v : p) {
          double u1 = (rand() + 1.0) / (RAND_MAX + 2.0), u2 = (rand() + 1.0) / (RAND_MAX + 2.0);
          v = 0.7 * sqrt(-2 * log(u1)) * cos(6.283185307179586 * u2);
        }
        std::vector<double> hs = {1.5, 0.7, 0.3, 0.12, 0.05};
        double *dp, *dh, *dout;
        hipMalloc(&dp, N * 8); hipMalloc(&dh, hs.size() * 8); hipMalloc(&dout, hs.size() * 32);
        hipMemcpy(dp, p.data(), N * 8, hipMemcpyHostToDevice);
        hipMemcpy(dh, hs.data(), hs.size() * 8, hipMemcpyHostToDevice);
        const size_t lds = (2 * (size_t)N + (size_t)P * Npad + (size_t)(P * Npad / 64) * 2 * N + NBP_RED + NBP_FITTAB) * 8;
        k_vals<<<1, P * Npad, lds>>>(dp, N, Npad, circ, dh, (int)hs.size(), dout);
        std::vector<double> o(hs.size() * 4);
        hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost);
        for (size_t k = 0; k < hs.size(); k++) {
          double ref = 0;  // the likelihood on the host
          for (int i = 0; i < N; i++) {
            double sm = 0;
            for (int j = 0; j < N; j++) {
              if (j == i) continue;
              double d = p[i] - p[j];
              if (circ) d = remainder(d, 6.283185307179586);
              sm += exp(-d * d / (2 * hs[k] * hs[k]));
            }
            if (sm < 1e-300) sm = 1e-300;
            ref += log(sm) - (log(hs[k]) + 0.5 * log(6.283185307179586) + log((double)(N - 1)));
          }
          ref = -ref / N;
          printf("host %.12f  ", ref);
          printf("P=%d circ=%d N=%3d h=%5.2f  f64 %.12f  f32 %.12f  diff %9.2e  bound %9.2e  %s %s\n", P, circ, N, hs[k], o[4 * k], o[4 * k + 1],
                 o[4 * k + 1] - o[4 * k], o[4 * k + 3], fabs(o[4 * k + 1] - o[4 * k]) <= o[4 * k + 3] || o[4 * k + 1] != o[4 * k + 1] ? "ok" : "BOUND VIOLATED",
                 o[4 * k + 2] == o[4 * k] ? "" : "DOUBLE-AFTER-SINGLE DIFFERS");
        }
        hipFree(dp); hipFree(dh); hipFree(dout);
      }
  (void)Ncap;
  return 0;
}
