// Which arithmetic variant of the Nelder-Mead restatement produces output differences, and how large (LinearRelative, 2-D and 3-D):
//   gcc -O2 -ffp-contract=off tools/exp/nm_arithmetic.c -lm -o /tmp/nm_arithmetic && /tmp/nm_arithmetic
// (profiles/r04_nelder_mead_arithmetic.txt)
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static double T[3];
static int OBJV;
static double objective(const double *x, int n) {
  double acc = 0;
  for (int i = 0; i < n; i++) { double r = x[i] - T[i]; if (OBJV) acc = fma(r, r, acc); else acc += r * r; }
  return acc;
}
static void nm_sort(int m, const double *f, int *ord) {
  for (int i = 0; i < m; i++) ord[i] = i;
  for (int i = 1; i < m; i++) { int k = ord[i], j = i - 1; while (j >= 0 && f[ord[j]] > f[k]) { ord[j + 1] = ord[j]; j--; } ord[j + 1] = k; }
}
static int nm_converged(int m, int n, const double *f) {
  double a = 0; for (int i = 0; i < m; i++) a += f[i]; a = a / m;
  double v = 0; for (int i = 0; i < m; i++) v += (f[i] - a) * (f[i] - a);
  return sqrt(v / (m - 1) * ((double)n / (double)m)) <= 1e-8;
}
// variant bits: 1 = contracted vertex formulas, 2 = centroid in sorted order
static int nm(int n, double *x, int variant, int *iters) {
  const int m = n + 1;
  const double alpha = 1.0, beta = 1.0 + 2.0 / n, gamma = 0.75 - 1.0 / (2.0 * n), delta = 1.0 - 1.0 / n, rn = 1.0 / n;
  double sx[4][3], f[4], xc[3], xr[3], xcache[3], xl[3]; int ord[4];
  for (int i = 0; i < m; i++) for (int d = 0; d < n; d++) sx[i][d] = x[d];
  for (int j = 0; j < n; j++) sx[j + 1][j] = (1.0 + 0.5) * sx[j + 1][j] + 0.025;
  for (int i = 0; i < m; i++) f[i] = objective(sx[i], n);
  nm_sort(m, f, ord);
  int converged = nm_converged(m, n, f), it = 0;
  while (!converged && it < 1000) {
    it++;
    int shrink = 0, ih = ord[m - 1];
    for (int d = 0; d < n; d++) {
      double s = 0;
      if (variant & 2) { for (int i = 0; i < n; i++) s += sx[ord[i]][d]; }
      else for (int i = 0; i < m; i++) if (i != ih) s += sx[i][d];
      xc[d] = s * rn; xl[d] = sx[ord[0]][d];
    }
    double f_lowest = f[ord[0]], f_second = f[ord[n - 1]], f_highest = f[ih];
    for (int d = 0; d < n; d++) xr[d] = xc[d] + alpha * (xc[d] - sx[ih][d]);
    double f_reflect = objective(xr, n);
#define COMB(c, co, a, b) ((variant & 1) ? fma((co), (a) - (b), (c)) : (c) + (co) * ((a) - (b)))
    if (f_reflect < f_lowest) {
      for (int d = 0; d < n; d++) xcache[d] = COMB(xc[d], beta, xr[d], xc[d]);
      double f_expand = objective(xcache, n);
      if (f_expand < f_reflect) { for (int d = 0; d < n; d++) sx[ih][d] = xcache[d]; f[ih] = f_expand; }
      else { for (int d = 0; d < n; d++) sx[ih][d] = xr[d]; f[ih] = f_reflect; }
      for (int i = m - 1; i >= 1; i--) ord[i] = ord[i - 1]; ord[0] = ih;
    } else if (f_reflect < f_second) {
      for (int d = 0; d < n; d++) sx[ih][d] = xr[d]; f[ih] = f_reflect; nm_sort(m, f, ord);
    } else {
      if (f_reflect < f_highest) {
        for (int d = 0; d < n; d++) xcache[d] = COMB(xc[d], gamma, xr[d], xc[d]);
        double fc = objective(xcache, n);
        if (fc < f_reflect) { for (int d = 0; d < n; d++) sx[ih][d] = xcache[d]; f[ih] = fc; nm_sort(m, f, ord); } else shrink = 1;
      } else {
        for (int d = 0; d < n; d++) xcache[d] = COMB(xc[d], -gamma, xr[d], xc[d]);
        double fc = objective(xcache, n);
        if (fc < f_highest) { for (int d = 0; d < n; d++) sx[ih][d] = xcache[d]; f[ih] = fc; nm_sort(m, f, ord); } else shrink = 1;
      }
    }
    if (shrink) {
      for (int i = 1; i < m; i++) { int oi = ord[i]; for (int d = 0; d < n; d++) sx[oi][d] = COMB(xl[d], delta, sx[oi][d], xl[d]); f[oi] = objective(sx[oi], n); }
      nm_sort(m, f, ord);
    }
    converged = nm_converged(m, n, f);
  }
  nm_sort(m, f, ord);
  int ih = ord[m - 1];
  for (int d = 0; d < n; d++) {
    double s = 0;
    if (variant & 2) { for (int i = 0; i < n; i++) s += sx[ord[i]][d]; } else for (int i = 0; i < m; i++) if (i != ih) s += sx[i][d];
    xc[d] = s * rn;
  }
  double fcen = objective(xc, n);
  if (fcen < f[ord[0]]) for (int d = 0; d < n; d++) x[d] = xc[d]; else for (int d = 0; d < n; d++) x[d] = sx[ord[0]][d];
  if (iters) *iters = it;
  return converged;
}
static double urand(void) { return rand() / (RAND_MAX + 1.0); }
int main(void) {
  for (int n = 2; n <= 3; n++)
    for (int variant = 1; variant <= 4; variant++) {   // 4: reference arithmetic, contracted objective only
      srand(1);
      int cnt12 = 0, cnt10 = 0, trials = 20000, itdiff = 0; double worst = 0;
      for (int t = 0; t < trials; t++) {
        double x0[3], a[3], b[3];
        for (int d = 0; d < 3; d++) { T[d] = 4 * urand() - 2; x0[d] = T[d] + 2.0 * (urand() - 0.5); a[d] = b[d] = x0[d]; }
        int ia, ib;
        OBJV = 0; nm(n, a, 0, &ia);
        OBJV = (variant == 4); nm(n, b, variant == 4 ? 0 : variant, &ib);
        double e = 0; for (int d = 0; d < n; d++) e = fmax(e, fabs(a[d] - b[d]));
        cnt12 += e > 1e-12; cnt10 += e > 1e-10; itdiff += ia != ib; if (e > worst) worst = e;
      }
      printf("n=%d variant %d (%s): > 1e-12: %.2f%%  > 1e-10: %.2f%%  worst %.2e  searches with another iteration count %d\n", n, variant,
             variant == 1 ? "contracted vertex formulas" : variant == 2 ? "sorted-order centroid" : variant == 3 ? "both" : "contracted objective only",
             100.0 * cnt12 / trials, 100.0 * cnt10 / trials, worst, itdiff);
    }
  return 0;
}
