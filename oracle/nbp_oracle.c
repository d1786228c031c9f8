/*
 * nbp_oracle.c -- CPU restatement (plain C, serial) of the IncrementalInference.jl per-clique
 * nonparametric Chapman-Kolmogorov hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT THE PRODUCT.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  The product path is libnbp (HIP) and never links,
 * loads or calls anything in this directory.
 *
 * PARITY STATUS: "parity unpinned" at sample level.  Julia is not installed here and the
 * reference ships no golden vectors; half of the arithmetic lives in
 * un-vendored third-party packages, whose *published algorithms* are restated below:
 *    KernelDensityEstimate.jl 0.5.6 -- leave-one-out likelihood CV bandwidth, golden section
 *                                      (Ihler's KDE toolbox `ksize(.,'lcv')`), and the multiscale
 *                                      sequential Gibbs product sampler (Ihler, Sudderth, Freeman,
 *                                      Willsky, "Efficient Multiscale Sampling from Products of
 *                                      Gaussian Mixtures", NIPS 2003) = `prodAppxMSGibbsS`
 *    ApproxManifoldProducts 0.9     -- manikde!/manifoldProduct: the same on tangent coordinates
 *                                      at the identity, circular coordinates wrapped
 *    Optim 1                        -- NelderMead (Gao-Han adaptive parameters, AffineSimplexer),
 *                                      BFGS for 1-D decision variables
 *    Manifolds 0.10 / ManifoldsBase -- exp/log/compose/vee/hat for TranslationGroup(D),
 *                                      RealCircleGroup, SpecialEuclidean(2; Hybrid tangent repr.)
 * The oracle is pinned *in distribution* by the acceptance bands of the reference's own tests
 * (tests/test_oracle_reference_bands.py and tests/band_cases.py cite each test file:line), by the exact
 * Gaussian posterior of a linear chain (tests/exact_gaussian.py) and by closed-form moments of its
 * operations (tests/test_analytic_ops.py, tests/test_product_unbiased.py).
 *
 * All file:line citations are relative to the IncrementalInference.jl v0.35.6 source tree.
 *
 * Data layout (shared with libnbp): an arena of "slots"; slot s starts at
 * arena + s*(3N+8) doubles: coordinate d of particle n at [d*N+n], bandwidth at [3N+d].
 */
#define _GNU_SOURCE
#include "../include/nbp.h"
#include "../include/nbp_math.h" /* log / sincos / atan2 / wrap of the values that travel from op to op: one definition for
                                    the checker and the product (the header says why) */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PI 3.14159265358979323846
#define TWO_PI 6.28318530717958647692

/* RNG purposes (counter word 1).  Shared contract with the HIP kernels. */
#define PURP_MEAS 1
#define PURP_MIXLBL 2
#define PURP_HYPO 3
#define PURP_ENTROPY 4
#define PURP_KDESEL 5
#define PURP_KDENOISE 6
#define PURP_PGIBBS 9
#define PURP_PFINAL 10
#define PURP_PLEVEL 14 /* samplePoint! between the levels of the product sampler */
#define PURP_PINDEX 15 /* one block per (sample, pass, density): ua -> sampleIndices!, ub -> the first sweep's sampleIndex */
#define PURP_ANYN 11
#define PURP_OLDSEL 12
#define PURP_OLDNOISE 13
#define NBP_TAG 0x4E4250u

typedef struct {
  int64_t solves, nonconverged, nan_results, residual_evals, lcv_evals;
} orc_diag;

/* Two levels of parallelism in ONE OpenMP team (the CPU baseline of bench.py): the ops of a stage are tasks, and inside an op
 * the particles of a proposal, the samples of a product and the rows of a leave-one-out likelihood are taskloops -- a stage
 * near the root of a tree holds a handful of ops, and the threads that got none of them pick up pieces of the ones that run
 * (round 5: ops only, the baseline peaked at 16 threads -- the critical path of the tree; nested parallel regions, tried first
 * in round 6, were slower still: libgomp builds a new team for every inner region).  g_inner > 1: the inner level is on (off in
 * the test suite).  Every inner loop is over independent items and every sum is still taken serially in the same order: the
 * results are the serial ones bit for bit, whatever the thread count. */
static int g_inner = 1;
static orc_diag g_diag;               /* merged totals */
static __thread orc_diag t_diag;      /* per-thread counters, merged at the end of every op */

static void diag_merge(void) {
#pragma omp critical(orc_diag)
  {
    g_diag.solves += t_diag.solves;
    g_diag.nonconverged += t_diag.nonconverged;
    g_diag.nan_results += t_diag.nan_results;
    g_diag.residual_evals += t_diag.residual_evals;
    g_diag.lcv_evals += t_diag.lcv_evals;
  }
  memset(&t_diag, 0, sizeof(t_diag));
}

void orc_diag_read(orc_diag *out, int reset) {
  diag_merge();
  *out = g_diag;
  if (reset) memset(&g_diag, 0, sizeof(g_diag));
}

int64_t orc_slot_stride(int32_t N) { return 3 * (int64_t)N + 8; }

/* the shared elementary functions (include/nbp_math.h) as gcc compiles them: the CPU side of tests/test_gpu_device_math.py
 * (same numbering as nbp_math_eval, include/nbp.h) */
void orc_math_eval(int32_t fn, const double *a, const double *b, double *o0, double *o1, int64_t n) {
  for (int64_t i = 0; i < n; i++) {
    double r0 = 0.0, r1 = 0.0;
    switch (fn) {
    case 0: r0 = nbpm_log(a[i]); break;
    case 1: nbpm_sincos(a[i], &r0, &r1); break;
    case 2: r0 = nbpm_atan2(a[i], b[i]); break;
    case 3: r0 = nbpm_wrap_pi(a[i]); break;
    default: nbpm_box_muller(a[i], b[i], &r0, &r1); break;
    }
    o0[i] = r0;
    if (o1) o1[i] = r1;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* Philox4x32-10 (Salmon et al., SC'11) -- counter-based so that CPU and GPU draw identical     */
/* streams irrespective of the execution order.                                                */
/* ------------------------------------------------------------------------------------------ */
void orc_philox(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                uint32_t out[4]) {
  for (int r = 0; r < 10; r++) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static double u01_from(uint32_t lo, uint32_t hi) {
  uint64_t v = ((uint64_t)hi << 32) | lo;
  return ((double)(v >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

/* two uniforms in (0,1) for (seed, n, purpose, k) */
void orc_uniform_pair(uint64_t seed, uint32_t n, uint32_t purpose, uint32_t k, double *ua, double *ub) {
  uint32_t o[4];
  orc_philox(n, purpose, k, NBP_TAG, (uint32_t)seed, (uint32_t)(seed >> 32), o);
  *ua = u01_from(o[0], o[1]);
  *ub = u01_from(o[2], o[3]);
}

/* two standard normals (Box-Muller) */
void orc_normal_pair(uint64_t seed, uint32_t n, uint32_t purpose, uint32_t k, double *na, double *nb) {
  double ua, ub;
  orc_uniform_pair(seed, n, purpose, k, &ua, &ub);
  nbpm_box_muller(ua, ub, na, nb); /* sqrt(-2 log ua) (cos, sin)(2 pi ub), include/nbp_math.h */
}

/* particle count of a belief: a slot holds up to N points, slot[3N + 6] = the count (0 = N) */
static int slot_count(const double *s, int N) {
  const double c = s[3 * N + 6];
  return (c > 0.0 && c < (double)N) ? (int)c : N;
}
/* _getindex_anyn(vec, n) = vec[n <= len ? n : rand(1:len)]  (NumericalCalculations.jl:377-381).  The reference
 * draws again at every evaluation of the residual (the objective of that particle's search is then not a function);
 * the restatement draws once per (op, particle, operand). */
static int anyn_index(int n, int cnt, uint64_t seed, int operand) {
  if (n < cnt) return n;
  double ua, ub;
  orc_uniform_pair(seed, n, PURP_ANYN, (uint32_t)operand, &ua, &ub);
  int i = (int)(ua * cnt);
  return i < cnt ? i : cnt - 1;
}

/* ------------------------------------------------------------------------------------------ */
/* Manifolds in tangent coordinates at the identity                                            */
/* ------------------------------------------------------------------------------------------ */
static int mani_dim(int m) {
  switch (m) {
  case NBP_EUCLID1: return 1;
  case NBP_EUCLID2: return 2;
  case NBP_EUCLID3: return 3;
  case NBP_CIRCULAR: return 1;
  case NBP_SE2: return 3;
  }
  return 0;
}
static int mani_P(int m) { return m == NBP_SE2 ? 6 : mani_dim(m); }
static int is_circ(int m, int d) { return (m == NBP_CIRCULAR && d == 0) || (m == NBP_SE2 && d == 2); }

/* Manifolds.sym_rem: wrap to [-pi, pi) */
double orc_wrap(double a) { return nbpm_wrap_pi(a); }

int32_t orc_manifold_dim(int32_t m) { return mani_dim(m); }
int32_t orc_manifold_P(int32_t m) { return mani_P(m); }

/* host AoS points (N x P) -> slot coordinates (vee(M, e, log(M, e, p)), what AMP.manikde! stores) */
void orc_belief_write(double *arena, int32_t N, int32_t slot, int32_t manifold, const double *pts, int32_t n_pts, const double *bw);
void orc_slot_write(double *arena, int32_t N, int32_t slot, int32_t manifold, const double *pts, const double *bw) {
  orc_belief_write(arena, N, slot, manifold, pts, N, bw);
}
/* n_pts < N: the belief keeps its count; n_pts > N: the first N points (GraphProductOperations.jl:44) */
void orc_belief_write(double *arena, int32_t N, int32_t slot, int32_t manifold, const double *pts, int32_t n_pts, const double *bw) {
  double *s = arena + orc_slot_stride(N) * slot;
  int D = mani_dim(manifold), P = mani_P(manifold);
  const int cnt = n_pts < N ? n_pts : N;
  memset(s, 0, sizeof(double) * (3 * N + 8));
  s[3 * N + 6] = cnt < N ? (double)cnt : 0.0;
  for (int n = 0; n < cnt; n++) {
    const double *p = pts + (size_t)n * P;
    if (manifold == NBP_SE2) {
      s[0 * N + n] = p[0];
      s[1 * N + n] = p[1];
      s[2 * N + n] = nbpm_atan2(p[3], p[2]); /* R = [c -s; s c] column-major: R11,R21,R12,R22 */
    } else if (manifold == NBP_CIRCULAR) {
      s[n] = orc_wrap(p[0]);
    } else {
      for (int d = 0; d < D; d++) s[d * N + n] = p[d];
    }
  }
  for (int d = 0; d < 3; d++) s[3 * N + d] = (bw && d < D) ? bw[d] : 0.0;
  for (int d = 0; d < 3; d++) s[3 * N + 3 + d] = 0.0; /* infoPerCoord of a fresh VariableNodeData */
}
int32_t orc_slot_count(const double *arena, int32_t N, int32_t slot) { return slot_count(arena + orc_slot_stride(N) * slot, N); }
/* infoPerCoord of a slot (TreeBelief.infoPerCoord, BeliefTypes.jl:47-57) */
void orc_slot_ipc_write(double *arena, int32_t N, int32_t slot, int32_t manifold, const double *ipc) {
  double *s = arena + orc_slot_stride(N) * slot;
  for (int d = 0; d < 3; d++) s[3 * N + 3 + d] = d < mani_dim(manifold) ? ipc[d] : 0.0;
}
void orc_slot_ipc_read(const double *arena, int32_t N, int32_t slot, int32_t manifold, double *ipc) {
  const double *s = arena + orc_slot_stride(N) * slot;
  for (int d = 0; d < mani_dim(manifold); d++) ipc[d] = s[3 * N + 3 + d];
}

void orc_slot_read(const double *arena, int32_t N, int32_t slot, int32_t manifold, double *pts, double *bw) {
  const double *s = arena + orc_slot_stride(N) * slot;
  int D = mani_dim(manifold), P = mani_P(manifold);
  for (int n = 0; n < N; n++) {
    double *p = pts + (size_t)n * P;
    if (manifold == NBP_SE2) {
      double th = s[2 * N + n];
      p[0] = s[n]; p[1] = s[N + n];
      double sn, cs;
      nbpm_sincos(th, &sn, &cs);
      p[2] = cs; p[3] = sn; p[4] = -sn; p[5] = cs;
    } else {
      for (int d = 0; d < D; d++) p[d] = s[d * N + n];
    }
  }
  if (bw) for (int d = 0; d < D; d++) bw[d] = s[3 * N + d];
}

/* ------------------------------------------------------------------------------------------ */
/* Sums in the order the kernels take them.                                                   */
/* A belief's spread statistics (calcStdBasicSpread, the means of calcVariableDistanceExpectedFractional) go INTO every   */
/* per-particle search: an ulp of difference there comes out of a 3-D Nelder-Mead search at 1e-9.  A sum of  */
/* N doubles has no canonical value -- Manifolds.jl's running mean, a pairwise tree and a serial loop all differ in the    */
/* last bit -- so ONE order is fixed for both sides, the one the device's reductions have: the 64 values of a chunk by the  */
/* butterfly of a wave (pairs 32 apart, then 16, ... 1), chunks added one after the other (block_sum / wave_chunks_sum in   */
/* csrc/nbp_device.h); prefix sums in the order of the wave scan (wave_inclusive_scan).  The reference's own definition,     */
/* the running geodesic mean, stays below as orc_mean_geodesic_walk and tests/test_spread_statistics.py holds the two to     */
/* 1e-13 of each other.                                                                                                       */
/* ------------------------------------------------------------------------------------------ */
static double chunked_tree_sum(const double *v, int n) {
  double tot = 0.0;
  for (int base = 0; base < n; base += 64) {
    double t[64];
    for (int i = 0; i < 64; i++) t[i] = (base + i < n) ? v[base + i] : 0.0; /* idle lanes hold zeros */
    for (int o = 32; o > 0; o >>= 1)
      for (int i = 0; i < o; i++) t[i] = t[i] + t[i + o];
    tot = (base == 0) ? t[0] : tot + t[0];
  }
  return tot;
}
/* inclusive prefix sums over 64 lanes in the association wave_inclusive_scan() produces them in: three row shifts of the
 * input, shifts by 4 and 8 of the partial sums inside each row of 16, then the row totals broadcast to the rows behind */
static void wave_scan64(const double *x, double *s) {
  double a[64];
  for (int i = 0; i < 64; i++) {
    const int r = i & 15;
    double v = x[i] + (r >= 1 ? x[i - 1] : 0.0);
    v = v + (r >= 2 ? x[i - 2] : 0.0);
    v = v + (r >= 3 ? x[i - 3] : 0.0);
    a[i] = v;
  }
  for (int i = 0; i < 64; i++) s[i] = a[i] + ((i & 15) >= 4 ? a[i - 4] : 0.0);
  for (int i = 0; i < 64; i++) a[i] = s[i] + ((i & 15) >= 8 ? s[i - 8] : 0.0);
  for (int i = 0; i < 64; i++) s[i] = a[i] + (((i >> 4) & 1) ? a[(i & ~15) - 1] : 0.0); /* row_bcast:15 into rows 1 and 3 */
  for (int i = 0; i < 64; i++) a[i] = s[i] + ((i >> 5) ? s[31] : 0.0);                   /* row_bcast:31 into rows 2 and 3 */
  for (int i = 0; i < 64; i++) s[i] = a[i];
}

/* The reference's definition: mean(M, pts, GeodesicInterpolation()), the running geodesic interpolation of Manifolds.jl
 * (services/VariableStatistics.jl:30), point by point. */
double orc_mean_geodesic_walk(const double *x, int cnt, int circ) {
  double m = x[0];
  for (int i = 1; i < cnt; i++) {
    double dl = x[i] - m;
    if (circ) dl = orc_wrap(dl);
    m = fma(dl, 1.0 / (double)(i + 1), m);
    if (circ) m = orc_wrap(m);
  }
  return m;
}

/* The same mean of a circular coordinate, the way the kernels reach it (mean_geodesic_coord, csrc/nbp_device.h):
 *  (1) every point within an arc shorter than ~pi of the first: no step of the walk wraps, the recurrence is the arithmetic
 *      mean of the offsets from the first point;
 *  (2) otherwise the walk's lifts X_i = x_i + 2 pi k_i (k_i puts X_i within pi of the mean of the points before it) are
 *      iterated to their fixed point with prefix sums -- the head of 64 alone first, then every chunk -- and the mean is the
 *      sum of the lifted points over the count;
 *  (3) lifts that have not settled within the kernels' sweep budget: the walk itself. */
static double mean_geodesic_circ(const double *x, int cnt) {
  const double x0 = x[0];
  double *d = (double *)malloc(sizeof(double) * (cnt + 64) * 3), *k = d + cnt + 64, *X = k + cnt + 64;
  double hi = -INFINITY, lo = -INFINITY, res;
  for (int i = 0; i < cnt; i++) {
    d[i] = orc_wrap(x[i] - x0);
    hi = fmax(hi, d[i]);
    lo = fmax(lo, -d[i]);
  }
  if (hi + lo < 3.0) {
    const double mo = chunked_tree_sum(d, cnt) / (double)cnt;
    res = orc_wrap(x0 + mo);
    free(d);
    return res;
  }
  const int nch = (cnt + 63) / 64;
  const double r2pi = 1.0 / TWO_PI;
  for (int i = 0; i < nch * 64; i++) k[i] = 0.0;
  /* head: the first 64 points alone, up to 24 sweeps */
  for (int sweep = 0; sweep < 24; sweep++) {
    double Xh[64], inc[64];
    int moved = 0;
    for (int i = 0; i < 64; i++) Xh[i] = (i < cnt) ? fma(TWO_PI, k[i], x[i]) : 0.0;
    wave_scan64(Xh, inc);
    for (int i = 1; i < 64 && i < cnt; i++) {
      const double before = inc[i] - Xh[i];
      const double kn = k[i] + rint((before / (double)i - Xh[i]) * r2pi);
      if (kn != k[i]) moved = 1;
      k[i] = kn;
    }
    if (!moved) break;
  }
  /* every chunk: up to 32 sweeps; a sweep after which nothing moved holds the sums of the fixed point */
  int fixed = 0, any_prev = 1;
  double tot = 0.0;
  for (int sweep = 0; sweep < 32; sweep++) {
    double wt[16], inc[16 * 64];
    for (int i = 0; i < nch * 64; i++) X[i] = (i < cnt) ? fma(TWO_PI, k[i], x[i]) : 0.0;
    for (int w = 0; w < nch; w++) { wave_scan64(X + 64 * w, inc + 64 * w); wt[w] = inc[64 * w + 63]; }
    tot = 0.0;
    for (int w = 0; w < nch; w++) tot += wt[w];
    if (sweep > 0 && !any_prev) { fixed = 1; break; }
    int any = 0;
    double off = 0.0;
    for (int w = 0; w < nch; w++) {
      for (int l = 0; l < 64; l++) {
        const int i = 64 * w + l;
        if (i < 1 || i >= cnt) continue;
        const double before = off + inc[i] - X[i];
        const double kn = k[i] + rint((before / (double)i - X[i]) * r2pi);
        if (kn != k[i]) any = 1;
        k[i] = kn;
      }
      off += wt[w];
    }
    any_prev = any;
  }
  res = fixed ? orc_wrap(tot / (double)cnt) : orc_mean_geodesic_walk(x, cnt, 1);
  free(d);
  return res;
}

/* mean(M, pts, GeodesicInterpolation()), services/VariableStatistics.jl:30.  N = stride of the coordinate arrays,
 * cnt = points held.  Euclidean coordinates: the arithmetic mean (what the running mean is, up to rounding). */
static void mean_geodesic_n(int manifold, const double *x, int N, int cnt, double *mu) {
  int D = mani_dim(manifold);
  for (int d = 0; d < D; d++)
    mu[d] = is_circ(manifold, d) ? mean_geodesic_circ(x + d * N, cnt) : chunked_tree_sum(x + d * N, cnt) / (double)cnt;
}
static void mean_geodesic(int manifold, const double *x, int N, double *mu) { mean_geodesic_n(manifold, x, N, N, mu); }

/* default mean(M, pts): arithmetic on Euclidean coordinates, extrinsic on the circle */
static void mean_default_n(int manifold, const double *x, int N, int cnt, double *mu) {
  int D = mani_dim(manifold);
  for (int d = 0; d < D; d++) {
    if (is_circ(manifold, d)) {
      double *sn = (double *)malloc(sizeof(double) * 2 * cnt), *cs = sn + cnt;
      for (int i = 0; i < cnt; i++) nbpm_sincos(x[d * N + i], &sn[i], &cs[i]);
      const double ss = chunked_tree_sum(sn, cnt), sc = chunked_tree_sum(cs, cnt);
      mu[d] = nbpm_atan2(ss, sc);
      free(sn);
    } else
      mu[d] = chunked_tree_sum(x + d * N, cnt) / (double)cnt;
  }
}
static void mean_default(int manifold, const double *x, int N, double *mu) { mean_default_n(manifold, x, N, N, mu); }

/* calcStdBasicSpread, services/VariableStatistics.jl:22-36: sigma = sqrt(sum d(mu,x_i)^2/(N-1)),
 * 1.0 if sigma < 1e-10.  Rotation part of SE(2) carries the Frobenius metric (||skew(w)||^2 = 2w^2). */
double orc_std_basic_spread(int manifold, const double *x, int N) {
  double mu[3];
  int D = mani_dim(manifold);
  mean_geodesic(manifold, x, N, mu);
  double *acc = (double *)malloc(sizeof(double) * N);
  for (int i = 0; i < N; i++) {
    double a = 0;
    for (int d = 0; d < D; d++) {
      double dl = x[d * N + i] - mu[d];
      if (is_circ(manifold, d)) {
        dl = orc_wrap(dl);
        a += (manifold == NBP_SE2 ? 2.0 : 1.0) * dl * dl;
      } else
        a += dl * dl;
    }
    acc[i] = a;
  }
  double sg = sqrt(chunked_tree_sum(acc, N) / (double)(N - 1));
  free(acc);
  return (1e-10 < sg) ? sg : 1.0;
}
/* exposed for tests/test_spread_statistics.py */
double orc_mean_geodesic_device_order(const double *x, int32_t cnt, int32_t circ) {
  return circ ? mean_geodesic_circ(x, cnt) : chunked_tree_sum(x, cnt) / (double)cnt;
}

/* ------------------------------------------------------------------------------------------ */
/* Residual functors.  a = first active variable, b = second, z = measurement      */
/* tangent coordinates.  Returns sum(r.^2) (CalcFactorNormSq, NumericalCalculations.jl:386-396) */
/* ------------------------------------------------------------------------------------------ */
static int factor_zdim(int kind, int manifold) {
  switch (kind) {
  case NBP_F_LINREL: return mani_dim(manifold);
  case NBP_F_CIRCULAR: return 1;
  case NBP_F_SE2: return 3;
  case NBP_F_EUCLIDDIST: return 1;
  default: return mani_dim(manifold); /* priors sample points */
  }
}

int32_t orc_residual(int32_t kind, int32_t manifold, const double *z, const double *a, const double *b, double *r) {
  int D = mani_dim(manifold);
  t_diag.residual_evals++;
  switch (kind) {
  case NBP_F_LINREL: /* Factors/LinearRelative.jl:42-49 */
    for (int d = 0; d < D; d++) r[d] = z[d] - (b[d] - a[d]);
    return D;
  case NBP_F_CIRCULAR: { /* Factors/Circular.jl:24-28 -> GenericFunctions.jl:47-52 */
    double qhat = a[0] + z[0]; /* exp(M, p, X) */
    r[0] = orc_wrap(qhat - b[0]); /* vee(log(M, q, qhat)) */
    return 1;
  }
  case NBP_F_SE2: { /* Factors/GenericFunctions.jl:39-44 */
    double c, s;
    nbpm_sincos(a[2], &s, &c);
    double qx = a[0] + c * z[0] - s * z[1]; /* compose(p, exp(e, X)) */
    double qy = a[1] + s * z[0] + c * z[1];
    double qt = a[2] + z[2];
    r[0] = qx - b[0];
    r[1] = qy - b[1];
    r[2] = orc_wrap(qt - b[2]);
    return 3;
  }
  case NBP_F_EUCLIDDIST: { /* Factors/EuclidDistance.jl:20 */
    double acc = 0;
    for (int d = 0; d < D; d++) acc += (b[d] - a[d]) * (b[d] - a[d]);
    r[0] = z[0] - sqrt(acc);
    return 1;
  }
  }
  return 0;
}

typedef struct {
  int kind, manifold, D, solve_b; /* solve_b == 2: the measurement is the decision variable (deconv) */
  double z[3], other[3];          /* deconv: z = first variable's point, other = second variable's point */
  int rmask;                      /* 0: every residual component; else bit k = component k counts (a partial relative
                                     factor whose residual "deals with the partial" itself, NumericalCalculations.jl:429) */
} objective_t;

static double objective(const objective_t *o, const double *x) {
  double r[3];
  int nr = o->solve_b == 2 ? orc_residual(o->kind, o->manifold, x, o->z, o->other, r)
           : o->solve_b    ? orc_residual(o->kind, o->manifold, o->z, o->other, x, r)
                           : orc_residual(o->kind, o->manifold, o->z, x, o->other, r);
  double acc = 0;
  for (int i = 0; i < nr; i++)
    if (!o->rmask || ((o->rmask >> i) & 1)) acc += r[i] * r[i];
  return acc;
}

/* ------------------------------------------------------------------------------------------ */
/* Optim.NelderMead restated: AdaptiveParameters (Gao & Han 2012), AffineSimplexer(a=0.025,     */
/* b=0.5), g_tol = 1e-8 on nmobjective (see nm_converged), iterations = 1000.                   */
/* Call site: NumericalCalculations.jl:108,122-126.                                            */
/* ------------------------------------------------------------------------------------------ */
static void nm_sort(int m, const double *f, int *ord) {
  for (int i = 0; i < m; i++) ord[i] = i;
  for (int i = 1; i < m; i++) { /* stable insertion sort */
    int k = ord[i], j = i - 1;
    while (j >= 0 && f[ord[j]] > f[k]) { ord[j + 1] = ord[j]; j--; }
    ord[j + 1] = k;
  }
}
/* Optim's convergence measure, written the way Optim writes it (multivariate/solvers/zeroth_order/nelder_mead.jl,
   Optim 1.x):
       nmobjective(y::Vector, m::Integer, n::Integer) = sqrt(var(y) * (m / n))
   called as  nmobjective(f_simplex, n, m)  with n = length(x), m = n + 1 vertices -- i.e. the corrected sample
   variance (Statistics.var divides by m - 1 = n) times n / (n + 1): the population standard deviation of the
   vertex values, sqrt(sum((y - mean(y))^2) / (n + 1)), compared with g_tol = 1e-8 (Optim.Options default;
   the reference passes no g_tol, NumericalCalculations.jl:122-126).
   Plain divisions and the square root on purpose: this file follows the reference's arithmetic, not the kernels'
   (the HIP kernel tests the same predicate as sum <= g_tol^2 * (n + 1); the two can only disagree when the sum is
   within rounding of the threshold, which the parity tolerances absorb). */
static int nm_converged(int m, int n, const double *f) {
  double a = 0;
  for (int i = 0; i < m; i++) a += f[i];
  a = a / m;
  double v = 0;
  for (int i = 0; i < m; i++) v += (f[i] - a) * (f[i] - a);
  const double var = v / (m - 1);                    /* Statistics.var, corrected */
  return sqrt(var * ((double)n / (double)m)) <= 1e-8; /* nmobjective(f_simplex, n, m) <= g_abstol */
}

int orc_nelder_mead(const objective_t *o, int n, double *x /* in: start, out: minimizer */, int *iters) {
  const int m = n + 1;
  const double alpha = 1.0, beta = 1.0 + 2.0 / n, gamma = 0.75 - 1.0 / (2.0 * n), delta = 1.0 - 1.0 / n;
  const double rn = 1.0 / n;
  double sx[4][3], f[4], xc[3], xr[3], xcache[3], xl[3];
  int ord[4];
  for (int i = 0; i < m; i++)
    for (int d = 0; d < n; d++) sx[i][d] = x[d];
  for (int j = 0; j < n; j++) sx[j + 1][j] = (1.0 + 0.5) * sx[j + 1][j] + 0.025;
  for (int i = 0; i < m; i++) f[i] = objective(o, sx[i]);
  nm_sort(m, f, ord);
  int converged = nm_converged(m, n, f);
  int it = 0;
  while (!converged && it < 1000) {
    it++;
    int shrink = 0;
    int ih = ord[m - 1];
    for (int d = 0; d < n; d++) {
      double s = 0;
      for (int i = 0; i < m; i++) if (i != ih) s += sx[i][d];
      xc[d] = s * rn; /* Optim's centroid!: sum over the other vertices, then rmul!(c, T(1)/n) */
      xl[d] = sx[ord[0]][d];
    }
    double f_lowest = f[ord[0]], f_second = f[ord[n - 1]], f_highest = f[ih];
    for (int d = 0; d < n; d++) xr[d] = xc[d] + alpha * (xc[d] - sx[ih][d]);
    double f_reflect = objective(o, xr);
    if (f_reflect < f_lowest) {
      for (int d = 0; d < n; d++) xcache[d] = xc[d] + beta * (xr[d] - xc[d]);
      double f_expand = objective(o, xcache);
      if (f_expand < f_reflect) {
        for (int d = 0; d < n; d++) sx[ih][d] = xcache[d];
        f[ih] = f_expand;
      } else {
        for (int d = 0; d < n; d++) sx[ih][d] = xr[d];
        f[ih] = f_reflect;
      }
      for (int i = m - 1; i >= 1; i--) ord[i] = ord[i - 1];
      ord[0] = ih;
    } else if (f_reflect < f_second) {
      for (int d = 0; d < n; d++) sx[ih][d] = xr[d];
      f[ih] = f_reflect;
      nm_sort(m, f, ord);
    } else {
      if (f_reflect < f_highest) { /* outside contraction */
        for (int d = 0; d < n; d++) xcache[d] = xc[d] + gamma * (xr[d] - xc[d]);
        double fc = objective(o, xcache);
        if (fc < f_reflect) {
          for (int d = 0; d < n; d++) sx[ih][d] = xcache[d];
          f[ih] = fc;
          nm_sort(m, f, ord);
        } else
          shrink = 1;
      } else { /* inside contraction */
        for (int d = 0; d < n; d++) xcache[d] = xc[d] - gamma * (xr[d] - xc[d]);
        double fc = objective(o, xcache);
        if (fc < f_highest) {
          for (int d = 0; d < n; d++) sx[ih][d] = xcache[d];
          f[ih] = fc;
          nm_sort(m, f, ord);
        } else
          shrink = 1;
      }
    }
    if (shrink) {
      for (int i = 1; i < m; i++) {
        int oi = ord[i];
        for (int d = 0; d < n; d++) sx[oi][d] = xl[d] + delta * (sx[oi][d] - xl[d]);
        f[oi] = objective(o, sx[oi]);
      }
      nm_sort(m, f, ord);
    }
    converged = nm_converged(m, n, f);
  }
  /* after_while!: better of best vertex and centroid of the n best */
  nm_sort(m, f, ord);
  int ih = ord[m - 1];
  for (int d = 0; d < n; d++) {
    double s = 0;
    for (int i = 0; i < m; i++) if (i != ih) s += sx[i][d];
    xc[d] = s * rn; /* centroid!, as above */
  }
  double fcen = objective(o, xc);
  if (fcen < f[ord[0]])
    for (int d = 0; d < n; d++) x[d] = xc[d];
  else
    for (int d = 0; d < n; d++) x[d] = sx[ord[0]][d];
  if (iters) *iters = it;
  return converged;
}

/* 1-D decision variable: Optim.BFGS with central finite differences (islen1 branch,
 * NumericalCalculations.jl:108).  Restated with an Armijo/quadratic-interpolation line search in
 * place of HagerZhang (documented deviation; identical fixed point for the residual set). */
static double fd_grad1(const objective_t *o, double x) {
  double h = 6.0554544523933395e-06 * fmax(1.0, fabs(x)); /* cbrt(eps(Float64)) */
  double xp = x + h, xm = x - h;
  return (objective(o, &xp) - objective(o, &xm)) / (2.0 * h);
}

int orc_bfgs_1d(const objective_t *o, double *x, int *iters) {
  double xc = *x, fx = objective(o, &xc), g = fd_grad1(o, xc), H = 1.0;
  int converged = 0, it = 0;
  for (it = 0; it < 1000; it++) {
    if (fabs(g) <= 1e-8) { converged = 1; break; }
    double s = -H * g;
    if (s * g >= 0) { H = 1.0; s = -g; }
    double al = 1.0, dphi0 = g * s, xn = xc, fn = fx;
    int ok = 0;
    for (int ls = 0; ls < 50; ls++) {
      xn = xc + al * s;
      fn = objective(o, &xn);
      if (fn <= fx + 1e-4 * al * dphi0) { ok = 1; break; }
      double aq = -dphi0 * al * al / (2.0 * (fn - fx - dphi0 * al));
      if (!(aq >= 0.1 * al)) aq = 0.1 * al;
      if (aq > 0.5 * al) aq = 0.5 * al;
      al = aq;
    }
    if (!ok) break;
    double gn = fd_grad1(o, xn), dx = xn - xc, dg = gn - g;
    if (dx == 0.0) { converged = fabs(gn) <= 1e-8; break; }
    if (dx * dg > 0) H = dx / dg;
    xc = xn; fx = fn; g = gn;
  }
  *x = xc;
  if (iters) *iters = it;
  return converged;
}

/* Optim.BFGS, n-dimensional: the solver of every partial factor (`alg = islen1 ? BFGS : NelderMead`, islen1 true when
 * ccw.partial, NumericalCalculations.jl:108,424).  Initial inverse Hessian I, central finite differences, g_tol = 1e-8 on
 * the max-norm of the gradient; the line search is the Armijo / quadratic-interpolation one of orc_bfgs_1d (documented
 * deviation from HagerZhang), to which this reduces for n = 1. */
static void fd_grad_nd(const objective_t *o, int n, const double *x, double *g) {
  for (int k = 0; k < n; k++) {
    double h = 6.0554544523933395e-06 * fmax(1.0, fabs(x[k])), xp[3], xm[3];
    for (int q = 0; q < n; q++) { xp[q] = x[q]; xm[q] = x[q]; }
    xp[k] += h; xm[k] -= h;
    g[k] = (objective(o, xp) - objective(o, xm)) / (2.0 * h);
  }
}
int orc_bfgs_nd(const objective_t *o, int n, double *x, int *iters) {
  double xc[3], g[3], H[3][3], fx;
  for (int k = 0; k < n; k++) { xc[k] = x[k]; for (int q = 0; q < n; q++) H[k][q] = k == q ? 1.0 : 0.0; }
  fx = objective(o, xc);
  fd_grad_nd(o, n, xc, g);
  int converged = 0, it = 0;
  for (it = 0; it < 1000; it++) {
    double gmax = 0;
    for (int k = 0; k < n; k++) gmax = fmax(gmax, fabs(g[k]));
    if (gmax <= 1e-8) { converged = 1; break; }
    double s[3], dphi0 = 0;
    for (int k = 0; k < n; k++) {
      double a = 0;
      for (int q = 0; q < n; q++) a -= H[k][q] * g[q];
      s[k] = a;
      dphi0 += g[k] * a;
    }
    if (dphi0 >= 0) {
      dphi0 = 0;
      for (int k = 0; k < n; k++) {
        for (int q = 0; q < n; q++) H[k][q] = k == q ? 1.0 : 0.0;
        s[k] = -g[k];
        dphi0 -= g[k] * g[k];
      }
    }
    double al = 1.0, fn = fx, xn[3];
    int ok = 0;
    for (int ls = 0; ls < 50; ls++) {
      for (int k = 0; k < n; k++) xn[k] = xc[k] + al * s[k];
      fn = objective(o, xn);
      if (fn <= fx + 1e-4 * al * dphi0) { ok = 1; break; }
      double aq = -dphi0 * al * al / (2.0 * (fn - fx - dphi0 * al));
      if (!(aq >= 0.1 * al)) aq = 0.1 * al;
      if (aq > 0.5 * al) aq = 0.5 * al;
      al = aq;
    }
    if (!ok) break;
    double gn[3], dx[3], dg[3], sy = 0, gnmax = 0;
    int moved = 0;
    fd_grad_nd(o, n, xn, gn);
    for (int k = 0; k < n; k++) {
      dx[k] = xn[k] - xc[k]; dg[k] = gn[k] - g[k];
      sy += dx[k] * dg[k];
      moved |= dx[k] != 0.0;
      gnmax = fmax(gnmax, fabs(gn[k]));
    }
    if (!moved) { converged = gnmax <= 1e-8; break; }
    if (sy > 0) { /* H <- (I - rho dx dg') H (I - rho dg dx') + rho dx dx' */
      double rho = 1.0 / sy, Hy[3], yHy = 0;
      for (int k = 0; k < n; k++) { double a = 0; for (int q = 0; q < n; q++) a += H[k][q] * dg[q]; Hy[k] = a; }
      for (int k = 0; k < n; k++) yHy += dg[k] * Hy[k];
      for (int k = 0; k < n; k++)
        for (int q = 0; q < n; q++) H[k][q] += rho * ((1.0 + rho * yHy) * dx[k] * dx[q] - Hy[k] * dx[q] - dx[k] * Hy[q]);
    }
    for (int k = 0; k < n; k++) { xc[k] = xn[k]; g[k] = gn[k]; }
    fx = fn;
  }
  for (int k = 0; k < n; k++) x[k] = xc[k];
  if (iters) *iters = it;
  return converged;
}

/* _solveCCWNumeric! for one particle, NumericalCalculations.jl:413-452 + :90-133 */
static void solve_particle(int kind, int manifold, const double *z, const double *other, int solve_b, double *x) {
  objective_t o;
  o.kind = kind; o.manifold = manifold; o.D = mani_dim(manifold); o.solve_b = solve_b; o.rmask = 0;
  for (int i = 0; i < 3; i++) { o.z[i] = z[i]; o.other[i] = other[i]; }
  double xc[3];
  int D = o.D;
  for (int d = 0; d < D; d++) xc[d] = x[d]; /* X0c = vee(M, e, log(M, e, u0)) */
  int conv;
  t_diag.solves++;
  if (D == 1) conv = orc_bfgs_1d(&o, xc, 0); /* islen1 -> BFGS */
  else conv = orc_nelder_mead(&o, D, xc, 0);
  if (!conv) t_diag.nonconverged++; /* @warn only; the result is still used (:128-131) */
  for (int d = 0; d < D; d++)
    if (isnan(xc[d])) { t_diag.nan_results++; return; }
  for (int d = 0; d < D; d++) x[d] = is_circ(manifold, d) ? orc_wrap(xc[d]) : xc[d]; /* exp(M, e, hat(..)) */
}

/* exposed for unit tests */
int32_t orc_solve_particle(int32_t kind, int32_t manifold, const double *z, const double *other, int32_t solve_b, double *x) {
  solve_particle(kind, manifold, z, other, solve_b, x);
  return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* _prepareHypoRecipe!, services/ExplicitDiscreteMarginalizations.jl:142-232 (multihypo) and   */
/* :234-289 (none / nullhypo).  INTEGER, exact-match item.  Indices are 1-based like the        */
/* reference; hypothesis 0 is the null hypothesis.                                              */
/* Output, for group g = 0..ngroups-1:  hypo[g] = pidx, nact[g], act[g][..] = iterah,           */
/* and `empty[g]` = 1 when the reference forces allelements[g] = Int[].                          */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  int ngroups;
  int hypo[NBP_MAXV + 1];
  int nact[NBP_MAXV + 1];
  int act[NBP_MAXV + 1][NBP_MAXV];
  int empty[NBP_MAXV + 1];
  int ncertain;
  int certain[NBP_MAXV];
  int sf_uncertain;
  double cat_p[NBP_MAXV + 1]; /* categorical over groups used to draw mhidx */
  int cat_first;              /* hypothesis index of cat_p[0] */
  int ncat;
} recipe_t;

static int in_list(const int *l, int n, int v) {
  for (int i = 0; i < n; i++) if (l[i] == v) return 1;
  return 0;
}
static void sorted_union(const int *a, int na, int v, int *out, int *nout) {
  int tmp[NBP_MAXV + 1], n = 0;
  for (int i = 0; i < na; i++) tmp[n++] = a[i];
  if (!in_list(a, na, v)) tmp[n++] = v;
  for (int i = 1; i < n; i++) { int k = tmp[i], j = i - 1; while (j >= 0 && tmp[j] > k) { tmp[j + 1] = tmp[j]; j--; } tmp[j + 1] = k; }
  for (int i = 0; i < n; i++) out[i] = tmp[i];
  *nout = n;
}

void orc_build_recipe(int has_multihypo, const double *mhp, int nvars, int sfidx1, double nullhypo, recipe_t *R) {
  memset(R, 0, sizeof(*R));
  if (!has_multihypo) {
    /* :234-289 */
    R->ncertain = nvars;
    for (int i = 0; i < nvars; i++) R->certain[i] = i + 1;
    R->ngroups = nvars + 1;
    for (int g = 0; g <= nvars; g++) {
      R->hypo[g] = g;
      if (g == 0) { R->nact[g] = 1; R->act[g][0] = sfidx1; }
      else if (g == 1) { R->nact[g] = nvars; for (int i = 0; i < nvars; i++) R->act[g][i] = i + 1; }
      else { R->nact[g] = 0; R->empty[g] = 1; }
    }
    R->ncat = 2; R->cat_first = 0;
    R->cat_p[0] = nullhypo; R->cat_p[1] = 1.0 - nullhypo;
    return;
  }
  /* getHypothesesVectors :17-24 */
  int unc[NBP_MAXV], nunc = 0;
  for (int i = 0; i < nvars; i++) {
    if (mhp[i] == 0.0) R->certain[R->ncertain++] = i + 1;
    else if (0.0 < mhp[i]) unc[nunc++] = i + 1;
  }
  int sfunc = in_list(unc, nunc, sfidx1), sfincer = in_list(R->certain, R->ncertain, sfidx1);
  R->sf_uncertain = sfunc;
  /* :161-172 -- select only hypotheses that can be used: when fewer than nvars-1 variables are
     initialised, the uninitialised ones (except the solve-for variable) get probability 0 in the
     draw of mhidx; certainidx / uncertnidx keep following the factor's own p.  The isinit flags
     travel in has_multihypo: bit 7 = flags present, bit 8+k = variable k initialised. */
  double mhs[NBP_MAXV];
  for (int i = 0; i < nvars; i++) mhs[i] = mhp[i];
  if (has_multihypo & 0x80) {
    int ninit = 0;
    for (int i = 0; i < nvars; i++) ninit += (has_multihypo >> (8 + i)) & 1;
    if (ninit < nvars - 1) {
      double tot = 0;
      for (int i = 0; i < nvars; i++) {
        if (!((has_multihypo >> (8 + i)) & 1) && i + 1 != sfidx1) mhs[i] = 0.0;
        tot += mhs[i];
      }
      for (int i = 0; i < nvars; i++) mhs[i] /= tot;
    }
  }
  mhp = mhs;
  double p[NBP_MAXV + 1];
  int np = 0, pidx0;
  if (sfunc) { /* :176-183 prepend the bad-init null hypothesis */
    double nhw = (double)(nunc + 1), tot = 0;
    p[np++] = 1.0 / nhw;
    for (int i = 0; i < nvars; i++) p[np++] = (double)nunc / nhw * mhp[i];
    for (int i = 0; i < np; i++) tot += p[i];
    for (int i = 0; i < np; i++) p[i] /= tot;
    pidx0 = 0;
  } else {
    for (int i = 0; i < nvars; i++) p[np++] = mhp[i];
    pidx0 = 1;
  }
  R->ncat = np; R->cat_first = pidx0;
  for (int i = 0; i < np; i++) R->cat_p[i] = p[i];
  R->ngroups = np;
  for (int g = 0; g < np; g++) { /* :195-222 */
    int pidx = pidx0 + g;
    int pidxincer = in_list(R->certain, R->ncertain, pidx);
    R->hypo[g] = pidx;
    if (!pidxincer && sfincer && pidx != 0) {
      sorted_union(R->certain, R->ncertain, pidx, R->act[g], &R->nact[g]);
    } else if (((pidxincer && !sfincer) || sfidx1 == pidx) && pidx != 0) {
      sorted_union(R->certain, R->ncertain, sfidx1, R->act[g], &R->nact[g]);
    } else if (pidxincer && sfincer && pidx != 0) {
      R->nact[g] = 0; R->empty[g] = 1;
    } else if (!pidxincer && !sfincer && pidx != 0) {
      R->nact[g] = nunc; for (int i = 0; i < nunc; i++) R->act[g][i] = unc[i];
    } else { /* pidx == 0 */
      R->nact[g] = 1; R->act[g][0] = sfidx1;
    }
  }
}

/* flat export for the exact-match tests (test/testExplicitMultihypo.jl:8-240) */
int32_t orc_hypo_recipe(int32_t has_multihypo, const double *mhp, int32_t nvars, int32_t sfidx1, double nullhypo,
                        const int32_t *mhidx, int32_t N, int32_t *certain_out, int32_t *ncertain_out,
                        int32_t *hypo_out, int32_t *nact_out, int32_t *act_out /* [(MAXV+1)*MAXV] */,
                        int32_t *nelem_out, int32_t *elem_out /* [(MAXV+1)*N], 1-based */) {
  recipe_t R;
  orc_build_recipe(has_multihypo, mhp, nvars, sfidx1, nullhypo, &R);
  *ncertain_out = R.ncertain;
  for (int i = 0; i < R.ncertain; i++) certain_out[i] = R.certain[i];
  for (int g = 0; g < R.ngroups; g++) {
    hypo_out[g] = R.hypo[g];
    nact_out[g] = R.nact[g];
    for (int i = 0; i < R.nact[g]; i++) act_out[g * NBP_MAXV + i] = R.act[g][i];
    int c = 0;
    if (!R.empty[g])
      for (int n = 0; n < N; n++) if (mhidx[n] == R.hypo[g]) elem_out[(size_t)g * N + c++] = n + 1;
    nelem_out[g] = c;
  }
  return R.ngroups;
}

/* rand(Categorical(p)) by inverse CDF on one uniform */
static int categorical(const double *p, int n, double u) {
  double c = 0;
  int last = 0;
  for (int i = 0; i < n; i++) {
    if (p[i] > 0) last = i;
    c += p[i];
    if (u < c) return i;
  }
  return last;
}

/* ------------------------------------------------------------------------------------------ */
/* Bandwidth: KernelDensityEstimate `kde!(pts, :lcv)` per coordinate -- golden-section search   */
/* of the leave-one-out negative log likelihood (Ihler ksize 'lcv'); call sites                 */
/* services/ApproxConv.jl:38,41 and AMP.manifoldProduct's rebandwidth.                          */
/* ------------------------------------------------------------------------------------------ */
static double neg_loo_ll(const double *x, int N, int circ, double h) {
  double inv2h2 = 1.0 / (2.0 * h * h), acc = 0;
  double lognorm = log(h) + 0.5 * log(TWO_PI) + log((double)(N - 1));
  t_diag.lcv_evals++;
  double term[NBP_MAXN];
#pragma omp taskloop grainsize(16) default(shared) if (g_inner > 1)
  for (int i = 0; i < N; i++) {
    double s = 0;
    for (int j = 0; j < N; j++) {
      if (j == i) continue;
      double d = x[i] - x[j];
      if (circ) d = orc_wrap(d);
      s += exp(-d * d * inv2h2);
    }
    if (s < 1e-300) s = 1e-300;
    term[i] = log(s) - lognorm;
  }
  for (int i = 0; i < N; i++) acc += term[i]; /* the rows added in order: the serial sum */
  return -acc / N;
}

double orc_lcv_bandwidth_1d(const double *x, int32_t N, int32_t circ) {
  double minm = INFINITY, lo = INFINITY, hi = -INFINITY;
  for (int i = 0; i < N; i++) {
    if (x[i] < lo) lo = x[i];
    if (x[i] > hi) hi = x[i];
    if (i + 1 < N) {
      double d = fabs(circ ? orc_wrap(x[i] - x[i + 1]) : x[i] - x[i + 1]);
      if (d < minm) minm = d;
    }
  }
  double maxm = circ ? TWO_PI : (hi - lo);
  if (!(maxm > 0)) return 1.0;
  if (minm < 1e-6 * maxm) minm = 1e-6 * maxm;
  double sc = 0.5 * (minm + maxm);
  double ax = minm / sc, bx = 1.0, cx = maxm / sc; /* = 2minm/(minm+maxm), 1, 2maxm/(minm+maxm) */
  const double R = 0.61803399, C = 1.0 - R, tol = 1e-2;
  double x0 = ax, x3 = cx, x1, x2;
  if (fabs(cx - bx) > fabs(bx - ax)) { x1 = bx; x2 = bx + C * (cx - bx); }
  else { x2 = bx; x1 = bx - C * (bx - ax); }
  double f1 = neg_loo_ll(x, N, circ, x1 * sc), f2 = neg_loo_ll(x, N, circ, x2 * sc);
  while (fabs(x3 - x0) > tol * (fabs(x1) + fabs(x2))) {
    if (f2 < f1) { x0 = x1; x1 = x2; x2 = R * x1 + C * x3; f1 = f2; f2 = neg_loo_ll(x, N, circ, x2 * sc); }
    else { x3 = x2; x2 = x1; x1 = R * x2 + C * x0; f2 = f1; f1 = neg_loo_ll(x, N, circ, x1 * sc); }
  }
  return (f1 < f2 ? x1 : x2) * sc;
}

static void fit_bandwidth(double *slot, int N, int manifold) {
  int D = mani_dim(manifold);
  const int cnt = slot_count(slot, N); /* manikde! of the points the belief holds */
  for (int d = 0; d < D; d++) {
    /* a circular coordinate is fitted on its representative in [-pi, pi): the identity for a stored belief, not for the
     * predicted measurements of a deconvolution (orc_run_deconv leaves b - a as the search found it), whose neighbour
     * distances would otherwise round differently from those of the wrapped angles */
    if (is_circ(manifold, d)) {
      double w[NBP_MAXN];
      for (int i = 0; i < cnt; i++) w[i] = orc_wrap(slot[d * N + i]);
      slot[3 * N + d] = orc_lcv_bandwidth_1d(w, cnt, 1);
    } else
      slot[3 * N + d] = orc_lcv_bandwidth_1d(slot + d * N, cnt, 0);
  }
  for (int d = D; d < 3; d++) slot[3 * N + d] = 0.0;
}

void orc_run_bandwidth(double *arena, int32_t N, int32_t slot, int32_t manifold) {
  fit_bandwidth(arena + orc_slot_stride(N) * slot, N, manifold);
}

/* ------------------------------------------------------------------------------------------ */
/* Proposal = approxConvBelief (ApproxConv.jl:4-45)                                            */
/* ------------------------------------------------------------------------------------------ */
/* rand(::AliasingScalarSampler) (entities/AliasScalarSampling.jl:57-66): a domain value drawn by its weight */
static double table_draw(const double *tb, int N, double u) {
  const int K = slot_count(tb, N);
  int i = 0;
  while (i < K - 1 && !(u < tb[N + i])) i++; /* row 1: cumulative weights */
  return tb[i];
}
static void sample_measurement(const nbp_proposal_desc *d, int n, int zdim, double *z, int *label, const double *arena, int N) {
  /* needFreshMeasurements = false (SolveTree.jl:119): the samples are the ones an earlier op drew */
  const uint64_t mseed = d->meas_seed ? d->meas_seed : d->seed;
  if (d->meas_kde > 0) {
    /* the measurement is a KDE (LinearRelative(::MKD) / CircularCircular(::MKD), the differential factors
     * of TreeMessageUtils.jl:279-335): sampleTangent(M, ::MKD) = sample(belief, 1) = random kernel +
     * bw*randn (manifolds/services/ManifoldSampling.jl:13-19) */
    const double *msg = arena + orc_slot_stride(N) * (d->meas_kde - 1);
    const int cm = slot_count(msg, N);
    double ua, ub, nn[4];
    orc_uniform_pair(mseed, n, PURP_KDESEL, 0, &ua, &ub);
    int i = (int)(ua * cm);
    if (i >= cm) i = cm - 1;
    orc_normal_pair(mseed, n, PURP_KDENOISE, 0, &nn[0], &nn[1]);
    if (zdim > 2) orc_normal_pair(mseed, n, PURP_KDENOISE, 1, &nn[2], &nn[3]);
    for (int k = 0; k < 3; k++) z[k] = k < zdim ? msg[k * N + i] + msg[3 * N + k] * nn[k] : 0.0;
    if (label) *label = 0;
    return;
  }
  /* Mixture.sampleFactor (Factors/Mixture.jl:114-155): label ~ Categorical(diversity) */
  int c = 0;
  if (d->ncomp > 1) {
    double w[NBP_MAXC], ua, ub;
    for (int i = 0; i < d->ncomp; i++) w[i] = d->comp[i][0];
    orc_uniform_pair(mseed, n, PURP_MIXLBL, 0, &ua, &ub);
    c = categorical(w, d->ncomp, ua);
  }
  if (label) *label = c;
  const double *cp = d->comp[c];
  if (zdim == 1 && cp[12] != 0.0) { /* rand(Uniform(a, b)) / rand(Rayleigh(sigma)): enum nbp_dist, include/nbp.h */
    double ua, ub;
    orc_uniform_pair(mseed, n, PURP_MEAS, 0, &ua, &ub);
    if (cp[12] == (double)NBP_DIST_TABLE) z[0] = table_draw(arena + orc_slot_stride(N) * d->var_slot[NBP_MAXV - 1], N, ua);
    else z[0] = cp[12] == (double)NBP_DIST_UNIFORM ? fma(cp[4], ua, cp[1]) : cp[4] * sqrt(-2.0 * nbpm_log(ua));
    z[1] = z[2] = 0;
    return;
  }
  double nn[4];
  orc_normal_pair(mseed, n, PURP_MEAS, 0, &nn[0], &nn[1]);
  if (zdim > 2) orc_normal_pair(mseed, n, PURP_MEAS, 1, &nn[2], &nn[3]);
  for (int i = 0; i < zdim; i++) { /* rand(MvNormal(mu, L L')) */
    double acc = cp[1 + i];
    for (int j = 0; j <= i; j++) acc += cp[4 + i * 3 + j] * nn[j];
    z[i] = acc;
  }
  for (int i = zdim; i < 3; i++) z[i] = 0;
}

/* addEntropyOnManifold!, EvalFactor.jl:95-132 */
static void add_entropy(int manifold, double *X, int N, int n, double spread, uint64_t seed, int kbase, int mask) {
  /* mask: coordinates that receive entropy (the `p` argument, :99,114); 0 = all */
  int D = mani_dim(manifold);
  double u[4];
  orc_uniform_pair(seed, n, PURP_ENTROPY, kbase, &u[0], &u[1]);
  if (D > 2) orc_uniform_pair(seed, n, PURP_ENTROPY, kbase + 1, &u[2], &u[3]);
  for (int d = 0; d < D; d++) {
    if (mask && !((mask >> d) & 1)) continue;
    double v = X[d * N + n] + spread * (u[d] - 0.5);
    X[d * N + n] = is_circ(manifold, d) ? orc_wrap(v) : v;
  }
}

/* calcVariableDistanceExpectedFractional, EvalFactor.jl:40-92 */
static double var_distance_expected_fractional(const nbp_proposal_desc *d, const recipe_t *R, const double *arena,
                                               int N, const double *X, double kappa) {
  int sf1 = d->sfidx + 1;
  if (in_list(R->certain, R->ncertain, sf1)) return kappa * orc_std_basic_spread(d->manifold, X, N);
  int D = mani_dim(d->manifold);
  double ref[3], mu[3], best = 1e-2;
  mean_default(d->manifold, X, N, ref);
  for (int i = 1; i <= d->nvars; i++) {
    const double *pts = (i == sf1) ? X : arena + orc_slot_stride(N) * d->var_slot[i - 1];
    const int ci = (i == sf1) ? N : slot_count(pts, N); /* the scratch copy of the target always has N entries */
    if (in_list(R->certain, R->ncertain, i)) mean_geodesic_n(d->manifold, pts, N, ci, mu);
    else mean_default_n(d->manifold, pts, N, ci, mu);
    double acc = 0;
    for (int k = 0; k < D; k++) acc += (ref[k] - mu[k]) * (ref[k] - mu[k]);
    double dist = sqrt(acc);
    if (dist > best) best = dist;
  }
  return kappa * best;
}

int32_t orc_run_proposal(double *arena, int32_t N, int32_t *side, const nbp_proposal_desc *d) {
  const int64_t S = orc_slot_stride(N);
  const int D = mani_dim(d->manifold);
  double *out = arena + S * d->out_slot;
  const int sf1 = d->sfidx + 1;
  int *mhidx = (int *)malloc(sizeof(int) * N);
  recipe_t R;
  orc_build_recipe(d->has_multihypo, d->multihypo, d->nvars, sf1, d->nullhypo, &R);

  /* mhidx: injected, or rand(Categorical) -- ExplicitDiscreteMarginalizations.jl:186,261 */
  for (int n = 0; n < N; n++) {
    if (d->mhidx_in >= 0) mhidx[n] = side[d->mhidx_in + n];
    else if (!d->has_multihypo && d->nullhypo == 0.0) mhidx[n] = 1;
    else {
      double ua, ub;
      orc_uniform_pair(d->seed, n, PURP_HYPO, 0, &ua, &ub);
      mhidx[n] = R.cat_first + categorical(R.cat_p, R.ncat, ua);
    }
    if (d->mhidx_out >= 0) side[d->mhidx_out + n] = mhidx[n];
  }

  if (d->factor_kind == NBP_F_PASSTHROUGH) {
    /* calcProposalBelief(::PartialPriorPassThrough), ApproxConv.jl:196-227: the proposal is the density itself
     * (fctFnc.Z.heatmap.densityFnc), placed on the partial coordinates (antimarginal); nothing is sampled or fitted.
     * keep_count = 0: multinomial resampling to the N points a product needs from every input. */
    const double *den = arena + S * d->var_slot[1], *cur = arena + S * d->var_slot[0];
    const int cd = slot_count(den, N), ct = slot_count(cur, N), pm = d->partial_mask ? d->partial_mask : 7;
    double *X = (double *)malloc(sizeof(double) * 3 * N);
    for (int n = 0; n < N; n++)
      for (int k = 0; k < 3; k++) X[k * N + n] = (k < D && n < ct) ? cur[k * N + n] : 0.0;
    for (int n = 0; n < N; n++) {
      int idx = n;
      if (!d->keep_count && cd < N) {
        double ua, ub;
        orc_uniform_pair(d->seed, n, PURP_KDESEL, 0, &ua, &ub);
        idx = (int)(ua * cd);
        if (idx >= cd) idx = cd - 1;
      }
      double nz[4] = {0, 0, 0, 0};
      if (d->keep_count == 2 && cd < N && n >= cd) { /* resample(bel, N) of graph initialisation, GraphInit.jl:174-177 */
        double ua, ub;
        orc_uniform_pair(d->seed, n, PURP_OLDSEL, 0, &ua, &ub);
        idx = (int)(ua * cd);
        if (idx >= cd) idx = cd - 1;
        orc_normal_pair(d->seed, n, PURP_OLDNOISE, 0, &nz[0], &nz[1]);
        if (D > 2) orc_normal_pair(d->seed, n, PURP_OLDNOISE, 1, &nz[2], &nz[3]);
      }
      if (idx < cd)
        for (int k = 0; k < D; k++)
          if ((pm >> k) & 1) {
            const double v = den[k * N + idx] + den[3 * N + k] * nz[k];
            X[k * N + n] = (nz[k] != 0.0 && is_circ(d->manifold, k)) ? orc_wrap(v) : v;
          }
    }
    memcpy(out, X, sizeof(double) * 3 * N);
    free(X);
    free(mhidx);
    for (int k = 0; k < 3; k++) {
      const int in = k < D && ((pm >> k) & 1);
      out[3 * N + k] = in ? den[3 * N + k] : 0.0;
      out[3 * N + 3 + k] = in ? 1.0 : 0.0;
    }
    out[3 * N + 6] = (d->keep_count == 1 && cd < N) ? (double)cd : 0.0;
    return NBP_OK;
  }
  if (d->factor_kind == NBP_F_PRIOR || d->factor_kind == NBP_F_MSGPRIOR) {
    /* evalPotentialSpecific(prior), EvalFactor.jl:400-542 */
    const double *cur = arena + S * d->var_slot[0];
    double *X = (double *)malloc(sizeof(double) * 3 * N);
    memcpy(X, cur, sizeof(double) * 3 * N); /* addEntr = deepcopy(solveForPts) */
    for (int n = slot_count(cur, N); n < N; n++) /* resized to N: new entries are the point default */
      for (int k = 0; k < 3; k++) X[k * N + n] = 0.0;
    double spread = d->spread_nh * orc_std_basic_spread(d->manifold, X, N); /* :464 */
    for (int n = 0; n < N; n++) {
      if (mhidx[n] == 1) {
        if (d->factor_kind == NBP_F_PRIOR && d->partial_mask) {
          /* partial prior: setPointPartial! on the partial coordinates only, :457-538 */
          double z[3];
          int zd = 0, pk = 0;
          for (int k = 0; k < D; k++) zd += (d->partial_mask >> k) & 1;
          sample_measurement(d, n, zd, z, 0, arena, N);
          for (int k = 0; k < D; k++)
            if ((d->partial_mask >> k) & 1) { X[k * N + n] = is_circ(d->manifold, k) ? orc_wrap(z[pk]) : z[pk]; pk++; }
        } else if (d->factor_kind == NBP_F_PRIOR) {
          double z[3];
          sample_measurement(d, n, D, z, 0, arena, N);
          for (int k = 0; k < D; k++) X[k * N + n] = is_circ(d->manifold, k) ? orc_wrap(z[k]) : z[k];
        } else { /* MsgPrior{MKD}: sample(belief,1): random kernel + bw*randn, Factors/MsgPrior.jl:27-30 */
          const double *msg = arena + S * d->var_slot[1];
          const int cm = slot_count(msg, N);
          const uint64_t mseed = d->meas_seed ? d->meas_seed : d->seed;
          double ua, ub, nn[4];
          orc_uniform_pair(mseed, n, PURP_KDESEL, 0, &ua, &ub);
          int i = (int)(ua * cm);
          if (i >= cm) i = cm - 1;
          orc_normal_pair(mseed, n, PURP_KDENOISE, 0, &nn[0], &nn[1]);
          if (D > 2) orc_normal_pair(mseed, n, PURP_KDENOISE, 1, &nn[2], &nn[3]);
          for (int k = 0; k < D; k++) {
            double v = msg[k * N + i] + msg[3 * N + k] * nn[k];
            X[k * N + n] = is_circ(d->manifold, k) ? orc_wrap(v) : v;
          }
        }
      } else { /* nullhypo particles keep their value + entropy, :476 */
        add_entropy(d->manifold, X, N, n, spread, d->seed, 0, d->partial_mask); /* partialCoords, :532 */
      }
    }
    memcpy(out, X, sizeof(double) * 3 * N);
    free(X);
  } else {
    /* evalPotentialSpecific(relative), EvalFactor.jl:321-395 */
    int zdim = factor_zdim(d->factor_kind, d->manifold);
    int pdim = -1, pdim2 = -1; /* the partial coordinates (one or two) of a partial relative factor */
    if (d->partial_mask) {
      int cnt = 0;
      for (int k = 0; k < D; k++) if ((d->partial_mask >> k) & 1) { if (cnt == 0) pdim = k; else pdim2 = k; cnt++; }
      if (d->factor_kind == NBP_F_SE2) {
        /* a partial ManifoldFactor on SE(2): the measurement stays the full group element; the residual counts only the
         * components in `.partial` (it "must deal with the partial" itself), entropy goes on those coordinates only, and
         * the search is BFGS over the WHOLE point (`islen1 = ... || ccwl.partial`, NumericalCalculations.jl:424-446) */
        pdim = pdim2 = -1;
      } else {
        if (d->factor_kind != NBP_F_LINREL || cnt < 1 || cnt > 2) { free(mhidx); return NBP_ERR_ARG; }
        zdim = cnt;
      }
    }
    double *X = out; /* ccwl.varValsAll[sfidx] = deepcopy(target), CalcFactor.jl:543-548 */
    {
      const double *tsrc = arena + S * d->var_slot[d->sfidx];
      const int ct = slot_count(tsrc, N);
      memmove(X, tsrc, sizeof(double) * 3 * N);
      for (int n = ct; n < N; n++) /* resize!(..., N) + getPointDefault for the new entries, CalcFactor.jl:555-565 */
        for (int k = 0; k < 3; k++) X[k * N + n] = 0.0;
    }
    double *Z = (double *)malloc(sizeof(double) * 3 * N);
    for (int n = 0; n < N; n++) sample_measurement(d, n, zdim, Z + 3 * n, 0, arena, N); /* sampleFactor!, :578 */

    /* computeAcrossHypothesis!, EvalFactor.jl:145-237 */
    for (int g = 0; g < R.ngroups; g++) {
      int hyp = R.hypo[g];
      int solve_case = (in_list(R.certain, R.ncertain, sf1) && hyp != 0) || in_list(R.certain, R.ncertain, hyp) || hyp == sf1;
      if (R.empty[g]) continue;
      int nelem = 0;
      for (int n = 0; n < N; n++) nelem += (mhidx[n] == hyp);
      if (nelem == 0) continue; /* empty allelements[g]: every loop of the reference is a no-op */
      if (solve_case) {
        if (R.nact[g] != 2) { free(Z); free(mhidx); return NBP_ERR_ARG; } /* binary mechanics only */
        int va = R.act[g][0], vb = R.act[g][1];
        int solve_b = (vb == sf1);
        int vother = solve_b ? va : vb;
        const double *O = arena + S * d->var_slot[vother - 1];
        for (int c = 0; c < d->inflate_cycles; c++) { /* :184-207 */
          double spread = var_distance_expected_fractional(d, &R, arena, N, X, d->inflation);
          for (int n = 0; n < N; n++)
            if (mhidx[n] == hyp) add_entropy(d->manifold, X, N, n, spread, d->seed, (g * 8 + c) * 2, d->partial_mask);
          /* (the searches of the particles are independent: the inner OpenMP level of the CPU baseline; every inner thread
             hands its counters in before the team ends) */
          {
#pragma omp taskloop grainsize(8) default(shared) if (g_inner > 1)
          for (int n = 0; n < N; n++) { /* approxConvOnElements!, :14-27 */
            if (mhidx[n] != hyp) continue;
            double x[3], oth[3];
            const int io = anyn_index(n, slot_count(O, N), d->seed, vother); /* _getindex_anyn */
            for (int k = 0; k < D; k++) { x[k] = X[k * N + n]; oth[k] = O[k * N + io]; }
            if (d->partial_mask && d->factor_kind == NBP_F_SE2) { /* partial SE(2) factor: BFGS over the whole point */
              objective_t o3;
              o3.kind = NBP_F_SE2; o3.manifold = NBP_SE2; o3.D = 3; o3.solve_b = solve_b; o3.rmask = d->partial_mask;
              for (int k = 0; k < 3; k++) { o3.z[k] = Z[3 * n + k]; o3.other[k] = oth[k]; }
              double x3[3] = {x[0], x[1], x[2]};
              t_diag.solves++;
              if (!orc_bfgs_nd(&o3, 3, x3, 0)) t_diag.nonconverged++;
              if (isnan(x3[0]) || isnan(x3[1]) || isnan(x3[2])) t_diag.nan_results++;
              else { x[0] = x3[0]; x[1] = x3[1]; x[2] = orc_wrap(x3[2]); }
            } else if (pdim2 >= 0) { /* two partial coordinates: n-D BFGS on the pair */
              objective_t o2;
              o2.kind = NBP_F_LINREL; o2.manifold = NBP_EUCLID2; o2.D = 2; o2.solve_b = solve_b; o2.rmask = 0;
              for (int k = 0; k < 3; k++) { o2.z[k] = Z[3 * n + k]; o2.other[k] = 0.0; }
              o2.other[0] = oth[pdim]; o2.other[1] = oth[pdim2];
              double x2[3] = {x[pdim], x[pdim2], 0};
              t_diag.solves++;
              if (!orc_bfgs_nd(&o2, 2, x2, 0)) t_diag.nonconverged++;
              if (isnan(x2[0]) || isnan(x2[1])) t_diag.nan_results++;
              else { x[pdim] = x2[0]; x[pdim2] = x2[1]; }
            } else if (pdim >= 0) /* `.partial` -> islen1 -> BFGS (NumericalCalculations.jl:424); the gradient is
                              zero off the partial coordinate, so the search runs on that coordinate alone */
              solve_particle(NBP_F_LINREL, NBP_EUCLID1, Z + 3 * n, oth + pdim, solve_b, x + pdim);
            else
              solve_particle(d->factor_kind, d->manifold, Z + 3 * n, oth, solve_b, x);
            for (int k = 0; k < D; k++) X[k * N + n] = x[k];
          }
          }
        }
      } else { /* other-hypothesis (:208-220) and nullhypo (:222-231): entropy only */
        double spread = var_distance_expected_fractional(d, &R, arena, N, X, d->spread_nh);
        for (int n = 0; n < N; n++)
          if (mhidx[n] == hyp) add_entropy(d->manifold, X, N, n, spread, d->seed, (g * 8) * 2, 0);
      }
    }
    free(Z);
  }
  free(mhidx);
  out[3 * N + 6] = 0.0; /* a proposal always holds N points */
  /* ipc = ones(D), zeroed outside `.partial` (EvalFactor.jl:383-391 relative, :534-540 prior) */
  for (int k = 0; k < 3; k++)
    out[3 * N + 3 + k] = (k < mani_dim(d->manifold) && (!d->partial_mask || ((d->partial_mask >> k) & 1))) ? 1.0 : 0.0;
  if (!d->skip_bandwidth) fit_bandwidth(out, N, d->manifold); /* manikde!, ApproxConv.jl:36-42 */
  return NBP_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* AMP.manifoldProduct (GraphProductOperations.jl:53-60): multiscale sequential Gibbs sampler.  */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
  int N, L;
  int cnt[12];
  int *lo[12], *hi[12], *child[12]; /* child[l][k] = index at level l+1 of the LAST child of node k */
} levels_t;

static void levels_build(levels_t *T, int N) {
  T->N = N;
  int l = 0;
  T->cnt[0] = 1;
  T->lo[0] = (int *)malloc(sizeof(int)); T->hi[0] = (int *)malloc(sizeof(int));
  T->lo[0][0] = 0; T->hi[0][0] = N;
  for (;;) {
    int all_leaf = 1;
    for (int k = 0; k < T->cnt[l]; k++) if (T->hi[l][k] - T->lo[l][k] > 1) all_leaf = 0;
    if (all_leaf) break;
    int c = 0;
    T->lo[l + 1] = (int *)malloc(sizeof(int) * N); T->hi[l + 1] = (int *)malloc(sizeof(int) * N);
    T->child[l] = (int *)malloc(sizeof(int) * T->cnt[l]);
    for (int k = 0; k < T->cnt[l]; k++) {
      int lo = T->lo[l][k], hi = T->hi[l][k];
      if (hi - lo == 1) { T->lo[l + 1][c] = lo; T->hi[l + 1][c] = hi; c++; } /* leaf: left child = itself */
      else {
        int mid = lo + (hi - lo + 1) / 2;
        T->lo[l + 1][c] = lo; T->hi[l + 1][c] = mid; c++;
        T->lo[l + 1][c] = mid; T->hi[l + 1][c] = hi; c++;
      }
      T->child[l][k] = c - 1;
    }
    T->cnt[l + 1] = c;
    l++;
  }
  T->L = l;
}
static void levels_free(levels_t *T) {
  for (int l = 0; l <= T->L; l++) { free(T->lo[l]); free(T->hi[l]); if (l < T->L) free(T->child[l]); }
}

typedef struct { const double *x; int N, dim; } sortctx_t;
static int cmp_idx(const void *a, const void *b, void *ctx) {
  const sortctx_t *g = (const sortctx_t *)ctx;
  int ia = *(const int *)a, ib = *(const int *)b;
  double xa = g->x[g->dim * g->N + ia], xb = g->x[g->dim * g->N + ib];
  if (xa < xb) return -1;
  if (xa > xb) return 1;
  return ia - ib;
}
/* KD-tree permutation: split the widest coordinate at the median (BallTree build) */
static void kd_build(const double *x, int N, int D, int mask, int *idx, int lo, int hi) {
  /* mask: the coordinates the density informs (all of them unless it is a partial density) */
  if (hi - lo <= 1) return;
  int best = 0;
  double bext = -1;
  for (int d = 0; d < D; d++) {
    if (!((mask >> d) & 1)) continue;
    double mn = INFINITY, mx = -INFINITY;
    for (int i = lo; i < hi; i++) { double v = x[d * N + idx[i]]; if (v < mn) mn = v; if (v > mx) mx = v; }
    if (mx - mn > bext) { bext = mx - mn; best = d; }
  }
  sortctx_t g;
  g.x = x; g.N = N; g.dim = best;
  qsort_r(idx + lo, hi - lo, sizeof(int), cmp_idx, &g);
  int mid = lo + (hi - lo + 1) / 2;
  kd_build(x, N, D, mask, idx, lo, mid);
  kd_build(x, N, D, mask, idx, mid, hi);
}

/* sample(oldBel, nn) (GraphProductOperations.jl:39-45): a belief with fewer than N points is topped up with draws
 * from its own KDE -- random kernel + bw * randn (manifolds/services/ManifoldSampling.jl:13-19); the points it holds
 * stay where they are */
void orc_topup_slot(double *s, int32_t N, int32_t manifold, uint64_t seed) {
  const int cnt = slot_count(s, N), D = mani_dim(manifold);
  if (cnt >= N) return;
  for (int n = cnt; n < N; n++) {
    double ua, ub, nn[4] = {0, 0, 0, 0};
    orc_uniform_pair(seed, n, PURP_OLDSEL, 0, &ua, &ub);
    int i = (int)(ua * cnt);
    if (i >= cnt) i = cnt - 1;
    orc_normal_pair(seed, n, PURP_OLDNOISE, 0, &nn[0], &nn[1]);
    if (D > 2) orc_normal_pair(seed, n, PURP_OLDNOISE, 1, &nn[2], &nn[3]);
    for (int k = 0; k < 3; k++) {
      double v = k < D ? s[k * N + i] + s[3 * N + k] * nn[k] : 0.0;
      s[k * N + n] = is_circ(manifold, k) ? orc_wrap(v) : v;
    }
  }
  s[3 * N + 6] = 0.0;
}
void orc_run_resample(double *arena, int32_t N, const int32_t *slots, const int32_t *manifolds, int32_t n, uint64_t seed) {
  for (int i = 0; i < n; i++)
    orc_topup_slot(arena + orc_slot_stride(N) * slots[i], N, manifolds[i], seed + 0x9E3779B97F4A7C15ull * (uint64_t)(i + 1));
}

int32_t orc_run_product(double *arena, int32_t N, int32_t *side, const nbp_product_desc *d) {
  const int64_t S = orc_slot_stride(N);
  const int D = mani_dim(d->manifold), F = d->nfactors, M = d->manifold;
  double *out = arena + S * d->out_slot;
  if (F == 1) { /* single density: AMP returns it as is */
    memmove(out, arena + S * d->in_slot[0], sizeof(double) * (3 * N + 3));
    for (int k = 0; k < 3; k++) out[3 * N + 3 + k] = k < D ? 1.0 : 0.0; /* proposalbeliefs!: ipc = sum of ones(D) */
    out[3 * N + 6] = (arena + S * d->in_slot[0])[3 * N + 6];
    if (d->labels_out >= 0) for (int n = 0; n < N; n++) side[d->labels_out + n] = n;
    return NBP_OK;
  }
  /* partial densities (AMP.marginal, ApproxConv.jl:287-291) multiply in on their coordinates only;
     a coordinate no density informs keeps the old point (GraphProductOperations.jl:39-45) */
  int pm[NBP_MAXF];
  for (int j = 0; j < F; j++) pm[j] = d->in_partial[j] ? d->in_partial[j] : (1 << D) - 1;
  if (d->old_slot >= 0) orc_topup_slot(arena + S * d->old_slot, N, M, d->seed); /* oldPoints, GraphProductOperations.jl:39-45 */
  const double *old = d->old_slot >= 0 ? arena + S * d->old_slot : 0;
  levels_t T;
  levels_build(&T, N);
  /* per density: sorted, centred coordinates and per-level node statistics */
  int *idx[NBP_MAXF];
  double *xs[NBP_MAXF], ctr[NBP_MAXF][3], h2[NBP_MAXF][3];
  double *nmean[NBP_MAXF][12], *nvar[NBP_MAXF][12], *nprec[NBP_MAXF][12];
  for (int j = 0; j < F; j++) {
    const double *x = arena + S * d->in_slot[j];
    idx[j] = (int *)malloc(sizeof(int) * N);
    for (int i = 0; i < N; i++) idx[j][i] = i;
    kd_build(x, N, D, pm[j], idx[j], 0, N);
    xs[j] = (double *)malloc(sizeof(double) * 3 * N);
    for (int k = 0; k < D; k++) {
      /* the centre of the density: the sum in the order the KD build's workgroup reduces it (chunked_tree_sum above) */
      double c = chunked_tree_sum(x + k * N, N) / (double)N;
      ctr[j][k] = c;
      h2[j][k] = x[3 * N + k] * x[3 * N + k];
      for (int i = 0; i < N; i++) xs[j][k * N + i] = x[k * N + idx[j][i]] - c;
    }
    for (int l = 0; l <= T.L; l++) {
      nmean[j][l] = (double *)malloc(sizeof(double) * 3 * T.cnt[l]);
      nvar[j][l] = (double *)malloc(sizeof(double) * 3 * T.cnt[l]);
      nprec[j][l] = (double *)malloc(sizeof(double) * 3 * T.cnt[l]);
      for (int z = 0; z < T.cnt[l]; z++) {
        int lo = T.lo[l][z], hi = T.hi[l][z], n = hi - lo;
        for (int k = 0; k < D; k++) {
          double s1 = 0, s2 = 0;
          /* (the root: the sums of its two children added -- the order the device takes them in) */
          int mid = (l == 0 && T.L > 0) ? T.hi[1][0] : hi;
          for (int i = lo; i < mid; i++) { double v = xs[j][k * N + i]; s1 += v; s2 += v * v; }
          if (mid < hi) {
            double t1 = 0, t2 = 0;
            for (int i = mid; i < hi; i++) { double v = xs[j][k * N + i]; t1 += v; t2 += v * v; }
            s1 += t1; s2 += t2;
          }
          double mu = s1 / n, var = s2 / n - mu * mu;
          if (var < 0) var = 0;
          nmean[j][l][k * T.cnt[l] + z] = ctr[j][k] + mu;
          nvar[j][l][k * T.cnt[l] + z] = var + h2[j][k]; /* moment-matched Gaussian of the sub-mixture */
          nprec[j][l][k * T.cnt[l] + z] = 1.0 / (var + h2[j][k]); /* its precision, once per node */
        }
      }
    }
  }
  double *res = (double *)malloc(sizeof(double) * 3 * N);
  /* (the output samples are independent -- every draw is keyed by the sample index: the inner OpenMP level of the CPU baseline) */
#pragma omp taskloop grainsize(8) default(shared) if (g_inner > 1)
  for (int s = 0; s < N; s++) {
    int ind[NBP_MAXF];
    for (int j = 0; j < F; j++) ind[j] = 0; /* levelInit! / initIndices!: root */
    /* passes: KernelDensityEstimate's Nlevels = floor(log2(maxNp) + 1) = the depth of the tree, except that a power of two
       gets one pass more than its tree has levels -- the leaf level twice (levelDown! leaves a leaf as its own child) */
    int npass = 1;
    for (int n2 = N; n2 > 1; n2 >>= 1) npass++;
    if (npass < T.L) npass = T.L;
    for (int ps = 1; ps <= npass; ps++) {
      const int l = ps < T.L ? ps : T.L, lp = (ps - 1) < T.L ? ps - 1 : T.L; /* tree level of this pass / of the labels it starts from */
      /* One level of the multiscale sampler as published (Ihler, Sudderth, Freeman, Willsky, "Efficient multiscale
         sampling from products of Gaussian mixtures", NIPS 2003, sec. 4; KernelDensityEstimate.jl's gibbs1 loop follows
         it: samplePoint!, levelDown!, sampleIndices!, then Niter sweeps of sampleIndex):
           samplePoint!:    x ~ the product of the Gaussians selected on the level above,
           levelDown!:      the candidates of every density become ALL nodes of this level,
           sampleIndices!:  every density draws its label given x, independently:  p(z) ~ w_z N(x; mean_z, var_z). */
      const int cp = T.cnt[lp], cnt = T.cnt[l];
      double xp[3];
      int xinf[3];
      {
        double nn[4] = {0, 0, 0, 0};
        orc_normal_pair(d->seed, s, PURP_PLEVEL, (uint32_t)(2 * ps), &nn[0], &nn[1]);
        if (D > 2) orc_normal_pair(d->seed, s, PURP_PLEVEL, (uint32_t)(2 * ps + 1), &nn[2], &nn[3]);
        for (int k = 0; k < D; k++) {
          double prec = 0, acc = 0, ss = 0, sc = 0;
          for (int q = 0; q < F; q++) {
            if (!((pm[q] >> k) & 1)) continue;
            double mq = nmean[q][lp][k * cp + ind[q]], rq = nprec[q][lp][k * cp + ind[q]];
            prec += rq;
            if (is_circ(M, k)) { double sq, cq; nbpm_sincos(mq, &sq, &cq); ss += sq * rq; sc += cq * rq; }
            else acc += mq * rq;
          }
          xinf[k] = prec > 0;
          if (!xinf[k]) { xp[k] = 0.0; continue; } /* no density informs this coordinate: it enters no weight */
          double mu = is_circ(M, k) ? nbpm_atan2(ss, sc) : acc / prec;
          double v = mu + sqrt(1.0 / prec) * nn[k];
          xp[k] = is_circ(M, k) ? orc_wrap(v) : v;
        }
      }
      /* One Philox block per (sample, pass, density) feeds both draws a density makes on a level with Niter = 1: its first
         uniform the label given the point (sampleIndices!), its second the first sweep's sampleIndex; further sweeps
         (it = 1 .. 7) have blocks of their own (PURP_PGIBBS). */
      double ub_first[NBP_MAXF];
      for (int j = 0; j < F; j++) {
        double ua;
        orc_uniform_pair(d->seed, s, PURP_PINDEX, (uint32_t)(ps * NBP_MAXF + j), &ua, &ub_first[j]);
        double ev[NBP_MAXN]; double m = -INFINITY;
        for (int z = 0; z < cnt; z++) {
          double e = 0;
          for (int k = 0; k < D; k++) {
            if (!((pm[j] >> k) & 1) || !xinf[k]) continue;
            double tmp = nmean[j][l][k * cnt + z] - xp[k];
            if (is_circ(M, k)) tmp = orc_wrap(tmp);
            double v = nvar[j][l][k * cnt + z];
            e += tmp * tmp / v + log(v);
          }
          e = -0.5 * e + log((double)(T.hi[l][z] - T.lo[l][z]) / N);
          ev[z] = e; if (e > m) m = e;
        }
        double tot = 0; for (int z = 0; z < cnt; z++) { ev[z] = exp(ev[z] - m); tot += ev[z]; }
        double target = ua * tot, c = 0; int choice = -1;
        for (int z = 0; z < cnt; z++) { c += ev[z]; if (target < c) { choice = z; break; } }
        ind[j] = choice < 0 ? cnt - 1 : choice;
      }
      for (int it = 0; it < d->niter; it++) {
        for (int j = 0; j < F; j++) { /* sequential Gibbs sweep: sampleIndex(j) */
          double mn[3], vn[3];
          int use[3]; /* coordinates that enter the weight: informed by j and by at least one other */
          for (int k = 0; k < D; k++) { /* product of all but the jth selected Gaussians */
            double prec = 0, acc = 0, ss = 0, sc = 0;
            for (int q = 0; q < F; q++) {
              if (q == j || !((pm[q] >> k) & 1)) continue;
              double mq = nmean[q][l][k * cnt + ind[q]], rq = nprec[q][l][k * cnt + ind[q]];
              prec += rq;
              if (is_circ(M, k)) { double sq, cq; nbpm_sincos(mq, &sq, &cq); ss += sq * rq; sc += cq * rq; }
              else acc += mq * rq;
            }
            use[k] = ((pm[j] >> k) & 1) && prec > 0;
            vn[k] = 1.0 / prec;
            mn[k] = is_circ(M, k) ? nbpm_atan2(ss, sc) : acc * vn[k]; /* getMu: Euclid / getCircMu */
          }
          double ua = ub_first[j], ub = 0;
          (void)ub;
          if (it > 0) orc_uniform_pair(d->seed, s, PURP_PGIBBS, (uint32_t)((ps * 8 + it) * NBP_MAXF + j), &ua, &ub);
          /* rand(Categorical(p)) by inverse CDF (max-stabilised weights) */
          double u = ua; int choice = -1;
          {
            double ev[NBP_MAXN]; double m = -INFINITY;
            for (int z = 0; z < cnt; z++) {
              double e = 0;
              for (int k = 0; k < D; k++) {
                if (!use[k]) continue;
                double tmp = nmean[j][l][k * cnt + z] - mn[k];
                if (is_circ(M, k)) tmp = orc_wrap(tmp);
                double v = nvar[j][l][k * cnt + z] + vn[k];
                e += tmp * tmp / v + log(v);
              }
              e = -0.5 * e + log((double)(T.hi[l][z] - T.lo[l][z]) / N);
              ev[z] = e; if (e > m) m = e;
            }
            double tot = 0; for (int z = 0; z < cnt; z++) { ev[z] = exp(ev[z] - m); tot += ev[z]; }
            double target = u * tot, c = 0;
            for (int z = 0; z < cnt; z++) { c += ev[z]; if (target < c) { choice = z; break; } }
            if (choice < 0) choice = cnt - 1;
          }
          if (choice >= 0) ind[j] = choice;
        }
      }
    }
    /* samplePoint!: draw from the product of the F selected leaf kernels */
    double nn[4];
    orc_normal_pair(d->seed, s, PURP_PFINAL, 0, &nn[0], &nn[1]);
    if (D > 2) orc_normal_pair(d->seed, s, PURP_PFINAL, 1, &nn[2], &nn[3]);
    const int cnt = T.cnt[T.L];
    for (int k = 0; k < D; k++) {
      double prec = 0, acc = 0, ss = 0, sc = 0;
      for (int q = 0; q < F; q++) {
        if (!((pm[q] >> k) & 1)) continue;
        double mq = nmean[q][T.L][k * cnt + ind[q]], rq = nprec[q][T.L][k * cnt + ind[q]];
        prec += rq;
        if (is_circ(M, k)) { double sq, cq; nbpm_sincos(mq, &sq, &cq); ss += sq * rq; sc += cq * rq; }
        else acc += mq * rq;
      }
      if (!(prec > 0)) { res[k * N + s] = old ? old[k * N + s] : 0.0; continue; } /* uninformed coordinate */
      double mu = is_circ(M, k) ? nbpm_atan2(ss, sc) : acc / prec;
      double v = mu + sqrt(1.0 / prec) * nn[k];
      res[k * N + s] = is_circ(M, k) ? orc_wrap(v) : v;
    }
    if (d->labels_out >= 0)
      for (int j = 0; j < F; j++) side[d->labels_out + s * F + j] = idx[j][T.lo[T.L][ind[j]]];
  }
  for (int k = 0; k < D; k++) memcpy(out + k * N, res + k * N, sizeof(double) * N);
  for (int k = D; k < 3; k++) memset(out + k * N, 0, sizeof(double) * N);
  free(res);
  for (int j = 0; j < F; j++) {
    free(idx[j]); free(xs[j]);
    for (int l = 0; l <= T.L; l++) { free(nmean[j][l]); free(nvar[j][l]); free(nprec[j][l]); }
  }
  levels_free(&T);
  /* proposalbeliefs! (ApproxConv.jl:277,298-303): fct_ipc = ones(vardim) for every factor, summed */
  for (int k = 0; k < 3; k++) out[3 * N + 3 + k] = k < D ? (double)F : 0.0;
  out[3 * N + 6] = 0.0; /* N points */
  fit_bandwidth(out, N, M); /* rebandwidth of the product */
  return NBP_OK;
}

void orc_run_copy(double *arena, int32_t N, const nbp_copy_desc *c) {
  const int64_t S = orc_slot_stride(N);
  memmove(arena + S * c->dst_slot, arena + S * c->src_slot, sizeof(double) * S);
}

/* ------------------------------------------------------------------------------------------ */
/* approxDeconv (services/DeconvUtils.jl:32-106 legacy, :108-160 AbstractManifoldMinimize):       */
/* per particle, the measurement that zeroes the residual between the stored variable points,  */
/* searched from a freshly sampled measurement (NelderMead, BFGS when zDim == 1, :94,139).      */
/* out_slot <- predicted measurement coordinates, meas_slot <- the sampled ones.                */
/* ------------------------------------------------------------------------------------------ */
int32_t orc_run_deconv(double *arena, int32_t N, const nbp_proposal_desc *d, int32_t meas_slot) {
  const int64_t S = orc_slot_stride(N);
  if (d->factor_kind < NBP_F_LINREL || d->has_multihypo || d->nvars != 2 || d->partial_mask) return NBP_ERR_ARG;
  const int D = mani_dim(d->manifold), zdim = factor_zdim(d->factor_kind, d->manifold);
  const double *A = arena + S * d->var_slot[0], *B = arena + S * d->var_slot[1];
  double *out = arena + S * d->out_slot, *ms = meas_slot >= 0 ? arena + S * meas_slot : 0;
  for (int n = 0; n < N; n++) {
    double z[3];
    sample_measurement(d, n, zdim, z, 0, arena, N);
    if (ms) for (int k = 0; k < 3; k++) ms[k * N + n] = k < zdim ? z[k] : 0.0;
    objective_t o;
    o.kind = d->factor_kind; o.manifold = d->manifold; o.D = D; o.solve_b = 2; o.rmask = 0;
    const int ia = anyn_index(n, slot_count(A, N), d->seed, 1), ib = anyn_index(n, slot_count(B, N), d->seed, 2);
    for (int k = 0; k < 3; k++) { o.z[k] = k < D ? A[k * N + ia] : 0.0; o.other[k] = k < D ? B[k * N + ib] : 0.0; }
    double zc[3] = {z[0], z[1], z[2]};
    int conv;
    t_diag.solves++;
    if (zdim == 1) conv = orc_bfgs_1d(&o, zc, 0);
    else conv = orc_nelder_mead(&o, zdim, zc, 0);
    if (!conv) t_diag.nonconverged++;
    int bad = 0;
    for (int k = 0; k < zdim; k++) bad |= isnan(zc[k]);
    if (bad) { t_diag.nan_results++; for (int k = 0; k < zdim; k++) zc[k] = z[k]; }
    for (int k = 0; k < 3; k++) out[k * N + n] = k < zdim ? zc[k] : 0.0;
  }
  for (int k = 0; k < 3; k++) { out[3 * N + k] = 0.0; if (ms) ms[3 * N + k] = 0.0; }
  out[3 * N + 6] = 0.0;
  if (ms) ms[3 * N + 6] = 0.0;
  return NBP_OK;
}

/* batch drivers, used by the CPU baseline: the same stage semantics as nbp_program_run */
/* ops of one stage are independent by construction (distinct out slots), so the multi-core
 * baseline runs them with OpenMP -- the analogue of the reference's task-per-clique concurrency
 * (services/SolverAPI.jl:59-97). */
/* (a team no larger than the stage: the rounds near the root of a tree hold a handful of ops, and waking 128 threads for
 *  five of them -- a few hundred times per solve -- is what made the baseline SLOWER beyond 16 threads in rounds 3 and 4) */
#ifdef _OPENMP
#include <omp.h>
static int team_for(int n) { const int t = omp_get_max_threads(); return n < t ? (n > 0 ? n : 1) : t; }
#else
static int team_for(int n) { (void)n; return 1; }
#endif
/* the ops of a stage as tasks of one team; with the inner level on (orc_set_nested) the whole team is present whatever the
 * number of ops, and its idle threads take the taskloops inside the ops that run.  The counters of the searches are per
 * thread: every thread of the team hands its own in before the team ends. */
static int g_nested = 0;
int32_t orc_run_proposals(double *arena, int32_t N, int32_t *side, const nbp_proposal_desc *d, int32_t n) {
  int rc = NBP_OK;
  g_inner = g_nested ? 2 : 1;
#pragma omp parallel num_threads(g_nested ? team_for(1 << 30) : team_for(n))
  {
#pragma omp single
    for (int i = 0; i < n; i++) {
#pragma omp task firstprivate(i) shared(rc)
      { int r = orc_run_proposal(arena, N, side, d + i); if (r) rc = r; }
    }
    diag_merge();
  }
  g_inner = 1;
  return rc;
}
int32_t orc_run_products(double *arena, int32_t N, int32_t *side, const nbp_product_desc *d, int32_t n) {
  int rc = NBP_OK;
  g_inner = g_nested ? 2 : 1;
#pragma omp parallel num_threads(g_nested ? team_for(1 << 30) : team_for(n))
  {
#pragma omp single
    for (int i = 0; i < n; i++) {
#pragma omp task firstprivate(i) shared(rc)
      { int r = orc_run_product(arena, N, side, d + i); if (r) rc = r; }
    }
    diag_merge();
  }
  g_inner = 1;
  return rc;
}
int32_t orc_run_deconvs(double *arena, int32_t N, const nbp_proposal_desc *d, const int32_t *meas_slots, int32_t n) {
  int rc = NBP_OK;
#pragma omp parallel for schedule(dynamic, 1) num_threads(team_for(n))
  for (int i = 0; i < n; i++) { int r = orc_run_deconv(arena, N, d + i, meas_slots ? meas_slots[i] : -1); if (r) rc = r; diag_merge(); }
  return rc;
}
void orc_set_threads(int32_t n);
void orc_set_nested(int32_t on);
void orc_run_copies(double *arena, int32_t N, const nbp_copy_desc *c, int32_t n) {
  for (int i = 0; i < n; i++) orc_run_copy(arena, N, c + i);
}

#ifdef _OPENMP
/* the inner level of parallelism (the CPU baseline): on = 1 lets the idle threads of a stage's team work inside its ops */
void orc_set_nested(int32_t on) { g_nested = on; }
void orc_set_threads(int32_t n) { omp_set_num_threads(n); }
int32_t orc_get_max_threads(void) { return omp_get_max_threads(); }
#else
void orc_set_nested(int32_t on) { (void)on; }
void orc_set_threads(int32_t n) { (void)n; }
int32_t orc_get_max_threads(void) { return 1; }
#endif
