/*
 * nbp_math.h -- the elementary functions of the continuous-valued data path, written ONCE.
 *
 * Why this file exists ("One arithmetic for the values that travel"): a per-particle Nelder-Mead
 * search in three dimensions stops ~1e-4 from its root, and where in that ball is a piecewise-affine function of its
 * start with a heavy-tailed slope -- an ulp of difference in what goes INTO a search (a measurement drawn through two
 * different libm's, a heading's sine) comes out at 1e-9 and flips a label of the product sampler four rounds later.
 * The values a solve hands from op to op therefore have to be the same to the last bit on the device and on the CPU
 * checker, and a libm is not a specification: glibc's and the ROCm device library's log / sin / cos / atan2 each round
 * "within an ulp", differently.  So the functions those values pass through are defined here, in plain C with every
 * multiply-add spelled as fma() and nothing left to a compiler's contraction choice, and BOTH sides include this file:
 * csrc/nbp_device.h (the product, compiled by hipcc for gfx950) and oracle/nbp_oracle.c (the checker, compiled by gcc).
 * IEEE-754 then makes the results identical: +, -, *, /, sqrt and fma are correctly rounded on both, and this file uses
 * nothing else.  (The product does not include anything under oracle/; the oracle already includes ../include/nbp.h.)
 *
 * What is here: log on the uniforms of Box-Muller, sin and cos of angles of moderate size, atan2, the wrap to [-pi, pi)
 * and the normal pair built from them.  What is NOT here and does not need to be: exp / log inside the weights of the
 * product sampler and inside the leave-one-out likelihood of the bandwidth fit -- those values decide comparisons and
 * label draws (integers), they never travel as coordinates.
 *
 * Accuracy (tests/test_nbp_math.py against the host libm on 10^6 arguments each): nbpm_log within 1 ulp, nbpm_atan2 within
 * 1.5 ulp, nbpm_sincos within 1 ulp + |a| * 1e-26 (the reduction carries pi/2 to 86 bits: next to a zero of the function
 * the error is absolute, not relative).  Domain: nbpm_log positive normal finite arguments; nbpm_sincos |a| < 1e5 (sums of a
 * few wrapped angles; Cody-Waite in two parts, no Payne-Hanek); nbpm_atan2 finite arguments.
 *
 * The polynomial coefficients and the reduction schemes are the classical ones of Sun's fdlibm (e_log.c, k_sin.c,
 * k_cos.c, s_atan.c: minimax coefficients published with the library), evaluated here in fma-Horner form.
 */
#ifndef NBP_MATH_H
#define NBP_MATH_H

#include <math.h>
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define NBPM_FN __host__ __device__ __forceinline__
#else
#define NBPM_FN static inline
#endif

/* One rounding per written operation in every function below (the multiply-adds that are wanted are written as fma()):
 * clang takes the pragma at the top of a compound statement; gcc has no equivalent pragma and is given -ffp-contract=off
 * on the command line (oracle/Makefile). */
#if defined(__clang__)
#define NBPM_EXACT _Pragma("clang fp contract(off)")
#else
#define NBPM_EXACT
#endif

#define NBPM_PI 3.14159265358979323846
#define NBPM_TWO_PI 6.28318530717958647692

NBPM_FN int64_t nbpm_bits(double x) {
  int64_t b;
  __builtin_memcpy(&b, &x, 8);
  return b;
}
NBPM_FN double nbpm_from_bits(int64_t b) {
  double x;
  __builtin_memcpy(&x, &b, 8);
  return x;
}

/* ------------------------------------------------------------------------------------------------------------------
 * log(x), x positive, normal, finite.  x = 2^k m with m in [sqrt(2)/2, sqrt(2)); f = m - 1, s = f / (2 + f):
 * log(m) = f - f^2/2 + s (f^2/2 + R(s^2)), R = the degree-14 even minimax polynomial of fdlibm's e_log.c.
 * ------------------------------------------------------------------------------------------------------------------ */
NBPM_FN double nbpm_log(double x) {
  NBPM_EXACT
  const int64_t b = nbpm_bits(x);
  int32_t hx = (int32_t)(b >> 32);
  int32_t k = (hx >> 20) - 1023;
  hx &= 0x000fffff;
  const int32_t i = (hx + 0x95f64) & 0x100000; /* mantissa above sqrt(2): halve it, k + 1 */
  k += i >> 20;
  const double m = nbpm_from_bits(((int64_t)(hx | (i ^ 0x3ff00000)) << 32) | (b & 0xffffffffll));
  const double f = m - 1.0;
  const double dk = (double)k;
  const double s = f / (2.0 + f);
  const double z = s * s, w = z * z;
  const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                            6.666666666666735130e-01);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  /* k ln2 in two parts (ln2_hi has 32 trailing zero bits: k * ln2_hi is exact) */
  return fma(dk, 6.93147180369123816490e-01, -((hfsq - fma(s, hfsq + R, dk * 1.90821492927058770002e-10)) - f));
}

/* ------------------------------------------------------------------------------------------------------------------
 * sin and cos of an angle of moderate size.  Reduction by pi/2 in two parts (Cody-Waite; k * pio2_1 is exact for
 * |k| < 2^20), then the fdlibm kernels on |r| <= pi/4.  No branches.
 * ------------------------------------------------------------------------------------------------------------------ */
NBPM_FN void nbpm_sincos(double a, double *sn, double *cs) {
  NBPM_EXACT
  const double t = fma(a, 6.36619772367581382433e-01, 6755399441055744.0); /* a * 2/pi + 1.5 * 2^52: the integer in the low word */
  const int32_t k = (int32_t)(uint32_t)(nbpm_bits(t) & 0xffffffffll);
  const double kf = t - 6755399441055744.0;
  double r = fma(kf, -1.57079632673412561417e+00, a); /* pio2_1 (33 bits) */
  r = fma(kf, -6.07710050650619224932e-11, r);        /* pio2_1t */
  const double z = r * r;
  const double ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08), 2.75573137070700676789e-06),
                                         -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
  const double s = fma(r * z, ps, r);
  const double pc = z * fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09), -2.75573143513906633035e-07),
                                          2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
  const double ar = fabs(r);
  /* cos near pi/4: 1 - (z/2 - qq) with qq ~ |r|/4 taken off both terms first (keeps < 1 ulp up to pi/4) */
  const double q4 = nbpm_from_bits((int64_t)((uint64_t)(uint32_t)((int32_t)(nbpm_bits(ar) >> 32) - 0x00200000) << 32));
  const double qx = (ar > 0.78125) ? 0.28125 : q4;
  const double qq = (ar < 0.3) ? 0.0 : qx;
  const double c = (1.0 - qq) - ((0.5 * z - qq) - z * pc);
  const int swap = k & 1;
  const double ss = swap ? c : s, cc = swap ? s : c;
  *sn = (k & 2) ? -ss : ss;
  *cs = ((k + 1) & 2) ? -cc : cc;
}

/* ------------------------------------------------------------------------------------------------------------------
 * atan(t), t >= 0 finite: four break points, one division, the odd minimax polynomial of fdlibm's s_atan.c.
 * ------------------------------------------------------------------------------------------------------------------ */
NBPM_FN double nbpm_atan_nonneg(double ax) {
  NBPM_EXACT
  /* r = (ax - c) / (1 + c ax) for c = 0, 1/2, 1, 3/2 and -1 / ax beyond 39/16, by selects: one division */
  const int id = (ax < 0.4375) ? -1 : (ax < 0.6875) ? 0 : (ax < 1.1875) ? 1 : (ax < 2.4375) ? 2 : 3;
  const double num = (id < 0) ? ax : (id == 0) ? (2.0 * ax - 1.0) : (id == 1) ? (ax - 1.0) : (id == 2) ? (ax - 1.5) : -1.0;
  const double den = (id < 0) ? 1.0 : (id == 0) ? (2.0 + ax) : (id == 1) ? (ax + 1.0) : (id == 2) ? fma(1.5, ax, 1.0) : ax;
  const double r = num / den;
  const double z = r * r, w = z * z;
  const double s1 = z * fma(w, fma(w, fma(w, fma(w, fma(w, 1.62858201153657823623e-02, 4.97687799461593236017e-02), 6.66107313738753120669e-02),
                                          9.09088713343650656196e-02), 1.42857142725034663711e-01), 3.33333333333329318027e-01);
  const double s2 = w * fma(w, fma(w, fma(w, fma(w, -3.65315727442169155270e-02, -5.83357013379057348645e-02), -7.69187620504482999495e-02),
                                   -1.11111104054623557880e-01), -1.99999999998764832476e-01);
  const double hi = (id == 0) ? 4.63647609000806093515e-01 : (id == 1) ? 7.85398163397448278999e-01 : (id == 2) ? 9.82793723247329054082e-01
                                                                                                             : 1.57079632679489655800e+00;
  const double lo = (id == 0) ? 2.26987774529616870924e-17 : (id == 1) ? 3.06161699786838301793e-17 : (id == 2) ? 1.39033110312309984516e-17
                                                                                                             : 6.12323399573676603587e-17;
  const double p = r * (s1 + s2);
  return (id < 0) ? r - p : hi - ((p - lo) - r);
}

/* atan2(y, x) of finite arguments; atan2(0, 0) = 0 (an empty resultant of a circular mean) */
NBPM_FN double nbpm_atan2(double y, double x) {
  NBPM_EXACT
  const double ax = fabs(x), ay = fabs(y);
  double z;
  if (ay == 0.0) z = 0.0;
  else if (ax == 0.0) z = 1.57079632679489655800e+00;
  else {
    const int32_t ex = (int32_t)((nbpm_bits(ax) >> 52) & 0x7ff), ey = (int32_t)((nbpm_bits(ay) >> 52) & 0x7ff);
    if (ey - ex > 60) z = 1.57079632679489655800e+00; /* |y / x| > 2^60 */
    else if (x < 0.0 && ey - ex < -60) z = 0.0;       /* |y / x| < 2^-60 and the answer is +-pi */
    else z = nbpm_atan_nonneg(ay / ax);
  }
  const double pi_lo = 1.2246467991473531772e-16;
  if (x < 0.0 || (x == 0.0 && nbpm_bits(x) < 0)) z = NBPM_PI - (z - pi_lo);
  return (y < 0.0 || (y == 0.0 && nbpm_bits(y) < 0)) ? -z : z;
}

/* ------------------------------------------------------------------------------------------------------------------
 * Manifolds.sym_rem: mod(a + pi, 2 pi) - pi, into [-pi, pi); the identity on the principal interval.  Sums and
 * differences of wrapped angles stay within a few pi, where fmod is one exact subtraction (Sterbenz), so that range is
 * done with selects; anything larger (or NaN) takes fmod, which is exact as well.
 * ------------------------------------------------------------------------------------------------------------------ */
NBPM_FN double nbpm_wrap_pi(double a) {
  NBPM_EXACT
  const double t = a + NBPM_PI;
  double r = (t >= NBPM_TWO_PI) ? t - NBPM_TWO_PI : t;
  r = (t < 0.0) ? t + NBPM_TWO_PI : r;
  double res = (a >= -NBPM_PI && a < NBPM_PI) ? a : r - NBPM_PI;
  if (!(fabs(a) < 2.9 * NBPM_PI)) {
    double q = fmod(t, NBPM_TWO_PI);
    if (q < 0) q += NBPM_TWO_PI;
    res = q - NBPM_PI;
  }
  return res;
}

/* two standard normals from two uniforms in (0, 1) (Box-Muller) */
NBPM_FN void nbpm_box_muller(double ua, double ub, double *na, double *nb) {
  NBPM_EXACT
  const double r = sqrt(-2.0 * nbpm_log(ua));
  double s, c;
  nbpm_sincos(NBPM_TWO_PI * ub, &s, &c);
  *na = r * c;
  *nb = r * s;
}

#endif /* NBP_MATH_H */
