// nbp_device.h -- CDNA4 (gfx950) device code for libnbp.
//
// Hand-written HIP for the per-clique nonparametric Chapman-Kolmogorov hot path of
// IncrementalInference.jl (IIF).  Two kernels do all the arithmetic:
//
//   nbp_proposal_kernel : one workgroup per `approxConvBelief` (ApproxConv.jl:4-45)
//                         = measurement sampling + hypothesis recipe + entropy injection +
//                           per-particle numerical solve + LCV bandwidth fit.
//   nbp_product_kernel  : one workgroup per `AMP.manifoldProduct` (GraphProductOperations.jl:53-60)
//                         = KD-tree build + multiscale Gibbs label sampling + product draw +
//                           LCV bandwidth fit + belief write-back.
//
// Mapping: one lane per particle (wave64; workgroup = roundup(N,64) lanes), particle coordinates
// SoA in HBM (coalesced 8 B/lane loads), every operand staged once into LDS, cross-lane
// reductions with DPP/shuffle inside a wave and a fixed-order LDS tree across waves (bitwise
// reproducible run to run).  No MFMA: this is particle-wise nonlinear residual evaluation and
// O(N^2) kernel sums in FP64, not a dense contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/nbp.h"
// log / sin / cos / atan2 / wrap of the values that travel from op to op: ONE definition for the kernels and the CPU checker
// (the header says why; this library is compiled with -ffp-contract=off, one rounding per written operation, for the same reason)
#include "../../include/nbp_math.h"

#define NBP_PI 3.14159265358979323846
#define NBP_TWO_PI 6.28318530717958647692

// RNG purposes -- shared contract with the CPU oracle (DESIGN.md "RNG")
#define PURP_MEAS 1
#define PURP_MIXLBL 2
#define PURP_HYPO 3
#define PURP_ENTROPY 4
#define PURP_KDESEL 5
#define PURP_KDENOISE 6
#define PURP_PGIBBS 9
#define PURP_PFINAL 10
#define PURP_ANYN 11      // _getindex_anyn: the element a particle beyond the end of a shorter operand reads
#define PURP_OLDSEL 12    // sample(oldBel, nn): kernel pick / noise of the top-up of a belief with fewer than N points
#define PURP_OLDNOISE 13
#define PURP_PLEVEL 14    // samplePoint! between the levels of the product sampler (k = 2 l, 2 l + 1)
#define PURP_PINDEX 15    // one block per (sample, pass, density), k = pass * NBP_MAXF + density: ua -> sampleIndices!, ub -> the first sweep's sampleIndex
#define NBP_TAG 0x4E4250u

#define NBP_MAXLEVELS 12
#define NBP_RED 64

// Level tables of the balanced KD-tree over N leaves: data-independent, built once per context.
struct nbp_levels {
  int32_t N, L;                       // L = depth (levels 0..L)
  int32_t cnt[NBP_MAXLEVELS];         // nodes per level
  int32_t off[NBP_MAXLEVELS];         // offset of the level in the node arrays
  const int32_t *node_lo;             // [total] first leaf position of the node
  const int32_t *node_hi;             // [total] one past the last
  const int32_t *node_child;          // [total] index (within the next level) of the LAST child
  const int32_t *pos_node;            // [(L+1)*N] node (within its level) owning position i
  const double *node_logw;            // [total] log((hi-lo)/N)
  const double *node_w;               // [total] (hi-lo)/N
};

struct nbp_counters {
  unsigned long long solves, nonconverged, nan_results, residual_evals, lcv_evals;
  unsigned long long lcv_evals_f32;  // single-precision bracketing evaluations of the bandwidth searches (neg_loo_ll_f32)
  unsigned long long flags;          // set by the host at context creation, never reset: bit 0 = every evaluation in double precision
};

// ------------------------------------------------------------------------------------------------
// Philox4x32-10
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
  for (int r = 0; r < 10; r++) {
    uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ double u01_from(uint32_t lo, uint32_t hi) {
  unsigned long long v = ((unsigned long long)hi << 32) | lo;
  return ((double)(v >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}

__device__ __forceinline__ void uniform_pair(uint64_t seed, uint32_t n, uint32_t purpose, uint32_t k,
                                             double &ua, double &ub) {
  uint32_t o[4];
  philox4x32_10(n, purpose, k, NBP_TAG, (uint32_t)seed, (uint32_t)(seed >> 32), o);
  ua = u01_from(o[0], o[1]);
  ub = u01_from(o[2], o[3]);
}

__device__ __forceinline__ void sincos_fast(double a, double *sn, double *cs);
// two standard normals: Box-Muller on the shared log and sincos (include/nbp_math.h) -- the CPU checker's normals bit for bit
__device__ __forceinline__ void normal_pair(uint64_t seed, uint32_t n, uint32_t purpose, uint32_t k,
                                            double &na, double &nb) {
  double ua, ub;
  uniform_pair(seed, n, purpose, k, ua, ub);
  nbpm_box_muller(ua, ub, &na, &nb);
}

// the same as a leaf call (the product sampler draws a point per tree level: inlined, the constants of log and sincos are
// hoisted out of the level loop and kept live through the Gibbs sweeps -- 40-50 registers spilled at four waves per SIMD)
__device__ __attribute__((noinline)) double2 normal_pair_call(uint64_t seed, uint32_t n, uint32_t purpose, uint32_t k) {
  double ua, ub, na, nb;
  uniform_pair(seed, n, purpose, k, ua, ub);
  nbpm_box_muller(ua, ub, &na, &nb);
  return make_double2(na, nb);
}

// a uniform pair as a leaf call (the latency-mode product kernels hold five manifolds' instances of the sampler at 252-256
// registers: the block inlined at the top of every pass put 16 B per lane into scratch)
__device__ __attribute__((noinline)) double2 uniform_pair_call(uint64_t seed, uint32_t n, uint32_t purpose, uint32_t k) {
  double ua, ub;
  uniform_pair(seed, n, purpose, k, ua, ub);
  return make_double2(ua, ub);
}

// ------------------------------------------------------------------------------------------------
// particle count of a belief: a slot holds up to N points; slot[3N + 6] = the count, 0 meaning N (what every
// kernel output has).  Beliefs with fewer points come from the host (nbp_belief_write): a variable initialised with
// another N, a message of a clique solved with another N.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int slot_count(const double *s, int N) {
  const double c = s[3 * N + 6];
  return (c > 0.0 && c < (double)N) ? (int)c : N;
}
// _getindex_anyn(vec, n) = vec[n <= len ? n : rand(1:len)] (NumericalCalculations.jl:377-381).  The reference draws
// a new element at every evaluation of the residual; here the draw is one per (op, particle, operand), which keeps the
// objective of a particle's search a function.
__device__ __forceinline__ int anyn_index(int n, int cnt, uint64_t seed, int operand) {
  if (n < cnt) return n;
  double ua, ub;
  uniform_pair(seed, n, PURP_ANYN, (uint32_t)operand, ua, ub);
  const int i = (int)(ua * cnt);
  return i < cnt ? i : cnt - 1;
}

// ------------------------------------------------------------------------------------------------
// manifolds in tangent coordinates at the identity
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int mani_dim(int m) { return m == NBP_SE2 ? 3 : (m == NBP_CIRCULAR ? 1 : m); }
__device__ __forceinline__ bool is_circ(int m, int d) {
  return (m == NBP_CIRCULAR && d == 0) || (m == NBP_SE2 && d == 2);
}
// Manifolds.sym_rem -> [-pi, pi); exact identity on the principal interval
__device__ __forceinline__ double readlane_f64(double v, int lane) {  // lane: wave-uniform
  const long long b = __double_as_longlong(v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, lane);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// Manifolds.sym_rem -> [-pi, pi): nbpm_wrap_pi (include/nbp_math.h: selects within a few pi, fmod beyond)
__device__ __forceinline__ double wrap_pi(double a) { return nbpm_wrap_pi(a); }

// sin and cos of an angle of moderate size (|a| < ~1e5; here: sums of a few wrapped angles): nbpm_sincos (include/nbp_math.h) --
// Cody-Waite reduction by pi/2 in two parts + the fdlibm kernels, ~35 FP64 operations, no branches, no Payne-Hanek path.  The
// per-particle searches on SE(2) call it at every evaluation of a reverse residual; the CPU checker calls the same function.
__device__ __forceinline__ void sincos_fast(double a, double *sn, double *cs) { nbpm_sincos(a, sn, cs); }

// exp(x) for x <= ~0 in the O(N^2) kernel sums (arguments are -d^2/(2h^2) or weights relative to
// their max).  Table-driven: x = (32k + j) ln2/32 + r, |r| <= ln2/64, exp(x) = 2^k * 2^(j/32) * p(r)
// with a degree-6 polynomial; the integer n = 32k + j is taken from the low mantissa bits after
// adding 1.5*2^52 and 2^k is applied by an integer add on the exponent field, so every operation is
// a full-rate FP64/INT32 VALU op (no v_rndne/v_cvt/v_ldexp quarter-rate ops).  <= 2 ulp.
// `tab` = the 256-entry table of nbp_lcv_table.h staged in LDS (nbp_exp_tab_init): entry k holds the bits of 2^(k/256)
// with k << 12 taken off the high word, so adding n << 12 puts 2^(n >> 8) on without masking (tools/gen_lcv_table.py)
#include "nbp_lcv_table.h"
#define NBP_EXPTAB NBP_LCVTAB
__device__ __forceinline__ void nbp_exp_tab_init(double *tab) {
  for (int q = threadIdx.x; q < NBP_LCVTAB; q += blockDim.x) tab[q] = __longlong_as_double((long long)NBP_LCV_TAB[q]);
}
// the bandwidth fit's own table: 2^NBP_LCV_TLOG entries.  8: the table of exp_nonpos and a quartic remainder polynomial;
// 11 (experiment): 16 KB of LDS buy one Horner step per kernel pair -- the remainder is within ln2/4096 and a cubic reaches
// 3.4e-17.  Measured (profiles/r04_lcv_pair_loop_experiments.txt): the same bandwidths bit for bit and the same 0.78 ps per
// pair -- the pair loop is not bound by the number of its vector instructions
#ifndef NBP_LCV_TLOG
#define NBP_LCV_TLOG 8
#endif
#define NBP_FITTAB (1 << NBP_LCV_TLOG)
__device__ __forceinline__ void nbp_fit_tab_init(double *tab) {
#if NBP_LCV_TLOG == 11
  for (int q = threadIdx.x; q < NBP_FITTAB; q += blockDim.x) tab[q] = __longlong_as_double((long long)NBP_LCV_TAB2K[q]);
#elif NBP_LCV_TLOG == 8
  for (int q = threadIdx.x; q < NBP_FITTAB; q += blockDim.x) tab[q] = __longlong_as_double((long long)NBP_LCV_TAB[q]);
#else
#error "NBP_LCV_TLOG: 8 or 11"
#endif
}
__device__ __forceinline__ double exp_nonpos(double x, const double *tab) {
  // clamp instead of branching: exp(-700) = 1e-304 is as good as 0 for every sum it enters
  x = fmax(x, -700.0);
  const double t = fma(x, 369.3299304675746, 6755399441055744.0);  // 256/ln2
  const int n = __double2loint(t);
  const double tf = t - 6755399441055744.0;
  // one-constant reduction: |tf| <= 2.6e5, so the rounding of ln2/256 (<= 2.2e-19) moves r by <= 6e-14
  // at the clamp and by < 1e-15 where the result still matters to a sum
  const double r = fma(tf, -2.7076061740622863e-03, x);
  // |r| <= ln2/512: degree 4 (next term 3.8e-17).  Horner with the coefficients pinned in SGPRs: one v_fma_f64 per step
  // (hipcc otherwise keeps them in VGPRs and pays a v_mov_b64 + v_fmac_f64 per step)
  double p;
#define NBP_FMA_S(dst, a, b, cst) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(dst) : "v"(a), "v"(b), "s"(cst))
  {
    const double c4 = 4.16666666666666666667e-02, c3 = 1.66666666666666666667e-01;
    double q;
    NBP_FMA_S(q, c4, r, c3);
    p = fma(q, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
  }
#undef NBP_FMA_S
  const double w = tab[n & (NBP_LCVTAB - 1)];
  int hi;
  asm("v_lshl_add_u32 %0, %1, 12, %2" : "=v"(hi) : "v"(n), "v"(__double2hiint(w)));  // 2^(n >> 8) onto the exponent field
  return __hiloint2double(hi, __double2loint(w)) * p;
}

// 1/sqrt(x) of a positive, finite, normal x (products of node variances): the device library's rsqrt -- v_rsq_f64 and one
// coupled refinement step -- without its class test and the two selects behind it (three of ~50 vector instructions per
// node weight of the product sampler's sweep); the same value bit for bit on such x
__device__ __forceinline__ double rsqrt_pos(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  const double e = fma(y0 * -x, y0, 1.0);
  return fma(y0 * e, fma(e, 0.375, 0.5), y0);
}

// ------------------------------------------------------------------------------------------------
// workgroup reductions: wave64 butterfly + fixed-order combine across waves (deterministic)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
  return v;
}

// `red` = LDS scratch of NBP_RED doubles: [0,32) wave partials, [32,48) broadcast slots.
// All threads of the block must call.
__device__ __forceinline__ double block_sum(double v, double *red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_sum(v);
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  double t = red[0];
  for (int i = 1; i < nw; i++) t += red[i];
  return t;
}
__device__ __forceinline__ double block_min(double v, double *red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_min(v);
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  double t = red[0];
  for (int i = 1; i < nw; i++) t = fmin(t, red[i]);
  return t;
}
__device__ __forceinline__ double block_max(double v, double *red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  v = wave_max(v);
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  double t = red[0];
  for (int i = 1; i < nw; i++) t = fmax(t, red[i]);
  return t;
}

// a double through one DPP pattern (lanes without a source, or masked out, receive 0)
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, BANK_MASK, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, BANK_MASK, false);
  return __hiloint2double(hi, lo);
}
// inclusive prefix sum over a wave64 in lane order, the DPP way: three row shifts of the input, shifts by 4 and 8 of the
// partial sums inside each row of 16, then the row totals broadcast to the rows behind (row_bcast:15 / :31)
__device__ __forceinline__ double wave_inclusive_scan(double x) {
  double s = x + dpp_f64<0x111, 0xf, 0xf>(x);   // row_shr:1
  s += dpp_f64<0x112, 0xf, 0xf>(x);             // row_shr:2
  s += dpp_f64<0x113, 0xf, 0xf>(x);             // row_shr:3
  s += dpp_f64<0x114, 0xf, 0xe>(s);             // row_shr:4, lanes 4..15 of every row
  s += dpp_f64<0x118, 0xf, 0xc>(s);             // row_shr:8, lanes 8..15
  s += dpp_f64<0x142, 0xa, 0xf>(s);             // row_bcast:15 into rows 1 and 3
  s += dpp_f64<0x143, 0xc, 0xf>(s);             // row_bcast:31 into rows 2 and 3
  return s;
}
// ---- reductions over an aligned group of HL adjacent lanes (HL = 2 .. 32: the helper lanes of one product sample) --------
// Data moves through DPP (quad permutes, row mirrors, row shifts, row_bcast:15) instead of ds_bpermute: a dependent chain of
// five bpermute steps is ~600 cycles of LDS-crossbar latency, the same steps on the VALU ~150 -- what a lone product's draws
// are made of.  Only the step across the two rows of a 32-lane group, and the broadcast of a group's last lane for HL >= 8,
// still take a bpermute.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ int dpp_i32(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, ROW_MASK, BANK_MASK, false);
}
template <int HL>
__device__ __forceinline__ double group_max(double v) {  // every lane of the group receives the maximum
  if (HL >= 2) v = fmax(v, dpp_f64<0xB1, 0xf, 0xf>(v));    // quad_perm [1,0,3,2]
  if (HL >= 4) v = fmax(v, dpp_f64<0x4E, 0xf, 0xf>(v));    // quad_perm [2,3,0,1]
  if (HL >= 8) v = fmax(v, dpp_f64<0x141, 0xf, 0xf>(v));   // row_half_mirror: the other quad of the 8
  if (HL >= 16) v = fmax(v, dpp_f64<0x140, 0xf, 0xf>(v));  // row_mirror: the other half of the row
  if (HL >= 32) v = fmax(v, __shfl_xor(v, 16, 32));
  return v;
}
template <int HL>
__device__ __forceinline__ int group_max(int v) {
  if (HL >= 2) v = max(v, dpp_i32<0xB1, 0xf, 0xf>(v));
  if (HL >= 4) v = max(v, dpp_i32<0x4E, 0xf, 0xf>(v));
  if (HL >= 8) v = max(v, dpp_i32<0x141, 0xf, 0xf>(v));
  if (HL >= 16) v = max(v, dpp_i32<0x140, 0xf, 0xf>(v));
  if (HL >= 32) v = max(v, __shfl_xor(v, 16, 32));
  return v;
}
// inclusive prefix sum in lane order inside the group; h = the lane's place in its group; *total = the group's sum
template <int HL>
__device__ __forceinline__ double group_inclusive_scan(double x, int h, double *total) {
  const int r = h & 15;  // place in the row of 16 (groups are aligned: up to 16 lanes share a row)
  double s = x, v;
  if (HL >= 2) { v = dpp_f64<0x111, 0xf, 0xf>(s); if (r >= 1 && (HL >= 16 || h >= 1)) s += v; }  // row_shr:1
  if (HL >= 4) { v = dpp_f64<0x112, 0xf, 0xf>(s); if (r >= 2 && (HL >= 16 || h >= 2)) s += v; }  // row_shr:2
  if (HL >= 8) { v = dpp_f64<0x114, 0xf, 0xf>(s); if (r >= 4 && (HL >= 16 || h >= 4)) s += v; }  // row_shr:4
  if (HL >= 16) { v = dpp_f64<0x118, 0xf, 0xf>(s); if (r >= 8) s += v; }                        // row_shr:8
  if (HL >= 32) s += dpp_f64<0x142, 0xa, 0xf>(s);  // row_bcast:15: the first row's total into the second row of the group
  if (HL == 2) *total = dpp_f64<0xF5, 0xf, 0xf>(s);       // quad_perm [1,1,3,3]
  else if (HL == 4) *total = dpp_f64<0xFF, 0xf, 0xf>(s);  // quad_perm [3,3,3,3]
  else *total = __shfl(s, HL - 1, HL);
  return s;
}

// exclusive prefix sum over the workgroup in lane order (wave scans, wave totals through `red`); *total = the sum
__device__ __forceinline__ double block_exclusive_scan(double v, double *red, double *total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  const double inc = wave_inclusive_scan(v);
  __syncthreads();
  if (lane == 63) red[w] = inc;
  __syncthreads();
  double off = 0, tot = 0;
  for (int q = 0; q < nw; q++) {
    if (q < w) off += red[q];
    tot += red[q];
  }
  *total = tot;
  return off + inc - v;
}

// ------------------------------------------------------------------------------------------------
// statistics: mean(M, pts, GeodesicInterpolation()) and calcStdBasicSpread
// (services/VariableStatistics.jl:22-36).  x = LDS or global SoA [d*stride + n].
// Euclidean coordinates: arithmetic mean by tree reduction (equal to the running mean up to
// rounding).  Circular coordinates: the running geodesic interpolation is order dependent, so lane
// 0 walks it sequentially exactly like Manifolds.jl does.
// ------------------------------------------------------------------------------------------------
// Spread around the circle: the running mean m_i = m_{i-1} + wrap(x_i - m_{i-1}) / (i + 1) is the running ARITHMETIC mean
// of the lifted points X_i = x_i + 2 pi k_i, where k_i puts X_i within pi of m_{i-1}.  The lifts depend on the
// trajectory, the trajectory is a prefix mean of the lifts: iterate (prefix sums -> lifts) to the fixed point, which
// by induction on i is the sequential walk's own assignment (every sweep fixes at least one more leading lift; on
// beliefs spread over the circle 5-15 sweeps of a workgroup-wide scan instead of N dependent steps; equal to the walk
// up to the rounding of the sums).  A belief that needs more sweeps is walked as before.
// A real call: inlined into the proposal kernel the sweeps cost that kernel 13 VGPRs and 48 B of scratch per lane.
// Returns the mean, or NaN when the lifts have not settled after 32 sweeps (by value: a result handed back through a
// pointer to a local of the caller costs every kernel that inlines the caller a stack slot in scratch).
__device__ __attribute__((noinline)) double mean_geodesic_lifts(const double *x, int N, double *red) {
  // One barrier per sweep: the wave totals and the "a lift changed in the previous sweep" flags go through
  // alternating halves of `red`, and the loop ends one sweep after the last change (that sweep's sums are the final ones).
  const int i = threadIdx.x, lane = i & 63, w = i >> 6, nw = (blockDim.x + 63) >> 6;
  const double xi = (i < N) ? x[i] : 0.0;
  double ki = 0.0, tot = 0.0;
  bool changed = true, fixed = false;  // `changed`: this lane's lift moved in the previous sweep
  if (i < 64) {
    // the head of the trajectory, where the weights 1/(i+1) are large and the lifts take most sweeps to settle, is
    // iterated by wave 0 alone: no barriers, no LDS -- the workgroup-wide sweeps below then start from a settled head
    for (int sweep = 0; sweep < 24; sweep++) {
      const double Xi = (i < N) ? fma(NBP_TWO_PI, ki, xi) : 0.0;
      const double before = wave_inclusive_scan(Xi) - Xi;
      double kn = ki;
      if (i >= 1 && i < N) kn = ki + rint((before / (double)i - Xi) * (1.0 / NBP_TWO_PI));
      const bool moved = __builtin_amdgcn_ballot_w64(kn != ki) != 0;
      ki = kn;
      if (!moved) break;
    }
  }
  __syncthreads();                      // the reductions above are done with `red`
  for (int sweep = 0; sweep < 32; sweep++) {
    double *buf = red + (sweep & 1) * 32;
    const double Xi = (i < N) ? fma(NBP_TWO_PI, ki, xi) : 0.0;
    const double inc = wave_inclusive_scan(Xi);
    const bool wave_changed = __builtin_amdgcn_ballot_w64(changed) != 0;
    if (lane == 63) buf[w] = inc;
    if (lane == 0) buf[16 + w] = wave_changed ? 1.0 : 0.0;
    __syncthreads();
    double off = 0.0, any = 0.0;
    tot = 0.0;
    for (int q = 0; q < nw; q++) {
      if (q < w) off += buf[q];
      tot += buf[q];
      any += buf[16 + q];
    }
    if (sweep > 0 && any == 0.0) { fixed = true; break; }  // nothing moved last time: these are the sums of the fixed point
    const double before = off + inc - Xi;
    double kn = ki;
    if (i >= 1 && i < N) kn = ki + rint((before / (double)i - Xi) * (1.0 / NBP_TWO_PI));
    changed = kn != ki;
    ki = kn;
  }
  __syncthreads();  // `red` is free again
  return fixed ? wrap_pi(tot / (double)N) : __longlong_as_double(0x7ff8000000000000ll);
}

__device__ __forceinline__ double mean_geodesic_coord(const double *x, int N, int manifold, int d, double *red) {
  double mu;
  if (is_circ(manifold, d)) {
    // When every point lies within an arc shorter than pi (the rule for a belief that is not spread around the circle),
    // no step of the running geodesic mean wraps relative to the first point, and the recurrence
    // m <- m + (x_i - m) / (i + 1) is the arithmetic mean of the offsets d_i = wrap(x_i - x_0): a parallel reduction
    // instead of N dependent steps (equal to the walk up to rounding).  Otherwise: the walk below.
    {
      const double x0 = x[0];
      const double di = (threadIdx.x < N) ? wrap_pi(x[threadIdx.x] - x0) : 0.0;
      // min and max in one pass: max(d) and max(-d) ride through the same shuffles and one barrier pair
      double hi = (threadIdx.x < N) ? di : -INFINITY, lo = (threadIdx.x < N) ? -di : -INFINITY;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        hi = fmax(hi, __shfl_xor(hi, o, 64));
        lo = fmax(lo, __shfl_xor(lo, o, 64));
      }
      {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
        __syncthreads();
        if (lane == 0) { red[w] = hi; red[16 + w] = lo; }
        __syncthreads();
        hi = red[0];
        lo = red[16];
        for (int i = 1; i < nw; i++) { hi = fmax(hi, red[i]); lo = fmax(lo, red[16 + i]); }
      }
      if (hi + lo < 3.0) {  // max - min, block-uniform
        const double mo = block_sum(di, red) / (double)N;
        return wrap_pi(x0 + mo);
      }
    }
    {  // spread around the circle: lifts and prefix means iterated to the walk's fixed point (mean_geodesic_lifts)
      const double mlift = mean_geodesic_lifts(x, N, red);
      if (mlift == mlift) return mlift;  // (block-uniform: every lane computes the same sums)
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      // Order-dependent running mean: N dependent steps.  Wave 0 walks it with every lane carrying the
      // same m; the points and the weights 1/(i+1) are produced 64 at a time by the lanes in parallel
      // (the divisions are off the chain) and handed to the walk through v_readlane.
      const int lane = threadIdx.x;
      double m = x[0];
      for (int base = 0; base < N; base += 64) {
        const int idx = base + lane;
        const double xv = idx < N ? x[idx] : 0.0;
        const double rv = 1.0 / (double)(idx + 1);
        const int cnt = (N - base < 64) ? N - base : 64;
        const int k0 = (base == 0) ? 1 : 0;
        // Speculate that no step of this block of 64 wraps (the rule, for a belief that does not straddle
        // +-pi): the chain is then one subtract and one fma per step, the range checks ride beside it.
        // A block that did wrap is walked again with the wraps in place -- same values either way.
        const double m0 = m;
        bool inside = true;
        for (int k = k0; k < cnt; k++) {
          const double xi = readlane_f64(xv, k), ri = readlane_f64(rv, k);
          const double dl = xi - m;
          m = fma(dl, ri, m);
          inside = inside && (fabs(dl) < NBP_PI) && (fabs(m) < NBP_PI);
        }
        if (!inside) {
          m = m0;
          for (int k = k0; k < cnt; k++) {
            const double xi = readlane_f64(xv, k), ri = readlane_f64(rv, k);
            // every lane carries the same values: the (frequent) no-wrap case is skipped by a scalar branch
            double dl = xi - m;
            if (__builtin_amdgcn_ballot_w64(!(fabs(dl) < NBP_PI)) != 0) dl = wrap_pi(dl);
            m = fma(dl, ri, m);
            if (__builtin_amdgcn_ballot_w64(!(fabs(m) < NBP_PI)) != 0) m = wrap_pi(m);
          }
        }
      }
      if (lane == 0) red[32] = m;
    }
    __syncthreads();
    mu = red[32];
  } else {
    double v = (threadIdx.x < N) ? x[threadIdx.x] : 0.0;
    mu = block_sum(v, red) / (double)N;
  }
  return mu;
}

// default mean(M, pts): arithmetic / extrinsic circular mean
__device__ __forceinline__ double mean_default_coord(const double *x, int N, int manifold, int d, double *red) {
  if (is_circ(manifold, d)) {
    double s = 0, c = 0;
    if (threadIdx.x < N) sincos_fast(x[threadIdx.x], &s, &c);
    double ss = block_sum(s, red), sc = block_sum(c, red);
    return nbpm_atan2(ss, sc);
  }
  double v = (threadIdx.x < N) ? x[threadIdx.x] : 0.0;
  return block_sum(v, red) / (double)N;
}

__device__ __forceinline__ double std_basic_spread(const double *x, int stride, int N, int manifold, double *red) {
  const int D = mani_dim(manifold);
  double acc = 0;
  for (int d = 0; d < D; d++) {
    double mu = mean_geodesic_coord(x + d * stride, N, manifold, d, red);
    if (threadIdx.x < N) {
      double dl = x[d * stride + threadIdx.x] - mu;
      if (is_circ(manifold, d)) {
        dl = wrap_pi(dl);
        acc += (manifold == NBP_SE2 ? 2.0 : 1.0) * dl * dl;
      } else
        acc += dl * dl;
    }
  }
  double sg = sqrt(block_sum(acc, red) / (double)(N - 1));
  return (1e-10 < sg) ? sg : 1.0;
}

// ------------------------------------------------------------------------------------------------
// residual functors -> sum(r.^2)   (CalcFactorNormSq, NumericalCalculations.jl:386-396)
// KIND and DN are compile-time so that every array lives in registers (no scratch).
// ------------------------------------------------------------------------------------------------
template <int KIND, int DN>
struct objective_t {
  double z[3], other[3];
  double sn_fixed, cs_fixed;  // SE(2), solve_b: sin/cos of the fixed pose's heading (hoisted out of the search)
  int solve_b;
  int rmask = 7;  // SE(2): the residual components that count (a partial ManifoldFactor: its `.partial` coordinates)
  unsigned int evals;
  __device__ __forceinline__ double normsq(const double *a, const double *b) const {
    double acc = 0;
    if (KIND == NBP_F_LINREL) {  // Factors/LinearRelative.jl:42-49
#pragma unroll
      for (int d = 0; d < DN; d++) {
        double r = z[d] - (b[d] - a[d]);
        acc += r * r;
      }
    } else if (KIND == NBP_F_CIRCULAR) {  // Factors/Circular.jl:24-28
      double r = wrap_pi((a[0] + z[0]) - b[0]);
      acc = r * r;
    } else if (KIND == NBP_F_SE2) {  // Factors/GenericFunctions.jl:39-44
      double s, c;
      if (solve_b) { s = sn_fixed; c = cs_fixed; }  // a is the fixed pose: its rotation was evaluated once
      else sincos_fast(a[2], &s, &c);
      double r0 = (a[0] + c * z[0] - s * z[1]) - b[0];
      double r1 = (a[1] + s * z[0] + c * z[1]) - b[1];
      double r2 = wrap_pi((a[2] + z[2]) - b[2]);
      if (rmask == 7) acc = r0 * r0 + r1 * r1 + r2 * r2;
      else {  // the same order of additions as the full sum, components outside the mask left out
        acc = 0;
        if (rmask & 1) acc += r0 * r0;
        if (rmask & 2) acc += r1 * r1;
        if (rmask & 4) acc += r2 * r2;
      }
    } else {  // NBP_F_EUCLIDDIST, Factors/EuclidDistance.jl:20
      double q = 0;
#pragma unroll
      for (int d = 0; d < DN; d++) q += (b[d] - a[d]) * (b[d] - a[d]);
      double r = z[0] - sqrt(q);
      acc = r * r;
    }
    return acc;
  }
  __device__ __forceinline__ double operator()(const double (&x)[DN]) {
    evals++;
    if (KIND == NBP_F_LINREL) {
      // z - (x - other) when x is the second variable, z - (other - x) = z + (x - other) when it is the first: the same
      // values as normsq() either way (a negation is exact), without a wave-uniform branch on solve_b in front of every
      // evaluation -- a lone wave (the slowest search of a workgroup) pays a taken branch with its instruction fetch
      // (the squares summed with one rounding per operation: the oracle's sum, and the same sum in every kernel the search is
      //  inlined into -- a contraction the compiler chooses per instance made the one-wave-per-proposal kernels differ from the
      //  workgroup kernels in one coordinate of a few hundred at 3e-12)
      const double sg = solve_b ? -1.0 : 1.0;
      double acc = 0;
#pragma unroll
      for (int d = 0; d < DN; d++) {
#pragma clang fp contract(off)
        const double r = fma(sg, x[d] - other[d], z[d]);
        const double rr = r * r;
        acc = acc + rr;
      }
      return acc;
    }
    // which end is searched is chosen value by value (wave-uniform selects): two array arguments picked at run time
    // would have to live in memory, i.e. in scratch
    double a[DN], b[DN];
#pragma unroll
    for (int d = 0; d < DN; d++) {
      a[d] = solve_b ? other[d] : x[d];
      b[d] = solve_b ? x[d] : other[d];
    }
    return normsq(a, b);
  }
};

// ------------------------------------------------------------------------------------------------
// Optim.NelderMead restated (Gao-Han adaptive parameters, AffineSimplexer a=0.025 b=0.5,
// g_tol 1e-8 on nmobjective, 1000 iterations) -- NumericalCalculations.jl:108,122-126.
// One lane = one particle; the simplex lives in registers.
// ------------------------------------------------------------------------------------------------
// The simplex is kept PHYSICALLY sorted by f (vertex 0 = best, vertex DN = worst) with
// compare-exchange steps on compile-time indices, so every access is a register access (an index
// permutation `ord[]` makes LLVM spill the simplex to scratch for the runtime-indexed reads).
// The arithmetic is Optim's, operation for operation (round 6): every vertex formula rounds once per operation (nm_lin; the
// library is compiled without contraction) and, in three dimensions, every vertex carries its slot of Optim's simplex array so
// that the centroid is summed in the order Optim's centroid! sums it (nm_centroid).  In two dimensions the coefficients are
// 2, 1/2, 1/2 and the centroid is a + b: any order and any contraction is the same search bit for bit.  The CPU checker runs
// the same search from the same start and ends on the same bits (tests/test_gpu_stagewise_parity.py); what it costs is in
// profiles/r06_nm_optim_order.txt.
// TB: ties broken by slot (true), or ties only NOTICED (false: `tied` is set when two values compared equal, and the caller
// runs the search again with TB -- two dimensions, where nothing else needs the slots and carrying them through every
// compare-exchange cost config 2's proposals 18 %: profiles/r06_nm_tie_order.txt)
template <int DN, bool OPT, bool TB>
__device__ __forceinline__ void nm_cswap(double (&sx)[DN + 1][DN], double (&f)[DN + 1], int (&id)[DN + 1], int i, bool &tied) {
  // the order is Optim's sortperm!(i_order, f_simplex): by value, EQUAL VALUES BY THEIR SLOT in the simplex array (Base's Perm
  // ordering breaks ties by index).  Ties do not happen between the values of a search that is converging; they do when a
  // search stalls with an objective of ~1e8 (a pose 1e4 away from its start): the values sit on a lattice of 1.5e-8 there, two
  // vertices tie, and which of them counts as the worst decides the rest of the search -- one particle of 29 136 fuzzed
  // proposals ended 0.1 from the oracle's before the slots broke the ties (tools/exp/fuzz_proposals.py, seed 19).
  bool sw = f[i] < f[i - 1];
  if constexpr (TB) sw = sw || (f[i] == f[i - 1] && id[i] < id[i - 1]);
  else tied = tied || (f[i] == f[i - 1]);
  const double fa = f[i - 1], fb = f[i];
  f[i - 1] = sw ? fb : fa;
  f[i] = sw ? fa : fb;
  if (DN > 2 || TB) {  // the vertex's place in Optim's simplex array travels with it (ties; nm_centroid in three dimensions)
    const int ia = id[i - 1], ib = id[i];
    id[i - 1] = sw ? ib : ia;
    id[i] = sw ? ia : ib;
  }
#pragma unroll
  for (int d = 0; d < DN; d++) {
    const double a = sx[i - 1][d], b = sx[i][d];
    sx[i - 1][d] = sw ? b : a;
    sx[i][d] = sw ? a : b;
  }
}
template <int DN, bool OPT, bool TB>
__device__ __forceinline__ void nm_sort_all(double (&sx)[DN + 1][DN], double (&f)[DN + 1], int (&id)[DN + 1], bool &tied) {
#pragma unroll
  for (int pass = 0; pass < DN; pass++)
#pragma unroll
    for (int i = DN; i >= 1 + pass; i--) nm_cswap<DN, OPT, TB>(sx, f, id, i, tied);
}
template <int DN, bool OPT, bool TB>
__device__ __forceinline__ void nm_sift_last(double (&sx)[DN + 1][DN], double (&f)[DN + 1], int (&id)[DN + 1], bool &tied) {
#pragma unroll
  for (int i = DN; i >= 1; i--) nm_cswap<DN, OPT, TB>(sx, f, id, i, tied);
}
// Optim's centroid!(c, simplex, h): the vertices other than the worst summed IN THE ORDER THE SIMPLEX ARRAY HOLDS THEM
// (a replaced vertex keeps the slot of the one it replaces), then rmul!(c, 1/n).  The simplex here is physically sorted
// by value, so each vertex carries its slot (`id`).  Two vertices: a + b commutes, nothing to do.  Three: only WHICH
// vertex is added last matters ((0 + p) + q is q + p) -- the one with the largest slot.  The rounding of this sum is
// what the vertex coordinates of a 3-D search inherit: summed in sorted order the search parts from Optim's (and the
// oracle's) in the last bit at the first iteration and by ~1e-8 at its end.
template <int DN, bool OPT>
__device__ __forceinline__ void nm_centroid(const double (&sx)[DN + 1][DN], const int (&id)[DN + 1], double (&xc)[DN]) {
  if (DN == 3) {
    const bool l0 = id[0] > id[1] && id[0] > id[2];
    const bool l1 = !l0 && id[1] > id[2];
#pragma unroll
    for (int d = 0; d < DN; d++) {
      NBPM_EXACT
      if constexpr (OPT) {
        // SUMS: the three sums "x added last" are all taken and the one Optim's order gives is selected -- six additions and two
        // selects per coordinate instead of four selects between elements of the simplex with their opaque register copies (below):
        // the same value bit for bit, ~6 instructions fewer per coordinate.  Config 5's proposals 87.0 -> 82.2 ms; the SE(2) search,
        // whose objective holds more registers, got slower with it (136.4 -> 140 ms) and keeps the selects
        // (profiles/r06_nm_centroid_forms.txt)
        const double a = sx[0][d], b = sx[1][d], c = sx[2][d];
        const double ra = (b + c) + a, rb = (a + c) + b, rc = (a + b) + c;
        xc[d] = (l0 ? ra : (l1 ? rb : rc)) * (1.0 / DN);
      } else {
        // (opaque copies: a select between elements of the simplex is rewritten by the compiler into an element read
        //  through a selected index, and a simplex that is indexed at run time lives in scratch)
        double a = sx[0][d], b = sx[1][d], c = sx[2][d];
        asm("" : "+v"(a), "+v"(b), "+v"(c));
        const double L = l0 ? a : (l1 ? b : c);
        const double U = l0 ? b : a;
        const double V = (l0 || l1) ? c : b;
        xc[d] = ((U + V) + L) * (1.0 / DN);
      }
    }
  } else {
#pragma unroll
    for (int d = 0; d < DN; d++) {
      double s = 0;
#pragma unroll
      for (int i = 0; i < DN; i++) s += sx[i][d];
      xc[d] = s * (1.0 / DN);
    }
  }
}

// Optim's convergence test: nmobjective(f_simplex, n, m) = sqrt(var(y) * n / (n + 1)) <= g_tol, i.e. the population
// standard deviation of the vertex values (oracle/nbp_oracle.c:nm_converged spells it the way Optim does).
// Here the same predicate without the square root and the divisions: sum((y - mean)^2) <= g_tol^2 * (n + 1)
// (one FP64 divide or sqrt costs ~30 VALU instructions, and this runs once per simplex iteration).
template <int DN>
__device__ __forceinline__ bool nm_converged(const double (&f)[DN + 1]) {
  constexpr double rm = 1.0 / (DN + 1);
  double a = 0;
#pragma unroll
  for (int i = 0; i <= DN; i++) a += f[i];
  a *= rm;
  double v = 0;
#pragma unroll
  for (int i = 0; i <= DN; i++) {  // (the multiply-add spelled out: the same in every instance)
    const double df = f[i] - a;
    v = fma(df, df, v);
  }
  return v <= 1e-16 * (DN + 1);
}

// a + c * b, one rounding per operation, as Julia evaluates Optim's vertex formulas -- a contracted multiply-add differs in the
// last bit whenever c * b is inexact (beta = 5/3, gamma = 7/12, delta = 2/3 in three dimensions, and 1.5 x + 0.025 of the initial
// simplex in any; 2, 1/2, 1/2 in two: exact)
template <bool OPT>
__device__ __forceinline__ double nm_lin(double a, double c, double b) {
  NBPM_EXACT
  const double p = c * b;
  return a + p;
}

// OPT: the form of the 3-D centroid (nm_centroid): sums taken and selected (true) or elements selected and summed (false)
template <class OBJ, int DN, bool OPT, bool TB>
__device__ __forceinline__ bool nelder_mead_tb(OBJ &o, double (&x)[DN], bool &tied) {
  constexpr int M = DN + 1;
  const double alpha = 1.0, beta = 1.0 + 2.0 / DN, gamma = 0.75 - 1.0 / (2.0 * DN), delta = 1.0 - 1.0 / DN;
  double sx[M][DN], f[M];
  int id[M];
#pragma unroll
  for (int i = 0; i < M; i++) id[i] = i;
#pragma unroll
  for (int i = 0; i < M; i++)
#pragma unroll
    for (int d = 0; d < DN; d++) sx[i][d] = x[d];
#pragma unroll
  for (int j = 0; j < DN; j++) sx[j + 1][j] = nm_lin<OPT>(0.025, 1.0 + 0.5, sx[j + 1][j]);
#pragma unroll
  for (int i = 0; i < M; i++) f[i] = o(sx[i]);
  nm_sort_all<DN, OPT, TB>(sx, f, id, tied);
  bool converged = nm_converged<DN>(f);
  int it = 0;
#ifdef NBP_X_NMNOUNROLL
#pragma clang loop unroll(disable)
#endif
  while (!converged && it < 1000) {
    it++;
    double xc[DN], xr[DN], xcache[DN];
    nm_centroid<DN, OPT>(sx, id, xc);
    const double f_lowest = f[0], f_second = f[DN - 1], f_highest = f[DN];
#pragma unroll
    for (int d = 0; d < DN; d++) xr[d] = xc[d] + alpha * (xc[d] - sx[DN][d]);
    const double f_reflect = o(xr);
    const bool do_expand = f_reflect < f_lowest;
    const bool accept_reflect = !do_expand && f_reflect < f_second;
    // straight-line: the second point of the iteration (expansion, outside or inside contraction) is always computed
    // and dropped by the lanes that accept the reflection -- the lanes of a wave take different branches in almost
    // every iteration anyway, and one copy of the insertion below serves all of them
    const bool outside = f_reflect < f_highest;
    const double coef = do_expand ? beta : (outside ? gamma : -gamma);
#pragma unroll
    for (int d = 0; d < DN; d++) xcache[d] = nm_lin<OPT>(xc[d], coef, xr[d] - xc[d]);
    const double f2 = o(xcache);
    o.evals -= accept_reflect ? 1 : 0;  // Optim does not evaluate it
    // expansion: the better of (expand, reflect) replaces the worst; contraction: accepted only
    // if it improves on min(reflect, highest)
    const bool take2 = !accept_reflect && (do_expand ? (f2 < f_reflect) : (f2 < (outside ? f_reflect : f_highest)));
    const bool shrink = !accept_reflect && !do_expand && !take2;
#pragma unroll
    for (int d = 0; d < DN; d++) sx[DN][d] = shrink ? sx[DN][d] : (take2 ? xcache[d] : xr[d]);
    f[DN] = shrink ? f[DN] : (take2 ? f2 : f_reflect);
    nm_sift_last<DN, OPT, TB>(sx, f, id, tied);  // leaves a sorted simplex (the shrinking lanes') as it is
    if (shrink) {
#pragma unroll
      for (int q = 1; q < M; q++) {
#pragma unroll
        for (int d = 0; d < DN; d++) sx[q][d] = nm_lin<OPT>(sx[0][d], delta, sx[q][d] - sx[0][d]);
        f[q] = o(sx[q]);
      }
      nm_sort_all<DN, OPT, TB>(sx, f, id, tied);
    }
    converged = nm_converged<DN>(f);
  }
  // after_while!: the better of the best vertex and the centroid of the DN best
  double xc[DN];
  nm_centroid<DN, OPT>(sx, id, xc);
  const double fcen = o(xc);
  const bool usec = fcen < f[0];
#pragma unroll
  for (int d = 0; d < DN; d++) x[d] = usec ? xc[d] : sx[0][d];
  return converged;
}

// The search with Optim's order of equal values (nm_cswap).  Three dimensions: the slots travel anyway (nm_centroid) and break
// the ties where they are compared.  Two dimensions: the search runs without the slots and only notices a tie; the one search
// in tens of thousands that sees one starts again from its start with them (the evaluations counted are those of the search
// that stands).
template <class OBJ, int DN, bool OPT = true>
__device__ __forceinline__ bool nelder_mead(OBJ &o, double (&x)[DN]) {
  bool tied = false;
#ifdef NBP_X_NOTIEBREAK
  return nelder_mead_tb<OBJ, DN, OPT, false>(o, x, tied);
#else
  if constexpr (DN != 2) return nelder_mead_tb<OBJ, DN, OPT, true>(o, x, tied);
  else {
    double x0[DN];
#pragma unroll
    for (int d = 0; d < DN; d++) x0[d] = x[d];
    const unsigned int e0 = o.evals;
    bool conv = nelder_mead_tb<OBJ, DN, OPT, false>(o, x, tied);
    if (tied) {
#pragma unroll
      for (int d = 0; d < DN; d++) x[d] = x0[d];
      o.evals = e0;
      conv = nelder_mead_tb<OBJ, DN, OPT, true>(o, x, tied);
    }
    return conv;
  }
#endif
}

// Optim.BFGS for a 1-D decision variable (islen1 branch), central finite differences,
// Armijo / quadratic-interpolation line search (documented deviation from HagerZhang).
// (the decision variable travels as a scalar and becomes a one-element array only inside the call of the objective:
//  arrays that live across the line-search loop end up in scratch)
template <class OBJ>
__device__ __forceinline__ double eval1(OBJ &o, double x) {
  const double t[1] = {x};
  return o(t);
}
template <class OBJ>
__device__ __forceinline__ double fd_grad1(OBJ &o, double x) {
  double h = 6.0554544523933395e-06 * fmax(1.0, fabs(x));
  return (eval1(o, x + h) - eval1(o, x - h)) / (2.0 * h);
}

template <class OBJ>
__device__ __forceinline__ bool bfgs_1d(OBJ &o, double (&x)[1]) {
  double xc = x[0];
  double fx = eval1(o, xc), g = fd_grad1(o, xc), H = 1.0;
  bool converged = false;
  for (int it = 0; it < 1000; it++) {
    if (fabs(g) <= 1e-8) { converged = true; break; }
    double s = -H * g;
    if (s * g >= 0) { H = 1.0; s = -g; }
    double al = 1.0, dphi0 = g * s, fn = fx;
    double xn = xc;
    bool ok = false;
    for (int ls = 0; ls < 50; ls++) {
      xn = xc + al * s;
      fn = eval1(o, xn);
      if (fn <= fx + 1e-4 * al * dphi0) { ok = true; break; }
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
  x[0] = xc;
  return converged;
}

// Optim.BFGS for an n-dimensional decision variable: what a partial factor over more than one coordinate runs
// (`alg = islen1 ? BFGS : NelderMead` with islen1 true for every partial factor, NumericalCalculations.jl:108,424; the
// gradient is zero off the partial coordinates, so the search is one over those).  Initial inverse Hessian I, central
// finite differences, g_tol = 1e-8 on the max-norm of the gradient, the Armijo / quadratic-interpolation line search of
// bfgs_1d (to which this reduces for DN = 1).
template <class OBJ, int DN>
__device__ __forceinline__ void fd_grad_nd(OBJ &o, const double (&x)[DN], double (&g)[DN]) {
#pragma unroll
  for (int k = 0; k < DN; k++) {
    const double h = 6.0554544523933395e-06 * fmax(1.0, fabs(x[k]));
    double xp[DN], xm[DN];
#pragma unroll
    for (int q = 0; q < DN; q++) { xp[q] = x[q]; xm[q] = x[q]; }
    xp[k] += h;
    xm[k] -= h;
    g[k] = (o(xp) - o(xm)) / (2.0 * h);
  }
}
template <class OBJ, int DN>
__device__ __forceinline__ bool bfgs_nd(OBJ &o, double (&x)[DN]) {
  double xc[DN], g[DN], H[DN][DN];
#pragma unroll
  for (int k = 0; k < DN; k++) {
    xc[k] = x[k];
#pragma unroll
    for (int q = 0; q < DN; q++) H[k][q] = (k == q) ? 1.0 : 0.0;
  }
  double fx = o(xc);
  fd_grad_nd<OBJ, DN>(o, xc, g);
  bool converged = false;
  for (int it = 0; it < 1000; it++) {
    double gmax = 0;
#pragma unroll
    for (int k = 0; k < DN; k++) gmax = fmax(gmax, fabs(g[k]));
    if (gmax <= 1e-8) { converged = true; break; }
    double s[DN], dphi0 = 0;
#pragma unroll
    for (int k = 0; k < DN; k++) {
      double a = 0;
#pragma unroll
      for (int q = 0; q < DN; q++) a -= H[k][q] * g[q];
      s[k] = a;
      dphi0 += g[k] * a;
    }
    if (dphi0 >= 0) {  // not a descent direction: restart from steepest descent
      dphi0 = 0;
#pragma unroll
      for (int k = 0; k < DN; k++) {
#pragma unroll
        for (int q = 0; q < DN; q++) H[k][q] = (k == q) ? 1.0 : 0.0;
        s[k] = -g[k];
        dphi0 -= g[k] * g[k];
      }
    }
    double al = 1.0, fn = fx, xn[DN];
    bool ok = false;
    for (int ls = 0; ls < 50; ls++) {
#pragma unroll
      for (int k = 0; k < DN; k++) xn[k] = xc[k] + al * s[k];
      fn = o(xn);
      if (fn <= fx + 1e-4 * al * dphi0) { ok = true; break; }
      double aq = -dphi0 * al * al / (2.0 * (fn - fx - dphi0 * al));
      if (!(aq >= 0.1 * al)) aq = 0.1 * al;
      if (aq > 0.5 * al) aq = 0.5 * al;
      al = aq;
    }
    if (!ok) break;
    double gn[DN], dx[DN], dg[DN], sy = 0, gnmax = 0;
    fd_grad_nd<OBJ, DN>(o, xn, gn);
    bool moved = false;
#pragma unroll
    for (int k = 0; k < DN; k++) {
      dx[k] = xn[k] - xc[k];
      dg[k] = gn[k] - g[k];
      sy += dx[k] * dg[k];
      moved |= dx[k] != 0.0;
      gnmax = fmax(gnmax, fabs(gn[k]));
    }
    if (!moved) { converged = gnmax <= 1e-8; break; }
    if (sy > 0) {  // H <- (I - rho dx dg') H (I - rho dg dx') + rho dx dx'
      const double rho = 1.0 / sy;
      double Hy[DN], yHy = 0;
#pragma unroll
      for (int k = 0; k < DN; k++) {
        double a = 0;
#pragma unroll
        for (int q = 0; q < DN; q++) a += H[k][q] * dg[q];
        Hy[k] = a;
      }
#pragma unroll
      for (int k = 0; k < DN; k++) yHy += dg[k] * Hy[k];
#pragma unroll
      for (int k = 0; k < DN; k++)
#pragma unroll
        for (int q = 0; q < DN; q++)
          H[k][q] += rho * ((1.0 + rho * yHy) * dx[k] * dx[q] - Hy[k] * dx[q] - dx[k] * Hy[q]);
    }
#pragma unroll
    for (int k = 0; k < DN; k++) { xc[k] = xn[k]; g[k] = gn[k]; }
    fx = fn;
  }
#pragma unroll
  for (int k = 0; k < DN; k++) x[k] = xc[k];
  return converged;
}

// _solveCCWNumeric! for one particle (NumericalCalculations.jl:413-452, :90-133)
#ifdef NBP_SOLVE_NOINLINE
#define NBP_SOLVE_ATTR __attribute__((noinline))
#else
#define NBP_SOLVE_ATTR __forceinline__
#endif
template <int KIND, int DN, bool PARTIAL_BFGS = false>
__device__ NBP_SOLVE_ATTR void solve_particle_t(int manifold, const double *z, const double *other, int solve_b, double *x,
                                                 unsigned int &n_solves, unsigned int &n_nonconv, unsigned int &n_nan,
                                                 unsigned int &n_evals, int rmask = 7) {
  objective_t<KIND, DN> o;
  o.solve_b = solve_b;
  o.evals = 0;
  o.rmask = PARTIAL_BFGS ? rmask : 7;
#pragma unroll
  for (int i = 0; i < 3; i++) { o.z[i] = z[i]; o.other[i] = other[i]; }
  o.sn_fixed = 0.0;
  o.cs_fixed = 1.0;
  if (KIND == NBP_F_SE2 && solve_b) sincos_fast(other[2], &o.sn_fixed, &o.cs_fixed);
  double xc[DN];
#pragma unroll
  for (int d = 0; d < DN; d++) xc[d] = x[d];
  bool conv;
  if constexpr (DN == 1) conv = bfgs_1d(o, xc);
  else if constexpr (PARTIAL_BFGS) conv = bfgs_nd<objective_t<KIND, DN>, DN>(o, xc);
  else conv = nelder_mead<objective_t<KIND, DN>, DN, KIND != NBP_F_SE2>(o, xc);
  n_solves++;
  n_evals += o.evals;
  if (!conv) n_nonconv++;
  bool bad = false;
#pragma unroll
  for (int d = 0; d < DN; d++) bad |= isnan(xc[d]);
  if (bad) { n_nan++; return; }
#pragma unroll
  for (int d = 0; d < DN; d++) x[d] = is_circ(manifold, d) ? wrap_pi(xc[d]) : xc[d];
}
// a partial LinearRelative over two coordinates: BFGS on those two (the Euclid(2) objective with their values)
__device__ __forceinline__ void solve_particle_partial2(const double *z, const double *other, int solve_b, double *x, unsigned int &a,
                                                        unsigned int &b, unsigned int &c, unsigned int &e) {
  solve_particle_t<NBP_F_LINREL, 2, true>(NBP_EUCLID2, z, other, solve_b, x, a, b, c, e);
}

// a partial ManifoldFactor on SE(2): the residual counts the components of `.partial` only, the search is BFGS over the
// whole point (NumericalCalculations.jl:424-446: `islen1 = ... || ccwl.partial`, the decision variable is the full u0)
__device__ __forceinline__ void solve_particle_partial_se2(const double *z, const double *other, int solve_b, double *x, int rmask, unsigned int &a,
                                                           unsigned int &b, unsigned int &c, unsigned int &e) {
  solve_particle_t<NBP_F_SE2, 3, true>(NBP_SE2, z, other, solve_b, x, a, b, c, e, rmask);
}

// wave-uniform dispatch on (factor kind, tangent dimension)
__device__ __forceinline__ void solve_particle(int kind, int manifold, const double *z, const double *other, int solve_b,
                                               double *x, unsigned int &a, unsigned int &b, unsigned int &c, unsigned int &e) {
  const int D = mani_dim(manifold);
  switch (kind) {
  case NBP_F_LINREL:
    if (D == 1) solve_particle_t<NBP_F_LINREL, 1>(manifold, z, other, solve_b, x, a, b, c, e);
    else if (D == 2) solve_particle_t<NBP_F_LINREL, 2>(manifold, z, other, solve_b, x, a, b, c, e);
    else solve_particle_t<NBP_F_LINREL, 3>(manifold, z, other, solve_b, x, a, b, c, e);
    break;
  case NBP_F_CIRCULAR: solve_particle_t<NBP_F_CIRCULAR, 1>(manifold, z, other, solve_b, x, a, b, c, e); break;
  case NBP_F_SE2: solve_particle_t<NBP_F_SE2, 3>(manifold, z, other, solve_b, x, a, b, c, e); break;
  default:
    if (D == 1) solve_particle_t<NBP_F_EUCLIDDIST, 1>(manifold, z, other, solve_b, x, a, b, c, e);
    else if (D == 2) solve_particle_t<NBP_F_EUCLIDDIST, 2>(manifold, z, other, solve_b, x, a, b, c, e);
    else solve_particle_t<NBP_F_EUCLIDDIST, 3>(manifold, z, other, solve_b, x, a, b, c, e);
    break;
  }
}

// ------------------------------------------------------------------------------------------------
// approxDeconv (DeconvUtils.jl:32-160): the measurement is the decision variable, both variable
// points are fixed.  ZD = measurement dimension (NelderMead; BFGS when ZD == 1, :94,139).
// ------------------------------------------------------------------------------------------------
template <int KIND, int DN, int ZD>
struct deconv_objective_t {
  double a[3], b[3];
  unsigned int evals;
  __device__ __forceinline__ double operator()(const double (&zz)[ZD]) {
    evals++;
    objective_t<KIND, DN> o;
    o.solve_b = 0;
    o.sn_fixed = 0.0;
    o.cs_fixed = 1.0;
#pragma unroll
    for (int k = 0; k < 3; k++) o.z[k] = 0.0;
#pragma unroll
    for (int k = 0; k < ZD; k++) o.z[k] = zz[k];
    return o.normsq(a, b);
  }
};

template <int KIND, int DN, int ZD>
__device__ __forceinline__ void deconv_particle_t(const double *a, const double *b, double *z, unsigned int &n_solves,
                                                  unsigned int &n_nonconv, unsigned int &n_nan, unsigned int &n_evals) {
  deconv_objective_t<KIND, DN, ZD> o;
  o.evals = 0;
#pragma unroll
  for (int k = 0; k < 3; k++) { o.a[k] = a[k]; o.b[k] = b[k]; }
  double zc[ZD];
#pragma unroll
  for (int k = 0; k < ZD; k++) zc[k] = z[k];
  bool conv;
  if constexpr (ZD == 1) conv = bfgs_1d(o, zc);
  else conv = nelder_mead<deconv_objective_t<KIND, DN, ZD>, ZD, KIND != NBP_F_SE2>(o, zc);
  n_solves++;
  n_evals += o.evals;
  if (!conv) n_nonconv++;
  bool bad = false;
#pragma unroll
  for (int k = 0; k < ZD; k++) bad |= isnan(zc[k]);
  if (bad) { n_nan++; return; }
#pragma unroll
  for (int k = 0; k < ZD; k++) z[k] = zc[k];
}

__device__ __forceinline__ void deconv_particle(int kind, int manifold, const double *a, const double *b, double *z,
                                                unsigned int &s, unsigned int &nc, unsigned int &nn, unsigned int &e) {
  const int D = mani_dim(manifold);
  switch (kind) {
  case NBP_F_LINREL:
    if (D == 1) deconv_particle_t<NBP_F_LINREL, 1, 1>(a, b, z, s, nc, nn, e);
    else if (D == 2) deconv_particle_t<NBP_F_LINREL, 2, 2>(a, b, z, s, nc, nn, e);
    else deconv_particle_t<NBP_F_LINREL, 3, 3>(a, b, z, s, nc, nn, e);
    break;
  case NBP_F_CIRCULAR: deconv_particle_t<NBP_F_CIRCULAR, 1, 1>(a, b, z, s, nc, nn, e); break;
  case NBP_F_SE2: deconv_particle_t<NBP_F_SE2, 3, 3>(a, b, z, s, nc, nn, e); break;
  default:
    if (D == 1) deconv_particle_t<NBP_F_EUCLIDDIST, 1, 1>(a, b, z, s, nc, nn, e);
    else if (D == 2) deconv_particle_t<NBP_F_EUCLIDDIST, 2, 1>(a, b, z, s, nc, nn, e);
    else deconv_particle_t<NBP_F_EUCLIDDIST, 3, 1>(a, b, z, s, nc, nn, e);
    break;
  }
}

// ------------------------------------------------------------------------------------------------
// Bandwidth: leave-one-out likelihood cross validation, golden-section search (KDE.jl `:lcv`).
// x = LDS array of N coordinates.  The workgroup is P x Npad lanes: lane (i, p) owns point i and
// walks the points j = p, p+P, ... (wave-uniform j -> LDS broadcast reads, 4 independent exp chains
// in flight); the P partial row sums are combined in a fixed order through LDS (`part`), the N row
// terms are tree-reduced.  O(N^2) exp per evaluation, ~16-21 evaluations per coordinate.
// ------------------------------------------------------------------------------------------------
// Pair-symmetric evaluation: K(x_i - x_j) = K(x_j - x_i), so lane i only visits the partners
// j = i + t (mod N), t = 1 .. (N-1)/2 (plus t = N/2 for the lower half when N is even): every
// unordered pair is computed once, added to lane i's own row sum (register) and to row j's
// accumulator acc[wave][j] in LDS (one private accumulator row per wave: lanes of a wave hit
// consecutive j -> conflict-free; a wave's LDS ops execute in order -> deterministic sums).
// in-kernel phase timing (tools/phase_timing.py, tools/lcv_phase_timing.py): debug builds only
#ifdef NBP_PHASE_TIMING
__device__ long long nbp_phase_clk[64];
#define NBP_TICK(k) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) nbp_phase_clk[k] += (long long)wall_clock64() - t_last_; t_last_ = wall_clock64(); } while (0)
#define NBP_TICK_INIT() long long t_last_ = wall_clock64()
// shader-clock variant for the short phases of one LCV evaluation, measured in the LAST workgroup
// of the launch (so that a chip-filling batch is in flight around it)
#define NBP_CTICK(k) do { if (blockIdx.x == gridDim.x - 1 && blockIdx.y == 0 && threadIdx.x == 0) nbp_phase_clk[k] += (long long)__builtin_readcyclecounter() - c_last_; c_last_ = __builtin_readcyclecounter(); } while (0)
#define NBP_CTICK_INIT() long long c_last_ = __builtin_readcyclecounter()
// begin / end (100 MHz wall clock) and hardware id of every workgroup of the last launch (tools/exp/block_timeline.py)
__device__ long long nbp_block_clk[8192][3];
#define NBP_BLOCK_BEGIN() const long long blk_t0_ = (long long)wall_clock64(); do { if (threadIdx.x == 0 && blockIdx.x < 8192) { nbp_block_clk[blockIdx.x][0] = (long long)wall_clock64(); nbp_block_clk[blockIdx.x][2] = (long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)); } } while (0)
// histogram of workgroup durations over all launches since the last read, by grid-size class (<= 8, <= 64, <= 300, more
// workgroups) in bins of 16 us
__device__ unsigned int nbp_block_hist[4][64];
#define NBP_BLOCK_END() do { if (threadIdx.x == 0) { const long long e_ = (long long)wall_clock64(); if (blockIdx.x < 8192) nbp_block_clk[blockIdx.x][1] = e_; \
  const int g_ = gridDim.x <= 8 ? 0 : (gridDim.x <= 64 ? 1 : (gridDim.x <= 300 ? 2 : 3)); long long b_ = (e_ - blk_t0_) / 1600; if (b_ > 63) b_ = 63; \
  atomicAdd(&nbp_block_hist[g_][b_], 1u); } } while (0)
#else
#define NBP_BLOCK_BEGIN()
#define NBP_BLOCK_END()
#define NBP_TICK(k)
#define NBP_TICK_INIT()
#define NBP_CTICK(k)
#define NBP_CTICK_INIT()
#endif

typedef __attribute__((address_space(3))) double nbp_lds_double;
__device__ __forceinline__ void lds_add(nbp_lds_double *p, double v) { (void)__builtin_amdgcn_ds_atomic_fadd_f64(p, v); }

// squared geodesic distance on the circle for a difference within (-3pi, 3pi): min(|d|, ||d| - 2pi|)^2,
// two VALU operations instead of a wrap (equal to wrap_pi(d)^2 up to the rounding of one subtraction)
__device__ __forceinline__ double circ_sq(double d) {
  const double a = fmin(fabs(d), fabs(fabs(d) - NBP_TWO_PI));
  return a * a;
}

// exp(-c q), q >= 0, with everything that depends on the bandwidth folded into per-evaluation constants that live in
// SGPRs: n = round(-c q 256/ln2) by the magic-number trick, z = -c q - n ln2/256 = -c r with r = q + n ln2/(256 c), and
// exp(z) = 1 + a1 r + ... + a4 r^4 with a_k = (-c)^k / k!  (|z| <= ln2/512: the next term is 3.8e-17), times
// 2^((n & 255)/256) * 2^(n >> 8), which is ONE table read and ONE integer add (lcv_tab_scaled).  15 VALU operations per
// pair where exp(-(q c)) by the general-purpose exp_nonpos takes 20 (the multiplication by c, two Horner steps and the
// masking of n are gone).
struct lcv_exp_k {
  double A, negM, B, qmax, a1, a2, a3, a4;
};

// a wave-uniform double into an SGPR pair.  Opaque to the compiler on purpose: it folds __builtin_amdgcn_readfirstlane
// of a value it can prove uniform and then keeps the result of the (vector) arithmetic in VGPRs -- eight constants of the
// pair loop would cost sixteen VGPRs.
__device__ __forceinline__ double sgpr_double(double v) {
  int lo, hi;
  // wait states on both sides: the compiler's hazard recognizer does not look inside inline asm, so neither the VALU
  // instruction that has just written the source VGPR nor the one that is about to read the SGPR gets its distance
  asm volatile("s_nop 4\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 4" : "=s"(lo) : "v"(__double2loint(v)));
  asm volatile("s_nop 4\n\tv_readfirstlane_b32 %0, %1\n\ts_nop 4" : "=s"(hi) : "v"(__double2hiint(v)));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ lcv_exp_k lcv_exp_consts(double c, double h) {  // c = 1/(2 h^2): 1/c = 2 h^2 without a division
  lcv_exp_k K;
  const double rc = 2.0 * h * h;
  K.A = sgpr_double(-c * (369.3299304675746 * (NBP_FITTAB / 256)));       // 2^TLOG/ln2 (256/ln2 times a power of two)
  K.negM = -6755399441055744.0;
  K.B = sgpr_double((0.0027076061740622863 / (NBP_FITTAB / 256)) * rc);   // ln2/2^TLOG
  K.qmax = sgpr_double(700.0 * rc);                 // exp(-700) = 1e-304 is as good as 0 for every sum it enters
  const double m = -c;
  K.a1 = sgpr_double(m);
  K.a2 = sgpr_double(m * m * 0.5);
  // the leading coefficient stays in a VGPR pair (one scalar operand per VOP3)
#if NBP_LCV_TLOG >= 11
  K.a3 = m * m * m * 1.66666666666666666667e-01;
  K.a4 = 0.0;
#else
  K.a3 = sgpr_double(m * m * m * 1.66666666666666666667e-01);
  K.a4 = m * m * m * m * 4.16666666666666666667e-02;
#endif
  return K;
}
#define NBP_FMA_VVS(dst, a, b, cst) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(dst) : "v"(a), "v"(b), "s"(cst))
#define NBP_FMA_VSV(dst, a, cst, c) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(dst) : "v"(a), "s"(cst), "v"(c))
// the table entry of n with 2^(n >> 8) already on it: entry k = n & 255 holds the bits of 2^(k/256) with k << 12 taken off
// the high word (tools/gen_lcv_table.py), so adding n << 12 = (e << 20) + (k << 12) leaves e on the exponent field
__device__ __forceinline__ double lcv_tab_scaled(const double *tab, int n) {
  const double w = tab[n & (NBP_FITTAB - 1)];
  int hi;
  asm("v_lshl_add_u32 %0, %1, %3, %2" : "=v"(hi) : "v"(n), "v"(__double2hiint(w)), "n"(20 - NBP_LCV_TLOG));
  return __hiloint2double(hi, __double2loint(w));
}
// the remainder polynomial 1 + a1 r + ... (cubic or quartic, see NBP_LCV_TLOG) without its last step
#if NBP_LCV_TLOG >= 11
#define NBP_LCV_HORNER(p, r, K) NBP_FMA_VVS(p, K.a3, r, K.a2); NBP_FMA_VVS(p, p, r, K.a1)
#else
#define NBP_LCV_HORNER(p, r, K) NBP_FMA_VVS(p, K.a4, r, K.a3); NBP_FMA_VVS(p, p, r, K.a2); NBP_FMA_VVS(p, p, r, K.a1)
#endif
__device__ __forceinline__ double lcv_clamp(double q, const lcv_exp_k &K) {
  double r;
  asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(q), "s"(K.qmax));  // fmin() canonicalises the scalar operand into a VGPR pair first
  return r;
}
__device__ __forceinline__ double lcv_exp(double q, const lcv_exp_k &K, const double *tab) {
  q = lcv_clamp(q, K);
  double t, r, p;
  const double M = 6755399441055744.0;
  NBP_FMA_VSV(t, q, K.A, M);
  const int n = __double2loint(t);
  const double tf = t + K.negM;
  NBP_FMA_VSV(r, tf, K.B, q);
  NBP_LCV_HORNER(p, r, K);
  p = fma(p, r, 1.0);
  return lcv_tab_scaled(tab, n) * p;
}

// four independent chains, written step by step so that the instruction stream keeps them interleaved (the scheduler
// otherwise runs one Horner chain after the other when registers are tight)
__device__ __forceinline__ void lcv_exp4(double &q0, double &q1, double &q2, double &q3, const lcv_exp_k &K, const double *tab) {
  const double M = 6755399441055744.0;
  q0 = lcv_clamp(q0, K); q1 = lcv_clamp(q1, K); q2 = lcv_clamp(q2, K); q3 = lcv_clamp(q3, K);
  double t0, t1, t2, t3;
  NBP_FMA_VSV(t0, q0, K.A, M); NBP_FMA_VSV(t1, q1, K.A, M); NBP_FMA_VSV(t2, q2, K.A, M); NBP_FMA_VSV(t3, q3, K.A, M);
#if NBP_LCV_KO & 2
  const double w0 = __hiloint2double(__double2loint(t0) << 9, 1), w1 = __hiloint2double(__double2loint(t1) << 9, 1);
  const double w2 = __hiloint2double(__double2loint(t2) << 9, 1), w3 = __hiloint2double(__double2loint(t3) << 9, 1);
#else
  const double w0 = lcv_tab_scaled(tab, __double2loint(t0)), w1 = lcv_tab_scaled(tab, __double2loint(t1));
  const double w2 = lcv_tab_scaled(tab, __double2loint(t2)), w3 = lcv_tab_scaled(tab, __double2loint(t3));
#endif
  t0 += K.negM; t1 += K.negM; t2 += K.negM; t3 += K.negM;
  double r0, r1, r2, r3, p0, p1, p2, p3;
  NBP_FMA_VSV(r0, t0, K.B, q0); NBP_FMA_VSV(r1, t1, K.B, q1); NBP_FMA_VSV(r2, t2, K.B, q2); NBP_FMA_VSV(r3, t3, K.B, q3);
#if NBP_LCV_TLOG >= 11
  NBP_FMA_VVS(p0, K.a3, r0, K.a2); NBP_FMA_VVS(p1, K.a3, r1, K.a2); NBP_FMA_VVS(p2, K.a3, r2, K.a2); NBP_FMA_VVS(p3, K.a3, r3, K.a2);
#else
  NBP_FMA_VVS(p0, K.a4, r0, K.a3); NBP_FMA_VVS(p1, K.a4, r1, K.a3); NBP_FMA_VVS(p2, K.a4, r2, K.a3); NBP_FMA_VVS(p3, K.a4, r3, K.a3);
  NBP_FMA_VVS(p0, p0, r0, K.a2); NBP_FMA_VVS(p1, p1, r1, K.a2); NBP_FMA_VVS(p2, p2, r2, K.a2); NBP_FMA_VVS(p3, p3, r3, K.a2);
#endif
  NBP_FMA_VVS(p0, p0, r0, K.a1); NBP_FMA_VVS(p1, p1, r1, K.a1); NBP_FMA_VVS(p2, p2, r2, K.a1); NBP_FMA_VVS(p3, p3, r3, K.a1);
  p0 = fma(p0, r0, 1.0); p1 = fma(p1, r1, 1.0); p2 = fma(p2, r2, 1.0); p3 = fma(p3, r3, 1.0);
  q0 = w0 * p0; q1 = w1 * p1; q2 = w2 * p2; q3 = w3 * p3;
}

// EXPERIMENT, off (NBP_LCV_PIPE = 1 to build it): the pair loop as a two-stage software pipeline.  Stage A of a group of four
// partners -- difference, square, clamp, t = q A + M, the read of the table entry -- is issued one group AHEAD of stage B
// (remainder, polynomial, scaling by the table entry, the sums), so that the table entry a group needs was requested a whole
// group of arithmetic earlier and no wave sits at an s_waitcnt for it (written straight through, the compiler puts the wait
// directly behind the four reads).  Two register sets, the loop body written twice: no copies at the back edge.  The same
// operations on the same values in the same order as the straight form -- bit-identical sums.  Measured: 4.29 ms against
// 4.18 ms for 8192 chip-filling fits (128 VGPRs, 32 B of scratch, one idle stage A per call): the LDS LATENCY a wave sees is
// not what the loop waits for.  The knock-out builds (NBP_LCV_KO, wrong sums) say what is: without the partner atomics
// 0.61 ps per pair instead of 0.78, without the table reads 0.63, without either 0.54 = the vector instructions alone
// (profiles/r04_lcv_pair_loop_experiments.txt).
#ifndef NBP_LCV_KO
#define NBP_LCV_KO 0
#endif
#ifndef NBP_LCV_PIPE
#define NBP_LCV_PIPE 0
#endif
struct lcv_group {
  double q0, q1, q2, q3, t0, t1, t2, t3, w0, w1, w2, w3;
};
template <bool CIRC>
__device__ __forceinline__ void lcv_stage_a(lcv_group &G, double xi, double y0, double y1, double y2, double y3, const lcv_exp_k &K,
                                            const double *tab) {
  const double M = 6755399441055744.0;
  const double d0 = xi - y0, d1 = xi - y1, d2 = xi - y2, d3 = xi - y3;
  G.q0 = lcv_clamp(CIRC ? circ_sq(d0) : d0 * d0, K);
  G.q1 = lcv_clamp(CIRC ? circ_sq(d1) : d1 * d1, K);
  G.q2 = lcv_clamp(CIRC ? circ_sq(d2) : d2 * d2, K);
  G.q3 = lcv_clamp(CIRC ? circ_sq(d3) : d3 * d3, K);
  NBP_FMA_VSV(G.t0, G.q0, K.A, M); NBP_FMA_VSV(G.t1, G.q1, K.A, M); NBP_FMA_VSV(G.t2, G.q2, K.A, M); NBP_FMA_VSV(G.t3, G.q3, K.A, M);
  G.w0 = tab[__double2loint(G.t0) & (NBP_FITTAB - 1)];
  G.w1 = tab[__double2loint(G.t1) & (NBP_FITTAB - 1)];
  G.w2 = tab[__double2loint(G.t2) & (NBP_FITTAB - 1)];
  G.w3 = tab[__double2loint(G.t3) & (NBP_FITTAB - 1)];
}
// stage B: e = 2^(n/256) (1 + a1 r + ... + a4 r^4); the table entry enters last (a false dependence on the polynomial keeps
// the scheduler from pulling its scaling -- and with it the wait for the read -- to the front)
__device__ __forceinline__ double lcv_scale_late(double w, double t, double p) {
  int hi;
  asm("v_lshl_add_u32 %0, %1, %4, %2" : "=v"(hi) : "v"(__double2loint(t)), "v"(__double2hiint(w)), "v"(p), "n"(20 - NBP_LCV_TLOG));
  return __hiloint2double(hi, __double2loint(w));
}
__device__ __forceinline__ void lcv_stage_b(const lcv_group &G, const lcv_exp_k &K, double &e0, double &e1, double &e2, double &e3) {
  const double u0 = G.t0 + K.negM, u1 = G.t1 + K.negM, u2 = G.t2 + K.negM, u3 = G.t3 + K.negM;
  double r0, r1, r2, r3, p0, p1, p2, p3;
  NBP_FMA_VSV(r0, u0, K.B, G.q0); NBP_FMA_VSV(r1, u1, K.B, G.q1); NBP_FMA_VSV(r2, u2, K.B, G.q2); NBP_FMA_VSV(r3, u3, K.B, G.q3);
  NBP_LCV_HORNER(p0, r0, K); NBP_LCV_HORNER(p1, r1, K); NBP_LCV_HORNER(p2, r2, K); NBP_LCV_HORNER(p3, r3, K);
  p0 = fma(p0, r0, 1.0); p1 = fma(p1, r1, 1.0); p2 = fma(p2, r2, 1.0); p3 = fma(p3, r3, 1.0);
  e0 = p0; e1 = p1; e2 = p2; e3 = p3;
}
#define NBP_LCV_FINISH(G, AP)                                                                                    \
  {                                                                                                               \
    const double f0 = lcv_scale_late(G.w0, G.t0, e0) * e0, f1 = lcv_scale_late(G.w1, G.t1, e1) * e1;              \
    const double f2 = lcv_scale_late(G.w2, G.t2, e2) * e2, f3 = lcv_scale_late(G.w3, G.t3, e3) * e3;              \
    s0 += f0; s1 += f1; s2 += f2; s3 += f3;                                                                       \
    if (!(NBP_LCV_KO & 1)) { lds_add(AP, f0); lds_add(AP + 1, f1); lds_add(AP + 2, f2); lds_add(AP + 3, f3); }    \
  }

template <bool CIRC>
__device__ __forceinline__ double loo_symmetric(const double *x, int pi, int ta, int n4, int nt, bool extra, double xi, const lcv_exp_k &K,
                                                double *accw, const double *tab) {
#if NBP_LCV_PIPE
  const double *xp = x + pi + ta;
  nbp_lds_double *ap = (nbp_lds_double *)(accw + pi + ta);
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  if (n4 > 0) {
    // x is followed by the row-sum partials in LDS: the reads run up to eleven entries past the last partner of the lane
    // (two groups ahead), whose values go through a stage A nobody finishes
    lcv_group GA, GB;
    double e0, e1, e2, e3;
    lcv_stage_a<CIRC>(GA, xi, xp[0], xp[1], xp[2], xp[3], K, tab);
    double y0 = xp[4], y1 = xp[5], y2 = xp[6], y3 = xp[7];
    // four additions of zero behind the first reads: the loop is then entered with the LDS queue it has at its back edge
    // (table reads, coordinate reads, four atomics), and the wait for the coordinates at its top becomes lgkmcnt(4) -- the
    // atomics of the group before stay in flight -- instead of the lgkmcnt(0) that merging the two entries gives
    lds_add(ap, 0.0); lds_add(ap + 1, 0.0); lds_add(ap + 2, 0.0); lds_add(ap + 3, 0.0);
    int k = 0;
    for (; k + 2 <= n4; k += 2, xp += 8, ap += 8) {
      lcv_stage_b(GA, K, e0, e1, e2, e3);
      lcv_stage_a<CIRC>(GB, xi, y0, y1, y2, y3, K, tab);
      y0 = xp[8]; y1 = xp[9]; y2 = xp[10]; y3 = xp[11];
      NBP_LCV_FINISH(GA, ap)
      lcv_stage_b(GB, K, e0, e1, e2, e3);
      lcv_stage_a<CIRC>(GA, xi, y0, y1, y2, y3, K, tab);
      y0 = xp[12]; y1 = xp[13]; y2 = xp[14]; y3 = xp[15];
      NBP_LCV_FINISH(GB, ap + 4)
    }
    if (k < n4) {
      lcv_stage_b(GA, K, e0, e1, e2, e3);
      NBP_LCV_FINISH(GA, ap)
      xp += 4;
      ap += 4;
    }
  }
  for (int k = 0; k < nt; k++) {
    const double d0 = xi - xp[k];
    const double e0 = lcv_exp(CIRC ? circ_sq(d0) : d0 * d0, K, tab);
    s0 += e0;
    lds_add(ap + k, e0);
  }
  if (extra) {
    const double d0 = xi - xp[nt];
    const double e0 = lcv_exp(CIRC ? circ_sq(d0) : d0 * d0, K, tab);
    s1 += e0;
    lds_add(ap + nt, e0);
  }
  return (s0 + s1) + (s2 + s3);
#else
  // x and the accumulator row are stored twice over ([0,2N)): partner pi+t never wraps, so both
  // addresses are one base register plus an immediate that advances with t.
  // The partner accumulation is an LDS atomic without return (ds_add_f64): lane i's slot at step t+1
  // is lane i+1's slot at step t, so plain read-modify-writes of consecutive steps would have to stay
  // strictly ordered (and exposed to the LDS latency); the atomic is applied by the LDS unit in the
  // wave's program order, so sums stay deterministic (one private row per wave).
  // Steps [ta, ta + 4*n4 + nt) plus one more when `extra`; n4 and nt are wave-uniform (scalar loops).
  const double *xp = x + pi + ta;
  nbp_lds_double *ap = (nbp_lds_double *)(accw + pi + ta);
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  // the coordinates of the next four partners are fetched while the current four are in flight
  double y0 = 0, y1 = 0, y2 = 0, y3 = 0;
  if (n4 > 0) { y0 = xp[0]; y1 = xp[1]; y2 = xp[2]; y3 = xp[3]; }
  for (int k = 0; k < n4; k++, xp += 4, ap += 4) {
    const double d0 = xi - y0, d1 = xi - y1, d2 = xi - y2, d3 = xi - y3;
    // unconditionally (x is followed by the row-sum partials in LDS: the last group reads up to four entries past the
    // lane's partners and drops them): behind a condition the four coordinates are copied at every back edge
#if NBP_LCV_KO & 4  // knock-out experiments (wrong sums; tools/exp/lcv_ko.sh): 1 no partner atomics, 2 no table reads, 4 no coordinate reads
    y0 += 1e-3; y1 += 1e-3; y2 += 1e-3; y3 += 1e-3;
#else
    y0 = xp[4]; y1 = xp[5]; y2 = xp[6]; y3 = xp[7];
#endif
    double e0 = CIRC ? circ_sq(d0) : d0 * d0, e1 = CIRC ? circ_sq(d1) : d1 * d1;
    double e2 = CIRC ? circ_sq(d2) : d2 * d2, e3 = CIRC ? circ_sq(d3) : d3 * d3;
    lcv_exp4(e0, e1, e2, e3, K, tab);
    s0 += e0;
    s1 += e1;
    s2 += e2;
    s3 += e3;
#if !(NBP_LCV_KO & 1)
    lds_add(ap, e0);
    lds_add(ap + 1, e1);
    lds_add(ap + 2, e2);
    lds_add(ap + 3, e3);
#endif
  }
  for (int k = 0; k < nt; k++) {
    const double d0 = xi - xp[k];
    const double e0 = lcv_exp(CIRC ? circ_sq(d0) : d0 * d0, K, tab);
    s0 += e0;
    lds_add(ap + k, e0);
  }
  if (extra) {
    const double d0 = xi - xp[nt];
    const double e0 = lcv_exp(CIRC ? circ_sq(d0) : d0 * d0, K, tab);
    s1 += e0;
    lds_add(ap + nt, e0);
  }
  return (s0 + s1) + (s2 + s3);
#endif
}

// log() as a real call from the evaluation: inlined, the double constants of its polynomial are hoisted out of the search
// as VGPR pairs and then spilled (40 B per lane that reach HBM once per fit); the callee is a leaf with a handful of
// registers, called once per evaluation
__device__ __attribute__((noinline)) double lcv_log(double v) { return log(v); }

// LDS: x[2N] (the coordinate, twice), part[P][Npad] row-sum partials, acc[NW][2N] per-wave partner
// accumulators (entry j and j+N both belong to point j).
// `acc` and `part` must be all-zero on entry and are all-zero again on exit (the readers clear what
// they read), so one evaluation costs three barriers: compute | combine+log | cross-wave sum.
template <bool REROLE = true>
__device__ __forceinline__ double neg_loo_ll(const double *x, int N, int Npad, bool circ, double h, double lognorm0, double *part,
                                             double *red, const double *tab) {
  // log(s_i) - log(h) - lognorm0 = log(s_i / h) - lognorm0: one log sequence per evaluation instead of
  // three (every lane of a wave pays for a log whether one lane needs it or all of them do)
  NBP_CTICK_INIT();
  const double inv_h = 1.0 / h, inv2h2 = 0.5 * inv_h * inv_h;
  const lcv_exp_k K = lcv_exp_consts(inv2h2, h);
  // The lane's role in the pair loop is recomputed in every evaluation, behind an opaque copy of the lane id: hoisted
  // out of the search (it does not depend on the bandwidth) the dozen role integers stay live across everything and go
  // to scratch -- which reaches HBM once per fit and workgroup (tens of MB per chip-filling launch).
  // (REROLE = false in the speculative search: its launches are small, the spill does not matter there.)
  int tid = threadIdx.x;
  if (REROLE) asm volatile("" : "+v"(tid));
  // throughput mode has one helper row (workgroup = Npad lanes): no divisions by run-time values for the lane's place
  const int P = ((int)blockDim.x == Npad) ? 1 : (int)blockDim.x / Npad;
  const int p = (P == 1) ? 0 : tid / Npad, i = (P == 1) ? tid : tid - p * Npad;
  const int w = tid >> 6, NW = blockDim.x >> 6;
  double *acc = part + P * Npad;
  {
    // Lane roles of the pair loop.  Normally lane (i, p) owns point i and the p-th share of the partner
    // steps.  The last wave of a row holds only A = N - 64*floor((N-1)/64) points; when it is at most
    // half full its 64 lanes are re-dealt as A2 points x HH sub-helpers (A2 = A rounded up to a power
    // of two, HH = 64/A2) that split the row's share HH ways, so that the wave leaves the SIMD after
    // 1/HH of the steps instead of idling most of its lanes through all of them (N = 200: 8 points
    // x 8 sub-helpers).  Row sums and partner sums are LDS atomics, so any dealing gives the same sets.
    const int H = (N - 1) / 2;  // full partner steps
    const int t0 = (P == 1) ? 1 : 1 + (p * H) / P, t1 = (P == 1) ? 1 + H : 1 + ((p + 1) * H) / P;
    const int lastbase = (N - 1) & ~63, A = N - lastbase, l = tid & 63;
    int A2 = 1, lg2 = 0;  // powers of two: every division by A2 or HH below is a shift
    while (A2 < A) { A2 <<= 1; lg2++; }
    const bool lastwave = i >= lastbase;            // wave-uniform
    // a wave BEHIND the last one: a belief that holds fewer points than the slot (N = its count, Npad = the slot's row) leaves
    // whole waves of the row without a point.  They take no part in the pair loop -- dealt the roles of the last wave, as they
    // were through round 4, they added its pairs a second time (the fitted bandwidth of such a belief was off by up to 15 %;
    // found by tests/test_gpu_fit_bracketing.py against the single-precision evaluation, pinned on the oracle there)
    const bool behind = i >= lastbase + 64;         // wave-uniform
    const bool redeal = lastwave && A2 <= 32;
    const int lgH = redeal ? 6 - lg2 : 0, HH = 1 << lgH;
    const int len = t1 - t0;
    // Balance between the waves of a row: after its own points (len / HH steps when re-dealt) the last wave takes the
    // tail steps [t0 + Lw, t1) of every full wave's points, so that all waves of the row leave the pair loop together
    // (N = 200, P = 1: 78 steps in the full waves, 13 + 3 x 21 in the last one, instead of 99 and 13).  A row sum then
    // has two contributors -- its owner and the last wave -- each with ONE atomic add into a slot that starts at zero:
    // a + b = b + a, so the sums do not depend on which comes first.
    const int nfull = lastbase >> 6, own_ws = (A2 <= 32) ? (len + (64 >> lg2) - 1) >> (6 - lg2) : len;
    // Fold (throughput mode, a row of 4k full waves and one more: N = 257 .. 320, the N = 300 of BASELINE's config 5): a
    // workgroup of five busy waves runs at two thirds of the rate of one of four -- two of its waves share a SIMD and the
    // other three wait for them at every barrier of the evaluation (0.96 against 0.69 ps per pair at N = 300 / 256,
    // profiles/r04_lcv_five_wave_rows.txt).  So the full waves take the last wave's points between them, a 1/nfull share
    // of the steps each (second phase below), and the last wave sits the pair loop out: four busy waves, one per SIMD.
    // A wave's share of a folded point's row sum goes into the wave's OWN accumulator row (the combine adds the rows in wave
    // order), not into `part`, where four atomic adds would arrive in any order.
#ifdef NBP_X_NOFOLD
    const bool fold = false;
#else
    const bool fold = (P == 1) && nfull >= 4 && (nfull & 3) == 0;
#endif
    int Lw = len;
    if (A < 64 && nfull > 0 && !fold) Lw = min(len, (own_ws + nfull * len + nfull) / (nfull + 1));
    const int hand = len - Lw;
    double *accw = acc + w * 2 * N;
    // second phase of a wave (wave-uniform): nf2 groups of 64 points from base2 on, c2 steps from step ta2 on
    int nf2 = 0, base2 = 0, ta2 = 0, c2 = 0;
    if (!lastwave) {
      const int n4 = __builtin_amdgcn_readfirstlane(Lw >> 2), nt = __builtin_amdgcn_readfirstlane(Lw & 3);
      const double xi = x[i];
      const double s = circ ? loo_symmetric<true>(x, i, t0, n4, nt, false, xi, K, accw, tab)
                            : loo_symmetric<false>(x, i, t0, n4, nt, false, xi, K, accw, tab);
      lds_add((nbp_lds_double *)(part + p * Npad + i), s);
      if (fold) {  // this wave's share of the steps of the last wave's points (scalar: the wave index through readfirstlane)
        const int ws = __builtin_amdgcn_readfirstlane(w), q = (len + nfull - 1) / nfull, wa = min(len, ws * q);
        nf2 = 1; base2 = lastbase; ta2 = t0 + wa; c2 = min(len, wa + q) - wa;
      }
    } else if (!fold && !behind) {
      const int pi = redeal ? lastbase + (l & (A2 - 1)) : i, hh = redeal ? l >> lg2 : 0;
      const int lenmin = len >> lgH, rem = len - (lenmin << lgH);
      const int ta = t0 + hh * lenmin + min(hh, rem);
      const int n4 = __builtin_amdgcn_readfirstlane(lenmin >> 2), nt = __builtin_amdgcn_readfirstlane(lenmin & 3);
      if (pi < N) {
        const double xi = x[pi];
        const double s = circ ? loo_symmetric<true>(x, pi, ta, n4, nt, hh < rem, xi, K, accw, tab)
                              : loo_symmetric<false>(x, pi, ta, n4, nt, hh < rem, xi, K, accw, tab);
        lds_add((nbp_lds_double *)(part + p * Npad + pi), s);
      }
      if (hand > 0) { nf2 = nfull; base2 = 0; ta2 = t0 + Lw; c2 = hand; }  // the tail steps of every full wave's points
    }
    nf2 = __builtin_amdgcn_readfirstlane(nf2);
    c2 = __builtin_amdgcn_readfirstlane(c2);
    base2 = __builtin_amdgcn_readfirstlane(base2);
    if (nf2 > 0 && c2 > 0) {
      const int h4 = c2 >> 2, ht = c2 & 3;
      const int sta = __builtin_amdgcn_readfirstlane(ta2);
      double *rowdst = fold ? accw : part + p * Npad;
      for (int f = 0; f < nf2; f++) {
        const int pj = base2 + 64 * f + l;
        if (pj < N) {
          const double xj = x[pj];
          const double s = circ ? loo_symmetric<true>(x, pj, sta, h4, ht, false, xj, K, accw, tab)
                                : loo_symmetric<false>(x, pj, sta, h4, ht, false, xj, K, accw, tab);
          lds_add((nbp_lds_double *)(rowdst + pj), s);
        }
      }
    }
    if ((N & 1) == 0 && p == P - 1 && i < N / 2) {  // antipodal partner, once per pair; both ends through the per-wave
      const int j = i + N / 2;                        // accumulators (program order within a wave: deterministic)
      const double d = x[i] - x[j];
      const double e = lcv_exp(circ ? circ_sq(d) : d * d, K, tab);
      lds_add((nbp_lds_double *)(accw + i), e);
      lds_add((nbp_lds_double *)(accw + j), e);
    }
  }
  NBP_CTICK(20);  // pair loop
  __syncthreads();
  NBP_CTICK(21);  // barrier 1
  double term = 0;
  if (P == 1) {
    // one helper row (throughput mode): lane i folds all accumulator rows of its point itself -- no second pass through
    // `part`, one barrier fewer; the same additions in the same order as the two-pass form
    if (i < N) {
      double s = part[i];
      part[i] = 0.0;
      double *a = acc + i;  // a running pointer and no unrolling: sixteen hoisted row addresses would live across the whole search
#pragma unroll 1
      for (int q = 0; q < NW; q++, a += 2 * N) {
        s += a[0] + a[N];
        a[0] = 0.0;
        a[N] = 0.0;
      }
      if (s < 1e-300) s = 1e-300;
      term = lcv_log(s * inv_h) - lognorm0;
    }
    NBP_CTICK(22);
    NBP_CTICK(23);
  } else {
    // combine: helper p folds the accumulator rows p, p+P, ... of point i (and clears them)
    if (i < N) {
      double s = part[p * Npad + i];
      double *a = acc + p * 2 * N + i;
#pragma unroll 1
      for (int q = p; q < NW; q += P, a += P * 2 * N) {
        s += a[0] + a[N];
        a[0] = 0.0;
        a[N] = 0.0;
      }
      part[p * Npad + i] = s;
    }
    NBP_CTICK(22);  // combine
    __syncthreads();
    NBP_CTICK(23);  // barrier 2
    if (p == 0 && i < N) {
      double s = 0;
      for (int q = 0; q < P; q++) {  // read-and-clear: the next evaluation accumulates into zeros
        s += part[q * Npad + i];
        part[q * Npad + i] = 0.0;
      }
      if (s < 1e-300) s = 1e-300;
      term = lcv_log(s * inv_h) - lognorm0;
    }
  }
  // only the first Npad lanes hold terms: reduce their waves
  term = wave_sum(term);
  if ((threadIdx.x & 63) == 0 && threadIdx.x < Npad) red[threadIdx.x >> 6] = term;
  NBP_CTICK(24);  // log + wave reduction
  __syncthreads();
  NBP_CTICK(25);  // barrier 3
  double tsum = red[0];
  for (int q = 1; q < (Npad >> 6); q++) tsum += red[q];
  return -tsum / (double)N;
}

// ------------------------------------------------------------------------------------------------
// The same likelihood in SINGLE precision, for the evaluations of the search that only BRACKET the minimum.  A golden-section
// search uses a likelihood value for one thing: the comparison f2 < f1.  While the two values are far apart -- the first nine
// or ten of the sixteen evaluations of a typical fit -- the comparison can be decided on values that carry an error bound far
// below their distance; the search below (lcv_bandwidth_1d) takes a single-precision value only when |f1 - f2| exceeds the
// sum of the two bounds, re-evaluates in double precision what it cannot decide, and from then on evaluates in double
// precision only.  Every comparison has the outcome the all-double search has, so the selected bandwidth is bit-identical.
//   exp(-d^2 / (2 h^2)) = 2^(-(d c)^2), c = sqrt(log2(e) / 2) / h: the points are centred, scaled by c and rounded to single
// precision ONCE per evaluation (LDS), a pair then costs v_sub, v_mul, v_exp_f32, v_add -- no table, no polynomial, no
// partner atomic: every lane sums over ALL partners of its own point (ordered pairs; the transcendental unit does the work
// the twelve double-precision operations of lcv_exp4 do), the partners arriving four at a time as broadcast ds_read_b128.
// The term of a point with itself is masked out (only in the sixteen groups of four that hold the points of the lane's own
// wave; the other groups run without the select).
// Error of the returned value against the double-precision one (u = 2^-24, X = the largest scaled coordinate, q = the
// exponent -log2 of a term): |df| <= max_i |d log s_i| <= the largest relative error of a term that matters plus that of the sum.
//   scaled points rounded to single precision: |dx| <= u X; the difference: |dd| <= 2 u X + u |d|;
//   q = d^2 rounded: |dq| <= 4 u sqrt(q) X + 3 u q; v_exp_f32: 1 ulp = 2 u  => a term: ln2 (4 u sqrt(q) X + 3 u q) + 2 u
//   terms that matter: s_i >= 2^-40 (below: invalid) and N <= 2^8 per 2^-24 of weight => q <= 72: u (23.5 X + 152)
//   the sums: <= N / 4 + 2 additions of positive terms per lane, u each; v_log_f32: 1 ulp of |log2| <= 46: 64 u; slack 34 u
//   => bound E = u (24 X + 314 + N / 4)   (lcv_f32_bound; the differences measured are ~1e-3 of it: rounding errors do not
//      line up over N^2 terms)
// Returns NaN when some point's sum is below 2^-40 (an isolated point: its nearest neighbour decides the sum and the bound
// above does not cover it) -- the caller then evaluates in double precision.
// LDS: the scaled points live in the first row of the partner accumulators (all zero between evaluations, restored here).
// ------------------------------------------------------------------------------------------------
#ifndef NBP_LCV_F32
#define NBP_LCV_F32 1
#endif
// CIRC: the angles live on the circle as 32-bit integers (x 2^32 / 2 pi): the difference of two of them, taken modulo 2^32, IS the
// geodesic difference -- no wrap, no min -- and it is exact; what single precision rounds is the difference, not the angle
// (the scaled angle 2 pi c can be hundreds of kernel widths for a belief with narrow modes: rounded to single precision first,
// it would carry that many times the error)
template <bool CIRC>
__device__ __forceinline__ void loo_f32_quad(const float4 y, float xi, float K, float &e0, float &e1, float &e2, float &e3) {
  float d0, d1, d2, d3;
  if (CIRC) {
    const unsigned int xq = __float_as_uint(xi);  // (unsigned: the wrap-around is the point)
    d0 = (float)(int)(xq - __float_as_uint(y.x)) * K;
    d1 = (float)(int)(xq - __float_as_uint(y.y)) * K;
    d2 = (float)(int)(xq - __float_as_uint(y.z)) * K;
    d3 = (float)(int)(xq - __float_as_uint(y.w)) * K;
  } else {
    d0 = xi - y.x; d1 = xi - y.y; d2 = xi - y.z; d3 = xi - y.w;
  }
  e0 = __builtin_amdgcn_exp2f(-d0 * d0);
  e1 = __builtin_amdgcn_exp2f(-d1 * d1);
  e2 = __builtin_amdgcn_exp2f(-d2 * d2);
  e3 = __builtin_amdgcn_exp2f(-d3 * d3);
}
// groups [ga, gb) of four partners, all of them points (the group that holds the padding is the caller's: loo_f32_tail)
template <bool CIRC, bool MASK>
__device__ __forceinline__ void loo_f32_groups(const float4 *x4, int ga, int gb, float xi, int i, float K, float &s0, float &s1, float &s2,
                                               float &s3) {
#pragma unroll 2
  for (int g = ga; g < gb; g++) {
    float e0, e1, e2, e3;
    loo_f32_quad<CIRC>(x4[g], xi, K, e0, e1, e2, e3);
    if (MASK) {
      const int j = 4 * g;
      e0 = (j == i) ? 0.f : e0;
      e1 = (j + 1 == i) ? 0.f : e1;
      e2 = (j + 2 == i) ? 0.f : e2;
      e3 = (j + 3 == i) ? 0.f : e3;
    }
    s0 += e0;
    s1 += e1;
    s2 += e2;
    s3 += e3;
  }
}
// one group with the point itself and the entries beyond the N-th masked out
template <bool CIRC>
__device__ __forceinline__ void loo_f32_tail(const float4 *x4, int g, bool on, float xi, int i, int N, float K, float &s0, float &s1, float &s2,
                                             float &s3) {
  float e0, e1, e2, e3;
  loo_f32_quad<CIRC>(x4[g], xi, K, e0, e1, e2, e3);
  const int j = 4 * g;
  s0 += (!on || j == i || j >= N) ? 0.f : e0;
  s1 += (!on || j + 1 == i || j + 1 >= N) ? 0.f : e1;
  s2 += (!on || j + 2 == i || j + 2 >= N) ? 0.f : e2;
  s3 += (!on || j + 3 == i || j + 3 >= N) ? 0.f : e3;
}
// a last wave of at most 32 points, re-dealt as A2 points x HH = 64 / A2 sub-helpers (A2 = the point count rounded up to a
// power of two): sub-helper hh of a point takes the hh-th part of the row's groups -- the wave leaves its SIMD after 1 / HH of
// the steps (N = 200: 8 points x 8 sub-helpers; the workgroup's four waves cost 3.1 wave-loops instead of 4) -- and the
// parts are added up by a butterfly over the lanes of the point.
template <bool CIRC>
__device__ __forceinline__ float loo_f32_redealt(const float4 *x4, int ga, int gb, float xi, int pi, int N, int hh, int lgH, float K) {
  const int len = gb - ga, a = ga + ((hh * len) >> lgH), b = ga + (((hh + 1) * len) >> lgH);
  const int steps = __builtin_amdgcn_readfirstlane((len + (1 << lgH) - 1) >> lgH);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  for (int t = 0; t < steps; t++) {
    const bool on = a + t < b;
    loo_f32_tail<CIRC>(x4, on ? a + t : ga, on, xi, pi, N, K, s0, s1, s2, s3);
  }
  float sf = (s0 + s1) + (s2 + s3);
  for (int o = 64 >> lgH; o < 64; o <<= 1) sf += __shfl_xor(sf, o, 64);
  return sf;
}
// |f_single - f_double| <= this (see above; u = 2^-24): u (24 X + 250 + N / 4) for the sums + 64 u for log2 in single precision
// (circular coordinates: the differences of the integer angles are exact and their conversion rounds like the subtraction it
// replaces; what is left of the representation term is the grid itself, 2 pi / 2^32 per angle: xmax = 0.0125 stands for it)
__device__ __forceinline__ double lcv_f32_bound(double xmax, double h, int N) {
  return 5.9604644775390625e-8 * (24.0 * (xmax * (0.84932180028801907 / h)) + 314.0 + 0.25 * (double)N);
}
__device__ __forceinline__ double neg_loo_ll_f32(const double *x, int N, int Npad, bool circ, double h, double lognorm0, double cen,
                                                 double *part, double *red) {
  const int tid = threadIdx.x;
  const int P = ((int)blockDim.x == Npad) ? 1 : (int)blockDim.x / Npad;
  const int p = (P == 1) ? 0 : tid / Npad, i = (P == 1) ? tid : tid - p * Npad;
  const double inv_h = 1.0 / h, scl = inv_h * 0.84932180028801907;  // sqrt(log2(e) / 2) / h
  double *acc = part + P * Npad;
  float *xf = (float *)(acc + (((uintptr_t)acc >> 3) & 1));  // 16-byte aligned (an offset, not a cast: the pointer stays an LDS pointer)
  const int G4 = (N + 3) >> 2;
  // (circular: the angle as a 32-bit integer, 2^32 to the turn -- the conversion through 64 bits wraps pi itself to -2^31)
  for (int j = tid; j < 4 * G4; j += blockDim.x) {
    if (circ) ((int *)xf)[j] = (j < N) ? (int)(long long)rint(x[j] * 683565275.57643159) : 0;
    else xf[j] = (j < N) ? (float)((x[j] - cen) * scl) : 0.f;
  }
  __syncthreads();
  const float K = (float)(scl * 1.4629180792671596e-9);  // circular: scaled radians per integer step (2 pi c / 2^32)
  const int full = N >> 2;                                // groups of four points; group `full` (if N % 4) holds the padding
  // row p of the workgroup takes the p-th share of the groups; the groups that hold the points of this lane's own wave are
  // the ones that need the self term masked (all bounds wave-uniform: scalar loops)
  const int ga = __builtin_amdgcn_readfirstlane((p * G4) / P), gb = __builtin_amdgcn_readfirstlane(((p + 1) * G4) / P);
  const int lastbase = (N - 1) & ~63, A = N - lastbase;
  int lg2 = 0;
  while ((1 << lg2) < A) lg2++;
  const bool redeal = __builtin_amdgcn_readfirstlane((int)(i >= lastbase && lg2 <= 5)) != 0;
  const bool behind = __builtin_amdgcn_readfirstlane((int)(i >= lastbase + 64)) != 0;  // a wave without a point (a belief of fewer points than the slot)
  const float4 *x4 = (const float4 *)xf;
  float sf = 0.f;
  if (behind) {
  } else if (redeal) {
    const int l = tid & 63, pi = lastbase + (l & ((1 << lg2) - 1)), hh = l >> lg2;
    const float xi = (pi < N) ? xf[pi] : 0.f;
    sf = circ ? loo_f32_redealt<true>(x4, ga, gb, xi, pi, N, hh, 6 - lg2, K) : loo_f32_redealt<false>(x4, ga, gb, xi, pi, N, hh, 6 - lg2, K);
  } else {
    const float xi = (i < N) ? xf[i] : 0.f;
    const int o0 = __builtin_amdgcn_readfirstlane((i & ~63) >> 2), gf = min(gb, full);
    const int m0 = max(ga, min(gf, o0)), m1 = max(ga, min(gf, o0 + 16));
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (circ) {
      loo_f32_groups<true, false>(x4, ga, m0, xi, i, K, s0, s1, s2, s3);
      loo_f32_groups<true, true>(x4, m0, m1, xi, i, K, s0, s1, s2, s3);
      loo_f32_groups<true, false>(x4, m1, gf, xi, i, K, s0, s1, s2, s3);
      if (gb > full) loo_f32_tail<true>(x4, full, true, xi, i, N, K, s0, s1, s2, s3);
    } else {
      loo_f32_groups<false, false>(x4, ga, m0, xi, i, K, s0, s1, s2, s3);
      loo_f32_groups<false, true>(x4, m0, m1, xi, i, K, s0, s1, s2, s3);
      loo_f32_groups<false, false>(x4, m1, gf, xi, i, K, s0, s1, s2, s3);
      if (gb > full) loo_f32_tail<false>(x4, full, true, xi, i, N, K, s0, s1, s2, s3);
    }
    sf = (s0 + s1) + (s2 + s3);
  }
  if (P > 1 && i < N) part[p * Npad + i] = (double)sf;  // (`part` is all zero between evaluations)
  __syncthreads();
  for (int j = tid; j < 2 * G4 + 2; j += blockDim.x) acc[j] = 0.0;  // the accumulator row as the double-precision evaluation expects it
  // f = -(ln2 / N) sum_i log2(s_i) + log(h) + lognorm0: the logarithm of a point's sum in single precision (v_log_f32),
  // log(h) in double precision by ONE wave (the last of row 0: the re-dealt one when there is one)
  double term = 0;
  if (p == 0 && i < N) {
    float s = sf;
    if (P > 1) {
      double sd = 0;
      for (int q = 0; q < P; q++) {
        sd += part[q * Npad + i];
        part[q * Npad + i] = 0.0;
      }
      s = (float)sd;
    }
    // 2^-40: below it the sum of an isolated point; NaN marks the evaluation invalid (and swallows a NaN / Inf coordinate)
    term = (s >= 9.094947017729282e-13f) ? (double)__builtin_amdgcn_logf(s) : __builtin_nan("");
  }
  term = wave_sum(term);
  if ((threadIdx.x & 63) == 0 && threadIdx.x < Npad) red[threadIdx.x >> 6] = term;
  if (__builtin_amdgcn_readfirstlane((int)(p == 0 && (i >> 6) == (lastbase >> 6)))) {
    const double lh = lcv_log(h);
    if ((tid & 63) == 0) red[32] = lh;
  }
  __syncthreads();
  double tsum = red[0];
  for (int q = 1; q < (Npad >> 6); q++) tsum += red[q];
  return fma(-0.69314718055994531 / (double)N, tsum, red[32] + lognorm0);
}

__device__ __forceinline__ double lcv_bandwidth_1d(const double *x, int N, int Npad, bool circ, double *part, double *red,
                                                   const double *tab, nbp_counters *ctr) {
  const int i = threadIdx.x;
  double lo = INFINITY, hi = -INFINITY, mn = INFINITY;
  if (i < N) {
    lo = hi = x[i];
    if (i + 1 < N) {
      double d = x[i] - x[i + 1];
      if (circ) d = wrap_pi(d);
      mn = fabs(d);
    }
  }
  {  // per-wave partner accumulators start at zero (neg_loo_ll keeps them zero between calls)
    double *acc = part + (blockDim.x / Npad) * Npad;
    for (int q = threadIdx.x; q < (int)(blockDim.x >> 6) * 2 * N; q += blockDim.x) acc[q] = 0.0;
    for (int q = threadIdx.x; q < (int)blockDim.x; q += blockDim.x) part[q] = 0.0;  // row sums are atomics too
  }
  double minm = block_min(mn, red);
  lo = block_min(lo, red);
  hi = block_max(hi, red);
  double maxm = circ ? NBP_TWO_PI : (hi - lo);
  if (!(maxm > 0)) return 1.0;
  if (minm < 1e-6 * maxm) minm = 1e-6 * maxm;
  const double sc = 0.5 * (minm + maxm);
  const double ax = minm / sc, bx = 1.0, cx = maxm / sc;
  const double R = 0.61803399, C = 1.0 - R, tol = 1e-2;
  const double lognorm0 = 0.5 * log(NBP_TWO_PI) + log((double)(N - 1));
  const double scs = sgpr_double(sc), ln0 = sgpr_double(lognorm0);
  double x0 = ax, x3 = cx, x1, x2;
  if (fabs(cx - bx) > fabs(bx - ax)) { x1 = bx; x2 = bx + C * (cx - bx); }
  else { x2 = bx; x1 = bx - C * (bx - ax); }
  // one call site for each evaluation (the six pair loops of the double-precision one are inlined once per kernel, not once
  // per call): the two initial values, then one new point per iteration.
  // Bracketing in single precision (neg_loo_ll_f32): e1 / e2 = the error bounds of f1 / f2 (0: a double-precision value).  A
  // comparison the bounds do not decide re-evaluates its single-precision sides in double precision -- `todo` 3 / 4 -- and ends
  // the single-precision phase; every decision is then the all-double search's, and so is every position and the result.
  const double cen = circ ? 0.0 : 0.5 * (lo + hi), xmax = circ ? 0.0125 : 0.5 * (hi - lo);  // (circular: the integer grid's half step on both angles, as a coordinate of that size)
  bool m32 = NBP_LCV_F32 && !(ctr && (ctr->flags & 1));
  double f1 = 0.0, f2 = 0.0, e1 = 0.0, e2 = 0.0, pt = x1;
  unsigned int nev = 0, nev32 = 0;
  int todo = 0;  // 0 / 1: the initial values at x1 / x2; 2: the new point of an iteration; 3 / 4: f1 / f2 again, in double precision
  bool c = false;
  while (true) {
    double v, ev = 0.0;
    if (m32 && todo <= 2) {
      v = neg_loo_ll_f32(x, N, Npad, circ, pt * scs, ln0, cen, part, red);
      nev32++;
      if (!(fabs(v) < INFINITY)) { m32 = false; continue; }  // invalid (an isolated point, NaN): the same point in double precision
      ev = lcv_f32_bound(xmax, pt * scs, N);
    } else {
      v = neg_loo_ll(x, N, Npad, circ, pt * scs, ln0, part, red, tab);
      nev++;
    }
    if (todo == 0) { f1 = v; e1 = ev; todo = 1; pt = x2; continue; }
    if (todo == 1 || (todo == 2 && c) || todo == 4) { f2 = v; e2 = ev; }
    else { f1 = v; e1 = ev; }
    if ((e1 > 0 || e2 > 0) && !(fabs(f1 - f2) > e1 + e2)) {  // undecided: the single-precision sides again
      m32 = false;
      if (e1 > 0) { todo = 3; pt = x1; }
      else { todo = 4; pt = x2; }
      continue;
    }
    if (!(fabs(x3 - x0) > tol * (fabs(x1) + fabs(x2)))) break;
    c = f2 < f1;
    // positions with one rounding per operation, as Julia evaluates KernelDensityEstimate's golden section (see golden_step)
    if (c) { x0 = x1; x1 = x2; x2 = R * x1 + C * x3; f1 = f2; e1 = e2; pt = x2; }
    else { x3 = x2; x2 = x1; x1 = R * x2 + C * x0; f2 = f1; e2 = e1; pt = x1; }
    todo = 2;
  }
  if (ctr && threadIdx.x == 0) {
    atomicAdd(&ctr->lcv_evals, (unsigned long long)nev);
    if (nev32) atomicAdd(&ctr->lcv_evals_f32, (unsigned long long)nev32);
  }
  return (f1 < f2 ? x1 : x2) * scs;
}

// ------------------------------------------------------------------------------------------------
// The same search for launches that cannot fill the chip (the top of the tree: a handful of fits, each pinned at what
// ONE CU does for 16-21 dependent likelihood evaluations).  K = 3 or 7 workgroups on as many CUs serve one fit: in every
// round they evaluate, concurrently, the point the search needs next (its position follows from values already known)
// and the points it would need in the one or two iterations after that, one per outcome of the comparisons in between --
// the positions of a golden-section search depend on comparison OUTCOMES only.  One device-scope rendezvous per round
// then advances the search two or three iterations.  Every likelihood value is computed by the same code at the same point as in the sequential
// search, so the selected bandwidth is bit-identical.  A workgroup that waits too long for its peers (they could in
// principle not be resident) stops waiting and evaluates all three points itself: slower, never stuck.
// ------------------------------------------------------------------------------------------------
#define NBP_SPEC_KMAX 7
#define NBP_SPEC_ROUNDS 24
struct nbp_spec_area {  // one per (fit job, coordinate); filled with ones (0xFF bytes) by the host before the launch
  unsigned long long f[NBP_SPEC_ROUNDS][8];  // published values (bit patterns), [round][role]
  unsigned int cnt[NBP_SPEC_ROUNDS];         // arrivals per round
  unsigned int pad_[8];
};
struct golden_state {
  double x0, x1, x2, x3, f1, f2;
};
// the positional update of one iteration given the outcome c = (f2 < f1); returns the point the iteration evaluates
// (its value belongs in f2 when c, in f1 otherwise)
__device__ __forceinline__ double golden_step(golden_state &g, bool c, double R, double C) {
  // one rounding per operation (the library is compiled without contraction): the sequential search, this one and the CPU
  // checker's walk through bit-identical positions
  // Both outcomes as selects (the values of `if (c) {x0 = x1; x1 = x2; x2 = R x1 + C x3; f1 = f2} else {x3 = x2; x2 = x1;
  // x1 = R x2 + C x0; f2 = f1}`): with branches the compiler merges the conditional stores into "store to field f1 or
  // f2" and the whole state goes to scratch (48 B per lane in the speculative kernels)
  const double up = R * g.x2 + C * g.x3, dn = R * g.x1 + C * g.x0;
  const double x0 = c ? g.x1 : g.x0, x1 = c ? g.x2 : dn, x2 = c ? up : g.x1, x3 = c ? g.x3 : g.x2;
  const double f1 = c ? g.f2 : g.f1, f2 = c ? g.f2 : g.f1;
  g.x0 = x0; g.x1 = x1; g.x2 = x2; g.x3 = x3; g.f1 = f1; g.f2 = f2;
  return c ? x2 : x1;
}
__device__ __forceinline__ bool golden_done(const golden_state &g, double tol) { return !(fabs(g.x3 - g.x0) > tol * (fabs(g.x1) + fabs(g.x2))); }

// DEPTH iterations per rendezvous, K = 2^DEPTH - 1 workgroups per fit.  Roles are the nodes of the outcome tree in heap
// order: node 1 = the iteration whose comparison is already decided by known values; node 2n / 2n+1 = the iteration that
// follows node n when the comparison after n comes out true / false.
#ifndef NBP_SPEC_PATIENCE
#define NBP_SPEC_PATIENCE 512  // polls before a role gives up on its peers (experiments: tools/exp/concurrency_probe.sh)
#endif
template <int DEPTH>
__device__ __forceinline__ double lcv_bandwidth_1d_spec(const double *x, int N, int Npad, bool circ, double *part, double *red, const double *tab,
                                                        nbp_counters *ctr, nbp_spec_area *area, int role) {
  constexpr int K = (1 << DEPTH) - 1;
  const int i = threadIdx.x;
  double lo = INFINITY, hi = -INFINITY, mn = INFINITY;
  if (i < N) {
    lo = hi = x[i];
    if (i + 1 < N) {
      double d = x[i] - x[i + 1];
      if (circ) d = wrap_pi(d);
      mn = fabs(d);
    }
  }
  {
    double *acc = part + (blockDim.x / Npad) * Npad;
    for (int q = threadIdx.x; q < (int)(blockDim.x >> 6) * 2 * N; q += blockDim.x) acc[q] = 0.0;
    for (int q = threadIdx.x; q < (int)blockDim.x; q += blockDim.x) part[q] = 0.0;
  }
  double minm = block_min(mn, red);
  lo = block_min(lo, red);
  hi = block_max(hi, red);
  const double maxm = circ ? NBP_TWO_PI : (hi - lo);
  if (!(maxm > 0)) return 1.0;
  if (minm < 1e-6 * maxm) minm = 1e-6 * maxm;
  const double sc = 0.5 * (minm + maxm);
  const double ax = minm / sc, bx = 1.0, cx = maxm / sc;
  const double R = 0.61803399, C = 1.0 - R, tol = 1e-2;
  const double lognorm0 = 0.5 * log(NBP_TWO_PI) + log((double)(N - 1));
  golden_state g;
  g.x0 = ax; g.x3 = cx;
  if (fabs(cx - bx) > fabs(bx - ax)) { g.x1 = bx; g.x2 = bx + C * (cx - bx); }
  else { g.x2 = bx; g.x1 = bx - C * (bx - ax); }
  // uniform state in SGPRs across the evaluations (see lcv_bandwidth_1d)
  const double scs = sgpr_double(sc), ln0 = sgpr_double(lognorm0);
  g.f1 = g.f2 = 0.0;
  auto eval = [&](double xs) { return neg_loo_ll<false>(x, N, Npad, circ, xs * scs, ln0, part, red, tab); };
  bool solo = false;  // gave up on the peers: the sequential search from here on
  int round = 0;
  unsigned int nev = 2;
  // the values of a round stay where the rendezvous put them (LDS): picked by a run-time node index below, a register
  // array would live in scratch (112 B per lane, written back to HBM once per launch)
  const double *vals = red + 48;
  // rendezvous: publish the value of this role's point, collect all K values; false = on our own from now on
  auto rendezvous = [&](double mine) -> bool {
    if (round >= NBP_SPEC_ROUNDS) return false;
    // one lane talks to memory: it publishes this role's value, waits for the K arrivals, fetches the K values and hands
    // them to the workgroup through the reduction scratch (slots 48 .. 63; its two halves alternate by
    // round so that one barrier per rendezvous is enough)
    double *box = red + 48 + (round & 1) * (K + 1);
    if (threadIdx.x == 0) {
      // Relaxed device-scope atomics only: a release / acquire pair at agent scope costs a write-back and an
      // invalidation of the XCD's L2 (microseconds -- as much as the evaluation it is meant to save).  The values
      // themselves are the flags: the host fills the area with ones (a NaN pattern no likelihood takes), a role's slot
      // turns into its value with one 8-byte atomic store, and the reader polls the K slots until none is blank.
      const unsigned long long blank = ~0ull;
      __hip_atomic_store(&area->f[round][role], (unsigned long long)__double_as_longlong(mine), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      int ok = 0;
      unsigned long long got[K];
      // Patience: a few hundred polls (a poll is a round trip to memory, ~1 us) = many evaluations.  Peers that have not
      // published by then are not running (other contexts' kernels hold the CUs): going on alone costs this fit its
      // speed-up, waiting longer would cost the whole solve its latency.
      for (int spin = 0; spin < NBP_SPEC_PATIENCE && !ok; spin++) {
        ok = 1;
#pragma unroll
        for (int r = 0; r < K; r++) {
          got[r] = __hip_atomic_load(&area->f[round][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok &= got[r] != blank;
        }
      }
      box[0] = (double)ok;
#pragma unroll
      for (int r = 0; r < K; r++) box[r + 1] = __longlong_as_double((long long)got[r]);
    }
    __syncthreads();
    const bool ok_ = box[0] != 0.0;
    vals = box;  // vals[r + 1] = the value of role r
    round++;
    return ok_;
  };
  // One call site for the evaluation: every pass of the loop decides which point (if any) this workgroup evaluates,
  // evaluates it, then consumes the value -- start-up (roles 0 and 1 hold the two initial points), speculative rounds,
  // or the sequential search once the peers are given up on.
  bool starting = true, c = false;
  int solo_stage = 0;
  while (true) {
    double pt = 0.0;
    bool doit = false;
    if (starting) {
      if (!solo) { doit = role < 2; pt = role == 0 ? g.x1 : g.x2; }
      else { doit = true; pt = solo_stage == 0 ? g.x1 : g.x2; }
    } else if (golden_done(g, tol)) {
      break;
    } else if (solo) {
      c = g.f2 < g.f1;
      pt = golden_step(g, c, R, C);
      doit = true;
      nev++;
    } else {
      // this role's point: replay the outcomes on the path from the root of the outcome tree to node role + 1
      const int node = role + 1;
      int depth = 0;
      while ((node >> (depth + 1)) != 0) depth++;
      golden_state gs = g;
      pt = golden_step(gs, g.f2 < g.f1, R, C);
      doit = true;  // false: the search stops before it gets to this node
      for (int lvl = depth - 1; lvl >= 0; lvl--) {
        if (golden_done(gs, tol)) { doit = false; break; }
        pt = golden_step(gs, ((node >> lvl) & 1) == 0, R, C);
      }
    }
    const double v = doit ? eval(pt) : 0.0;  // block-uniform
    if (starting) {
      if (!solo) {
        if (rendezvous(v)) { g.f1 = vals[1]; g.f2 = vals[2]; starting = false; }
        else solo = true;  // evaluate both initial points ourselves
      } else if (solo_stage == 0) { g.f1 = v; solo_stage = 1; }
      else { g.f2 = v; starting = false; }
    } else if (solo) {
      // (selects, not a store to "f1 or f2": a field picked at run time puts the whole state into scratch)
      g.f2 = c ? v : g.f2;
      g.f1 = c ? g.f1 : v;
    } else {
      if (!rendezvous(v)) { solo = true; continue; }
      // advance up to DEPTH iterations with the values now known
      int node = 1;
      bool cc = g.f2 < g.f1;
#pragma unroll
      for (int lvl = 0; lvl < DEPTH; lvl++) {
        (void)golden_step(g, cc, R, C);
        const double vn = vals[node];
        g.f2 = cc ? vn : g.f2;
        g.f1 = cc ? g.f1 : vn;
        nev++;
        if (lvl == DEPTH - 1 || golden_done(g, tol)) break;
        cc = g.f2 < g.f1;
        node = 2 * node + (cc ? 0 : 1);
      }
    }
  }
  if (ctr && threadIdx.x == 0 && role == 0) atomicAdd(&ctr->lcv_evals, (unsigned long long)nev);
  return (g.f1 < g.f2 ? g.x1 : g.x2) * scs;
}

// rand(Categorical(p)) by inverse CDF on one uniform
__device__ __forceinline__ int categorical(const double *p, int n, double u) {
  double c = 0;
  int last = 0;
  for (int i = 0; i < n; i++) {
    if (p[i] > 0) last = i;
    c += p[i];
    if (u < c) return i;
  }
  return last;
}

// ------------------------------------------------------------------------------------------------
// _prepareHypoRecipe! (ExplicitDiscreteMarginalizations.jl:142-289), integer part.  1-based
// variable indices like the reference; hypothesis 0 = null hypothesis.
// ------------------------------------------------------------------------------------------------
struct recipe_t {
  int ngroups;
  int hypo[NBP_MAXV + 1];
  int nact[NBP_MAXV + 1];
  int act[NBP_MAXV + 1][NBP_MAXV];
  int empty[NBP_MAXV + 1];
  int ncertain;
  int certain[NBP_MAXV];
  double cat_p[NBP_MAXV + 1];
  int cat_first, ncat;
};

__device__ __forceinline__ bool in_list(const int *l, int n, int v) {
  for (int i = 0; i < n; i++)
    if (l[i] == v) return true;
  return false;
}
__device__ __forceinline__ void sorted_union(const int *a, int na, int v, int *out, int *nout) {
  int n = 0;
  bool placed = in_list(a, na, v);
  for (int i = 0; i < na; i++) {  // `a` (certainidx) is ascending
    if (!placed && v < a[i]) { out[n++] = v; placed = true; }
    out[n++] = a[i];
  }
  if (!placed) out[n++] = v;
  *nout = n;
}

__device__ __forceinline__ void build_recipe(const nbp_proposal_desc *d, recipe_t *R) {
  const int nvars = d->nvars, sf1 = d->sfidx + 1;
  R->ncertain = 0;
  for (int g = 0; g <= NBP_MAXV; g++) { R->nact[g] = 0; R->empty[g] = 0; R->hypo[g] = 0; }
  if (!d->has_multihypo) {
    R->ncertain = nvars;
    for (int i = 0; i < nvars; i++) R->certain[i] = i + 1;
    R->ngroups = nvars + 1;
    for (int g = 0; g <= nvars; g++) {
      R->hypo[g] = g;
      if (g == 0) { R->nact[g] = 1; R->act[g][0] = sf1; }
      else if (g == 1) { R->nact[g] = nvars; for (int i = 0; i < nvars; i++) R->act[g][i] = i + 1; }
      else R->empty[g] = 1;
    }
    R->ncat = 2; R->cat_first = 0;
    R->cat_p[0] = d->nullhypo; R->cat_p[1] = 1.0 - d->nullhypo;
    return;
  }
  int unc[NBP_MAXV], nunc = 0;
  for (int i = 0; i < nvars; i++) {
    if (d->multihypo[i] == 0.0) R->certain[R->ncertain++] = i + 1;
    else if (0.0 < d->multihypo[i]) unc[nunc++] = i + 1;
  }
  const bool sfunc = in_list(unc, nunc, sf1), sfincer = in_list(R->certain, R->ncertain, sf1);
  // :161-172 -- uninitialised hypotheses are suppressed in the draw of mhidx when fewer than nvars-1
  // variables are initialised (isinit flags: has_multihypo bit 7 = present, bit 8+k = variable k)
  double mhs[NBP_MAXV];
  for (int i = 0; i < NBP_MAXV; i++) mhs[i] = (i < nvars) ? d->multihypo[i] : 0.0;
  if (d->has_multihypo & 0x80) {
    int ninit = 0;
    for (int i = 0; i < nvars; i++) ninit += (d->has_multihypo >> (8 + i)) & 1;
    if (ninit < nvars - 1) {
      double tot = 0;
      for (int i = 0; i < nvars; i++) {
        if (!((d->has_multihypo >> (8 + i)) & 1) && i + 1 != sf1) mhs[i] = 0.0;
        tot += mhs[i];
      }
      for (int i = 0; i < nvars; i++) mhs[i] /= tot;
    }
  }
  int np = 0, pidx0;
  if (sfunc) {
    double nhw = (double)(nunc + 1), tot = 0;
    R->cat_p[np++] = 1.0 / nhw;
    for (int i = 0; i < nvars; i++) R->cat_p[np++] = (double)nunc / nhw * mhs[i];
    for (int i = 0; i < np; i++) tot += R->cat_p[i];
    for (int i = 0; i < np; i++) R->cat_p[i] /= tot;
    pidx0 = 0;
  } else {
    for (int i = 0; i < nvars; i++) R->cat_p[np++] = mhs[i];
    pidx0 = 1;
  }
  R->ncat = np; R->cat_first = pidx0; R->ngroups = np;
  for (int g = 0; g < np; g++) {
    const int pidx = pidx0 + g;
    const bool pidxincer = in_list(R->certain, R->ncertain, pidx);
    R->hypo[g] = pidx;
    if (!pidxincer && sfincer && pidx != 0) sorted_union(R->certain, R->ncertain, pidx, R->act[g], &R->nact[g]);
    else if (((pidxincer && !sfincer) || sf1 == pidx) && pidx != 0) sorted_union(R->certain, R->ncertain, sf1, R->act[g], &R->nact[g]);
    else if (pidxincer && sfincer && pidx != 0) { R->nact[g] = 0; R->empty[g] = 1; }
    else if (!pidxincer && !sfincer && pidx != 0) { R->nact[g] = nunc; for (int i = 0; i < nunc; i++) R->act[g][i] = unc[i]; }
    else { R->nact[g] = 1; R->act[g][0] = sf1; }
  }
}
