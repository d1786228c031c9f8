// nbp_kernels.h -- the gfx950 kernels of libnbp (see nbp_device.h for the shared device code).
//
// Geometries: the proposal kernel runs one lane per particle; the bandwidth fits and the
// KD builds run P x Npad lanes (Npad = roundup(N, 64), lane (i, p): i = tid % Npad the point, p its
// helper index, all lanes of a wave share p); the product kernel runs HL adjacent lanes per output
// sample.  The host picks P / HL per launch from the batch size (latency vs throughput mode).
#pragma once
#include <type_traits>
#include "nbp_device.h"

// Translation units: libnbp is built from several .hip files in parallel (tools/build_lib.py), each defining one group of
// kernels (NBP_TU = the groups this file defines; everything else is declared only, so that the host code in nbp_api.hip
// can launch it).  A file that does not set NBP_TU defines every kernel (single-file build).
#define NBP_TU_PROPOSAL 1   // proposal / deconv kernels + the small copy / reseed / resample kernels
#define NBP_TU_PREP 2       // bandwidth fits + KD builds, sequential search
#define NBP_TU_PREPSPEC 4   // the same with the speculative search
#define NBP_TU_PRODLAT 8    // product kernels, latency geometries (y32, x16, l8)
#define NBP_TU_PRODTHR 16   // product kernels, throughput geometries, generic (m4, t2)
#define NBP_TU_PRODUNI 32   // product kernels, throughput geometries, one manifold per instance
#define NBP_TU_FUSED 64     // the fused variable-update kernels
#define NBP_TU_PRODUNI4 128 // product kernels, one manifold per instance, four helper lanes (the two-lane ones: NBP_TU_PRODUNI)
#define NBP_TU_PREPW5 256   // bandwidth fits + KD builds at five waves per SIMD (rows of 4k + 1 waves: N = 257 .. 320)
#define NBP_TU_PROPWAVE 512 // proposal kernels, one wave per proposal (chip-filling launches of simple Euclidean batches)
#ifndef NBP_TU
#define NBP_TU 0xFFFF
#endif

// ================================================================================================
// Proposal kernel: one workgroup = one approxConvBelief (ApproxConv.jl:4-45)
//   evalFactor -> evalPotentialSpecific (EvalFactor.jl:321-395 relative, :400-542 prior)
//   -> manikde! bandwidth.
// LDS: X[3][N] target scratch (the deepcopy of CalcFactor.jl:543-548), mhidx[N], LCV partial sums.
// HBM traffic: reads the operand beliefs once (coalesced, 8 B/lane), writes the proposal once.
// ================================================================================================
__device__ __forceinline__ void sample_measurement(const nbp_proposal_desc *d, int n, int zdim, double *z, const double *arena,
                                                   int64_t S, int N) {
  // (one exit, the three coordinates as scalars until then: with a return per measurement family the array the caller
  //  hands in ends up in scratch)
  const uint64_t mseed = d->meas_seed ? d->meas_seed : d->seed;  // stored measurement of an earlier op, or fresh
  double z0, z1 = 0.0, z2 = 0.0;
  if (d->meas_kde > 0) {
    // the measurement is a KDE (differential message factor): sample(belief) = random kernel + bw*randn
    // (manifolds/services/ManifoldSampling.jl:13-19), like the MsgPrior draw below
    const double *msg = arena + S * (d->meas_kde - 1);
    const int cm = slot_count(msg, N);
    double ua, ub, n0, n1, n2 = 0, n3 = 0;
    uniform_pair(mseed, n, PURP_KDESEL, 0, ua, ub);
    int i = (int)(ua * cm);
    if (i >= cm) i = cm - 1;
    normal_pair(mseed, n, PURP_KDENOISE, 0, n0, n1);
    if (zdim > 2) normal_pair(mseed, n, PURP_KDENOISE, 1, n2, n3);
    z0 = msg[i] + msg[3 * N] * n0;
    z1 = (zdim > 1) ? msg[N + i] + msg[3 * N + 1] * n1 : 0.0;
    z2 = (zdim > 2) ? msg[2 * N + i] + msg[3 * N + 2] * n2 : 0.0;
  } else {
    int c = 0;
    if (d->ncomp > 1) {  // Mixture.sampleFactor, Factors/Mixture.jl:114-155
      double ua, ub, cum = 0;
      uniform_pair(mseed, n, PURP_MIXLBL, 0, ua, ub);
      int last = 0;
      c = -1;
      for (int i = 0; i < d->ncomp; i++) {
        const double w = d->comp[i][0];
        if (w > 0) last = i;
        cum += w;
        if (c < 0 && ua < cum) c = i;
      }
      if (c < 0) c = last;
    }
    const double *cp = d->comp[c];
    if (zdim == 1 && cp[12] != 0.0) {  // scalar Uniform / Rayleigh component (enum nbp_dist)
      double ua, ub;
      uniform_pair(mseed, n, PURP_MEAS, 0, ua, ub);
      if (cp[12] == (double)NBP_DIST_TABLE) {  // rand(::AliasingScalarSampler): a domain value by its weight (inverse CDF)
        const double *tb = arena + S * d->var_slot[NBP_MAXV - 1];
        const int K = slot_count(tb, N);
        int i = 0;
        while (i < K - 1 && !(ua < tb[N + i])) i++;
        z0 = tb[i];
      } else
        z0 = cp[12] == (double)NBP_DIST_UNIFORM ? fma(cp[4], ua, cp[1]) : cp[4] * sqrt(-2.0 * nbpm_log(ua));
    } else {
      double n0 = 0, n1 = 0, n2 = 0, n3 = 0;
      normal_pair(mseed, n, PURP_MEAS, 0, n0, n1);
      if (zdim > 2) normal_pair(mseed, n, PURP_MEAS, 1, n2, n3);
      z0 = cp[1] + cp[4] * n0;
      z1 = (zdim > 1) ? cp[2] + cp[7] * n0 + cp[8] * n1 : 0.0;
      z2 = (zdim > 2) ? cp[3] + cp[10] * n0 + cp[11] * n1 + cp[12] * n2 : 0.0;
    }
  }
  z[0] = z0;
  z[1] = z1;
  z[2] = z2;
}

// addEntropyOnManifold!, EvalFactor.jl:95-132
// `mask`: the coordinates that receive entropy (the `p` argument, :99,114); 0 = all
__device__ __forceinline__ void add_entropy(int manifold, int D, double *x, int n, double spread, uint64_t seed, int kbase,
                                            int mask = 0) {
  double u0, u1, u2 = 0, u3 = 0;
  uniform_pair(seed, n, PURP_ENTROPY, kbase, u0, u1);
  if (D > 2) uniform_pair(seed, n, PURP_ENTROPY, kbase + 1, u2, u3);
  if (mask == 0) mask = 7;
  // (selects, not conditional stores: the compiler merges those into stores through a selected address, and the caller's
  //  point then lives in scratch)
  // (one rounding per operation, as the oracle and Julia evaluate it -- and the same in every kernel this is inlined into:
  //  left to the compiler, a multiply-add is contracted in one instance of the proposal kernels and not in another)
  double v0, v1, v2;
  {
#pragma clang fp contract(off)
    const double p0 = spread * (u0 - 0.5), p1 = spread * (u1 - 0.5), p2 = spread * (u2 - 0.5);
    v0 = x[0] + p0;
    v1 = x[1] + p1;
    v2 = x[2] + p2;
  }
  x[0] = (mask & 1) ? (is_circ(manifold, 0) ? wrap_pi(v0) : v0) : x[0];
  x[1] = (D > 1 && (mask & 2)) ? v1 : x[1];
  x[2] = (D > 2 && (mask & 4)) ? (is_circ(manifold, 2) ? wrap_pi(v2) : v2) : x[2];
}

// calcVariableDistanceExpectedFractional, EvalFactor.jl:40-92 (block-uniform result)
// `M`: the manifold (a compile-time constant in the single-class kernels: the circular-mean paths then fold away)
__device__ __forceinline__ double var_distance_expected_fractional(const nbp_proposal_desc *d, const recipe_t *R, const double *arena,
                                                   int64_t S, int N, const double *X, double kappa, double *red, const int M) {
  const int sf1 = d->sfidx + 1, D = mani_dim(M);
  if (in_list(R->certain, R->ncertain, sf1)) return kappa * std_basic_spread(X, N, N, M, red);
  double ref[3] = {0, 0, 0};
  for (int k = 0; k < D; k++) ref[k] = mean_default_coord(X + k * N, N, M, k, red);
  double best = 1e-2;
  for (int i = 1; i <= d->nvars; i++) {
    const double *pts = (i == sf1) ? X : arena + S * d->var_slot[i - 1];
    const int ci = (i == sf1) ? N : slot_count(pts, N);  // the scratch copy of the target always has N entries
    const bool cer = in_list(R->certain, R->ncertain, i);
    double acc = 0;
    for (int k = 0; k < D; k++) {
      double mu = cer ? mean_geodesic_coord(pts + k * N, ci, M, k, red) : mean_default_coord(pts + k * N, ci, M, k, red);
      acc += (ref[k] - mu) * (ref[k] - mu);
    }
    best = fmax(best, sqrt(acc));
  }
  return kappa * best;
}

// FIXK / FIXM: a batch whose relative factors are all of kind FIXK on the manifold FIXM, none of them partial (priors,
// message priors and pass-through densities of that manifold may ride along): the solver dispatch and the partial
// branches fold away, and with them the registers of the largest solver (the generic kernel holds the SE(2) simplex)
template <int FIXK, int FIXM>
// `d`: the op; `out`: where the proposal goes -- the slot d->out_slot of the arena, or (fused update kernel) a slot-shaped
// area of the workgroup's LDS
__device__ __forceinline__ void proposal_body(const nbp_proposal_desc *d, double *out, double *arena, int N, int Npad, int64_t S, int32_t *side,
                                              nbp_counters *ctr, double *smem) {
  double *X = smem;                 // [3][N]
  double *red = X + 3 * N;          // [NBP_RED]
  int *mh = (int *)(red + NBP_RED); // [N]
  __shared__ recipe_t R;
  const int n = threadIdx.x, M = FIXK ? FIXM : d->manifold, D = mani_dim(M), kind = d->factor_kind;
  const bool live = n < N;  // lanes (i < N, p == 0) own a particle
  unsigned int n_solves = 0, n_nonconv = 0, n_nan = 0, n_evals = 0;

  NBP_CTICK_INIT();
  NBP_BLOCK_BEGIN();
  if (n == 0) build_recipe(d, &R);
  {
    const double *src = arena + S * d->var_slot[(kind == NBP_F_PRIOR || kind == NBP_F_MSGPRIOR || kind == NBP_F_PASSTHROUGH) ? 0 : d->sfidx];
    // resize!(target copy, N): entries beyond the belief's own count are the point default (CalcFactor.jl:555-565)
    const int ct = slot_count(src, N);
    // (rows beyond the manifold's dimension hold zeros in every slot: not read)
    if (live)
      for (int k = 0; k < 3; k++) X[k * N + n] = (k < D && n < ct) ? src[k * N + n] : 0.0;
  }
  __syncthreads();
  // mhidx: injected or rand(Categorical)  (ExplicitDiscreteMarginalizations.jl:186,261)
  if (live) {
    int h;
    if (d->mhidx_in >= 0) h = side[d->mhidx_in + n];
    else if (!d->has_multihypo && d->nullhypo == 0.0) h = 1;
    else {
      double ua, ub;
      uniform_pair(d->seed, n, PURP_HYPO, 0, ua, ub);
      h = R.cat_first + categorical(R.cat_p, R.ncat, ua);
    }
    mh[n] = h;
    if (d->mhidx_out >= 0) side[d->mhidx_out + n] = h;
  }
  __syncthreads();

  if (kind == NBP_F_PASSTHROUGH) {
    // calcProposalBelief(::PartialPriorPassThrough) (ApproxConv.jl:196-227): the density itself, on the partial coordinates
    // (antimarginal); the other coordinates of the scratch copy stay the target's
    const double *den = arena + S * d->var_slot[1];
    const int cd = slot_count(den, N), pm = d->partial_mask ? d->partial_mask : 7;
    int idx = n;
    if (!d->keep_count && cd < N) {  // multinomial resampling to N (a product needs N points from every input)
      double ua, ub;
      uniform_pair(d->seed, n, PURP_KDESEL, 0, ua, ub);
      idx = (int)(ua * cd);
      if (idx >= cd) idx = cd - 1;
    }
    double nz0 = 0, nz1 = 0, nz2 = 0, nz3 = 0;  // (scalars: an array indexed by the loop below lives in scratch)
    if (d->keep_count == 2 && cd < N && n >= cd) {  // resample(bel, N) of graph initialisation (GraphInit.jl:174-177):
      double ua, ub;                                 // the density's points stay, the rest are draws from its KDE
      uniform_pair(d->seed, n, PURP_OLDSEL, 0, ua, ub);
      idx = (int)(ua * cd);
      if (idx >= cd) idx = cd - 1;
      normal_pair(d->seed, n, PURP_OLDNOISE, 0, nz0, nz1);
      if (D > 2) normal_pair(d->seed, n, PURP_OLDNOISE, 1, nz2, nz3);
    }
    if (live && idx < cd)
      for (int k = 0; k < D; k++)
        if ((pm >> k) & 1) {
          const double nzk = k == 0 ? nz0 : (k == 1 ? nz1 : nz2);
          const double v = den[k * N + idx] + den[3 * N + k] * nzk;
          X[k * N + n] = (nzk != 0.0 && is_circ(M, k)) ? wrap_pi(v) : v;
        }
    if (live)
      for (int k = 0; k < 3; k++) out[k * N + n] = (k < D) ? X[k * N + n] : 0.0;
    if (n < 3) {
      const bool in = n < D && ((pm >> n) & 1);
      out[3 * N + n] = in ? den[3 * N + n] : 0.0;   // the density's own bandwidth: nothing is fitted
      out[3 * N + 3 + n] = in ? 1.0 : 0.0;          // infoPerCoord
    }
    if (n == 0) out[3 * N + 6] = (d->keep_count == 1 && cd < N) ? (double)cd : 0.0;
    return;
  }
  if (kind == NBP_F_PRIOR || kind == NBP_F_MSGPRIOR) {
    // evalPotentialSpecific(prior), EvalFactor.jl:400-542
    const double spread = d->spread_nh * std_basic_spread(X, N, N, M, red);  // :464, before the overwrite
    __syncthreads();
    if (live) {
      double x[3] = {X[n], X[N + n], X[2 * N + n]};
      if (mh[n] == 1) {
        if (!FIXK && kind == NBP_F_PRIOR && d->partial_mask) {  // (a uniform batch holds no partial priors: proposals_uniform_class)
          // partial prior: setPointPartial! on the partial coordinates only (:457-538)
          const int pmk = d->partial_mask;
          double z[3];
          sample_measurement(d, n, __popc(pmk & 7), z, arena, S, N);
          int pk = 0;
          if (pmk & 1) { x[0] = is_circ(M, 0) ? wrap_pi(z[0]) : z[0]; pk = 1; }
          if (pmk & 2) { x[1] = pk ? z[1] : z[0]; pk++; }  // (z[pk] would put the array in scratch)
          if (pmk & 4) { const double v = (pk == 0) ? z[0] : (pk == 1 ? z[1] : z[2]); x[2] = is_circ(M, 2) ? wrap_pi(v) : v; }
        } else if (kind == NBP_F_PRIOR) {
          double z[3];
          sample_measurement(d, n, D, z, arena, S, N);
          x[0] = is_circ(M, 0) ? wrap_pi(z[0]) : z[0];
          x[1] = z[1];
          x[2] = is_circ(M, 2) ? wrap_pi(z[2]) : z[2];
        } else {  // MsgPrior{MKD}: sample(belief): random kernel + bw*randn (Factors/MsgPrior.jl:27-30)
          const double *msg = arena + S * d->var_slot[1];
          const int cm = slot_count(msg, N);
          const uint64_t mseed = d->meas_seed ? d->meas_seed : d->seed;
          double ua, ub, n0, n1, n2 = 0, n3 = 0;
          uniform_pair(mseed, n, PURP_KDESEL, 0, ua, ub);
          int i = (int)(ua * cm);
          if (i >= cm) i = cm - 1;
          normal_pair(mseed, n, PURP_KDENOISE, 0, n0, n1);
          if (D > 2) normal_pair(mseed, n, PURP_KDENOISE, 1, n2, n3);
          double v0 = msg[i] + msg[3 * N] * n0;
          x[0] = is_circ(M, 0) ? wrap_pi(v0) : v0;
          if (D > 1) x[1] = msg[N + i] + msg[3 * N + 1] * n1;
          if (D > 2) {
            double v2 = msg[2 * N + i] + msg[3 * N + 2] * n2;
            x[2] = is_circ(M, 2) ? wrap_pi(v2) : v2;
          }
        }
      } else {
        add_entropy(M, D, x, n, spread, d->seed, 0, d->partial_mask);  // :476, partialCoords :532
      }
      for (int k = 0; k < 3; k++) X[k * N + n] = x[k];
    }
    __syncthreads();
  } else {
    // evalPotentialSpecific(relative), EvalFactor.jl:321-395
    // a partial relative factor (one partial coordinate, validated on the host) measures, inflates
    // and solves that coordinate only (EvalFactor.jl:184-198, NumericalCalculations.jl:424)
    // a partial relative factor measures, inflates and solves its `.partial` coordinates only (validated on the host:
    // LinearRelative, one or two of the variable's coordinates)
    const int pmask = FIXK ? 0 : d->partial_mask, npd = __popc(pmask & 7);
    const int rkind = FIXK ? FIXK : kind;
    // a partial ManifoldFactor on SE(2) keeps its full measurement and searches over the whole point (its residual counts
    // the partial components only); a partial LinearRelative measures and searches its partial coordinates
    const bool pse2 = !FIXK && pmask && rkind == NBP_F_SE2;
    const int pdim = (pmask && !pse2) ? (pmask & 1 ? 0 : (pmask & 2 ? 1 : 2)) : -1;       // first partial coordinate
    const int pdim2 = (npd > 1 && !pse2) ? ((pmask & 1) && (pmask & 2) ? 1 : 2) : -1;     // second one
    const int zdim = (pmask && !pse2) ? npd : ((rkind == NBP_F_LINREL) ? D : (rkind == NBP_F_SE2 ? 3 : 1));
    double z[3] = {0, 0, 0};
    if (live) sample_measurement(d, n, zdim, z, arena, S, N);  // sampleFactor!, CalcFactor.jl:578
    const int sf1 = d->sfidx + 1;
    const int myh = live ? mh[n] : -1000;
    // computeAcrossHypothesis!, EvalFactor.jl:145-237
    // the recipe lives in LDS: what is read from it is wave-uniform but arrives in vector registers -- moved to scalar
    // ones, so that the group / cycle loops branch on SGPRs and keep no VGPRs live across the per-particle searches
    const int ngroups = __builtin_amdgcn_readfirstlane(R.ngroups);
    for (int g = 0; g < ngroups; g++) {
      if (__builtin_amdgcn_readfirstlane(R.empty[g])) continue;
      const int hyp = __builtin_amdgcn_readfirstlane(R.hypo[g]);
      if (!__syncthreads_or(myh == hyp)) continue;  // empty allelements[g]: nothing to do
      const bool solve_case = __builtin_amdgcn_readfirstlane(
          (int)((in_list(R.certain, R.ncertain, sf1) && hyp != 0) || in_list(R.certain, R.ncertain, hyp) || hyp == sf1)) != 0;
      if (solve_case) {
        const int va = __builtin_amdgcn_readfirstlane(R.act[g][0]), vb = __builtin_amdgcn_readfirstlane(R.act[g][1]);
        const int solve_b = (vb == sf1);
        const int vother = solve_b ? va : vb;
        const double *O = arena + S * d->var_slot[vother - 1];
        double oth[3] = {0, 0, 0};
        if (myh == hyp) {
          const int io = anyn_index(n, slot_count(O, N), d->seed, vother);  // _getindex_anyn
          oth[0] = O[io];
          if (D > 1) oth[1] = O[N + io];
          if (D > 2) oth[2] = O[2 * N + io];
        }
        for (int c = 0; c < d->inflate_cycles; c++) {  // :184-207
          NBP_CTICK(30);  // everything before / between cycles
          const double spread = var_distance_expected_fractional(d, &R, arena, S, N, X, d->inflation, red, M);
          __syncthreads();
          NBP_CTICK(31);  // spread statistics (workgroup reductions)
          if (myh == hyp) {
            double x[3] = {X[n], X[N + n], X[2 * N + n]};
            add_entropy(M, D, x, n, spread, d->seed, (g * 8 + c) * 2, pmask);
            if (pse2) {
              solve_particle_partial_se2(z, oth, solve_b, x, pmask, n_solves, n_nonconv, n_nan, n_evals);
            } else if (pdim2 >= 0) {  // two partial coordinates: BFGS on the pair (NumericalCalculations.jl:108,424)
              double x2[3] = {pdim == 0 ? x[0] : x[1], pdim2 == 1 ? x[1] : x[2], 0};
              const double o2[3] = {pdim == 0 ? oth[0] : oth[1], pdim2 == 1 ? oth[1] : oth[2], 0};
              solve_particle_partial2(z, o2, solve_b, x2, n_solves, n_nonconv, n_nan, n_evals);
              if (pdim == 0) x[0] = x2[0];
              else x[1] = x2[0];
              if (pdim2 == 1) x[1] = x2[1];
              else x[2] = x2[1];
            } else if (pdim >= 0) {
              double x1[3] = {pdim == 0 ? x[0] : (pdim == 1 ? x[1] : x[2]), 0, 0};
              const double o1[3] = {pdim == 0 ? oth[0] : (pdim == 1 ? oth[1] : oth[2]), 0, 0};
              solve_particle(NBP_F_LINREL, NBP_EUCLID1, z, o1, solve_b, x1, n_solves, n_nonconv, n_nan, n_evals);
              if (pdim == 0) x[0] = x1[0];
              else if (pdim == 1) x[1] = x1[0];
              else x[2] = x1[0];
            } else
              solve_particle(rkind, M, z, oth, solve_b, x, n_solves, n_nonconv, n_nan, n_evals);  // approxConvOnElements!
            X[n] = x[0];
            if (D > 1) X[N + n] = x[1];
            if (D > 2) X[2 * N + n] = x[2];
          }
          NBP_CTICK(32);  // entropy + per-particle solve (this lane)
          __syncthreads();
          NBP_CTICK(33);  // waiting for the slowest lane / wave
        }
      } else {  // other-hypothesis (:208-220) / nullhypo (:222-231): entropy only
        const double spread = var_distance_expected_fractional(d, &R, arena, S, N, X, d->spread_nh, red, M);
        __syncthreads();
        if (myh == hyp) {
          double x[3] = {X[n], X[N + n], X[2 * N + n]};
          add_entropy(M, D, x, n, spread, d->seed, (g * 8) * 2);
          X[n] = x[0];
          if (D > 1) X[N + n] = x[1];
          if (D > 2) X[2 * N + n] = x[2];
        }
        __syncthreads();
      }
    }
  }
  // manikde!(M, pts) (ApproxConv.jl:36-42): the bandwidth fit of this proposal runs in the prep launch
  // of its update (nbp_prep_kernel), or in nbp_bandwidth_kernel for the immediate-mode entry points;
  // this kernel keeps one lane per particle so that the Nelder-Mead simplex stays in registers.
  // (only the rows the manifold has: nothing reads a row beyond them -- fits, KD builds, products and belief reads go by
  //  the manifold; a third of a Euclid(2) proposal's write traffic)
  if (live)
    for (int k = 0; k < D; k++) out[k * N + n] = X[k * N + n];
  // infoPerCoord of the proposal: ones(D), zeroed outside the factor's `.partial` (EvalFactor.jl:383-391, :534-540)
  if (n < 3) out[3 * N + 3 + n] = (n < D && (!d->partial_mask || ((d->partial_mask >> n) & 1))) ? 1.0 : 0.0;
  if (n == 0) out[3 * N + 6] = 0.0;  // a proposal always holds N points
  NBP_BLOCK_END();
#ifdef NBP_PHASE_TIMING
  {  // lane utilisation of the per-particle searches: residual evaluations summed over the lanes vs 64 x the wave's slowest lane
    unsigned int mx = n_evals;
    for (int o = 32; o > 0; o >>= 1) mx = max(mx, (unsigned int)__shfl_xor((int)mx, o, 64));
    unsigned long long sm = n_evals;
    for (int o = 32; o > 0; o >>= 1) sm += __shfl_xor(sm, o, 64);
    if ((threadIdx.x & 63) == 0 && mx) {
      atomicAdd((unsigned long long *)&nbp_phase_clk[60], sm);
      atomicAdd((unsigned long long *)&nbp_phase_clk[61], 64ull * mx);
    }
  }
#endif
  // diagnostics: one atomic per wave
  {
    unsigned int v[4] = {n_solves, n_nonconv, n_nan, n_evals};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      unsigned int t = v[q];
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
      v[q] = t;
    }
    if ((threadIdx.x & 63) == 0 && (v[0] | v[3])) {
      atomicAdd(&ctr->solves, (unsigned long long)v[0]);
      atomicAdd(&ctr->nonconverged, (unsigned long long)v[1]);
      atomicAdd(&ctr->nan_results, (unsigned long long)v[2]);
      atomicAdd(&ctr->residual_evals, (unsigned long long)v[3]);
    }
  }
}

#define NBP_PROPOSAL_ARGS const nbp_proposal_desc *descs, double *arena, int N, int Npad, int64_t S, int32_t *side, nbp_counters *ctr
#if NBP_TU & NBP_TU_PROPOSAL
__global__ void __launch_bounds__(512)
nbp_proposal_kernel(NBP_PROPOSAL_ARGS) {
  extern __shared__ double smem[];
  proposal_body<0, 0>(descs + blockIdx.x, arena + S * descs[blockIdx.x].out_slot, arena, N, Npad, S, side, ctr, smem);
}
#else
__global__ void nbp_proposal_kernel(NBP_PROPOSAL_ARGS);
#endif
// one relative-factor kind on one manifold (the odometry chains of the BASELINE configs): lin2 = LinearRelative on
// Euclid(2) (configs 2 / 2p: -10 % proposal time), lin3 = LinearRelative on Euclid(3) (config 5: -7 %).  The circle
// (config 3) gains nothing from its own instance: the registers there are the spread statistics', not the solver's
#if NBP_TU & NBP_TU_PROPOSAL
#define NBP_PROPOSAL_UNIFORM(NAME, K_, M_, WAVES)                                                                          \
  __global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(WAVES)))                                       \
  NAME(NBP_PROPOSAL_ARGS) {                                                                                                \
    extern __shared__ double smem[];                                                                                       \
    proposal_body<K_, M_>(descs + blockIdx.x, arena + S * descs[blockIdx.x].out_slot, arena, N, Npad, S, side, ctr, smem);  \
  }
#else
#define NBP_PROPOSAL_UNIFORM(NAME, K_, M_, WAVES) __global__ void NAME(NBP_PROPOSAL_ARGS);
#endif
#ifndef NBP_W_LIN2
#define NBP_W_LIN2 3
#endif
NBP_PROPOSAL_UNIFORM(nbp_proposal_kernel_lin2, NBP_F_LINREL, NBP_EUCLID2, NBP_W_LIN2)
// Euclid(3) instance at FIVE waves per SIMD (96 VGPRs, 44 B of scratch per lane; 120 VGPRs = four waves per SIMD without the cap):
// BASELINE's config 5 runs N = 300, i.e. workgroups of five waves, of which a CU holds three at four waves per SIMD (15 waves, two
// waves of every workgroup on one SIMD) and four at five.  975 proposals at N = 300: 1018 -> 747 us, 4000: 3630 -> 3010 us; a lone
// proposal and N = 200 unchanged (168 / 499 us); config 5 415 -> 404 ms per solve (profiles/r04_lcv_five_wave_rows.txt, section 5)
#ifndef NBP_W_LIN3
#define NBP_W_LIN3 5
#endif
NBP_PROPOSAL_UNIFORM(nbp_proposal_kernel_lin3, NBP_F_LINREL, NBP_EUCLID3, NBP_W_LIN3)
// CircularCircular on the circle (config 3, incl. its multihypo sightings) and ManifoldFactor on SE(2) (config 4)
NBP_PROPOSAL_UNIFORM(nbp_proposal_kernel_circ, NBP_F_CIRCULAR, NBP_CIRCULAR, 3)
#ifndef NBP_W_SE2
#define NBP_W_SE2 2  // (3: 168 VGPRs and 80 B of scratch per lane; measured, profiles/r04_lcv_five_wave_rows.txt section 5)
#endif
NBP_PROPOSAL_UNIFORM(nbp_proposal_kernel_se2, NBP_F_SE2, NBP_SE2, NBP_W_SE2)

// ================================================================================================
// One WAVE per proposal: the geometry of chip-filling launches of "simple" Euclidean batches (thousands of LinearRelative
// proposals of one tree level of a long chain, no multihypo, no nullhypo, full factors; priors, message priors and
// pass-through densities ride along).  A workgroup of the kernels above takes, per inflation cycle, as long as the slowest
// search of its slowest wave, and its other waves wait at the barrier behind the spread statistics (SQ_WAIT_ANY = 0.5 of the
// wave-cycles, an instruction issuing in 0.6-0.7 of the SIMD-time: profiles/r04_proposal_wave_occupancy.txt).  Here lane l of
// the one wave owns the particles l, l + 64, l + 128, ... and runs their searches one after the other; the spread statistics
// are wave reductions over the chunks of 64 in the order block_sum() adds the wave partials (same sums, bit for bit); there is
// no barrier, so a wave is always ready to issue, and with >= ~4000 proposals in a launch every SIMD holds four or five.
// LDS: the scratch copy X[D][N] of the target belief only (each lane reads and writes the entries of its own particles:
// storage that can be indexed by the chunk loop, no exchange between lanes).  The measurements of a lane's NC particles stay
// in registers and are rotated by one place per chunk, so that the chunk loop is not unrolled (one copy of the search).
// Particle for particle the same operations as proposal_body<NBP_F_LINREL, FIXM>.
// ================================================================================================
template <int NC>
__device__ __forceinline__ double wave_chunks_sum(const double (&v)[NC]) {
  double t = wave_sum(v[0]);
#pragma unroll
  for (int p = 1; p < NC; p++) t += wave_sum(v[p]);
  return t;
}

template <int FIXM, int NC>
__device__ __forceinline__ void proposal_wave_body(const nbp_proposal_desc *d, double *out, double *arena, int N, int64_t S, int32_t *side,
                                                   nbp_counters *ctr, double *X) {
  constexpr int M = FIXM, D = FIXM;  // Euclid(D)
  const int lane = threadIdx.x & 63, kind = d->factor_kind;
  unsigned int n_solves = 0, n_nonconv = 0, n_nan = 0, n_evals = 0;
  {
    const double *src = arena + S * d->var_slot[(kind == NBP_F_PRIOR || kind == NBP_F_MSGPRIOR || kind == NBP_F_PASSTHROUGH) ? 0 : d->sfidx];
    const int ct = slot_count(src, N);
#pragma unroll
    for (int p = 0; p < NC; p++) {
      const int n = lane + 64 * p;
      if (n < N)
        for (int k = 0; k < D; k++) X[k * N + n] = (n < ct) ? src[k * N + n] : 0.0;
    }
  }
  if (d->mhidx_out >= 0) {
#pragma unroll
    for (int p = 0; p < NC; p++)
      if (lane + 64 * p < N) side[d->mhidx_out + lane + 64 * p] = 1;  // every particle on the factor's one hypothesis
  }
  if (kind == NBP_F_PASSTHROUGH) {  // (proposal_body, same branch)
    const double *den = arena + S * d->var_slot[1];
    const int cd = slot_count(den, N), pm = d->partial_mask ? d->partial_mask : 7;
#pragma unroll 1
    for (int p = 0; p < NC; p++) {
      const int n = lane + 64 * p;
      if (n >= N) continue;
      int idx = n;
      if (!d->keep_count && cd < N) {
        double ua, ub;
        uniform_pair(d->seed, n, PURP_KDESEL, 0, ua, ub);
        idx = (int)(ua * cd);
        if (idx >= cd) idx = cd - 1;
      }
      double nz0 = 0, nz1 = 0, nz2 = 0, nz3 = 0;
      if (d->keep_count == 2 && cd < N && n >= cd) {
        double ua, ub;
        uniform_pair(d->seed, n, PURP_OLDSEL, 0, ua, ub);
        idx = (int)(ua * cd);
        if (idx >= cd) idx = cd - 1;
        normal_pair(d->seed, n, PURP_OLDNOISE, 0, nz0, nz1);
        if (D > 2) normal_pair(d->seed, n, PURP_OLDNOISE, 1, nz2, nz3);
      }
      if (idx < cd)
        for (int k = 0; k < D; k++)
          if ((pm >> k) & 1) {
            const double nzk = k == 0 ? nz0 : (k == 1 ? nz1 : nz2);
            X[k * N + n] = den[k * N + idx] + den[3 * N + k] * nzk;
          }
      for (int k = 0; k < 3; k++) out[k * N + n] = (k < D) ? X[k * N + n] : 0.0;
    }
    if (lane < 3) {
      const bool in = lane < D && ((pm >> lane) & 1);
      out[3 * N + lane] = in ? den[3 * N + lane] : 0.0;
      out[3 * N + 3 + lane] = in ? 1.0 : 0.0;
    }
    if (lane == 0) out[3 * N + 6] = (d->keep_count == 1 && cd < N) ? (double)cd : 0.0;
    return;
  }
  if (kind == NBP_F_PRIOR || kind == NBP_F_MSGPRIOR) {
    // evalPotentialSpecific(prior) with every particle on the hypothesis (no nullhypo in this class: the spread is not needed)
#pragma unroll 1
    for (int p = 0; p < NC; p++) {
      const int n = lane + 64 * p;
      if (n >= N) continue;
      double x[3] = {0, 0, 0};
      if (kind == NBP_F_PRIOR) {
        sample_measurement(d, n, D, x, arena, S, N);
      } else {
        const double *msg = arena + S * d->var_slot[1];
        const int cm = slot_count(msg, N);
        const uint64_t mseed = d->meas_seed ? d->meas_seed : d->seed;
        double ua, ub, n0, n1, n2 = 0, n3 = 0;
        uniform_pair(mseed, n, PURP_KDESEL, 0, ua, ub);
        int i = (int)(ua * cm);
        if (i >= cm) i = cm - 1;
        normal_pair(mseed, n, PURP_KDENOISE, 0, n0, n1);
        if (D > 2) normal_pair(mseed, n, PURP_KDENOISE, 1, n2, n3);
        x[0] = msg[i] + msg[3 * N] * n0;
        if (D > 1) x[1] = msg[N + i] + msg[3 * N + 1] * n1;
        if (D > 2) x[2] = msg[2 * N + i] + msg[3 * N + 2] * n2;
      }
      for (int k = 0; k < D; k++) X[k * N + n] = x[k];
    }
  } else {
    // evalPotentialSpecific(relative): LinearRelative between two variables, one hypothesis group holding every particle
    // (build_recipe without multihypo: act = {1, 2}, both certain)
    const int sf1 = d->sfidx + 1, solve_b = (sf1 == 2), vother = solve_b ? 1 : 2;
    const double *O = arena + S * d->var_slot[vother - 1];
    const int co = slot_count(O, N);
    double z[NC][D];
#pragma unroll
    for (int p = 0; p < NC; p++) {
      double zz[3] = {0, 0, 0};
      if (lane + 64 * p < N) sample_measurement(d, lane + 64 * p, D, zz, arena, S, N);  // sampleFactor!, once per approxConv
#pragma unroll
      for (int k = 0; k < D; k++) z[p][k] = zz[k];
    }
    const double rN = (double)N;
    for (int c = 0; c < d->inflate_cycles; c++) {
      // calcStdBasicSpread of the scratch copy (std_basic_spread / block_sum, chunk by chunk)
      double acc[NC];
#pragma unroll
      for (int p = 0; p < NC; p++) acc[p] = 0.0;
#pragma unroll
      for (int k = 0; k < D; k++) {
        double v[NC];
#pragma unroll
        for (int p = 0; p < NC; p++) v[p] = (lane + 64 * p < N) ? X[k * N + lane + 64 * p] : 0.0;
        const double mu = wave_chunks_sum<NC>(v) / rN;
#pragma unroll
        for (int p = 0; p < NC; p++)
          if (lane + 64 * p < N) {
            const double dl = v[p] - mu;
            acc[p] += dl * dl;
          }
      }
      const double sg = sqrt(wave_chunks_sum<NC>(acc) / (double)(N - 1));
      const double spread = d->inflation * ((1e-10 < sg) ? sg : 1.0);
#pragma unroll 1
      for (int p = 0; p < NC; p++) {
        const int n = lane + 64 * p;
        if (n < N) {
          const int io = anyn_index(n, co, d->seed, vother);  // _getindex_anyn
          double oth[3] = {O[io], 0, 0};
          if (D > 1) oth[1] = O[N + io];
          if (D > 2) oth[2] = O[2 * N + io];
          double x[3] = {X[n], 0, 0};
          if (D > 1) x[1] = X[N + n];
          if (D > 2) x[2] = X[2 * N + n];
          double zz[3] = {z[0][0], 0, 0};
          if (D > 1) zz[1] = z[0][1];
          if (D > 2) zz[2] = z[0][D - 1];
          add_entropy(M, D, x, n, spread, d->seed, (1 * 8 + c) * 2, 0);
          solve_particle_t<NBP_F_LINREL, D>(M, zz, oth, solve_b, x, n_solves, n_nonconv, n_nan, n_evals);
          X[n] = x[0];
          if (D > 1) X[N + n] = x[1];
          if (D > 2) X[2 * N + n] = x[2];
        }
        // the next chunk's measurement moves to the front (NC rotations bring every one back to its place)
        double z0[D];
#pragma unroll
        for (int k = 0; k < D; k++) z0[k] = z[0][k];
#pragma unroll
        for (int q = 0; q + 1 < NC; q++)
#pragma unroll
          for (int k = 0; k < D; k++) z[q][k] = z[q + 1][k];
#pragma unroll
        for (int k = 0; k < D; k++) z[NC - 1][k] = z0[k];
      }
    }
  }
#pragma unroll
  for (int p = 0; p < NC; p++) {
    const int n = lane + 64 * p;
    if (n < N)
      for (int k = 0; k < D; k++) out[k * N + n] = X[k * N + n];
  }
  if (lane < 3) out[3 * N + 3 + lane] = (lane < D) ? 1.0 : 0.0;  // infoPerCoord (no partial factors in this class)
  if (lane == 0) out[3 * N + 6] = 0.0;
  {
    unsigned int v[4] = {n_solves, n_nonconv, n_nan, n_evals};
#pragma unroll
    for (int q = 0; q < 4; q++) {
      unsigned int t = v[q];
      for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
      v[q] = t;
    }
    if (lane == 0 && (v[0] | v[3])) {
      atomicAdd(&ctr->solves, (unsigned long long)v[0]);
      atomicAdd(&ctr->nonconverged, (unsigned long long)v[1]);
      atomicAdd(&ctr->nan_results, (unsigned long long)v[2]);
      atomicAdd(&ctr->residual_evals, (unsigned long long)v[3]);
    }
  }
}

#define NBP_PROPOSAL_WAVE_ARGS const nbp_proposal_desc *descs, int n, double *arena, int N, int64_t S, int32_t *side, nbp_counters *ctr
#if NBP_TU & NBP_TU_PROPWAVE
// workgroups of NBP_PW_WAVES independent waves (no barrier between them), one proposal each
#define NBP_PROPOSAL_WAVE(NAME, M_, NC_, WAVES)                                                                            \
  __global__ void __launch_bounds__(64 * NBP_PW_WAVES) __attribute__((amdgpu_waves_per_eu(WAVES)))                         \
  NAME(NBP_PROPOSAL_WAVE_ARGS) {                                                                                           \
    extern __shared__ double smem[];                                                                                       \
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), i = blockIdx.x * NBP_PW_WAVES + w;                                                     \
    if (i >= n) return;                                                                                                    \
    proposal_wave_body<M_, NC_>(descs + i, arena + S * descs[i].out_slot, arena, N, S, side, ctr, smem + (size_t)w * M_ * N); \
  }
#else
#define NBP_PROPOSAL_WAVE(NAME, M_, NC_, WAVES) __global__ void NAME(NBP_PROPOSAL_WAVE_ARGS);
#endif
#define NBP_PW_WAVES 4
#ifndef NBP_WW_LIN2
#define NBP_WW_LIN2 5  // 94 VGPRs
#endif
#ifndef NBP_WW_LIN3
#define NBP_WW_LIN3 3  // 133 / 140 VGPRs
#endif
NBP_PROPOSAL_WAVE(nbp_proposal_wave_kernel_lin2, NBP_EUCLID2, 4, NBP_WW_LIN2)
NBP_PROPOSAL_WAVE(nbp_proposal_wave_kernel_lin3, NBP_EUCLID3, 4, NBP_WW_LIN3)
NBP_PROPOSAL_WAVE(nbp_proposal_wave_kernel_lin3n5, NBP_EUCLID3, 5, NBP_WW_LIN3)
static inline size_t nbp_proposal_wave_lds_bytes(int N, int D) { return (size_t)NBP_PW_WAVES * D * N * 8; }

// ================================================================================================
// Deconvolution kernel: one workgroup = one approxDeconv(dfg, fct) (DeconvUtils.jl:32-160), one lane
// per particle: sample a measurement (the search start, returned as "measured"), then find the
// measurement that zeroes the residual between the two stored variable points ("predicted").
// ================================================================================================
#define NBP_DECONV_ARGS const nbp_proposal_desc *descs, const int32_t *meas_slots, double *arena, int N, int64_t S, nbp_counters *ctr
#if NBP_TU & NBP_TU_PROPOSAL
__global__ void __launch_bounds__(512)
nbp_deconv_kernel(const nbp_proposal_desc *descs, const int32_t *meas_slots, double *arena, int N, int64_t S, nbp_counters *ctr) {
  const nbp_proposal_desc *d = descs + blockIdx.x;
  const int n = threadIdx.x, M = d->manifold, D = mani_dim(M), kind = d->factor_kind;
  const int zdim = (kind == NBP_F_LINREL) ? D : (kind == NBP_F_SE2 ? 3 : 1);
  const double *A = arena + S * d->var_slot[0], *B = arena + S * d->var_slot[1];
  double *out = arena + S * d->out_slot;
  double *ms = (meas_slots && meas_slots[blockIdx.x] >= 0) ? arena + S * meas_slots[blockIdx.x] : nullptr;
  unsigned int n_solves = 0, n_nonconv = 0, n_nan = 0, n_evals = 0;
  if (n < N) {
    double z[3], a[3] = {0, 0, 0}, b[3] = {0, 0, 0};
    sample_measurement(d, n, zdim, z, arena, S, N);
    if (ms)
      for (int k = 0; k < 3; k++) ms[k * N + n] = (k < zdim) ? z[k] : 0.0;
    const int ia = anyn_index(n, slot_count(A, N), d->seed, 1), ib = anyn_index(n, slot_count(B, N), d->seed, 2);
    a[0] = A[ia]; b[0] = B[ib];
    if (D > 1) { a[1] = A[N + ia]; b[1] = B[N + ib]; }
    if (D > 2) { a[2] = A[2 * N + ia]; b[2] = B[2 * N + ib]; }
    deconv_particle(kind, M, a, b, z, n_solves, n_nonconv, n_nan, n_evals);
    for (int k = 0; k < 3; k++) out[k * N + n] = (k < zdim) ? z[k] : 0.0;
  }
  if (n < 3) {
    out[3 * N + n] = 0.0;
    if (ms) ms[3 * N + n] = 0.0;
  }
  if (n == 0) {
    out[3 * N + 6] = 0.0;
    if (ms) ms[3 * N + 6] = 0.0;
  }
  unsigned int v[4] = {n_solves, n_nonconv, n_nan, n_evals};
#pragma unroll
  for (int q = 0; q < 4; q++) {
    unsigned int t = v[q];
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    v[q] = t;
  }
  if ((threadIdx.x & 63) == 0 && (v[0] | v[3])) {
    atomicAdd(&ctr->solves, (unsigned long long)v[0]);
    atomicAdd(&ctr->nonconverged, (unsigned long long)v[1]);
    atomicAdd(&ctr->nan_results, (unsigned long long)v[2]);
    atomicAdd(&ctr->residual_evals, (unsigned long long)v[3]);
  }
}
#else
__global__ void nbp_deconv_kernel(NBP_DECONV_ARGS);
#endif

static inline size_t nbp_proposal_lds_bytes(int N) { return ((size_t)3 * N + NBP_RED) * 8 + (size_t)N * 4; }

// fit the bandwidth of coordinate k of a resident slot (block-uniform early exit for k >= D)
template <int SPEC>  // 0: sequential search; 2 / 3: speculative search, that many iterations per rendezvous
__device__ __forceinline__ void lcv_slot_coordinate(double *s, int M, int k, int Ncap, int Npad, double *smem, nbp_counters *ctr,
                                                    nbp_spec_area *area = nullptr, int role = 0) {
  const int D = mani_dim(M), n = threadIdx.x;
  if (k >= D) {
    if (n == 0) s[3 * Ncap + k] = 0.0;
    return;
  }
  const int N = slot_count(s, Ncap);  // manikde! of the points the belief holds (Ncap = slot capacity = stride)
  const int P = blockDim.x / Npad;
  // the table of lcv_exp first: its address is then a constant of the kernel (one shift for the address of an entry)
  double *tab = smem, *X = smem + NBP_FITTAB, *part = X + 2 * N, *red = part + P * Npad + (blockDim.x >> 6) * 2 * N;
  nbp_fit_tab_init(tab);
  // circular coordinates are staged wrapped (the identity for stored beliefs): every pair difference of
  // the fit is then within (-2pi, 2pi), which is what circ_sqdist relies on
  if (n < N) {
    const double v = s[k * Ncap + n];
    X[n] = X[n + N] = is_circ(M, k) ? wrap_pi(v) : v;
  }
  __syncthreads();
  if (SPEC) {  // latency mode: 2^depth - 1 workgroups per fit (lcv_bandwidth_1d_spec); a kernel of its own, so that the
               // registers the outcome tree needs do not cost the throughput kernel its occupancy
    const double hs = lcv_bandwidth_1d_spec<SPEC ? SPEC : 2>(X, N, Npad, is_circ(M, k), part, red, tab, ctr, area, role);
    if (n == 0 && role == 0) s[3 * Ncap + k] = hs;
    return;
  }
  double h = lcv_bandwidth_1d(X, N, Npad, is_circ(M, k), part, red, tab, ctr);
  if (n == 0) s[3 * Ncap + k] = h;
}

// ================================================================================================
// Bandwidth kernel: AMP.manikde!(M, pts) -- grid (njobs, 3): one workgroup per (slot, coordinate);
// the D per-coordinate fits of a KDE are independent.  Used for proposals (ApproxConv.jl:36-42),
// for the rebandwidth of products and for nbp_run_bandwidth.
// ================================================================================================
#define NBP_BANDWIDTH_ARGS const int32_t *slots, const int32_t *manifolds, double *arena, int N, int Npad, int64_t S, nbp_counters *ctr
#if NBP_TU & NBP_TU_PREP
#ifndef NBP_W_PREP
#define NBP_W_PREP 4
#endif
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(NBP_W_PREP)))
nbp_bandwidth_kernel(NBP_BANDWIDTH_ARGS) {
  extern __shared__ double smem[];  // grid (jobs, 3)
  lcv_slot_coordinate<0>(arena + S * slots[blockIdx.x], manifolds[blockIdx.x], blockIdx.y, N, Npad, smem, ctr);
}
#else
__global__ void nbp_bandwidth_kernel(NBP_BANDWIDTH_ARGS);
#endif
// The same kernel at FIVE waves per SIMD, for workgroups of 4k + 1 waves (N = 257 .. 320: BASELINE's config 5 runs N = 300).
// At four waves per SIMD a CU holds three such workgroups (15 waves); at five it holds four, 5 waves on every SIMD: 0.90 ->
// 0.78 ps per pair at N = 300, config 5's fits 189 -> 165 ms (profiles/r04_lcv_five_wave_rows.txt).  The cap costs this
// instance 16 B of scratch per lane (96 VGPRs); rows of 4k waves gain nothing from it (N = 200 / 256: 1 %) and keep the
// scratch-free kernel above.
#if NBP_TU & NBP_TU_PREPW5
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(5)))
nbp_bandwidth_kernel_w5(NBP_BANDWIDTH_ARGS) {
  extern __shared__ double smem[];  // grid (jobs, 3)
  lcv_slot_coordinate<0>(arena + S * slots[blockIdx.x], manifolds[blockIdx.x], blockIdx.y, N, Npad, smem, ctr);
}
#else
__global__ void nbp_bandwidth_kernel_w5(NBP_BANDWIDTH_ARGS);
#endif
#if NBP_TU & NBP_TU_PREPSPEC
template <int DEPTH>
__global__ void __launch_bounds__(1024)
nbp_bandwidth_kernel_spec(NBP_BANDWIDTH_ARGS, nbp_spec_area *spec) {
  extern __shared__ double smem[];  // grid (jobs, 3, 2^DEPTH - 1)
  lcv_slot_coordinate<DEPTH>(arena + S * slots[blockIdx.x], manifolds[blockIdx.x], blockIdx.y, N, Npad, smem, ctr,
                             spec + (blockIdx.x * 3 + blockIdx.y), blockIdx.z);
}
template __global__ void nbp_bandwidth_kernel_spec<2>(NBP_BANDWIDTH_ARGS, nbp_spec_area *spec);
template __global__ void nbp_bandwidth_kernel_spec<3>(NBP_BANDWIDTH_ARGS, nbp_spec_area *spec);
#else
template <int DEPTH> __global__ void nbp_bandwidth_kernel_spec(NBP_BANDWIDTH_ARGS, nbp_spec_area *spec);
#endif

// exp table | X[2N] | part[P][Npad] | acc[NW][2N] | red     (NW = P*Npad/64 waves)
static inline size_t nbp_bandwidth_lds_bytes(int N, int Npad, int P) {
  return (2 * (size_t)N + (size_t)P * Npad + (size_t)(P * Npad / 64) * 2 * N + NBP_RED + NBP_FITTAB) * 8;
}

#if NBP_TU & NBP_TU_PROPOSAL
__global__ void nbp_copy_kernel(const nbp_copy_desc *c, double *arena, int64_t S) {
  const double *src = arena + S * c[blockIdx.x].src_slot;
  double *dst = arena + S * c[blockIdx.x].dst_slot;
  for (int64_t i = threadIdx.x; i < S; i += blockDim.x) dst[i] = src[i];
}

// points only (NBP_STAGE_COPY_POINTS): 3N doubles, the bandwidth entries of the destination are not touched
__global__ void nbp_copy_points_kernel(const nbp_copy_desc *c, double *arena, int64_t S, int N) {
  const double *src = arena + S * c[blockIdx.x].src_slot;
  double *dst = arena + S * c[blockIdx.x].dst_slot;
  for (int i = threadIdx.x; i < 3 * N; i += blockDim.x) dst[i] = src[i];
  if (threadIdx.x == 0) dst[3 * N + 6] = src[3 * N + 6];  // the particle count belongs to the points
}
#else
__global__ void nbp_copy_kernel(const nbp_copy_desc *c, double *arena, int64_t S);
__global__ void nbp_copy_points_kernel(const nbp_copy_desc *c, double *arena, int64_t S, int N);
#endif

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
#if NBP_TU & NBP_TU_PROPOSAL
// re-key every op of a resident program: seed <- splitmix64(seed ^ splitmix64(salt)) (seeds.py mix_seed)
__global__ void nbp_reseed_kernel(char *blob, const int64_t *seed_off, int n, uint64_t salt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    uint64_t *s = (uint64_t *)(blob + seed_off[i]);
    *s = splitmix64(*s ^ splitmix64(salt));
  }
}
// blanks the rendezvous areas of a launch of speculative fits (all ones: the bit pattern no likelihood takes).  A kernel of
// the library's own in front of the fit kernel, not hipMemsetAsync: kernel -> kernel is the one ordering every path (plain
// stream, captured graph, several processes on the device) is known to keep
__global__ void nbp_spec_blank_kernel(unsigned long long *p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = ~0ull;
}
#else
__global__ void nbp_reseed_kernel(char *blob, const int64_t *seed_off, int n, uint64_t salt);
__global__ void nbp_spec_blank_kernel(unsigned long long *p, int n);
#endif

// ================================================================================================
// AMP.manifoldProduct(dens, M; Niter, N) + setBelief!  (GraphProductOperations.jl:53-60,
// SolveTree.jl:74), split over two kernels.
//
// Algorithm (Ihler et al. NIPS 2003, `prodAppxMSGibbsS`): every input KDE gets a balanced KD-tree
// whose nodes carry the moment-matched Gaussian of their leaves; all N output samples walk the
// trees root->leaves in lock step; at every level each sample runs `niter` sequential Gibbs sweeps
// re-drawing its label in density j from p(l_j | others) over ALL nodes of that level (inverse
// CDF); at the leaves the sample is drawn from the product of the F selected kernels.
//
// nbp_prep_kernel   (heterogeneous grid): workgroup b < 3*nbw fits the bandwidth of one (slot,
//                   coordinate); the others build the KD permutation of one (product, density) and
//                   leave the sorted, centred coordinates in an HBM workspace.  The tree build does
//                   not need bandwidths, so it runs beside the LCV fits of the same update instead
//                   of behind them.
// nbp_product_kernel_{l8,m4,t2}: grid (nprod, G): workgroup (p, g) draws the output samples
//                   [g*SPB, (g+1)*SPB) of product p with HL = 8/4/2 helper lanes per sample (adjacent
//                   lanes of one wave); results do not depend on the geometry beyond rounding (the
//                   RNG is keyed by the sample index).
// ================================================================================================
// HBM workspace of one (product, density): xs[3][N] | cen[4] | idx[N] (int32) | node sums st[3][2][TOT]
// (st[k][0][g] = sum, st[k][1][g] = sum of squares of the sorted, centred coordinate k over the leaves of
// node g, g = position in the level tables; TOT = nbp_kd_nodes_cap(N) >= number of nodes of all levels)
__host__ __device__ inline size_t nbp_kd_nodes_cap(int N) { return (size_t)3 * N + 8; }
__host__ __device__ inline size_t nbp_kd_stats_offset(int N) { return (size_t)3 * N + 4 + (size_t)(N + 1) / 2; }
__host__ __device__ inline size_t nbp_kd_ws_doubles(int N) { return nbp_kd_stats_offset(N) + 6 * nbp_kd_nodes_cap(N); }

// KD-tree permutation of one density: median split of the widest coordinate, by rank counting
// inside each segment (P helper lanes per position, no sort network).
#define NBP_KD_PARTS 512
template <int D>
// Outputs: xs[D][N] the sorted, centred coordinates (may be `x` itself: the points are staged before anything is
// written), cen[D] the centres, widx[N] the permutation, st = the node sums of every level (null: the caller takes them
// from xs itself -- the fused update kernel keeps the tree in LDS).
__device__ __forceinline__ void kd_build(const double *x, double *xs, double *cen_out, int *widx, double *st, int N, int Npad, const nbp_levels &T,
                                         double *smem, int mask /* coordinates the density informs */) {
  const int tid = threadIdx.x, TB = blockDim.x, P = TB / Npad, s = tid % Npad, sub = tid / Npad;
  double *raw = smem;                   // [D][N]
  double *ext = raw + (size_t)D * N;    // [3*Npad]
  double *red = ext + 3 * Npad;         // [NBP_RED]
  int *tmpA = (int *)(red + NBP_RED);   // [N]
  int *tmpB = tmpA + N;                 // [N]
  int *prk = tmpB + N;                  // [P*Npad]
  double *pmn = (double *)(prk + P * Npad + (P * Npad & 1));  // [NBP_KD_PARTS] partial minima of the extent pass
  double *pmx = pmn + NBP_KD_PARTS;     // [NBP_KD_PARTS]
  if (tid < N) {
#pragma unroll
    for (int k = 0; k < D; k++) raw[k * N + tid] = x[k * N + tid];
    tmpA[tid] = tid;
  }
  __syncthreads();
  int *pa = tmpA, *pb = tmpB;
  for (int l = 0; l < T.L; l++) {
    const int cnt = T.cnt[l], off = T.off[l];
    if (D > 1) {
      // extent of every segment in every coordinate.  Long segments (the top levels) are cut into parts of about
      // four positions, one thread each, and a second pass combines the parts: min/max do not care about the order
      const int len0 = T.node_hi[off] - T.node_lo[off];
      int Q = (len0 + 3) / 4;
      if (Q > NBP_KD_PARTS / (cnt * D)) Q = NBP_KD_PARTS / (cnt * D);
      if (Q > 1) {
        for (int item = tid; item < cnt * D * Q; item += TB) {
          const int zk = item / Q, part = item % Q, z = zk / D, k = zk % D;
          const int lo = T.node_lo[off + z], len = T.node_hi[off + z] - lo;
          const int a = lo + (part * len) / Q, b = lo + ((part + 1) * len) / Q;
          double mn = INFINITY, mx = -INFINITY;
#pragma unroll 4
          for (int p = a; p < b; p++) {
            const double v = raw[k * N + pa[p]];
            mn = fmin(mn, v);
            mx = fmax(mx, v);
          }
          pmn[item] = mn;
          pmx[item] = mx;
        }
        __syncthreads();
        for (int item = tid; item < cnt * D; item += TB) {
          double mn = INFINITY, mx = -INFINITY;
          for (int q = 0; q < Q; q++) {
            mn = fmin(mn, pmn[item * Q + q]);
            mx = fmax(mx, pmx[item * Q + q]);
          }
          ext[item] = mx - mn;
        }
      } else {
        for (int item = tid; item < cnt * D; item += TB) {
          const int z = item / D, k = item % D;
          const int lo = T.node_lo[off + z], hi = T.node_hi[off + z];
          double mn = INFINITY, mx = -INFINITY;
          for (int p = lo; p < hi; p++) {
            const double v = raw[k * N + pa[p]];
            mn = fmin(mn, v);
            mx = fmax(mx, v);
          }
          ext[item] = mx - mn;
        }
      }
      __syncthreads();
    }
    int lo = 0, hi = 0, me = 0;
    if (s < N) {
      const int node = T.pos_node[l * N + s];
      lo = T.node_lo[off + node];
      hi = T.node_hi[off + node];
      me = pa[s];
      int rank = 0;
      if (hi - lo > 1) {
        int best = 0;
        if (D > 1) {  // widest informed coordinate of the segment (first one on ties)
          double bext = -1.0;
#pragma unroll
          for (int k = 0; k < D; k++)
            if (((mask >> k) & 1) && ext[node * D + k] > bext) { bext = ext[node * D + k]; best = k; }
        }
        const double v = raw[best * N + me];
        const int len = hi - lo, a = lo + (sub * len) / P, b = lo + ((sub + 1) * len) / P;
#pragma unroll 4
        for (int p = a; p < b; p++) {
          const int ip = pa[p];
          const double vp = raw[best * N + ip];
          rank += (vp < v || (vp == v && ip < me)) ? 1 : 0;
        }
      }
      prk[sub * Npad + s] = rank;
    }
    __syncthreads();
    if (sub == 0 && s < N) {
      int rank = 0;
      for (int q = 0; q < P; q++) rank += prk[q * Npad + s];
      pb[(hi - lo > 1) ? lo + rank : s] = me;
    }
    __syncthreads();
    int *t = pa; pa = pb; pb = t;
  }
  double *srt = ext;  // [D][Npad] sorted, centred coordinates (the extent scratch is free now)
#pragma unroll
  for (int k = 0; k < D; k++) {
    double c = block_sum(tid < N ? raw[k * N + tid] : 0.0, red) / (double)N;
    if (tid == 0) cen_out[k] = c;
    if (tid < N) {
      const double v = raw[k * N + pa[tid]] - c;
      xs[k * N + tid] = v;
      srt[k * Npad + tid] = v;
    }
  }
  if (tid < N) widx[tid] = pa[tid];
  __syncthreads();
  // node sums of every level (the moment-matched Gaussians of the product's multiscale sampler): done here,
  // beside the bandwidth fits of the same launch, so that the product kernel only reads them
  if (st) {
    const int g0 = T.off[1], TOT = T.off[T.L] + T.cnt[T.L];
    const size_t cap = nbp_kd_nodes_cap(N);
    for (int item = tid; item < (TOT - g0) * D; item += TB) {
      const int g = g0 + item % (TOT - g0), k = item / (TOT - g0);
      const int lo = T.node_lo[g], hi = T.node_hi[g];
      double s1 = 0, s2 = 0;
      for (int p = lo; p < hi; p++) { const double v = srt[k * Npad + p]; s1 += v; s2 += v * v; }
      st[(size_t)(k * 2) * cap + g] = s1;
      st[(size_t)(k * 2 + 1) * cap + g] = s2;
    }
  }
}

// sample(oldBel, nn) (GraphProductOperations.jl:39-45): a belief that holds fewer than N points is topped up, in
// place, with draws from its own KDE (random kernel + bw * randn); the points it has stay where they are.  One
// workgroup per slot.  Used for the oldPoints of a product (prep launch) and by nbp_run_resample.
__device__ __forceinline__ void topup_slot(double *s, int N, int manifold, uint64_t seed) {
  const int cnt = slot_count(s, N), D = mani_dim(manifold);
  if (cnt >= N) return;  // block-uniform
  for (int n = cnt + threadIdx.x; n < N; n += blockDim.x) {
    double ua, ub, nn[4] = {0, 0, 0, 0};
    uniform_pair(seed, n, PURP_OLDSEL, 0, ua, ub);
    int i = (int)(ua * cnt);
    if (i >= cnt) i = cnt - 1;
    normal_pair(seed, n, PURP_OLDNOISE, 0, nn[0], nn[1]);
    if (D > 2) normal_pair(seed, n, PURP_OLDNOISE, 1, nn[2], nn[3]);
    for (int k = 0; k < 3; k++) {
      const double v = (k < D) ? s[k * N + i] + s[3 * N + k] * nn[k] : 0.0;
      s[k * N + n] = is_circ(manifold, k) ? wrap_pi(v) : v;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) s[3 * N + 6] = 0.0;
}
#if NBP_TU & NBP_TU_PROPOSAL
__global__ void nbp_resample_kernel(const int32_t *slots, const int32_t *manifolds, double *arena, int N, int64_t S, uint64_t seed) {
  topup_slot(arena + S * slots[blockIdx.x], N, manifolds[blockIdx.x], seed + 0x9E3779B97F4A7C15ull * (uint64_t)(blockIdx.x + 1));
}
#else
__global__ void nbp_resample_kernel(const int32_t *slots, const int32_t *manifolds, double *arena, int N, int64_t S, uint64_t seed);
#endif

static inline size_t nbp_kd_lds_bytes(int D, int N, int Npad, int P) {
  return ((size_t)D * N + 3 * Npad + NBP_RED + 2 * NBP_KD_PARTS) * 8 + ((size_t)2 * N + (size_t)P * Npad + 2) * 4;
}

// kdF: the largest nfactors of the batch in its low 16 bits; bit 16 = the product launch behind this one takes the node
// sums from the sorted coordinates itself (product_kernel_uniform<.., XS = true>): the KD builds leave them out
#define NBP_KD_NOSTATS 0x10000
template <int SPEC>
__device__ __forceinline__ void prep_body(const int32_t *bw_slots, const int32_t *bw_manis, int nbw, const nbp_product_desc *descs, int nprod, int kdF_,
                                          double *arena, double *ws, int N, int Npad, int64_t S, const nbp_levels &T, nbp_counters *ctr,
                                          nbp_spec_area *spec, double *smem) {
  const int kdF = kdF_ & 0xFFFF;
  const bool nostats = (kdF_ & NBP_KD_NOSTATS) != 0;
  const int b = blockIdx.x;
  constexpr int KS = SPEC ? (1 << SPEC) - 1 : 1;  // workgroups per (slot, coordinate)
  if (b < 3 * nbw * KS) {  // manikde! bandwidth of (slot, coordinate)
    const int job = b / (3 * KS), k = (b % (3 * KS)) / KS, role = b % KS;
    lcv_slot_coordinate<SPEC>(arena + S * bw_slots[job], bw_manis[job], k, N, Npad, smem, ctr, SPEC ? spec + (job * 3 + k) : nullptr, role);
    return;
  }
  const int q = b - 3 * nbw * KS, p = q / kdF, j = q % kdF;  // kdF = largest nfactors of the batch
  if (p >= nprod) return;
  const nbp_product_desc *d = descs + p;
  if (d->nfactors == 1 || j >= d->nfactors) return;
  const double *x = arena + S * d->in_slot[j];
  double *wsj = ws + (size_t)(p * kdF + j) * nbp_kd_ws_doubles(N);
  const int mask = d->in_partial[j] ? d->in_partial[j] : 7;
  if (j == 0 && d->old_slot >= 0) topup_slot(arena + S * d->old_slot, N, d->manifold, d->seed);  // oldPoints of the product
  double *cenj = wsj + 3 * N, *stj = nostats ? nullptr : wsj + nbp_kd_stats_offset(N);
  int *idxj = (int *)(wsj + 3 * N + 4);
  switch (mani_dim(d->manifold)) {
  case 1: kd_build<1>(x, wsj, cenj, idxj, stj, N, Npad, T, smem, 1); break;
  case 2: kd_build<2>(x, wsj, cenj, idxj, stj, N, Npad, T, smem, mask); break;
  default: kd_build<3>(x, wsj, cenj, idxj, stj, N, Npad, T, smem, mask); break;
  }
}
#define NBP_PREP_ARGS const int32_t *bw_slots, const int32_t *bw_manis, int nbw, const nbp_product_desc *descs, int nprod, int kdF, \
                      double *arena, double *ws, int N, int Npad, int64_t S, nbp_levels T, nbp_counters *ctr
#if NBP_TU & NBP_TU_PREP
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(NBP_W_PREP)))
nbp_prep_kernel(NBP_PREP_ARGS) {
  extern __shared__ double smem[];
  prep_body<0>(bw_slots, bw_manis, nbw, descs, nprod, kdF, arena, ws, N, Npad, S, T, ctr, nullptr, smem);
}
#else
__global__ void nbp_prep_kernel(NBP_PREP_ARGS);
#endif
#if NBP_TU & NBP_TU_PREPW5  // five waves per SIMD: see nbp_bandwidth_kernel_w5
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(5)))
nbp_prep_kernel_w5(NBP_PREP_ARGS) {
  extern __shared__ double smem[];
  prep_body<0>(bw_slots, bw_manis, nbw, descs, nprod, kdF, arena, ws, N, Npad, S, T, ctr, nullptr, smem);
}
#else
__global__ void nbp_prep_kernel_w5(NBP_PREP_ARGS);
#endif
// latency mode: 2^DEPTH - 1 workgroups per fit (lcv_bandwidth_1d_spec)
#if NBP_TU & NBP_TU_PREPSPEC
template <int DEPTH>
__global__ void __launch_bounds__(1024)
nbp_prep_kernel_spec(NBP_PREP_ARGS, nbp_spec_area *spec) {
  extern __shared__ double smem[];
  prep_body<DEPTH>(bw_slots, bw_manis, nbw, descs, nprod, kdF, arena, ws, N, Npad, S, T, ctr, spec, smem);
}
template __global__ void nbp_prep_kernel_spec<2>(NBP_PREP_ARGS, nbp_spec_area *spec);
template __global__ void nbp_prep_kernel_spec<3>(NBP_PREP_ARGS, nbp_spec_area *spec);
#else
template <int DEPTH> __global__ void nbp_prep_kernel_spec(NBP_PREP_ARGS, nbp_spec_area *spec);
#endif

// Gibbs geometry: HL adjacent lanes of ONE wave serve one output sample (a wave carries 64/HL samples).
// The helpers of a sample split the nodes of a level into contiguous ranges and combine their shares
// with wave shuffles, so a conditional draw needs no barrier and no LDS round trip; the only
// workgroup barriers left are the two around the per-level node statistics.  The host picks HL per
// launch: 8 when the launch cannot fill the chip (latency: short ranges, several small workgroups per
// product), 4 or 2 when it can (throughput: one workgroup per product computes the node statistics once).
struct product_lds {
  double *lm, *lv, *lr, *ls, *lc, *lg, *cen, *h2, *nw, *tab;
  double *uu;  // [F][SPB][2]: the uniforms of this pass's draws (one Philox block per sample and density: sampleIndices! | first sweep)
  double *ck;  // [nch][2][TB]: chunk sums and running maxima of a lane's pass-1 range (throughput geometries; CK doubles)
  int *ind;
  int ns;  // nodes a (density, coordinate) row holds: N (one level at a time) or the node count of the whole tree
};
// RESIDENT LEVELS (bit 17 of a product kernel's F argument): the node statistics of EVERY level are staged once, at node
// index = position, and the workgroup's waves then walk the levels without a barrier between them (2.3 N nodes per row
// instead of N: taken where the LDS of two workgroups per CU allows it, launch_products)
// Largest workgroup of a throughput product kernel: 16 waves for the Euclidean instances (<= 128 VGPRs: the bound costs them
// nothing), 8 for the others (the SE(2) and generic kernels hold 186-221 VGPRs and would spill under a bound of 12 or 16 waves).
// A product whose samples need more than eight waves (N = 300 at two helper lanes: ten) then runs as ONE workgroup, with one
// staging of the node statistics, instead of a full one and a nearly empty one: config 5's products 162 -> 122 ms per solve.
#define NBP_PROD_WIDE(MANI) ((MANI) == NBP_EUCLID1 || (MANI) == NBP_EUCLID2 || (MANI) == NBP_EUCLID3)
#define NBP_PROD_LB(MANI) (NBP_PROD_WIDE(MANI) ? 1024 : 512)
#define NBP_PROD_ALL_LEVELS 0x20000
// bits 18 .. 21 of the same argument: chunks per helper range of the throughput geometries (their sums live in LDS; the host
// takes as many as the LDS of the launch allows, launch_products)
#define NBP_PROD_NCH_SHIFT 18
#define NBP_PROD_NCH(kdF) ((((kdF) >> NBP_PROD_NCH_SHIFT) & 15) ? (((kdF) >> NBP_PROD_NCH_SHIFT) & 15) : 2)

// `big` = the per-level node statistics (3 x F x D x N doubles) do not fit the LDS: they go to a scratch
// area private to the workgroup in global memory and are served by L1/L2; LDS then holds only the small
// per-product items.
__host__ __device__ inline size_t nbp_product_gstats_doubles(int F, int D, int N) { return (3 * (size_t)F * D + 3 * (size_t)F) * N; }
// CK: doubles of the chunk area (throughput geometries: nch x 2 x lanes of the workgroup; 0: chunk sums stay in registers);
// circ: the launch may hold a product with a circular coordinate (false: the sin / cos rows are left out)
__host__ __device__ inline size_t product_lds_layout(int F, int D, int N, int SPB, bool big, double *base, product_lds *L, int NS = 0,
                                                     size_t CK = 0, bool circ = true) {
  size_t o = 0;
  auto dbl = [&](size_t n) { size_t r = o; o += n; return r; };
  if (NS <= 0) NS = N;
  const size_t bulk = big ? 0 : (size_t)F * D * NS;
  size_t lm = dbl(bulk), lv = dbl(bulk), lr = dbl(bulk);
  // circular coordinate (one per manifold at most): sin / cos of every node mean times its precision, so that the
  // conditional mean of a draw is two sums and one atan2 instead of a sincos per density
  size_t ls = dbl((big || !circ) ? 0 : (size_t)F * NS), lc = dbl((big || !circ) ? 0 : (size_t)F * NS);
  // g_z = w_z / sqrt(prod_k var_zk): the weight of node z beside its exponential when the label is drawn on a POINT
  // (sampleIndices!: nothing is added to the node's own variance), once per node instead of once per draw
  size_t lg = dbl(big ? 0 : (size_t)F * NS);
  size_t cen = dbl((size_t)F * 3), h2 = dbl((size_t)F * 3);
  size_t nw = dbl((size_t)NS), tab = dbl(NBP_EXPTAB);
  size_t uu = dbl((size_t)2 * F * SPB), ck = dbl(CK);
  size_t ints0 = o;
  if (L) {
    L->lm = base + lm; L->lv = base + lv; L->lr = base + lr; L->ls = base + ls; L->lc = base + lc; L->lg = base + lg; L->cen = base + cen; L->h2 = base + h2;
    L->nw = base + nw; L->tab = base + tab;
    L->uu = base + uu; L->ck = base + ck;
    L->ind = (int *)(base + ints0);
    L->ns = NS;
  }
  return ints0 * 8 + (size_t)F * SPB * 4;  // ind[F][SPB]  (a second row, the labels of the next level, went with the round-4 sampler:
                                          //  3 KB at three densities -- what kept two chunks per range and two workgroups per CU apart)
}

// PARTIAL: some input density is partial (AMP.marginal(propBel, pardims), ApproxConv.jl:287-291): a
// density enters the conditionals and the final draw on its own coordinates only; a coordinate that
// no density informs keeps the old point (GraphProductOperations.jl:39-45).  Separate instantiation so
// that the all-full path carries no masks.
// BIG: the node statistics live in this workgroup's scratch in global memory (products with many densities; only the
// latency geometries HL = 16 / 8 run them).  A compile-time switch: with a run-time one every access to the statistics
// goes through a generic 64-bit pointer -- flat loads in the Gibbs loop and register pairs for what is an LDS offset.
// FUSED (the fused update kernel, nbp_fused.h): the densities are the proposals this workgroup has just made -- sorted,
// centred coordinates, centres, permutations and bandwidths in LDS (`fio`), node sums taken from the sorted coordinates
// level by level (the same leaf-order sums kd_build leaves in the HBM workspace), the result into an LDS slot.
#define NBP_FUSED_MAXF 4
struct nbp_fused_io {
  product_lds L;                      // the product's own LDS areas
  const double *xs;                   // sorted, centred coordinates of density j: [D][N] at xs + j * xs_stride
  size_t xs_stride;
  const int *idx;                     // [N] permutation of density j at idx + j * idx_stride (only for label output)
  size_t idx_stride;
  const double *cen;                  // [F][3]
  const double *bw;                   // [F][3]
  double *out;                        // slot-shaped LDS area that receives the product's points
};
// FUSED: 0 = densities from the KD workspaces; 1 = the fused update kernel (everything in LDS, `fio` made by the caller);
//        2 = the _xs product kernels (sorted coordinates staged in LDS, `fio->L` laid out by product_lds_layout)
// nch (throughput geometries, HL <= 4, not the update kernel): chunks per helper range, their sums kept in LDS (`L.ck`)
template <int MANI, bool PARTIAL, int HL, bool BIG, int FUSED = 0>
__device__ __forceinline__ void product_body(const nbp_product_desc *d, double *arena, const double *ws, int kdF, double *gstats,
                                             int N, int64_t S, int32_t *side, const nbp_levels &T, double *smem,
                                             const nbp_fused_io *fio = nullptr, bool all_levels = false, int nch = 2, bool lay_circ = true) {
  constexpr bool CKL = (HL <= 4) && (FUSED != 1);  // chunk sums in LDS
  constexpr int D = (MANI == NBP_SE2) ? 3 : (MANI == NBP_CIRCULAR ? 1 : MANI);
  constexpr bool circ[3] = {MANI == NBP_CIRCULAR, false, MANI == NBP_SE2};
  const int F = d->nfactors, tid = threadIdx.x, TB = blockDim.x;
  const int SPB = TB / HL;                 // output samples of this workgroup
  const int sl = tid / HL, h = tid % HL;   // sample (local), helper index
  const int s = blockIdx.y * SPB + sl;
  const bool live = s < N;
  constexpr bool big = BIG;
  product_lds L;
  const int TOT = T.off[T.L] + T.cnt[T.L];
  if constexpr (FUSED) L = fio->L;
  else product_lds_layout(F, D, N, SPB, big, smem, &L, (all_levels && !big) ? TOT : N, CKL ? (size_t)nch * 2 * TB : 0, lay_circ);
  const int NS = big ? N : L.ns;   // row length of the statistics
  const bool all = NS != N;        // every level resident
  double *cen = L.cen, *h2 = L.h2;
  int *ind = L.ind;
  double *out = FUSED ? fio->out : arena + S * d->out_slot;
  const double *wsp = FUSED ? nullptr : ws + (size_t)blockIdx.x * kdF * nbp_kd_ws_doubles(N);
  // node statistics: LDS, or (big) this workgroup's private scratch in global memory
  // (the stride is the LAUNCH's -- its largest density count, three coordinates: what launch_products sized the area by.  By the
  //  product's own count, as it was through round 5, the workgroups of a launch that mixes density counts wrote over each
  //  other's statistics: products of 3-D manifolds at N >= ~260 with four densities beside smaller ones came out wrong or
  //  non-finite -- found by the whole-solve differential check, tools/exp/se2_mixed_products.py)
  double *gs = big ? gstats + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * nbp_product_gstats_doubles(kdF & 0xFFFF, 3, N) : nullptr;
  double *lm = big ? gs : L.lm, *lv = big ? gs + (size_t)F * D * N : L.lv, *lr = big ? gs + 2 * (size_t)F * D * N : L.lr;
  double *lsn = big ? gs + 3 * (size_t)F * D * N : L.ls, *lcs = big ? gs + (3 * (size_t)F * D + F) * N : L.lc;
  double *lgw = big ? gs + (3 * (size_t)F * D + 2 * (size_t)F) * N : L.lg;
  constexpr int KC = (MANI == NBP_CIRCULAR) ? 0 : (MANI == NBP_SE2 ? 2 : -1);  // the circular coordinate, if any
  nbp_exp_tab_init(L.tab);
  NBP_CTICK_INIT();
  // ---- bandwidths and centres of every density -------------------------------------------------
  const size_t stcap = nbp_kd_nodes_cap(N);
  for (int t = tid; t < F * 3; t += TB) {
    const int j = t / 3, k = t % 3;
    const double bw = FUSED ? fio->bw[t] : arena[S * d->in_slot[j] + 3 * N + k];
    h2[t] = (k < D) ? bw * bw : 0.0;
    cen[t] = FUSED ? ((k < D) ? fio->cen[t] : 0.0) : wsp[(size_t)j * nbp_kd_ws_doubles(N) + 3 * N + k];
  }
  if (h == 0)
    for (int j = 0; j < F; j++) ind[j * SPB + sl] = 0;  // levelInit! / initIndices!: root
  // samplePoint!: a draw from the product of the Gaussians the labels select -- between the levels the point x the labels
  // of the next level are drawn on, after the last level the output sample.  Its mean and precision are taken at the end of
  // a level (the statistics of that level are still in place), the normals at the top of the next: one site each.  Every
  // helper lane of a sample carries them.
  double xp[D], xmu[D], xpr[D];  // the point drawn on the level above; mean and precision of the next one
#pragma unroll
  for (int k = 0; k < D; k++) { xp[k] = 0.0; xmu[k] = 0.0; xpr[k] = 0.0; }
  // ---- multiscale Gibbs (Ihler, Sudderth, Freeman, Willsky, NIPS 2003, sec. 4; the gibbs1 loop of KernelDensityEstimate.jl):
  //      per level  samplePoint! (x from the labels of the level above) -> levelDown! (the candidates of every density
  //      are ALL nodes of this level) -> sampleIndices! (every label given x, independently) -> Niter sweeps of sampleIndex
  // Passes: KernelDensityEstimate's Nlevels = floor(log2(maxNp) + 1) -- the depth of the tree, except that a particle count
  // that is a power of two gets one pass more than its tree has levels: the leaf level twice (a leaf is its own child)
  int npass = 1;
  for (int n2_ = N; n2_ > 1; n2_ >>= 1) npass++;
  if (npass < T.L) npass = T.L;
  int Lc = 0;  // the coarse levels 0 .. Lc share one staging (below): as many as one row of the statistics holds
  while (Lc + 1 <= T.L && T.off[Lc + 1] + T.cnt[Lc + 1] <= N) Lc++;
  for (int ps = 0; ps <= npass + 1; ps++) {
    const int l = ps < T.L ? ps : T.L;  // the tree level of this pass
    if (ps > 0 && live) {  // samplePoint!
      double n0, n1, n2 = 0, n3 = 0;
      const uint32_t purpose = (ps <= npass) ? PURP_PLEVEL : PURP_PFINAL, k0 = (ps <= npass) ? (uint32_t)(2 * ps) : 0u;
      const double2 na = normal_pair_call(d->seed, (uint32_t)s, purpose, k0);
      n0 = na.x;
      n1 = na.y;
      if (D > 2) { const double2 nb = normal_pair_call(d->seed, (uint32_t)s, purpose, k0 + 1); n2 = nb.x; n3 = nb.y; }
      (void)n3;
      const double nn[3] = {n0, n1, n2};
#pragma unroll
      for (int k = 0; k < D; k++) {
        // (xpr = 0, PARTIAL only: a coordinate no density informs -- it enters no weight and keeps the old point)
        const double v = (PARTIAL && !(xpr[k] > 0)) ? 0.0 : xmu[k] + sqrt(1.0 / xpr[k]) * nn[k];
        xp[k] = circ[k] ? wrap_pi(v) : v;
      }
    }
    NBP_CTICK(47);  // samplePoint!: the normals
    if (ps > npass) break;
    const int cnt = T.cnt[l], off = T.off[l];
    // THE COARSE LEVELS TOGETHER (round 6): the levels 0 .. Lc whose nodes fit one row of N statistics between them (N = 200:
    // 1 + 2 + ... + 64 = 127 nodes, levels 0 .. 6 of 8) are staged in ONE go in front of the first pass, at row position = node
    // index, and the waves then walk them without a barrier between the levels; only the fine levels (128 and 200 nodes) are
    // staged one by one.  A level's staging is two loops, three barriers and a dependent chain of table reads whatever its node
    // count -- six of nine of them per product were spent on levels whose draws take a few hundred cycles.  Same statistics,
    // same draws: only where a level's nodes sit in the row changes (`lb`).
#ifndef NBP_X_COARSE_BLOCK
#define NBP_X_COARSE_BLOCK 1
#endif
    const bool inblock = !all && NBP_X_COARSE_BLOCK && l <= Lc;
    const int lb = (all || inblock) ? off : 0;  // where this level's nodes start in a row of the statistics
    if ((all ? ps == 0 : (ps == 0 || !inblock)) && ps <= T.L) {  // (a repeated leaf pass finds its statistics in place)
    __syncthreads();
    NBP_CTICK(40);  // staging (first level) / Gibbs draws of the previous level
    // node statistics: of this level, of the coarse levels together, or (resident levels) of every level at once -- node g
    // of the tree at row position g - g0: a level's nodes are contiguous in the tree's node arrays
    const int g0 = (all || inblock) ? 0 : off, gn = all ? TOT : (inblock ? T.off[Lc] + T.cnt[Lc] : cnt);
    for (int item = tid; item < F * D * gn; item += TB) {
      const int z = item % gn, jk = item / gn, g = g0 + z;
      const int lo = T.node_lo[g], hi = T.node_hi[g];
      const int j = jk / D, k = jk % D;
      const bool root = (g == 0 && T.L > 0);  // the root: the sums of its two children added (the KD build leaves the levels below it)
      double s1, s2;
      if constexpr (FUSED) {  // the same sums in the same (leaf) order as kd_build's, from the tree in LDS
        const double *srt = fio->xs + j * fio->xs_stride + k * N;
        const int mid = root ? T.node_hi[T.off[1]] : hi;
        s1 = 0;
        s2 = 0;
        for (int p = lo; p < mid; p++) { const double v = srt[p]; s1 += v; s2 += v * v; }
        if (mid < hi) {
          double t1 = 0, t2 = 0;
          for (int p = mid; p < hi; p++) { const double v = srt[p]; t1 += v; t2 += v * v; }
          s1 += t1;
          s2 += t2;
        }
      } else {  // the node sums were left by the KD build of this density (nbp_prep_kernel)
        const double *st = wsp + (size_t)j * nbp_kd_ws_doubles(N) + nbp_kd_stats_offset(N) + (size_t)(k * 2) * stcap;
        if (root) {
          const int c0 = T.off[1];
          s1 = st[c0] + st[c0 + 1];
          s2 = st[stcap + c0] + st[stcap + c0 + 1];
        } else {
          s1 = st[g];
          s2 = st[stcap + g];
        }
      }
      const double nn = (double)(hi - lo), mu = s1 / nn;
      double var = s2 / nn - mu * mu;
      if (var < 0) var = 0;
      lm[jk * NS + z] = cen[j * 3 + k] + mu;
      const double vz = var + h2[j * 3 + k];
      lv[jk * NS + z] = vz;
      const double rz = 1.0 / vz;  // the precision, once per node: the draws below only multiply
      lr[jk * NS + z] = rz;
      if (KC >= 0 && k == KC) {
        double sn, cs_;
        sincos_fast(cen[j * 3 + k] + mu, &sn, &cs_);
        lsn[j * NS + z] = sn * rz;
        lcs[j * NS + z] = cs_ * rz;
      }
    }
    for (int z = tid; z < gn; z += TB) L.nw[z] = (double)(T.node_hi[g0 + z] - T.node_lo[g0 + z]) / (double)N;
    __syncthreads();
    for (int item = tid; item < F * gn; item += TB) {  // g_z of every node (over the coordinates the density informs)
      const int z = item % gn, j = item / gn;
      const int pmz = PARTIAL ? (d->in_partial[j] ? d->in_partial[j] : 7) : 7;
      double pv = 1.0;
#pragma unroll
      for (int k = 0; k < D; k++)
        if (!PARTIAL || ((pmz >> k) & 1)) pv *= lv[(j * D + k) * NS + z];
      lgw[j * NS + z] = rsqrt(pv) * L.nw[z];
    }
    __syncthreads();
    NBP_CTICK(41);  // node statistics
    }
    // The uniforms of this pass: ONE Philox block per (sample, density) -- its first uniform for sampleIndices!, its second for
    // the first sweep's sampleIndex.  The helper lanes of a sample take turns at the densities (lane h makes the blocks of
    // j = h, h + HL, ...: with two densities and two helpers one block's worth of instructions per pass instead of four) and
    // hand them over through LDS; the lanes of a sample are lanes of one wave, whose LDS operations stay in order.
    // (the latency geometries -- 8 to 32 helpers per sample, a CU to themselves, five manifolds' instances in one kernel at
    //  252 registers -- make the block where it is used, once for each of its two uniforms: the hand-over cost them scratch)
#ifndef NBP_X_UUL_LAT
#define NBP_X_UUL_LAT 1
#endif
    constexpr bool UUL = HL <= 4 || NBP_X_UUL_LAT;
    if (UUL && ps > 0 && live)
      for (int j0 = 0; j0 < F; j0 += HL) {
        const int j = j0 + h;
        if (j < F) {
          double u0, u1;
          uniform_pair(d->seed, s, PURP_PINDEX, (uint32_t)(ps * NBP_MAXF + j), u0, u1);
          L.uu[(j * SPB + sl) * 2] = u0;
          L.uu[(j * SPB + sl) * 2 + 1] = u1;
        }
      }
    NBP_CTICK(43);  // the uniforms of the pass
    // mean and precision of the product of the selected Gaussians (for the samplePoint! at the top of the next pass)
    auto point_moments = [&]() {
#pragma unroll
      for (int k = 0; k < D; k++) {
        double prec = 0, acc = 0, ss = 0, sc = 0;
        for (int q = 0; q < F; q++) {
          if (PARTIAL && d->in_partial[q] && !((d->in_partial[q] >> k) & 1)) continue;
          const int iq = ind[q * SPB + sl];
          const double rq = lr[(q * D + k) * NS + lb + iq];
          prec += rq;
          if (circ[k]) {
            ss += lsn[q * NS + lb + iq];
            sc += lcs[q * NS + lb + iq];
          } else
            acc += lm[(q * D + k) * NS + lb + iq] * rq;
        }
        xpr[k] = prec;
        xmu[k] = (PARTIAL && !(prec > 0)) ? 0.0 : (circ[k] ? nbpm_atan2(ss, sc) : acc / prec);
      }
    };
    const int z0 = (h * cnt) / HL, z1 = ((h + 1) * cnt) / HL;  // this helper's node range (the root, l = 0: nothing to draw)
    // the sweep is instantiated for the leaf level and for the levels above it: `leaf` as a run-time flag is a
    // wave-uniform branch in front of every node weight (two taken branches per node in the pass-1 loop)
    auto sweep = [&](auto leaf_c) {
    constexpr bool leaf = decltype(leaf_c)::value;
    for (int it = -1; ps > 0 && it < d->niter; it++) {  // it = -1: sampleIndices! (every label given the point x of the level above)
      for (int j = 0; j < F; j++) {  // sampleIndex(j): sequential Gibbs sweep
        // Draw l_j ~ p(l_j | others) by inverse CDF over the nodes of this level.
        //   weight_z = w_z * N(mean_z; mn, var_z + vn)  =  exp(a_z) * g_z,
        //   a_z = -0.5 * sum_k t_k^2 / v_k  (<= 0),   g_z = w_z / sqrt(prod_k v_k)
        // evaluated in the linear domain with ONE rsqrt per node (its square is the reciprocal of the
        // variance product) and the running max taken over a_z only; at the leaf level v_k and w_z
        // are the same for every node, so g_z cancels and the reciprocals are hoisted.
        // Pass 1 (all helpers, NCH chunks each): rescaled totals -> shuffle max / prefix sum.
        // Pass 2 (the helper whose share holds u * total): locate the chunk, re-evaluate just that chunk.
        auto draw = [&](auto xp_c) {
        constexpr bool XP = decltype(xp_c)::value;  // sampleIndices!: the label given the POINT x (nothing added to a node's variance)
#ifndef NBP_X_NCH
#define NBP_X_NCH 2  // (four chunks cost sixteen more registers: 51 spilled at four waves per SIMD, 142 MB of scratch traffic per chip-filling launch; with the helpers rescanning a chunk together the longer chunk costs nothing)
#endif
        constexpr int NCH = NBP_X_NCH;  // chunks per helper range where their sums stay in registers
        const int nchk = CKL ? nch : NCH;
        double mn[D], vn[D], ua = 0, ub = 0, m = -INFINITY, tot = 0;
        double cs[CKL ? 1 : NCH], ms[CKL ? 1 : NCH];
        double *ckc = L.ck + tid;  // chunk c of this lane: sum at ckc[2 c TB], running maximum at ckc[(2 c + 1) TB]
        (void)cs; (void)ms; (void)ckc;
        const double *mj = lm + j * D * NS + lb, *vj = lv + j * D * NS + lb, *rj = lr + j * D * NS + lb, *gj = lgw + j * NS + lb;
        double linv[D];
        bool use[D];  // PARTIAL: coordinates informed by density j and by at least one other
#pragma unroll
        for (int k = 0; k < D; k++) use[k] = true;
        const int pmj = PARTIAL ? (d->in_partial[j] ? d->in_partial[j] : 7) : 7;
        auto node_w = [&](int z, double &a, double &g) {
          double t[D], v[D];
#pragma unroll
          for (int k = 0; k < D; k++) {
            const double tmp = mj[k * NS + z] - mn[k];  // node mean in [-2pi, 2pi), mn in [-pi, pi]
            t[k] = circ[k] ? circ_sq(tmp) : tmp * tmp;
            v[k] = vj[k * NS + z] + vn[k];
            if (PARTIAL && !use[k]) { t[k] = 0.0; v[k] = 1.0; }
          }
          if (leaf) {
            double q = 0;
#pragma unroll
            for (int k = 0; k < D; k++) q = fma(t[k], linv[k], q);
            a = -0.5 * q;
            g = 1.0;
          } else if constexpr (XP) {  // the node's own precisions and its g_z from the staging
            double q = 0;
#pragma unroll
            for (int k = 0; k < D; k++) q = fma(t[k], (PARTIAL && !use[k]) ? 0.0 : rj[k * NS + z], q);
            a = -0.5 * q;
            g = gj[z];
          } else {
            double pv, num;
            if constexpr (D == 1) { pv = v[0]; num = t[0]; }
            // (weights decide label draws, they never travel as coordinates: their multiply-adds are spelled out as fma -- the
            //  library is compiled without contraction for the sake of the values that do travel)
            else if constexpr (D == 2) { pv = v[0] * v[1]; num = fma(t[0], v[1], t[1] * v[0]); }
            else { const double v01 = v[0] * v[1]; pv = v01 * v[2]; num = fma(t[0], v[1] * v[2], fma(t[1], v[0] * v[2], t[2] * v01)); }
            const double r = rsqrt_pos(pv);
            a = -0.5 * num * (r * r);
            g = r * L.nw[lb + z];
          }
        };
        // chunk size of this helper's range: a multiple of 4, the pass-1 loop takes the nodes four at a time
        const int zr = z1 - z0, csz = (((zr + nchk - 1) / nchk) + 3) & ~3;
        // a level with at most SR nodes per helper -- every level of a 200-particle tree at 32 helpers, the three or four
        // coarsest levels at 2 or 4 -- keeps the weights in registers: no chunks, no rescan, one exponential per node
#ifndef NBP_X_SR_THR
#define NBP_X_SR_THR 0  // register slots of the short-range draw in the throughput geometries (0 = chunks on every level; 4: measured 1 % on config 2 for 8 spilled registers and an occupancy step on Euclid(3): off)
#endif
        constexpr int SR = (HL >= 8) ? 8 : (NBP_X_SR_THR > 0 ? NBP_X_SR_THR : 1);
        const int nzmax = (cnt + HL - 1) / HL;
        const bool shortr = (HL >= 8 || NBP_X_SR_THR > 0) && nzmax <= SR;
        double wr[SR], gr[SR];
        NBP_CTICK(40);
        if (live) {
#pragma unroll
          for (int k = 0; k < D; k++) {  // product of all but the jth selected Gaussians
            if (XP || it < 0) {  // sampleIndices!: p(z) ~ w_z N(x; mean_z, var_z)
              mn[k] = xp[k];
              vn[k] = 0.0;
              if (PARTIAL) use[k] = ((pmj >> k) & 1) != 0;  // (a coordinate the density informs has a point: xpr > 0)
              continue;
            }
            double prec = 0, acc = 0, ss = 0, sc = 0;
            for (int q = 0; q < F; q++) {
              if (q == j) continue;
              if (PARTIAL && d->in_partial[q] && !((d->in_partial[q] >> k) & 1)) continue;
              const int iq = ind[q * SPB + sl];
              const double rq = lr[(q * D + k) * NS + lb + iq];
              prec += rq;
              if (circ[k]) {
                ss += lsn[q * NS + lb + iq];
                sc += lcs[q * NS + lb + iq];
              } else
                acc += lm[(q * D + k) * NS + lb + iq] * rq;
            }
            if (PARTIAL) {
              use[k] = ((pmj >> k) & 1) && prec > 0;
              if (!use[k]) prec = 1.0;
            }
            vn[k] = 1.0 / prec;
            mn[k] = circ[k] ? nbpm_atan2(ss, sc) : acc * vn[k];
          }
          if (leaf) {
#pragma unroll
            for (int k = 0; k < D; k++) linv[k] = (PARTIAL && !use[k]) ? 0.0 : 1.0 / (h2[j * 3 + k] + vn[k]);
          }
          if (UUL && it <= 0) ua = L.uu[(j * SPB + sl) * 2 + (it < 0 ? 0 : 1)];
          else {  // further sweeps: a block each
            uniform_pair(d->seed, s, it <= 0 ? PURP_PINDEX : PURP_PGIBBS, (uint32_t)(it <= 0 ? ps * NBP_MAXF + j : (ps * 8 + it) * NBP_MAXF + j), ua, ub);
            if (it == 0) ua = ub;
          }
          NBP_CTICK(44);  // conditional mean / variance of the other densities
          if (shortr) {  // at most SR nodes per helper: their weights stay in registers, nothing is evaluated twice
            // (nzmax = the longest range of the level, wave-uniform: the coarse levels of a 32-helper geometry have one node
            //  per helper or none, and the unrolled slots beyond that are branched over, not predicated through)
#pragma unroll
            for (int i = 0; i < SR; i++) { wr[i] = 0.0; gr[i] = 0.0; }
#pragma unroll
            for (int i = 0; i < SR; i++) {
              if (i >= nzmax) break;
              double a = -INFINITY, g = 0.0;
              if (z0 + i < z1) node_w(z0 + i, a, g);
              wr[i] = a;
              gr[i] = g;
              m = fmax(m, a);
            }
#pragma unroll
            for (int i = 0; i < SR; i++) {
              if (i >= nzmax) break;
              wr[i] = (z0 + i < z1) ? exp_nonpos(wr[i] - m, L.tab) * gr[i] : 0.0;
              tot += wr[i];
            }
          } else {
          auto chunk = [&](int za, int zb, double &cur) {
            int z = za;
            // four nodes per step: four independent weight evaluations in flight (a lone wave on its SIMD is
            // bound by the dependent-chain latency of one) and one running-max update instead of four
            // (the register-capped throughput variants take this path only where it does not spill more: D <= 2)
            for (; (HL >= 8 || D <= 2) && z + 3 < zb; z += 4) {
              double a0, a1, a2, a3, g0, g1, g2, g3;
              node_w(z, a0, g0);
              node_w(z + 1, a1, g1);
              node_w(z + 2, a2, g2);
              node_w(z + 3, a3, g3);
              const double am = fmax(fmax(a0, a1), fmax(a2, a3));
              if (am > m) {
                const double f = exp_nonpos(m - am, L.tab);  // m == -inf -> 0
                tot *= f;
                cur *= f;
                m = am;
              }
              const double w = fma(exp_nonpos(a0 - m, L.tab), g0, exp_nonpos(a1 - m, L.tab) * g1) +
                               fma(exp_nonpos(a2 - m, L.tab), g2, exp_nonpos(a3 - m, L.tab) * g3);
              tot += w;
              cur += w;
            }
            for (; z < zb; z++) {
              double a, g;
              node_w(z, a, g);
              if (a > m) {
                const double f = exp_nonpos(m - a, L.tab);  // m == -inf -> 0
                tot *= f;
                cur *= f;
                m = a;
              }
              const double w = exp_nonpos(a - m, L.tab) * g;
              tot += w;
              cur += w;
            }
          };
          if constexpr (CKL) {  // chunk c leaves the cumulative sum of the range up to its end and the running maximum there
            for (int c = 0; c < nchk; c++) {
              const int za = z0 + c * csz;
              if (za >= z1) break;
              double unused = 0;
              chunk(za, min(z1, za + csz), unused);
              ckc[(2 * c) * TB] = tot;
              ckc[(2 * c + 1) * TB] = m;
            }
          } else {
#pragma unroll
            for (int c = 0; c < NCH; c++) {
              double cur = 0;
              const int za = z0 + c * csz;
              chunk(za, min(z1, za + csz), cur);
              cs[c] = cur;
              ms[c] = m;
            }
          }
          }
        }
        NBP_CTICK(56 + (leaf ? 2 : 0) + (XP ? 1 : 0));  // pass 1: node weights of this helper's range (56 sweep, 57 on the point; 58 / 59 leaf level)
        // combine over the HL helper lanes of the sample (adjacent lanes of this wave): common max,
        // shares rescaled to it, inclusive prefix sum -> the one helper whose interval holds u * total
        const double Mx = group_max<HL>(m);
        const double share = (tot > 0) ? tot * exp_nonpos(m - Mx, L.tab) : 0.0;
        double total;
        const double incl = group_inclusive_scan<HL>(share, h, &total);
        const double target = ua * total, before = incl - share;
        int choice = -1;
        NBP_CTICK(45);  // shuffle combine
        if (shortr) {
          if (live && share > 0 && target >= before && target < incl) {
            const double f = exp_nonpos(m - Mx, L.tab);
            double c = before;
            choice = z1 - 1;
            bool hit = false;
#pragma unroll
            for (int i = 0; i < SR; i++) {
              if (i >= nzmax) break;
              c = fma(wr[i], f, c);
              if (!hit && z0 + i < z1 && target < c) { choice = z0 + i; hit = true; }
            }
          }
        } else if constexpr (HL <= 4) {
          // pass 2, throughput geometries: the helper whose share holds the target locates the chunk; ALL helpers of the
          // sample then rescan it together -- a row of HL nodes at a time, prefix over the row through DPP -- instead of
          // idling while the one lane walks it (the rescan was a fifth of a chip-filling launch)
          const bool own = live && share > 0 && target >= before && target < incl;
          int pk = -1;
          double cb = -INFINITY;
          if (own) {
            double cacc = before;
            int za = z0, zb = z1;
            bool found = false;
            if constexpr (CKL) {  // the cumulative sums the chunks left in LDS, each on the running maximum of its end
              for (int c = 0; c < nchk; c++) {
                const int ca = z0 + c * csz;
                if (ca >= z1) break;
                const double tc = ckc[(2 * c) * TB], mc = ckc[(2 * c + 1) * TB];
                const double cum = (tc > 0) ? tc * exp_nonpos(mc - Mx, L.tab) : 0.0;
                za = ca; zb = min(z1, ca + csz);  // the last non-empty chunk is the fallback
                if (target < before + cum) { found = true; break; }
                cacc = before + cum;
              }
            } else {
#pragma unroll
              for (int c = 0; c < NCH; c++) {
                const double shc = (cs[c] > 0) ? cs[c] * exp_nonpos(ms[c] - Mx, L.tab) : 0.0;
                const int ca = z0 + c * csz, cb_ = min(z1, ca + csz);
                if (!found && ca < cb_) {
                  za = ca; zb = cb_;  // the last non-empty chunk is the fallback
                  if (target < cacc + shc) found = true;
                  else cacc += shc;
                }
              }
            }
            choice = zb - 1;
            if (found) { pk = (za << 10) | (zb - za); cb = cacc; }
          }
          pk = group_max<HL>(pk);
          cb = group_max<HL>(cb);
          if (pk >= 0) {
            const int za = pk >> 10, len = pk & 1023, rows = (len + HL - 1) / HL;
            constexpr int NOHIT = 1 << 20;
            int hitz = NOHIT;
            double base = cb;
            for (int r = 0; r < rows; r++) {
              const int z = za + r * HL + h;
              const bool valid = z < za + len;
              double w = 0.0;
              if (valid) {
                double a, g;
                node_w(z, a, g);
                w = exp_nonpos(a - Mx, L.tab) * g;
              }
              double rowtot;
              const double inc = group_inclusive_scan<HL>(w, h, &rowtot);
              if (valid && target < base + inc) hitz = min(hitz, z);
              if (group_max<HL>(hitz < NOHIT ? 1 : 0)) break;
              base += rowtot;
            }
            const int best = -group_max<HL>(-hitz);  // the first node, in node order, whose cumulative weight passes the target
            choice = best < NOHIT ? best : za + len - 1;
          }
        } else if (live && share > 0 && target >= before && target < incl) {
          // pass 2 inside this helper's own range: find the chunk that holds `target`, rescan only it
          double cacc = before;
          int za = z0, zb = z1;
          bool found = false;
#pragma unroll
          for (int c = 0; c < NCH; c++) {
            const double shc = (cs[c] > 0) ? cs[c] * exp_nonpos(ms[c] - Mx, L.tab) : 0.0;
            const int ca = z0 + c * csz, cb = min(z1, ca + csz);
            if (!found && ca < cb) {
              za = ca; zb = cb;  // the last non-empty chunk is the fallback
              if (target < cacc + shc) found = true;
              else cacc += shc;
            }
          }
          choice = zb - 1;
          if (found) {
            double c = cacc;
            for (int z = za; z < zb; z++) {
              double a, g;
              node_w(z, a, g);
              c = fma(exp_nonpos(a - Mx, L.tab), g, c);
              if (target < c) { choice = z; break; }
            }
          }
        }
        choice = group_max<HL>(choice);
        if (choice < 0) choice = cnt - 1;  // rounding left u * total beyond the last share
        if (h == 0 && live) ind[j * SPB + sl] = choice;
        NBP_CTICK(60 + (leaf ? 2 : 0) + (XP ? 1 : 0));  // pass 2: rescan of the chosen chunk + broadcast of the choice
        };
        // (the throughput geometries instantiate the draw on the point separately: g_z and the precisions from the staging,
        //  no rsqrt per node; in the latency kernels -- five manifolds x partial x big in one kernel -- the second copy costs
        //  300 spilled registers and goes through the general form)
#ifndef NBP_X_XP_HL
#define NBP_X_XP_HL 4
#endif
        if (HL <= NBP_X_XP_HL && it < 0) draw(std::true_type{});
        else draw(std::false_type{});
      }
    }
    };
    if (l == T.L) sweep(std::true_type{});
    else sweep(std::false_type{});
    NBP_CTICK(40);
    if (live) point_moments();
    NBP_CTICK(48);  // moments of the next point
  }
  NBP_CTICK(40);
  if constexpr (FUSED) __syncthreads();  // the result slot shares LDS with statistics the slower waves still read
  // ---- samplePoint!: draw from the product of the F selected leaf kernels -----------------------
  if (h == 0 && live) {
    double res[D];
#pragma unroll
    for (int k = 0; k < D; k++) {
      res[k] = xp[k];
      if (PARTIAL && !(xpr[k] > 0))  // uninformed coordinate: oldPoints (topped up to N by the prep launch of this stage)
        res[k] = (d->old_slot >= 0) ? arena[S * d->old_slot + k * N + s] : 0.0;
    }
    if (d->labels_out >= 0)
      for (int j = 0; j < F; j++) {
        const int *widx = FUSED ? fio->idx + j * fio->idx_stride : (const int *)(wsp + (size_t)j * nbp_kd_ws_doubles(N) + 3 * N + 4);
        side[d->labels_out + s * F + j] = widx[T.node_lo[T.off[T.L] + ind[j * SPB + sl]]];
      }
    // setBelief!: the rebandwidth rides with the next nbp_prep_kernel / nbp_bandwidth_kernel launch
#pragma unroll
    for (int k = 0; k < 3; k++) out[k * N + s] = (k < D) ? res[k < D ? k : 0] : 0.0;
  }
  NBP_CTICK(42);  // final draw
}

// single density: AMP returns it unchanged
__device__ __forceinline__ void product_passthrough(const nbp_product_desc *d, double *arena, int N, int64_t S, int32_t *side) {
  if (blockIdx.y != 0) return;
  const double *src = arena + S * d->in_slot[0];
  double *out = arena + S * d->out_slot;
  const int DN = mani_dim(d->manifold) * N;  // (the rows beyond the manifold's dimension are zeros: written, not read)
  for (int i = threadIdx.x; i < 3 * N + 3; i += blockDim.x) out[i] = (i < DN || i >= 3 * N) ? src[i] : 0.0;
  // infoPerCoord of the update: the sum over its factors of ones(D) (proposalbeliefs!, ApproxConv.jl:277,298-303)
  if (threadIdx.x < 3) out[3 * N + 3 + threadIdx.x] = (threadIdx.x < mani_dim(d->manifold)) ? 1.0 : 0.0;
  if (threadIdx.x == 0) out[3 * N + 6] = src[3 * N + 6];
  if (d->labels_out >= 0)
    for (int i = threadIdx.x; i < N; i += blockDim.x) side[d->labels_out + i] = i;
}
__device__ __forceinline__ void product_write_ipc(const nbp_product_desc *d, double *arena, int N, int64_t S) {
  if (blockIdx.y == 0 && threadIdx.x < 3)
    arena[S * d->out_slot + 3 * N + 3 + threadIdx.x] = (threadIdx.x < mani_dim(d->manifold)) ? (double)d->nfactors : 0.0;
  if (blockIdx.y == 0 && threadIdx.x == 0) arena[S * d->out_slot + 3 * N + 6] = 0.0;  // N points
}


#define NBP_PRODUCT_BODY(M_, P_)                                                                              \
  do {                                                                                                        \
    if (HL >= 8 && gstats) product_body<M_, P_, HL, (HL >= 8)>(d, arena, ws, kdF, gstats, N, S, side, T, smem); \
    else product_body<M_, P_, HL, false>(d, arena, ws, kdF, gstats, N, S, side, T, smem, nullptr, all_levels, nch, true);  \
  } while (0)
template <int HL>
__device__ __forceinline__ void product_kernel_body(const nbp_product_desc *descs, double *arena, const double *ws, int kdF,
                                                    double *gstats, int N, int64_t S, int32_t *side, const nbp_levels &T, double *smem) {
  const nbp_product_desc *d = descs + blockIdx.x;
  const bool all_levels = (kdF & NBP_PROD_ALL_LEVELS) != 0;
  const int nch = NBP_PROD_NCH(kdF);
  kdF &= 0xFFFF;
  if (d->nfactors == 1) { product_passthrough(d, arena, N, S, side); return; }
  product_write_ipc(d, arena, N, S);
  bool partial = false;
  for (int j = 0; j < d->nfactors; j++) partial |= (d->in_partial[j] != 0);
  if (partial) {  // validated on the host: D >= 2
    switch (d->manifold) {
    case NBP_EUCLID2: NBP_PRODUCT_BODY(NBP_EUCLID2, true); break;
    case NBP_EUCLID3: NBP_PRODUCT_BODY(NBP_EUCLID3, true); break;
    default: NBP_PRODUCT_BODY(NBP_SE2, true); break;
    }
    return;
  }
  switch (d->manifold) {
  case NBP_EUCLID1: NBP_PRODUCT_BODY(NBP_EUCLID1, false); break;
  case NBP_EUCLID2: NBP_PRODUCT_BODY(NBP_EUCLID2, false); break;
  case NBP_EUCLID3: NBP_PRODUCT_BODY(NBP_EUCLID3, false); break;
  case NBP_CIRCULAR: NBP_PRODUCT_BODY(NBP_CIRCULAR, false); break;
  default: NBP_PRODUCT_BODY(NBP_SE2, false); break;
  }
}

// A batch whose multi-density products all live on ONE manifold and have no partial inputs (every stage of a
// homogeneous graph) runs a kernel that holds that one instantiation: the register allocation of the throughput
// variants is then the body's own need instead of the maximum over all manifolds (scratch per lane at 4 waves per
// SIMD: generic 308-328 B, Euclid(1) 0, Euclid(2) 64 B, Euclid(3) 52 B).
// XS: the workgroup stages the sorted, centred coordinates of its F <= NBP_FUSED_MAXF densities in LDS (F x D x N doubles
// behind the product's own areas) and takes the node sums of every level from there -- the same leaf-order sums kd_build
// would have left in the workspace -- so that a KD workspace carries 4 KB per density (coordinates, centres, permutation)
// through HBM instead of 33 KB (the 2 x D x ~2N node sums written by the prep launch and read back here).
__host__ __device__ inline size_t nbp_product_xs_doubles(int F, int D, int N) { return (size_t)F * D * N + 6 * (size_t)F; }
template <int MANI, int HL, bool XS>
__device__ __forceinline__ void product_kernel_uniform(const nbp_product_desc *descs, double *arena, const double *ws, int kdF,
                                                       double *gstats, int N, int64_t S, int32_t *side, const nbp_levels &T, double *smem) {
  const nbp_product_desc *d = descs + blockIdx.x;
  const bool all_levels = (kdF & NBP_PROD_ALL_LEVELS) != 0;
  const int nch = NBP_PROD_NCH(kdF);
  constexpr bool lay_circ = (MANI == NBP_CIRCULAR || MANI == NBP_SE2);  // (the host lays the LDS out the same way: launch_products)
  kdF &= 0xFFFF;
  if (d->nfactors == 1) { product_passthrough(d, arena, N, S, side); return; }
  product_write_ipc(d, arena, N, S);
  if constexpr (XS) {
    constexpr int D = (MANI == NBP_SE2) ? 3 : (MANI == NBP_CIRCULAR ? 1 : MANI);
    const int F = d->nfactors, TB = blockDim.x, tid = threadIdx.x;
    nbp_fused_io fio;
    const size_t own = (product_lds_layout(F, D, N, TB / HL, false, smem, &fio.L, all_levels ? T.off[T.L] + T.cnt[T.L] : N,
                                           HL <= 4 ? (size_t)nch * 2 * TB : 0, lay_circ) + 7) / 8;
    double *xs = smem + own, *cen = xs + (size_t)F * D * N, *bw = cen + 3 * F;
    const size_t wsd = nbp_kd_ws_doubles(N);
    const double *wsp = ws + (size_t)blockIdx.x * kdF * wsd;
    for (int t = tid; t < F * D * N; t += TB) {
      const int j = t / (D * N), r = t - j * (D * N);
      xs[t] = wsp[(size_t)j * wsd + r];
    }
    for (int t = tid; t < F * 3; t += TB) {
      const int j = t / 3, k = t % 3;
      cen[t] = wsp[(size_t)j * wsd + 3 * N + k];
      bw[t] = arena[S * d->in_slot[j] + 3 * N + k];
    }
    fio.xs = xs;
    fio.xs_stride = (size_t)D * N;
    fio.idx = (const int *)(wsp + 3 * N + 4);
    fio.idx_stride = wsd * 2;  // in ints
    fio.cen = cen;
    fio.bw = bw;
    fio.out = arena + S * d->out_slot;
    __syncthreads();
    product_body<MANI, false, HL, false, 2>(d, arena, ws, kdF, gstats, N, S, side, T, smem, &fio, false, nch, lay_circ);
  } else
    product_body<MANI, false, HL, false>(d, arena, ws, kdF, gstats, N, S, side, T, smem, nullptr, all_levels, nch, lay_circ);  // HL = 4 / 2: never BIG (launch_products)
}

// Entry points: the latency variants (HL = 32 for fewer than 16 products, 16 on request, HL = 8; few workgroups in flight) and the
// throughput variants (one workgroup per product, per-manifold instances); none of them uses scratch
// (profiles/r03_kernel_resources.txt).
#define NBP_PRODUCT_ARGS const nbp_product_desc *descs, double *arena, const double *ws, int kdF, double *gstats, int N, int64_t S, int32_t *side, nbp_levels T
// Latency geometries (8 / 16 / 32 helper lanes per sample, workgroups of at most four waves).  Each in two instances: the
// plain one (launch bound 512: two waves per SIMD, 256 registers, 16 B of scratch per lane since the uniforms of a pass are made
// once per sample and handed over through LDS) and `_w1` for launches with at most one workgroup per CU (launch bound 256: one
// wave per SIMD may hold 512 registers, the compiler parks four values in accumulation registers: no scratch).  A lone product of
// 2 / 5 / 8 densities: 82.8 / 196.8 / 325.3 -> 78.4 / 178.6 / 289.4 us; 66 products (462 workgroups, the plain instance) 168.9 ->
// 162.4 us -- at one wave per SIMD they take 234 us (profiles/r05_latency_product_uniforms.txt)
#if NBP_TU & NBP_TU_PRODLAT
#define NBP_PRODUCT_LATENCY(NAME, HL_, BOUNDS)                                                \
  __global__ void __launch_bounds__(BOUNDS) NAME(NBP_PRODUCT_ARGS) {                          \
    extern __shared__ double smem[];                                                          \
    product_kernel_body<HL_>(descs, arena, ws, kdF, gstats, N, S, side, T, smem);             \
  }
#else
#define NBP_PRODUCT_LATENCY(NAME, HL_, BOUNDS) __global__ void NAME(NBP_PRODUCT_ARGS);
#endif
NBP_PRODUCT_LATENCY(nbp_product_kernel_x16, 16, 512)
NBP_PRODUCT_LATENCY(nbp_product_kernel_l8, 8, 512)
// fewer than 16 products alone on the chip: 32 helper lanes per sample halve the node range of every lane once more
// (a lone F = 2 product: 87 -> 77 us; 64 lanes per sample gain nothing more and cost twelve F = 3 products 209 instead of 109 us)
NBP_PRODUCT_LATENCY(nbp_product_kernel_y32, 32, 512)
NBP_PRODUCT_LATENCY(nbp_product_kernel_x16_w1, 16, 256)
NBP_PRODUCT_LATENCY(nbp_product_kernel_l8_w1, 8, 256)
NBP_PRODUCT_LATENCY(nbp_product_kernel_y32_w1, 32, 256)
#if NBP_TU & NBP_TU_PRODTHR
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) nbp_product_kernel_m4(NBP_PRODUCT_ARGS) {
  extern __shared__ double smem[];
  product_kernel_body<4>(descs, arena, ws, kdF, gstats, N, S, side, T, smem);
}
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) nbp_product_kernel_t2(NBP_PRODUCT_ARGS) {
  extern __shared__ double smem[];
  product_kernel_body<2>(descs, arena, ws, kdF, gstats, N, S, side, T, smem);
}
#else
__global__ void nbp_product_kernel_m4(NBP_PRODUCT_ARGS);
__global__ void nbp_product_kernel_t2(NBP_PRODUCT_ARGS);
#endif
// waves per SIMD of the single-manifold kernels, measured (profiles/r02_product_waves.txt): Euclid(1/2) are faster at 4
// (Euclid(2) then spills 76 B per lane; at 3 it is spill-free but 16 % slower on the 10 000-variable chain), Euclid(3),
// Circular and SE(2) at 3 (140-168 VGPRs: no scratch on Euclid(3), 36 / 120 B on the circular ones; 1-3 % faster)
#ifndef NBP_W_E2
#define NBP_W_E2 4
#endif
#ifndef NBP_W_SE
#define NBP_W_SE 2
#endif
#ifndef NBP_W_CI
#define NBP_W_CI 2  // (two waves: no spill -- three spilled 12 B per lane -- and 2 % faster on config 3)
#endif
#define NBP_PRODUCT_UNIFORM(NAME, MANI, HL) NBP_PRODUCT_UNIFORM_W(NAME, MANI, HL, ((MANI) == NBP_EUCLID1 ? 4 : (MANI) == NBP_EUCLID2 ? NBP_W_E2 : (MANI) == NBP_SE2 ? NBP_W_SE : (MANI) == NBP_CIRCULAR ? NBP_W_CI : 3))
#define NBP_PRODUCT_UNIFORM_DECL(NAME) __global__ void NAME(NBP_PRODUCT_ARGS); __global__ void NAME##_xs(NBP_PRODUCT_ARGS);
#if NBP_TU & (NBP_TU_PRODUNI | NBP_TU_PRODUNI4)
#define NBP_PRODUCT_UNIFORM_W(NAME, MANI, HL, NBP_UNIFORM_WAVES) NBP_PRODUCT_UNIFORM_W##HL(NAME, MANI, HL, NBP_UNIFORM_WAVES)
#define NBP_PRODUCT_UNIFORM_DEF(NAME, MANI, HL, NBP_UNIFORM_WAVES)                                                                        \
  __global__ void __launch_bounds__(NBP_PROD_LB(MANI)) __attribute__((amdgpu_waves_per_eu(NBP_UNIFORM_WAVES))) NAME(NBP_PRODUCT_ARGS) { \
    extern __shared__ double smem[];                                                                               \
    product_kernel_uniform<MANI, HL, false>(descs, arena, ws, kdF, gstats, N, S, side, T, smem);                    \
  }                                                                                                                \
  __global__ void __launch_bounds__(NBP_PROD_LB(MANI)) __attribute__((amdgpu_waves_per_eu(NBP_UNIFORM_WAVES))) NAME##_xs(NBP_PRODUCT_ARGS) { \
    extern __shared__ double smem[];                                                                               \
    product_kernel_uniform<MANI, HL, true>(descs, arena, ws, kdF, gstats, N, S, side, T, smem);                     \
  }
#if NBP_TU & NBP_TU_PRODUNI
#define NBP_PRODUCT_UNIFORM_W2(NAME, MANI, HL, W) NBP_PRODUCT_UNIFORM_DEF(NAME, MANI, HL, W)
#else
#define NBP_PRODUCT_UNIFORM_W2(NAME, MANI, HL, W) NBP_PRODUCT_UNIFORM_DECL(NAME)
#endif
#if NBP_TU & NBP_TU_PRODUNI4
#define NBP_PRODUCT_UNIFORM_W4(NAME, MANI, HL, W) NBP_PRODUCT_UNIFORM_DEF(NAME, MANI, HL, W)
#else
#define NBP_PRODUCT_UNIFORM_W4(NAME, MANI, HL, W) NBP_PRODUCT_UNIFORM_DECL(NAME)
#endif
#else
#define NBP_PRODUCT_UNIFORM_W(NAME, MANI, HL, NBP_UNIFORM_WAVES) NBP_PRODUCT_UNIFORM_DECL(NAME)
#endif
NBP_PRODUCT_UNIFORM(nbp_product_kernel_t2_e1, NBP_EUCLID1, 2)
NBP_PRODUCT_UNIFORM(nbp_product_kernel_t2_e2, NBP_EUCLID2, 2)
NBP_PRODUCT_UNIFORM(nbp_product_kernel_t2_e3, NBP_EUCLID3, 2)
NBP_PRODUCT_UNIFORM(nbp_product_kernel_t2_ci, NBP_CIRCULAR, 2)
NBP_PRODUCT_UNIFORM(nbp_product_kernel_t2_se, NBP_SE2, 2)
NBP_PRODUCT_UNIFORM(nbp_product_kernel_m4_e1, NBP_EUCLID1, 4)
NBP_PRODUCT_UNIFORM(nbp_product_kernel_m4_e2, NBP_EUCLID2, 4)
NBP_PRODUCT_UNIFORM(nbp_product_kernel_m4_e3, NBP_EUCLID3, 4)
NBP_PRODUCT_UNIFORM(nbp_product_kernel_m4_ci, NBP_CIRCULAR, 4)
NBP_PRODUCT_UNIFORM(nbp_product_kernel_m4_se, NBP_SE2, 4)

static inline size_t nbp_product_lds_bytes(int F, int D, int N, int SPB, bool big, size_t CK = 0, bool circ = true) {
  return product_lds_layout(F, D, N, SPB, big, nullptr, nullptr, 0, CK, circ);
}
