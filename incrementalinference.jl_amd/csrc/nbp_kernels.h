// nbp_kernels.h -- the gfx950 kernels of libnbp (see nbp_device.h for the shared device code).
#pragma once
#include "nbp_device.h"

// ================================================================================================
// Proposal kernel: one workgroup = one approxConvBelief (ApproxConv.jl:4-45)
//   evalFactor -> evalPotentialSpecific (EvalFactor.jl:321-395 relative, :400-542 prior)
//   -> manikde! bandwidth.
// LDS: X[3][N] target scratch (the deepcopy of CalcFactor.jl:543-548), Z[3][N] measurements,
//      mhidx[N], reduction scratch.  HBM traffic: reads the operand beliefs once (coalesced,
//      8 B/lane), writes the proposal once.
// ================================================================================================
__device__ void sample_measurement(const nbp_proposal_desc *d, int n, int zdim, double *z) {
  int c = 0;
  if (d->ncomp > 1) {  // Mixture.sampleFactor, Factors/Mixture.jl:114-155
    double w[NBP_MAXC], ua, ub;
    for (int i = 0; i < NBP_MAXC; i++) w[i] = (i < d->ncomp) ? d->comp[i][0] : 0.0;
    uniform_pair(d->seed, n, PURP_MIXLBL, 0, ua, ub);
    c = categorical(w, d->ncomp, ua);
  }
  const double *cp = d->comp[c];
  double nn[4] = {0, 0, 0, 0};
  normal_pair(d->seed, n, PURP_MEAS, 0, nn[0], nn[1]);
  if (zdim > 2) normal_pair(d->seed, n, PURP_MEAS, 1, nn[2], nn[3]);
  for (int i = 0; i < 3; i++) {
    double acc = 0;
    if (i < zdim) {
      acc = cp[1 + i];
      for (int j = 0; j <= i; j++) acc += cp[4 + i * 3 + j] * nn[j];
    }
    z[i] = acc;
  }
}

// addEntropyOnManifold!, EvalFactor.jl:95-132
__device__ __forceinline__ void add_entropy(int manifold, int D, double *x, int n, double spread, uint64_t seed, int kbase) {
  double u[4] = {0, 0, 0, 0};
  uniform_pair(seed, n, PURP_ENTROPY, kbase, u[0], u[1]);
  if (D > 2) uniform_pair(seed, n, PURP_ENTROPY, kbase + 1, u[2], u[3]);
  for (int k = 0; k < D; k++) {
    double v = x[k] + spread * (u[k] - 0.5);
    x[k] = is_circ(manifold, k) ? wrap_pi(v) : v;
  }
}

// calcVariableDistanceExpectedFractional, EvalFactor.jl:40-92 (block-uniform result)
__device__ double var_distance_expected_fractional(const nbp_proposal_desc *d, const recipe_t *R, const double *arena,
                                                   int64_t S, int N, const double *X, double kappa, double *red) {
  const int sf1 = d->sfidx + 1, M = d->manifold, D = mani_dim(M);
  if (in_list(R->certain, R->ncertain, sf1)) return kappa * std_basic_spread(X, N, N, M, red);
  double ref[3] = {0, 0, 0};
  for (int k = 0; k < D; k++) ref[k] = mean_default_coord(X + k * N, N, M, k, red);
  double best = 1e-2;
  for (int i = 1; i <= d->nvars; i++) {
    const double *pts = (i == sf1) ? X : arena + S * d->var_slot[i - 1];
    const bool cer = in_list(R->certain, R->ncertain, i);
    double acc = 0;
    for (int k = 0; k < D; k++) {
      double mu = cer ? mean_geodesic_coord(pts + k * N, N, M, k, red) : mean_default_coord(pts + k * N, N, M, k, red);
      acc += (ref[k] - mu) * (ref[k] - mu);
    }
    best = fmax(best, sqrt(acc));
  }
  return kappa * best;
}

__global__ void nbp_proposal_kernel(const nbp_proposal_desc *descs, double *arena, int N, int64_t S, int32_t *side,
                                    nbp_counters *ctr) {
  extern __shared__ double smem[];
  double *X = smem;          // [3][N]
  double *Z = X + 3 * N;     // [3][N]
  double *red = Z + 3 * N;   // [16]
  int *mh = (int *)(red + 16);
  __shared__ recipe_t R;
  const nbp_proposal_desc *d = descs + blockIdx.x;
  const int n = threadIdx.x, M = d->manifold, D = mani_dim(M), kind = d->factor_kind;
  const bool live = n < N;
  double *out = arena + S * d->out_slot;
  unsigned int n_solves = 0, n_nonconv = 0, n_nan = 0, n_evals = 0;

  if (n == 0) build_recipe(d, &R);
  {
    const double *src = arena + S * d->var_slot[(kind == NBP_F_PRIOR || kind == NBP_F_MSGPRIOR) ? 0 : d->sfidx];
    if (live)
      for (int k = 0; k < 3; k++) X[k * N + n] = src[k * N + n];
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

  if (kind == NBP_F_PRIOR || kind == NBP_F_MSGPRIOR) {
    // evalPotentialSpecific(prior), EvalFactor.jl:400-542
    const double spread = d->spread_nh * std_basic_spread(X, N, N, M, red);  // :464, before the overwrite
    __syncthreads();
    if (live) {
      double x[3] = {X[n], X[N + n], X[2 * N + n]};
      if (mh[n] == 1) {
        if (kind == NBP_F_PRIOR) {
          double z[3];
          sample_measurement(d, n, D, z);
          for (int k = 0; k < D; k++) x[k] = is_circ(M, k) ? wrap_pi(z[k]) : z[k];
        } else {  // MsgPrior{MKD}: sample(belief): random kernel + bw*randn (Factors/MsgPrior.jl:27-30)
          const double *msg = arena + S * d->var_slot[1];
          double ua, ub, nn[4] = {0, 0, 0, 0};
          uniform_pair(d->seed, n, PURP_KDESEL, 0, ua, ub);
          int i = (int)(ua * N);
          if (i >= N) i = N - 1;
          normal_pair(d->seed, n, PURP_KDENOISE, 0, nn[0], nn[1]);
          if (D > 2) normal_pair(d->seed, n, PURP_KDENOISE, 1, nn[2], nn[3]);
          for (int k = 0; k < D; k++) {
            double v = msg[k * N + i] + msg[3 * N + k] * nn[k];
            x[k] = is_circ(M, k) ? wrap_pi(v) : v;
          }
        }
      } else {
        add_entropy(M, D, x, n, spread, d->seed, 0);  // :476
      }
      for (int k = 0; k < 3; k++) X[k * N + n] = x[k];
    }
    __syncthreads();
  } else {
    // evalPotentialSpecific(relative), EvalFactor.jl:321-395
    const int zdim = (kind == NBP_F_LINREL) ? D : (kind == NBP_F_SE2 ? 3 : 1);
    double z[3] = {0, 0, 0};
    if (live) sample_measurement(d, n, zdim, z);  // sampleFactor!, CalcFactor.jl:578
    (void)Z;
    const int sf1 = d->sfidx + 1;
    const int myh = live ? mh[n] : -1000;
    // computeAcrossHypothesis!, EvalFactor.jl:145-237
    for (int g = 0; g < R.ngroups; g++) {
      if (R.empty[g]) continue;
      const int hyp = R.hypo[g];
      if (!__syncthreads_or(myh == hyp)) continue;  // empty allelements[g]: nothing to do
      const bool solve_case = (in_list(R.certain, R.ncertain, sf1) && hyp != 0) || in_list(R.certain, R.ncertain, hyp) || hyp == sf1;
      if (solve_case) {
        const int va = R.act[g][0], vb = R.act[g][1];
        const int solve_b = (vb == sf1);
        const int vother = solve_b ? va : vb;
        const double *O = arena + S * d->var_slot[vother - 1];
        double oth[3] = {0, 0, 0};
        if (myh == hyp)
          for (int k = 0; k < D; k++) oth[k] = O[k * N + n];
        for (int c = 0; c < d->inflate_cycles; c++) {  // :184-207
          const double spread = var_distance_expected_fractional(d, &R, arena, S, N, X, d->inflation, red);
          __syncthreads();
          if (myh == hyp) {
            double x[3] = {X[n], X[N + n], X[2 * N + n]};
            add_entropy(M, D, x, n, spread, d->seed, (g * 8 + c) * 2);
            solve_particle(kind, M, z, oth, solve_b, x, n_solves, n_nonconv, n_nan, n_evals);  // approxConvOnElements!
            for (int k = 0; k < D; k++) X[k * N + n] = x[k];
          }
          __syncthreads();
        }
      } else {  // other-hypothesis (:208-220) / nullhypo (:222-231): entropy only
        const double spread = var_distance_expected_fractional(d, &R, arena, S, N, X, d->spread_nh, red);
        __syncthreads();
        if (myh == hyp) {
          double x[3] = {X[n], X[N + n], X[2 * N + n]};
          add_entropy(M, D, x, n, spread, d->seed, (g * 8) * 2);
          for (int k = 0; k < D; k++) X[k * N + n] = x[k];
        }
        __syncthreads();
      }
    }
  }
  // manikde!(M, pts): bandwidth per coordinate (ApproxConv.jl:36-42)
  if (!d->skip_bandwidth) {
    for (int k = 0; k < D; k++) {
      double h = lcv_bandwidth_1d(X + k * N, N, is_circ(M, k), red);
      if (n == 0) out[3 * N + k] = h;
    }
    if (n == 0)
      for (int k = D; k < 3; k++) out[3 * N + k] = 0.0;
  }
  if (live)
    for (int k = 0; k < 3; k++) out[k * N + n] = (k < D) ? X[k * N + n] : 0.0;
  // diagnostics: one atomic per wave
  {
    unsigned int v[4] = {n_solves, n_nonconv, n_nan, n_evals};
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

// ================================================================================================
// Bandwidth kernel: AMP.manikde!(M, pts) for a resident slot
// ================================================================================================
__global__ void nbp_bandwidth_kernel(const int32_t *slots, const int32_t *manifolds, double *arena, int N, int64_t S) {
  extern __shared__ double smem[];
  double *X = smem, *red = smem + 3 * N;
  double *s = arena + S * slots[blockIdx.x];
  const int M = manifolds[blockIdx.x], D = mani_dim(M), n = threadIdx.x;
  if (n < N)
    for (int k = 0; k < 3; k++) X[k * N + n] = s[k * N + n];
  __syncthreads();
  for (int k = 0; k < D; k++) {
    double h = lcv_bandwidth_1d(X + k * N, N, is_circ(M, k), red);
    if (n == 0) s[3 * N + k] = h;
  }
}

__global__ void nbp_copy_kernel(const nbp_copy_desc *c, double *arena, int64_t S) {
  const double *src = arena + S * c[blockIdx.x].src_slot;
  double *dst = arena + S * c[blockIdx.x].dst_slot;
  for (int64_t i = threadIdx.x; i < S; i += blockDim.x) dst[i] = src[i];
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__global__ void nbp_reseed_proposals(nbp_proposal_desc *d, int n, uint64_t salt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i].seed = splitmix64(d[i].seed ^ splitmix64(salt));
}
__global__ void nbp_reseed_products(nbp_product_desc *d, int n, uint64_t salt) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i].seed = splitmix64(d[i].seed ^ splitmix64(salt));
}

// ================================================================================================
// Product kernel: one workgroup = one AMP.manifoldProduct(dens, M; Niter, N) + rebandwidth +
// setBelief!  (GraphProductOperations.jl:53-60, SolveTree.jl:74).
//
// Algorithm (Ihler et al. NIPS 2003, `prodAppxMSGibbsS`): every input KDE gets a balanced KD-tree
// whose nodes carry the moment-matched Gaussian of their leaves; all N output samples walk the
// trees root->leaves in lock step; at every level each sample runs `niter` sequential Gibbs sweeps
// re-drawing its label in density j from p(l_j | others) over ALL nodes of that level; at the
// leaves the sample is drawn from the product of the F selected kernels.
//
// Mapping: lane s = output sample s.  The KD permutation is built by rank counting inside each
// segment (no barriers inside a level); node statistics of the *current* level for all densities
// live in LDS (lm/lv) and are read with wave-uniform addresses (LDS broadcast) in the O(N) label
// loop; per-sample labels live in LDS (ind).
// ================================================================================================
template <int D>
__device__ void product_body(const nbp_product_desc *d, double *arena, int N, int64_t S, int32_t *side,
                             const nbp_levels &T, double *smem) {
  const int F = d->nfactors, M = d->manifold, tid = threadIdx.x, TB = blockDim.x;
  double *xs = smem;                 // [F][D][N] sorted, centred
  double *lm = xs + F * D * N;       // [F][D][N] node mean of the current level
  double *lv = lm + F * D * N;       // [F][D][N] node variance (+ bandwidth^2)
  double *cen = lv + F * D * N;      // [F][3]
  double *h2 = cen + F * 3;          // [F][3]
  double *red = h2 + F * 3;          // [16]
  int *idx = (int *)(red + 16);      // [F][N]
  int *ind = idx + F * N;            // [F][TB]
  int *tmpA = ind + F * TB;          // [N]
  int *tmpB = tmpA + N;              // [N]
  double *out = arena + S * d->out_slot;
  bool circ[D];
#pragma unroll
  for (int k = 0; k < D; k++) circ[k] = is_circ(M, k);

  // ---- KD-tree permutation per density ----------------------------------------------------
  for (int j = 0; j < F; j++) {
    const double *x = arena + S * d->in_slot[j];
    double *raw = lm + j * D * N;  // temporary home of the unsorted coordinates
    if (tid < N) {
#pragma unroll
      for (int k = 0; k < D; k++) raw[k * N + tid] = x[k * N + tid];
      tmpA[tid] = tid;
    }
    if (tid < 3) h2[j * 3 + tid] = (tid < D) ? x[3 * N + tid] * x[3 * N + tid] : 0.0;
    __syncthreads();
    int *pa = tmpA, *pb = tmpB;
    for (int l = 0; l < T.L; l++) {
      if (tid < N) {
        const int node = T.pos_node[l * N + tid];
        const int lo = T.node_lo[T.off[l] + node], hi = T.node_hi[T.off[l] + node];
        const int me = pa[tid];
        if (hi - lo > 1) {
          int best = 0;
          if (D > 1) {  // widest coordinate of the segment
            double bext = -1.0;
#pragma unroll
            for (int k = 0; k < D; k++) {
              double mn = INFINITY, mx = -INFINITY;
              for (int p = lo; p < hi; p++) {
                double v = raw[k * N + pa[p]];
                mn = fmin(mn, v);
                mx = fmax(mx, v);
              }
              if (mx - mn > bext) { bext = mx - mn; best = k; }
            }
          }
          const double v = raw[best * N + me];
          int rank = 0;
          for (int p = lo; p < hi; p++) {
            const int ip = pa[p];
            const double vp = raw[best * N + ip];
            rank += (vp < v || (vp == v && ip < me)) ? 1 : 0;
          }
          pb[lo + rank] = me;
        } else
          pb[tid] = me;
      }
      __syncthreads();
      int *t = pa; pa = pb; pb = t;
    }
#pragma unroll
    for (int k = 0; k < D; k++) {
      double c = block_sum(tid < N ? raw[k * N + tid] : 0.0, red) / (double)N;
      if (tid == 0) cen[j * 3 + k] = c;
      if (tid < N) xs[(j * D + k) * N + tid] = raw[k * N + pa[tid]] - c;
    }
    if (tid < N) idx[j * N + tid] = pa[tid];
    __syncthreads();
  }

  // ---- multiscale Gibbs ---------------------------------------------------------------------
  for (int j = 0; j < F; j++) ind[j * TB + tid] = 0;  // levelInit!: root
  for (int l = 1; l <= T.L; l++) {
    const int cnt = T.cnt[l], off = T.off[l];
    __syncthreads();
    for (int item = tid; item < F * D * cnt; item += TB) {  // node statistics of this level
      const int z = item % cnt, jk = item / cnt;
      const int lo = T.node_lo[off + z], hi = T.node_hi[off + z];
      const double *xv = xs + jk * N;
      double s1 = 0, s2 = 0;
      for (int p = lo; p < hi; p++) { double v = xv[p]; s1 += v; s2 += v * v; }
      const double nn = (double)(hi - lo), mu = s1 / nn;
      double var = s2 / nn - mu * mu;
      if (var < 0) var = 0;
      const int j = jk / D, k = jk % D;
      lm[jk * N + z] = cen[j * 3 + k] + mu;
      lv[jk * N + z] = var + h2[j * 3 + k];
    }
    __syncthreads();
    if (tid < N) {
      for (int j = 0; j < F; j++) ind[j * TB + tid] = T.node_child[T.off[l - 1] + ind[j * TB + tid]];  // levelDown!
      for (int it = 0; it < d->niter; it++) {
        for (int j = 0; j < F; j++) {  // sampleIndex(j)
          double mn[D], vn[D];
#pragma unroll
          for (int k = 0; k < D; k++) {
            double prec = 0, acc = 0, ss = 0, sc = 0;
            for (int q = 0; q < F; q++) {
              if (q == j) continue;
              const int iq = ind[q * TB + tid];
              const double mq = lm[(q * D + k) * N + iq], vq = lv[(q * D + k) * N + iq];
              prec += 1.0 / vq;
              if (circ[k]) {
                double sn, cs;
                sincos(mq, &sn, &cs);
                ss += sn / vq;
                sc += cs / vq;
              } else
                acc += mq / vq;
            }
            vn[k] = 1.0 / prec;
            mn[k] = circ[k] ? atan2(ss, sc) : acc * vn[k];
          }
          double ua, ub;
          uniform_pair(d->seed, tid, PURP_PGIBBS, (uint32_t)((l * 8 + it) * NBP_MAXF + j), ua, ub);
          // rand(Categorical(p)) by inverse CDF over the nodes of this level.  Pass 1: running
          // max + rescaled total; pass 2: cumulative sum until u*total (weights are recomputed,
          // not stored: N doubles per lane would not fit in registers or LDS).
          const double *mj = lm + j * D * N, *vj = lv + j * D * N;
          auto node_e = [&](int z) -> double {
            double e = 0;
#pragma unroll
            for (int k = 0; k < D; k++) {
              double tmp = mj[k * N + z] - mn[k];
              if (circ[k]) tmp = wrap_pi(tmp);
              const double v = vj[k * N + z] + vn[k];
              e += tmp * tmp / v + log(v);
            }
            return -0.5 * e + T.node_logw[off + z];
          };
          double m = -INFINITY, tot = 0;
          for (int z = 0; z < cnt; z++) {
            const double e = node_e(z);
            if (e > m) { tot = (tot > 0) ? tot * exp(m - e) : 0.0; m = e; }
            tot += exp(e - m);
          }
          const double target = ua * tot;
          double c = 0;
          int choice = -1;
          for (int z = 0; z < cnt; z++) {
            c += exp(node_e(z) - m);
            if (target < c) { choice = z; break; }
          }
          if (choice < 0) choice = cnt - 1;
          if (choice >= 0) ind[j * TB + tid] = choice;
        }
      }
    }
  }
  // ---- samplePoint!: draw from the product of the F selected leaf kernels -----------------------
  double res[D];
  if (tid < N) {
    double nn[4] = {0, 0, 0, 0};
    normal_pair(d->seed, tid, PURP_PFINAL, 0, nn[0], nn[1]);
    if (D > 2) normal_pair(d->seed, tid, PURP_PFINAL, 1, nn[2], nn[3]);
#pragma unroll
    for (int k = 0; k < D; k++) {
      double prec = 0, acc = 0, ss = 0, sc = 0;
      for (int q = 0; q < F; q++) {
        const int iq = ind[q * TB + tid];
        const double mq = lm[(q * D + k) * N + iq], vq = lv[(q * D + k) * N + iq];
        prec += 1.0 / vq;
        if (circ[k]) {
          double sn, cs;
          sincos(mq, &sn, &cs);
          ss += sn / vq;
          sc += cs / vq;
        } else
          acc += mq / vq;
      }
      const double mu = circ[k] ? atan2(ss, sc) : acc / prec;
      const double v = mu + sqrt(1.0 / prec) * nn[k];
      res[k] = circ[k] ? wrap_pi(v) : v;
    }
    if (d->labels_out >= 0)
      for (int j = 0; j < F; j++)
        side[d->labels_out + tid * F + j] = idx[j * N + T.node_lo[T.off[T.L] + ind[j * TB + tid]]];
  }
  __syncthreads();
  double *R = xs;  // reuse: [D][N] result
  if (tid < N) {
#pragma unroll
    for (int k = 0; k < D; k++) R[k * N + tid] = res[k];
  }
  __syncthreads();
  // rebandwidth + setBelief!
#pragma unroll
  for (int k = 0; k < D; k++) {
    double h = lcv_bandwidth_1d(R + k * N, N, circ[k], red);
    if (tid == 0) out[3 * N + k] = h;
  }
  if (tid == 0)
    for (int k = D; k < 3; k++) out[3 * N + k] = 0.0;
  if (tid < N)
    for (int k = 0; k < 3; k++) out[k * N + tid] = (k < D) ? R[k * N + tid] : 0.0;
}

__global__ void nbp_product_kernel(const nbp_product_desc *descs, double *arena, int N, int64_t S, int32_t *side,
                                   nbp_levels T) {
  extern __shared__ double smem[];
  const nbp_product_desc *d = descs + blockIdx.x;
  if (d->nfactors == 1) {  // single density: AMP returns it unchanged
    const double *src = arena + S * d->in_slot[0];
    double *out = arena + S * d->out_slot;
    for (int i = threadIdx.x; i < 3 * N + 3; i += blockDim.x) out[i] = src[i];
    if (d->labels_out >= 0 && threadIdx.x < N) side[d->labels_out + threadIdx.x] = threadIdx.x;
    return;
  }
  switch (mani_dim(d->manifold)) {
  case 1: product_body<1>(d, arena, N, S, side, T, smem); break;
  case 2: product_body<2>(d, arena, N, S, side, T, smem); break;
  default: product_body<3>(d, arena, N, S, side, T, smem); break;
  }
}

// LDS bytes of the product kernel for (F, D, N, threads)
static inline size_t nbp_product_lds_bytes(int F, int D, int N, int TB) {
  size_t dbl = (size_t)3 * F * D * N + 6 * F + 16;
  size_t ints = (size_t)F * N + (size_t)F * TB + 2 * N;
  return dbl * 8 + ints * 4;
}
static inline size_t nbp_proposal_lds_bytes(int N) { return ((size_t)6 * N + 16) * 8 + (size_t)N * 4; }
