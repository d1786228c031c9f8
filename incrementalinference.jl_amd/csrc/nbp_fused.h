// nbp_fused.h -- the fused variable-update kernel: one workgroup = one `propagateBelief` (GraphProductOperations.jl:16-64)
//
//   proposalbeliefs! (the F approxConvBelief calls of the update, ApproxConv.jl:238-304)  -> F proposals in LDS
//   manikde! of every proposal (ApproxConv.jl:36-42)                                      -> their bandwidths in LDS
//   AMP.manifoldProduct(dens; Niter, N) (GraphProductOperations.jl:53-60): KD trees in LDS, multiscale Gibbs, final draw
//   setBelief! (SolveTree.jl:74): manikde! of the result, ONE write of the new belief to its slot
//
// The three-launch form of an update (nbp_proposal_kernel -> nbp_prep_kernel -> nbp_product_kernel) passes the proposals
// (F x 4.9 KB at N = 200) and the KD workspaces (35 KB per density) through HBM; here nothing but the operand beliefs
// (read once, coalesced) and the new belief (written once) touches memory: the algorithmic bytes of SURVEY 8(d).  The
// phases are the SAME device functions the three kernels run (proposal_body, lcv_slot_coordinate, kd_build,
// product_body), with the same random streams, in the one-lane-per-particle geometry (workgroup = Npad lanes): the
// particles and bandwidths are those of the three-launch form at its throughput geometry, bit for bit
// (tests/test_gpu_fused_update.py).  Used for the stages that fill the chip (nbp_api.hip: fused_plan); the launches that
// cannot fill it keep the three-launch form, whose latency geometries spread ONE update over many CUs.
//
// LDS of a workgroup (Fmax = largest F of the launch, SL = 3N + 8 doubles = one slot):
//   tab[256] | slot[Fmax][SL] | bw[Fmax][3] | cen[Fmax][3] | transient area
// slot[j] holds proposal j in slot layout (points SoA, bandwidth, infoPerCoord, count), later -- in place -- its sorted,
// centred coordinates (the KD tree), and slot[0] finally the product's points (the trees are dead once the leaf level's
// node statistics are taken).  The transient area is, in turn, the proposal's scratch, the fit's accumulators, the KD
// build's rank tables and the product's per-level node statistics.
#pragma once
#include "nbp_kernels.h"

// one variable update of a fused stage (built by nbp_program_finalize)
struct nbp_update_desc {
  int32_t prod;                  // index of the product descriptor in its stage
  int32_t prop[NBP_FUSED_MAXF];  // indices of its proposals in their stage, in the product's input order
  int32_t flags;                 // bit 0: fit the bandwidth of the result; bit 1 + j: proposal j also goes to its arena slot
  int32_t pad_[2];
};
#define NBP_UPD_FIT_OUT 1

// doubles of kd_build's own LDS (its layout: raw | ext | red | tmpA | tmpB | prk | pmn | pmx)
__host__ __device__ inline size_t nbp_update_kd_doubles(int D, int N, int Npad, int P) {
  return ((size_t)D * N + 3 * Npad + NBP_RED + 2 * NBP_KD_PARTS) + ((size_t)2 * N + (size_t)P * Npad + 2 + 1) / 2;
}
struct fused_lds {
  double *tab, *slot, *bw, *cen, *tr;
  size_t SL;
};
// bytes of a fused workgroup's LDS; fills `L` when `base` is given
// P = helper rows of the workgroup (workgroup = P x Npad lanes): the fits and the KD builds use all of them, the product
// runs P lanes per output sample, a proposal one lane per particle (the other rows wait at its barriers)
__host__ __device__ inline size_t nbp_update_lds_layout(int Fmax, int D, int N, int Npad, int P, bool circ, double *base, fused_lds *L) {
  const size_t SL = 3 * (size_t)N + 8;
  size_t o = 0;
  auto dbl = [&](size_t n) { size_t r = o; o += n; return r; };
  const size_t tab = dbl(NBP_EXPTAB), slot = dbl((size_t)Fmax * SL), bw = dbl((size_t)Fmax * 3), cen = dbl((size_t)Fmax * 3);
  o = (o + 1) & ~(size_t)1;
  const size_t tr = o;
  // transient area: the largest of the four phases (all in doubles)
  const size_t prop = 3 * (size_t)N + NBP_RED + ((size_t)N + 1) / 2;
  const size_t fit = 2 * (size_t)N + (size_t)P * Npad + (size_t)(P * Npad / 64) * 2 * N + NBP_RED + NBP_FITTAB;
  const size_t kd = nbp_update_kd_doubles(D, N, Npad, P) + ((size_t)N + 1) / 2 /* idx */;
  const size_t bulk = (size_t)Fmax * D * N;
  const size_t prodL = 3 * bulk + (circ ? 2 * (size_t)Fmax * N : 0) + (size_t)Fmax * N /* lg */ + 6 * (size_t)Fmax + N + 2 * (size_t)Fmax * Npad /* uu */ +
                       ((size_t)Fmax * Npad * 2 + 1) / 2 /* ind | nxt */;
  size_t t = prop;
  if (fit > t) t = fit;
  if (kd > t) t = kd;
  if (prodL > t) t = prodL;
  if (L) {
    L->tab = base + tab; L->slot = base + slot; L->bw = base + bw; L->cen = base + cen; L->tr = base + tr;
    L->SL = SL;
  }
  return (tr + t) * 8;
}
static inline size_t nbp_update_lds_bytes(int Fmax, int D, int N, int Npad, int P, bool circ) {
  return nbp_update_lds_layout(Fmax, D, N, Npad, P, circ, nullptr, nullptr);
}

// a wave-uniform pointer that arrived in vector registers (function arguments do) back into scalar registers: the
// descriptor reads and the branches on them stay scalar
template <class T>
__device__ __forceinline__ T *uniform_ptr(T *p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
  return (T *)(((unsigned long long)hi << 32) | lo);
}
// One proposal as a real call.  Inlined into the update kernel the proposal's registers (the Nelder-Mead simplex, ~150
// VGPRs) are allocated together with everything the other phases keep live and the kernel spills; as a function it gets
// the allocation of the stand-alone proposal kernel, and nothing but uniform values is live across the call.  The LDS
// areas are passed as offsets into the kernel's dynamic LDS, so that the callee still addresses them as LDS (ds_read /
// ds_write) instead of through generic pointers.
template <int FIXK, int FIXM>
__device__ __attribute__((noinline)) void proposal_call(const nbp_proposal_desc *d, int out_off, int tr_off, double *arena, int N, int Npad,
                                                        int64_t S, int32_t *side, nbp_counters *ctr) {
  extern __shared__ double smem[];
  N = __builtin_amdgcn_readfirstlane(N);
  Npad = __builtin_amdgcn_readfirstlane(Npad);
  out_off = __builtin_amdgcn_readfirstlane(out_off);
  tr_off = __builtin_amdgcn_readfirstlane(tr_off);
  const unsigned long long s64 = (unsigned long long)S;
  S = (int64_t)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(s64 >> 32)) << 32) |
                (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)s64));
  proposal_body<FIXK, FIXM>(uniform_ptr(d), smem + out_off, uniform_ptr(arena), N, Npad, S, uniform_ptr(side), uniform_ptr(ctr), smem + tr_off);
}

// FIXK / FIXM: the class of the launch (proposals_uniform_class): every relative factor is a full factor of kind FIXK on
// the manifold FIXM, every product lives on FIXM and has only full inputs
template <int FIXK, int FIXM, int P>
__device__ __forceinline__ void update_body(const nbp_update_desc *uds, const nbp_proposal_desc *props, const nbp_product_desc *prods,
                                            double *arena, int N, int Npad, int64_t S, int32_t *side, const nbp_levels &T, nbp_counters *ctr,
                                            int Fmax, double *smem) {
  constexpr int M = FIXM;
  constexpr int D = (M == NBP_SE2) ? 3 : (M == NBP_CIRCULAR ? 1 : M);
  constexpr bool CIRC = (M == NBP_SE2 || M == NBP_CIRCULAR);
  const nbp_update_desc *u = uds + blockIdx.x;
  const nbp_product_desc *d = prods + u->prod;
  const int F = d->nfactors, tid = threadIdx.x, flags = u->flags;
  fused_lds FL;
  nbp_update_lds_layout(Fmax, D, N, Npad, P, CIRC, smem, &FL);
  const size_t SL = FL.SL;
  nbp_exp_tab_init(FL.tab);
  NBP_TICK_INIT();
  // ---- proposalbeliefs!: the F proposals of the update, one after the other, one lane per particle -----------------
  for (int j = 0; j < F; j++) {
#ifdef NBP_X_PROPCALL
    proposal_call<FIXK, FIXM>(props + u->prop[j], (int)(FL.slot + j * SL - smem), (int)(FL.tr - smem), arena, N, Npad, S, side, ctr);
#else
    proposal_body<FIXK, FIXM>(props + u->prop[j], FL.slot + j * SL, arena, N, Npad, S, side, ctr, FL.tr);
#endif
    __syncthreads();
  }
  NBP_TICK(50);  // proposals
  // ---- manikde! of every proposal (ApproxConv.jl:36-42; pass-through densities bring their own bandwidth) ------------
  for (int j = 0; j < F; j++) {
    const nbp_proposal_desc *pd = props + u->prop[j];
    const bool fit = !pd->skip_bandwidth && pd->factor_kind != NBP_F_PASSTHROUGH && (F > 1 || (flags & NBP_UPD_FIT_OUT));
    if (fit) {
#pragma unroll 1
      for (int k = 0; k < 3; k++) {
        lcv_slot_coordinate<0>(FL.slot + j * SL, M, k, N, Npad, FL.tr, ctr);
        __syncthreads();
      }
    }
    if ((flags >> (1 + j)) & 1) {  // somebody reads this proposal from its arena slot later
      double *o = arena + S * pd->out_slot;
      for (int i = tid; i < (int)SL; i += blockDim.x) o[i] = FL.slot[j * SL + i];
    }
  }
  NBP_TICK(51);  // fits of the proposals
  double *out = arena + S * d->out_slot;
  if (F == 1) {
    // a single density: AMP returns it (product_passthrough): points, bandwidth, count; infoPerCoord = ones(D)
    const double *src = FL.slot;
    for (int i = tid; i < 3 * N + 3; i += blockDim.x) out[i] = src[i];
    if (tid < 3) out[3 * N + 3 + tid] = (tid < D) ? 1.0 : 0.0;
    if (tid == 0) out[3 * N + 6] = src[3 * N + 6];
    return;
  }
  // ---- KD trees, in place ----------------------------------------------------------------------------------------------
  if (tid < F * 3) FL.bw[tid] = FL.slot[(tid / 3) * SL + 3 * N + tid % 3];
  __syncthreads();
  int *idx_tmp = (int *)(FL.tr + nbp_update_kd_doubles(D, N, Npad, P));  // the permutation: only a product that reports labels reads it (not fused)
  for (int j = 0; j < F; j++) {
    double *sj = FL.slot + j * SL;
    kd_build<D>(sj, sj, FL.cen + j * 3, idx_tmp, nullptr, N, Npad, T, FL.tr, D == 1 ? 1 : 7);
    __syncthreads();
  }
  NBP_TICK(52);  // KD builds
  // ---- AMP.manifoldProduct: multiscale Gibbs over the trees in LDS, result into slot 0 ---------------------------------
  nbp_fused_io fio;
  {
    const size_t bulk = (size_t)F * D * N;
    double *b = FL.tr;
    fio.L.lm = b; b += bulk;
    fio.L.lv = b; b += bulk;
    fio.L.lr = b; b += bulk;
    fio.L.ls = b; if (CIRC) b += (size_t)F * N;
    fio.L.lc = b; if (CIRC) b += (size_t)F * N;
    fio.L.lg = b; b += (size_t)F * N;
    fio.L.cen = b; b += (size_t)F * 3;
    fio.L.h2 = b; b += (size_t)F * 3;
    fio.L.nw = b; b += N;
    fio.L.tab = FL.tab;
    fio.L.uu = b; b += 2 * (size_t)F * Npad;  // (a workgroup of P x Npad lanes serves Npad samples)
    fio.L.ck = nullptr;                        // chunk sums in registers
    fio.L.ind = (int *)b;
    fio.L.ns = N;
    fio.xs = FL.slot;
    fio.xs_stride = SL;
    fio.idx = idx_tmp;
    fio.idx_stride = 0;
    fio.cen = FL.cen;
    fio.bw = FL.bw;
    fio.out = FL.slot;
  }
  product_body<M, false, P, false, 1>(d, arena, nullptr, 0, nullptr, N, S, side, T, FL.tr, &fio);
  __syncthreads();
  NBP_TICK(53);  // product
  // ---- setBelief! (SolveTree.jl:74): manikde! of the result (when anything reads it), one write of the new belief -----
  if (flags & NBP_UPD_FIT_OUT) {
    if (tid == 0) FL.slot[3 * N + 6] = 0.0;  // N points
    __syncthreads();
#pragma unroll 1
    for (int k = 0; k < 3; k++) {
      lcv_slot_coordinate<0>(FL.slot, M, k, N, Npad, FL.tr, ctr);
      __syncthreads();
    }
    if (tid < 3) out[3 * N + tid] = FL.slot[3 * N + tid];
  }
  NBP_TICK(54);  // fit of the result
  for (int i = tid; i < 3 * N; i += blockDim.x) out[i] = (i < D * N) ? FL.slot[i] : 0.0;
  // infoPerCoord of the update: the sum over its factors of ones(D) (proposalbeliefs!, ApproxConv.jl:277,298-303)
  if (tid < 3) out[3 * N + 3 + tid] = (tid < D) ? (double)F : 0.0;
  if (tid == 0) out[3 * N + 6] = 0.0;
}

#define NBP_UPDATE_ARGS const nbp_update_desc *uds, const nbp_proposal_desc *props, const nbp_product_desc *prods, double *arena, int N, \
                        int Npad, int64_t S, int32_t *side, nbp_levels T, nbp_counters *ctr, int Fmax
#if NBP_TU & NBP_TU_FUSED
#define NBP_UPDATE_KERNEL(NAME, K_, M_, P_, WAVES)                                                                 \
  __global__ void __launch_bounds__(P_ * 256) __attribute__((amdgpu_waves_per_eu(WAVES))) NAME(NBP_UPDATE_ARGS) {  \
    extern __shared__ double smem[];                                                                               \
    update_body<K_, M_, P_>(uds, props, prods, arena, N, Npad, S, side, T, ctr, Fmax, smem);                       \
  }
#else
#define NBP_UPDATE_KERNEL(NAME, K_, M_, P_, WAVES) __global__ void NAME(NBP_UPDATE_ARGS);
#endif
#ifndef NBP_W_UPD_E2
#define NBP_W_UPD_E2 4
#endif
// _p1: one lane per particle (workgroup = Npad lanes; N <= 256), for rounds with thousands of updates; _p2: two helper
// rows (2 Npad lanes), for the rounds that fill the chip only with twice the lanes per update
NBP_UPDATE_KERNEL(nbp_update_kernel_lin2_p1, NBP_F_LINREL, NBP_EUCLID2, 1, NBP_W_UPD_E2)
NBP_UPDATE_KERNEL(nbp_update_kernel_lin2_p2, NBP_F_LINREL, NBP_EUCLID2, 2, 4)
