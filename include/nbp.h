/*
 * nbp.h -- C ABI of libnbp: the MI355X-native replacement for the per-clique
 * nonparametric Chapman-Kolmogorov hot path of IncrementalInference.jl (IIF).
 *
 * The reference has NO FFI for this path: it is extended by Julia multiple
 * dispatch.  This header therefore *defines* the boundary a Julia `ccall` shim binds to
 * (INTEGRATION.md shows the shim).  Every entry point cites the reference function it replaces
 * (paths relative to the IncrementalInference.jl source tree, v0.35.6).
 *
 * Conventions
 *  - plain pointers and sizes only; no torch / C++ types.
 *  - all floating point is IEEE double (reference: Float64 everywhere).
 *  - the caller owns every host buffer; the library never keeps a host pointer past return.
 *  - return value: 0 = NBP_OK, <0 = hard error (the Julia shim turns it into `error()` so the
 *    clique Task fails and `monitorCSMs` propagates ERROR_STATUS, CliqStateMachineUtils.jl:184-246).
 *  - soft conditions are counted, not raised: non-converged per-particle solves are used anyway
 *    (NumericalCalculations.jl:128-131), NaN solves leave the particle unchanged (:348-351).
 *
 * Host point layout ("P doubles per point", packed AoS, exactly what Julia holds):
 *    NBP_EUCLID1/2/3 : P = D, SVector{D,Float64}                (Variables/DefaultVariables.jl:9-19)
 *    NBP_CIRCULAR    : P = 1, angle in [-pi,pi) (shim packs Vector{Vector{Float64}}, :52)
 *    NBP_SE2         : P = 6, ArrayPartition(t[2], R[2x2] column-major) = x,y,R11,R21,R12,R22
 *                      (test/testSpecialEuclidean2Mani.jl:14)
 * Device layout ("slot"): tangent coordinates at the identity, SoA: coord d of particle n at
 *    slot_base + d*N + n, d < D (SE2 is stored as x,y,theta), then bw[3], infoPerCoord[3], see DESIGN.md:
 *    a slot is the device form of a TreeBelief (val, bw, infoPerCoord; entities/BeliefTypes.jl:47-57).
 */
#ifndef NBP_H
#define NBP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NBP_MAXV 6   /* variables attached to one factor (multihypo door sighting uses 5) */
#define NBP_MAXF 128 /* densities multiplied in one manifoldProduct (a landmark with many sightings);
                        products whose node statistics do not fit the 160 KB LDS keep them in HBM/L2 */
#define NBP_MAXD 3   /* tangent dimension                                                   */
#define NBP_MAXC 4   /* Mixture components                                                  */
#define NBP_MAXN 512 /* particles per belief                                                */
#define NBP_COMP_STRIDE 13 /* per component: weight, mean[3], sqrt-cov L[3][3] row-major    */
/* Family of a SCALAR measurement component, carried in its last slot (comp[c][12] = L[2][2], which a one-dimensional
 * measurement does not use; read only when the measurement has one dimension).  Multi-dimensional measurements are
 * Gaussian (MvNormal).  `rand(Z)` of the reference's Distributions.jl objects (sampleFactor, CalcFactor.jl). */
enum nbp_dist {
  NBP_DIST_GAUSSIAN = 0, /* z = mean[0] + L[0][0] * randn                                     */
  NBP_DIST_UNIFORM = 1,  /* Uniform(a, b): mean[0] = a, L[0][0] = b - a; z = a + (b - a) u   */
  NBP_DIST_RAYLEIGH = 2, /* Rayleigh(sigma): L[0][0] = sigma; z = sigma sqrt(-2 log u), u in (0, 1] */
  NBP_DIST_TABLE = 3     /* AliasingScalarSampler(domain, weights) (entities/AliasScalarSampling.jl:13-74): z = domain[i],
                            i ~ Categorical(weights) -- StatsBase.alias_sample! draws from that categorical, here by inverse
                            CDF on one uniform.  The table lives in a slot: row 0 = the domain, row 1 = the CUMULATIVE
                            normalised weights, count = its length (<= N); written like a belief on NBP_EUCLID2
                            (nbp_belief_write; clique calls: factor_density[f]).  nbp_proposal_desc.var_slot[NBP_MAXV - 1]
                            names the slot (factors with a table have at most NBP_MAXV - 1 variables); one table per factor.
                            A slot has N rows: a table with MORE than N entries does not fit -- nbp_clique_* refuse it
                            (NBP_ERR_RANGE) and the hosts (Python `table_belief(N)`, ext/IIFNbpExt.jl `supported(fct, N)`)
                            refuse it before writing; nbp_belief_write itself cannot tell a table from a belief and keeps
                            the first N rows of whatever it is given */
};

typedef int32_t nbp_status;
#define NBP_OK 0
#define NBP_ERR_ARG (-1)
#define NBP_ERR_HIP (-2)
#define NBP_ERR_NOGPU (-3)
#define NBP_ERR_RANGE (-4)

/* getManifold(variableType): Position{N} -> TranslationGroup(N), Circular -> RealCircleGroup,
 * SE(2) -> SpecialEuclidean(2; vectors=HybridTangentRepresentation()) */
enum nbp_manifold {
  NBP_EUCLID1 = 1,
  NBP_EUCLID2 = 2,
  NBP_EUCLID3 = 3,
  NBP_CIRCULAR = 4,
  NBP_SE2 = 5
};

/* the closed set of residual functors */
enum nbp_factor {
  NBP_F_PRIOR = 1,      /* Prior / PriorCircular / ManifoldPrior: point = wrap(mean + L n)
                           Factors/DefaultPrior.jl:17, Circular.jl:56-60, GenericFunctions.jl:181-214 */
  NBP_F_MSGPRIOR = 2,   /* MsgPrior{ManifoldKernelDensity}: draw from the KDE held in var_slot[1]
                           Factors/MsgPrior.jl:27-30, services/TreeMessageUtils.jl:86-89 */
  NBP_F_LINREL = 3,     /* LinearRelative{D}: r = z - (x2 - x1)       Factors/LinearRelative.jl:42-49 */
  NBP_F_CIRCULAR = 4,   /* CircularCircular                            Factors/Circular.jl:24-28       */
  NBP_F_SE2 = 5,        /* ManifoldFactor(SpecialEuclidean(2))         Factors/GenericFunctions.jl:39-44,98-100 */
  NBP_F_EUCLIDDIST = 6, /* EuclidDistance: r = z - ||x2 - x1||         Factors/EuclidDistance.jl:20    */
  NBP_F_PASSTHROUGH = 7 /* PartialPriorPassThrough(Z, partial): the proposal IS the density Z.heatmap.densityFnc, placed on
                           the coordinates of `.partial` (calcProposalBelief dispatch, ApproxConv.jl:196-227;
                           Factors/PartialPriorPassThrough.jl).  var_slot[1] = slot holding that density (points in the
                           variable's coordinates -- only the partial ones are read --, bandwidth, count); nothing is
                           sampled, solved or fitted, and no hypothesis machinery applies (evalFactor is bypassed) */
};

/*
 * One proposal = one `approxConvBelief(dfg, fct, target)` (services/ApproxConv.jl:4-45):
 * evalFactor -> evalPotentialSpecific (services/EvalFactor.jl:321-395 relative, :400-542 prior)
 * followed by manikde! (bandwidth fit).  Carries exactly the knobs the reference reads from
 * SolverParams / the factor (entities/SolverParams.jl:12-75, services/FactorGraph.jl:824-839).
 */
typedef struct nbp_proposal_desc {
  int32_t factor_kind;        /* enum nbp_factor                                              */
  int32_t manifold;           /* enum nbp_manifold of the solve-for variable                  */
  int32_t nvars;              /* variables attached to the factor (1 for priors)              */
  int32_t sfidx;              /* 0-based index of the solve-for variable in var_slot          */
  int32_t var_slot[NBP_MAXV]; /* belief slot of every attached variable (getVariableOrder);
                                 NBP_F_MSGPRIOR: var_slot[1] = slot holding the message KDE   */
  int32_t out_slot;           /* scratch slot that receives the proposal (never the belief itself:
                                 services/CalcFactor.jl:543-548)                              */
  int32_t ncomp;              /* 1, or the number of Mixture components (Factors/Mixture.jl)  */
  int32_t has_multihypo;      /* 0: hypotheses === nothing; else bit 0 set.  Optional isinit flags of the
                                 attached variables (ExplicitDiscreteMarginalizations.jl:161-172): bit 7 =
                                 flags present, bit 8+k = variable k is initialised                */
  int32_t inflate_cycles;     /* SolverParams.inflateCycles (default 3)                       */
  int32_t mhidx_in;           /* >=0: offset into the ctx int32 side buffer holding an injected
                                 mhidx[N] (exact-match tests); -1: sample internally           */
  int32_t mhidx_out;          /* >=0: offset in the side buffer where the used mhidx[N] is stored */
  int32_t skip_bandwidth;     /* 1: do not fit the bandwidth (caller discards it)              */
  int32_t partial_mask;       /* 0: full factor.  Otherwise bit k set = tangent coordinate k is in the
                                 factor's `.partial` tuple (ccw.partialDims, CalcFactor.jl:369-486):
                                 the measurement has popcount(mask) dimensions; a partial prior sets
                                 only those coordinates (setPointPartial!, EvalFactor.jl:457-538), a
                                 partial relative factor solves and inflates only them (:184-198);
                                 all other coordinates keep the target's current values          */
  int32_t meas_kde;           /* 0: the measurement model is `comp`.  k+1: the measurement is a kernel density
                                 estimate held in slot k (points + bandwidth), sampled as random kernel +
                                 bw*randn: the "differential" relative factors LinearRelative(::MKD) /
                                 CircularCircular(::MKD) of the useMsgLikelihoods upward messages
                                 (TreeMessageUtils.jl:279-335, Factors/LinearRelative.jl:32,
                                 manifolds/services/ManifoldSampling.jl:13-19); relative factors only */
  int32_t keep_count;         /* NBP_F_PASSTHROUGH only.  1: the proposal keeps the density's own particle count (the
                                 density is the only factor of the update: "PassThrough transfers the full point count
                                 to the graph, unless a product is calculated", test/testSpecialEuclidean2Mani.jl:394);
                                 0: it is resampled (multinomially, no kernel noise: the bandwidth stays) to the N
                                 points a product needs from every input;
                                 2: graph initialisation of a variable from this density alone: "the graph should stay
                                 restricted to N" (resample(bel, N), GraphInit.jl:174-177) -- the density's points
                                 stay, the rest up to N are draws from its KDE (random kernel + bw * randn) */
  double multihypo[NBP_MAXV]; /* parsed Categorical p: certain variables carry 0.0
                                 (services/FactorGraph.jl:639-651)                            */
  double nullhypo;            /* max(ccw.nullhypo, nullSurplus)   EvalFactor.jl:352            */
  double inflation;           /* ccw.inflation (default SolverParams.inflation = 5.0)          */
  double spread_nh;           /* SolverParams.spreadNH (3.0)                                   */
  double comp[NBP_MAXC][NBP_COMP_STRIDE]; /* measurement model, see NBP_COMP_STRIDE            */
  uint64_t seed;              /* Philox key of this op (counter-based RNG)          */
  uint64_t meas_seed;         /* 0: fresh measurements, drawn from `seed`.  Otherwise the measurement samples
                                 (and Mixture labels / KDE draws) are those of the op with this seed: the
                                 stored ccw.measurement reused when needFreshMeasurements = false, i.e. Gibbs
                                 iterations > 1 with SolverParams.alwaysFreshMeasurements = false
                                 (SolveTree.jl:119, services/CalcFactor.jl:492-510)            */
} nbp_proposal_desc;

/*
 * One product = the `AMP.manifoldProduct(dens, M; Niter=1, N)` call of propagateBelief
 * (services/GraphProductOperations.jl:53-60) plus the bandwidth fit of the result and the
 * `setBelief!` write-back (services/SolveTree.jl:74, services/FactorGraph.jl:250-263).
 */
typedef struct nbp_product_desc {
  int32_t manifold;
  int32_t nfactors;            /* 1 = pass-through (AMP returns the single density)            */
  int32_t niter;               /* Gibbs sweeps per tree level, 1 .. 8 (reference passes Niter=1) */
  int32_t out_slot;            /* belief slot that receives points + bandwidth                 */
  int32_t in_slot[NBP_MAXF];   /* proposal slots                                               */
  int32_t labels_out;          /* >=0: offset in the int32 side buffer for labels[N][nfactors] */
  int32_t old_slot;            /* oldPoints (GraphProductOperations.jl:39-45): coordinates that no input
                                  density informs are copied from this slot; -1 when every input is full */
  uint8_t in_partial[NBP_MAXF]; /* per input density: 0 = full, else the coordinate bit mask of a partial
                                  density (AMP.marginal(propBel, pardims), ApproxConv.jl:287-291): it
                                  multiplies into the product on those coordinates only              */
  uint64_t seed;
} nbp_product_desc;

/* belief copy: tree message traffic (TreeBelief val+bw, entities/BeliefTypes.jl:47-57) */
typedef struct nbp_copy_desc {
  int32_t src_slot;
  int32_t dst_slot;
} nbp_copy_desc;

typedef struct nbp_diag {
  int64_t solves;        /* per-particle optimiser runs                        */
  int64_t nonconverged;  /* NumericalCalculations.jl:128-131                   */
  int64_t nan_results;   /* NumericalCalculations.jl:348-351                   */
  int64_t residual_evals;
  int64_t lcv_evals;     /* leave-one-out likelihood evaluations in double precision (each = N(N-1)/2 kernel pairs) */
  int64_t lcv_evals_f32; /* bracketing evaluations of the bandwidth searches in single precision (each = N(N-1) ordered
                            pairs): they decide comparisons whose two sides are further apart than their error bounds;
                            the bandwidth selected is that of the all-double search, bit for bit (NBP_FIT_F64=1 in the
                            environment of nbp_ctx_create: every evaluation in double precision)                        */
} nbp_diag;

typedef struct nbp_ctx nbp_ctx;

/* ---- context ---------------------------------------------------------------------------- */
/* `arena` may be NULL (library allocates with hipMalloc) or a device pointer owned by the caller
 * (e.g. a torch tensor, so torch.distributed/RCCL can exchange slots); arena_bytes must be at
 * least nbp_arena_bytes(N, n_slots).  `side_ints` = size of the int32 side buffer (mhidx, labels). */
int64_t nbp_arena_bytes(int32_t N, int32_t n_slots);
int64_t nbp_slot_stride_doubles(int32_t N);
nbp_status nbp_ctx_create(int32_t device, int32_t N, int32_t n_slots, void *arena,
                          int64_t arena_bytes, int32_t side_ints, nbp_ctx **out);
nbp_status nbp_ctx_destroy(nbp_ctx *ctx);
/* One object of the layer above rides with the context (the native host's plan cache): `destroy(obj)` is called at the top of
 * nbp_ctx_destroy, while the context and its programs are still whole, or when another object takes its place. */
nbp_status nbp_ctx_attach(nbp_ctx *ctx, void *obj, void (*destroy)(void *obj));
void *nbp_ctx_attached(const nbp_ctx *ctx);
const char *nbp_last_error(void);
nbp_status nbp_synchronize(nbp_ctx *ctx);
void *nbp_arena_ptr(nbp_ctx *ctx);
void *nbp_stream_ptr(nbp_ctx *ctx); /* hipStream_t the library launches on */
int32_t nbp_ctx_particles(const nbp_ctx *ctx); /* N the context was created for (0: null) */
int32_t nbp_ctx_slots(const nbp_ctx *ctx);     /* belief slots of its arena (0: null) */
int32_t nbp_ctx_device(const nbp_ctx *ctx);    /* the HIP device it was created on (-1: null) */
/* Resident slots: the LAST n slots of the arena are set aside for beliefs that STAY on the device between clique calls --
 * the deep-copied sub graph of a clique between its up and its down solve, the up message a parent reads from its child
 * (a LikelihoodMessage that never visits the host: nbp_tree_belief.handle, nbp_host.h).  Handle h (1 .. n) names slot
 * nbp_ctx_slots - h; the clique calls plan their own slots below the resident ones. */
nbp_status nbp_ctx_reserve_resident(nbp_ctx *ctx, int32_t n);
int32_t nbp_ctx_resident(const nbp_ctx *ctx);

/* ---- belief I/O: setValKDE!/getVal at the boundary (FactorGraph.jl:250-297) ---------------
 * A slot is read and written BY MANIFOLD: the rows of the manifold's D coordinates (and bw / infoPerCoord entries 0 .. D-1)
 * are the belief; rows beyond D are unspecified -- a write from the host zeroes them, a kernel that produces a belief
 * leaves them as they were (a third of a Euclid(2) proposal's write traffic), and nothing in the library reads them (fits,
 * KD builds, products, proposals and the reads below all go by the manifold; whole-slot copies carry them along unread).
 * Read a slot with the manifold of the belief that was put there. */
nbp_status nbp_slot_write(nbp_ctx *ctx, int32_t slot, int32_t manifold, const double *pts_NxP,
                          const double *bw_D /* nullable */);
nbp_status nbp_slot_read(nbp_ctx *ctx, int32_t slot, int32_t manifold, double *pts_NxP,
                         double *bw_D /* nullable */);
/* The full TreeBelief / VariableNodeData triple (val, bw, infoPerCoord; BeliefTypes.jl:47-57, FactorGraph.jl:250-263).
 * infoPerCoord is produced on the device like the reference produces it: a proposal carries ones(D), zeroed outside
 * the factor's `.partial` (EvalFactor.jl:383-391); a variable update carries the sum over its factors of ones(D)
 * (proposalbeliefs!, ApproxConv.jl:277,298-303 -- the reference's `fct_ipc = ones(vardim)`, partial factors
 * included); slot copies (tree messages) carry it along.  nbp_slot_write stores zeros (a fresh VariableNodeData).
 * n_pts: particle count of the belief.  A slot holds up to N points (N of the context).  Fewer: the belief keeps its
 * own count (it must come with its bandwidth) and every consumer does what the reference does with a belief shorter
 * than N -- a convolution reads a random element for the particles beyond its end (_getindex_anyn,
 * NumericalCalculations.jl:377-381), the scratch copy of a target is filled up with the point default
 * (CalcFactor.jl:555-565), a MsgPrior / KDE measurement samples among the points it has, the oldPoints of a product are
 * topped up with sample(oldBel, N - Npts) (GraphProductOperations.jl:39-45), manikde! fits the points there are; every
 * kernel output holds N points.  More than N: the first N are kept (`_pts[1:N]`, GraphProductOperations.jl:44).
 * nbp_belief_read returns the count in *n_pts and fills the first *n_pts points (at most N rows: size `pts` for the N of the
 * context unless the count is known); with n_pts = NULL (nbp_slot_read) all N rows of the slot are written. */
nbp_status nbp_belief_write(nbp_ctx *ctx, int32_t slot, int32_t manifold, const double *pts_NxP, int32_t n_pts,
                            const double *bw_D /* nullable */, const double *ipc_D /* nullable: zeros */);
nbp_status nbp_belief_read(nbp_ctx *ctx, int32_t slot, int32_t manifold, double *pts_NxP, int32_t *n_pts /* nullable */,
                           double *bw_D /* nullable */, double *ipc_D /* nullable */);
/* Many beliefs in one call: packed into a pinned staging buffer and moved with one asynchronous copy per run of consecutive
 * slots on the library's stream (a clique call moves all its beliefs in one or two copies; a caller loading a whole graph
 * moves it in one).  Arrays of n entries; n_pts / bw / ipc may be NULL as a whole (N points, no bandwidth, infoPerCoord 0)
 * or per entry (bw, ipc).  _write_batch returns once the copies are queued: they are ordered before whatever is launched
 * next.  _read_batch returns the rows each belief holds in n_pts[i] (pts[i] sized for N rows), like nbp_belief_read. */
nbp_status nbp_belief_write_batch(nbp_ctx *ctx, int32_t n, const int32_t *slots, const int32_t *manifolds, const double *const *pts,
                                  const int32_t *n_pts, const double *const *bw, const double *const *ipc);
nbp_status nbp_belief_read_batch(nbp_ctx *ctx, int32_t n, const int32_t *slots, const int32_t *manifolds, double *const *pts,
                                 int32_t *n_pts, double *const *bw, double *const *ipc);
/* The same transfers without a host synchronisation (the asynchronous clique seam, nbp_clique_submit_batch): staged in
 * pinned buffers of a pool of the context, queued on the library stream behind whatever runs there.  _write_batch_async
 * returns once the copies are queued (the caller's buffers are free again: they were packed).  _read_batch_begin queues the
 * copies out and returns a token; _read_batch_end waits for them (an event, not the whole stream), unpacks into the caller's
 * buffers (entries with pts[i] == NULL are skipped) and consumes the token. */
typedef struct nbp_read_token nbp_read_token;
nbp_status nbp_belief_write_batch_async(nbp_ctx *ctx, int32_t n, const int32_t *slots, const int32_t *manifolds, const double *const *pts,
                                        const int32_t *n_pts, const double *const *bw, const double *const *ipc);
nbp_status nbp_belief_read_batch_begin(nbp_ctx *ctx, int32_t n, const int32_t *slots, nbp_read_token **out);
nbp_status nbp_belief_read_batch_end(nbp_read_token *token, const int32_t *manifolds, double *const *pts, int32_t *n_pts,
                                     double *const *bw, double *const *ipc);
/* sample(oldBel, N - Npts) in place: beliefs with fewer than N points are topped up to N with draws from their own KDE
 * (random kernel + bw * randn); the points they hold stay.  Multinomial resampling of a belief to the solver's N. */
nbp_status nbp_run_resample(nbp_ctx *ctx, const int32_t *slots, const int32_t *manifolds, int32_t n, uint64_t seed);
nbp_status nbp_side_write(nbp_ctx *ctx, int32_t offset, const int32_t *src, int32_t n);
nbp_status nbp_side_read(nbp_ctx *ctx, int32_t offset, int32_t *dst, int32_t n);

/* ---- factor seam: approxConvOnElements!/evalFactor + manikde! (EvalFactor.jl:14-27,571-603) */
nbp_status nbp_run_proposals(nbp_ctx *ctx, const nbp_proposal_desc *descs, int32_t n);
/* ---- AMP.manikde!(M, pts) bandwidth selection for slots already resident ------------------ */
nbp_status nbp_run_bandwidth(nbp_ctx *ctx, const int32_t *slots, const int32_t *manifolds, int32_t n);
/* ---- variable seam: AMP.manifoldProduct + rebandwidth (GraphProductOperations.jl:53-60) --- */
nbp_status nbp_run_products(nbp_ctx *ctx, const nbp_product_desc *descs, int32_t n);
/* ---- host-buffer entry points: one call per reference function ----------------------------------
 * For callers that keep beliefs on the host (the factor / variable seams of the Julia shim).  Points are
 * packed AoS like nbp_slot_write; the calls stage through slots 0.. of the context (which they clobber)
 * and synchronise before returning.
 *
 * nbp_kde_bandwidth:     AMP.manikde!(M, pts) bandwidth selection (ApproxConv.jl:38,41; FGOSUtils.jl:118-128).
 * nbp_conv:              approxConvBelief(dfg, fct, target) (ApproxConv.jl:4-45): `tmpl` carries the factor
 *                        (kind, manifold, nvars, sfidx, comp, multihypo, nullhypo, ..., seed); its slot fields
 *                        are ignored.  var_pts[i] = points of variable i (MsgPrior: [target, message KDE] and
 *                        var_bw[1] = the KDE's bandwidth).  mhidx_in / out_mhidx nullable (needs 2N side ints);
 *                        out_bw NULL skips the bandwidth fit.  Context: >= nvars + 1 slots.
 * nbp_manifold_product:  AMP.manifoldProduct(dens, M; Niter, oldPoints) + rebandwidth
 *                        (GraphProductOperations.jl:53-60).  partial_masks / old_pts / out_labels nullable.
 *                        Context: >= F + 2 slots (and N*F side ints for the labels).                        */
nbp_status nbp_kde_bandwidth(nbp_ctx *ctx, int32_t manifold, const double *pts_NxP, double *bw_out_D);
nbp_status nbp_conv(nbp_ctx *ctx, const nbp_proposal_desc *tmpl, const double *const *var_pts, const double *const *var_bw,
                    const int32_t *mhidx_in, double *out_pts_NxP, double *out_bw_D, int32_t *out_mhidx);
nbp_status nbp_manifold_product(nbp_ctx *ctx, int32_t manifold, int32_t nfactors, const double *const *dens_pts,
                                const double *const *dens_bw, const uint8_t *partial_masks, const double *old_pts,
                                int32_t niter, uint64_t seed, double *out_pts_NxP, double *out_bw_D, int32_t *out_labels);

/* approxDeconv(dfg, fct) (services/DeconvUtils.jl:32-160): per particle, sample a measurement and
 * search from it for the measurement that zeroes the residual between the stored points of the two
 * variables (var_slot[0], var_slot[1]).  out_slot receives the predicted measurement (tangent
 * coordinates k < zDim, read it back with the Euclid manifold of that dimension), meas_slots[i] (may be
 * NULL / -1) the sampled one.  Relative factors without multihypo only (reference #467/#927). */
nbp_status nbp_run_deconv(nbp_ctx *ctx, const nbp_proposal_desc *descs, const int32_t *meas_slots, int32_t n);
nbp_status nbp_run_copies(nbp_ctx *ctx, const nbp_copy_desc *descs, int32_t n);
/* the same queued on the library stream without waiting (points_only != 0: the points and the count, not the bandwidth) */
nbp_status nbp_run_copies_async(nbp_ctx *ctx, const nbp_copy_desc *descs, int32_t n, int32_t points_only);

/* ---- clique seam: a whole up/down schedule resident on the device --------------------------
 * Replaces upGibbsCliqueDensity (SolveTree.jl:164-239) and solveCliqDownFrontalProducts!
 * (CliqStateMachineUtils.jl:479-571) for *all cliques of a tree level at once*: a program is an
 * ordered list of stages; a stage is a batch of independent proposals, products or copies.
 * Upload once, replay per solve.                                                              */
typedef struct nbp_program nbp_program;
enum nbp_stage_kind {
  NBP_STAGE_PROPOSALS = 1,
  NBP_STAGE_PRODUCTS = 2,
  NBP_STAGE_COPIES = 3,
  NBP_STAGE_DECONV = 4, /* nbp_proposal_desc[]: approxDeconv of a relative factor between var_slot[0] and
                           var_slot[1] (like nbp_run_deconv); out_slot receives the predicted measurements
                           AND their fitted bandwidth (manikde!), i.e. a KDE a later proposal can name in
                           meas_kde -- the child side of the differential messages
                           (addLikelihoodsDifferentialCHILD!, TreeMessageUtils.jl:279-335) */
  NBP_STAGE_COPY_POINTS = 5  /* nbp_copy_desc[]: copies the points of a slot only.  The destination's bandwidth is
                           left unspecified, nothing waits for a pending fit of the source: for values that are read
                           as points and never as a density -- the separator values a down message hands to a child
                           (updateSubFgFromDownMsgs!, TreeMessageUtils.jl:66-84) */
};
nbp_status nbp_program_create(nbp_ctx *ctx, nbp_program **out);
nbp_status nbp_program_add_stage(nbp_program *prog, int32_t kind, const void *descs, int32_t n);
/* Program options, to be set before nbp_program_finalize.
 * NBP_OPT_LAZY_BANDWIDTH (default 0): skip the bandwidth fit of a product output that a later stage
 * overwrites before anything reads its bandwidth (readers: a MsgPrior proposal sampling the slot, a
 * product taking it as input, slot copies; an EMPTY copy stage counts as "everything is read").  The
 * reference fits a bandwidth on every setBelief! (FactorGraph.jl:250-263), but inside a clique's Gibbs
 * sweeps only the last one of each variable is ever looked at, so no result changes.  With the option
 * on, the bandwidth of such an intermediate belief is undefined between stages.
 * NBP_OPT_GRAPH_REPLAY (default 1): nbp_program_run captures the launch sequence of a stage range into a hipGraph the
 * second time it runs and replays the graph afterwards (one submission instead of ~3 launches per variable update).
 * Ignored while per-kernel timing is enabled.
 * NBP_OPT_FUSED_UPDATES (default 1): a PROPOSALS stage followed by the PRODUCTS stage that multiplies exactly its proposals
 * (one round of Gibbs steps) may run as ONE launch of the fused update kernel -- proposals, their bandwidth fits, the KD
 * trees, the product and the fit of the result in one workgroup per variable, with nothing but the operand beliefs and
 * the new belief touching HBM -- when the round has at least NBP_FUSED_MIN updates (environment, read at nbp_ctx_create;
 * unset = never: the fused form trades time for traffic) and its factors are of a class the kernel is built
 * for; same particles and bandwidths as the three-launch form up to the rounding of sums taken in another order.
 * A stage range of nbp_program_run that ends between the two stages of a fused pair runs the pair in the three-launch form
 * (the proposals to their arena slots, their fits at the end of the range); a range that starts at the second stage of such a pair
 * fits every proposal of the pair again before the products (redundant after the previous range's end-of-range fits, never wrong). */
enum nbp_program_option { NBP_OPT_LAZY_BANDWIDTH = 1, NBP_OPT_GRAPH_REPLAY = 2, NBP_OPT_FUSED_UPDATES = 3,
                          NBP_OPT_ASYNC_UPLOAD = 4 /* (default 0) nbp_program_finalize sends the descriptors stream-ordered from a
                                                      pinned buffer and does not wait: for short-lived programs queued behind
                                                      running ones (the asynchronous clique seam, nbp_host.h) */ };
/* options are set before nbp_program_finalize; NBP_OPT_GRAPH_REPLAY alone may also be switched afterwards (it is a property of the
 * runs: graphs already captured stay with the program) */
nbp_status nbp_program_set_option(nbp_program *prog, int32_t option, int32_t value);
nbp_status nbp_program_finalize(nbp_program *prog);              /* uploads descriptors        */
nbp_status nbp_program_run(nbp_program *prog, int32_t first_stage, int32_t last_stage /* excl, -1=all */);
nbp_status nbp_program_reseed(nbp_program *prog, uint64_t salt); /* xor-mix all op seeds on device */
/* New seeds for every op of a finalized program, in stage order: per proposal / deconvolution descriptor its seed and, where the
 * descriptor named a stored measurement when the program was finalized (meas_seed != 0), that one behind it; per product
 * descriptor its seed.  n = nbp_program_num_seeds.  Stream-ordered.  (The native host's plan cache: a batch of clique requests
 * whose structure has not changed is the same program with other seeds -- include/nbp_host.h.) */
nbp_status nbp_program_num_seeds(nbp_program *prog, int32_t *out);
nbp_status nbp_program_set_seeds(nbp_program *prog, const uint64_t *seeds, int32_t n);
nbp_status nbp_program_num_stages(nbp_program *prog, int32_t *out);
/* rounds of a finalized program that run as one launch of the fused update kernel (NBP_OPT_FUSED_UPDATES) */
nbp_status nbp_program_num_fused(nbp_program *prog, int32_t *out);
/* rounds of a finalized program whose two halves run on two streams, one launch apart (environment NBP_PIPELINE_MIN =
 * smallest product batch that is split; same particles and bandwidths as the single-stream order, bit for bit) */
nbp_status nbp_program_num_two_stream(nbp_program *prog, int32_t *out);
nbp_status nbp_program_destroy(nbp_program *prog);
/* destroy without waiting: the program is dropped once everything queued on the library stream so far has run */
nbp_status nbp_program_retire(nbp_program *prog);

/* ---- separator exchange between ranks (one process per GPU) -------------------------------------------------------
 * The reference moves a LikelihoodMessage through a Channel per tree edge (JunctionTreeUtils.jl:943-956,
 * CliqueStateMachine.jl:221-234/617-629); across GPUs the payload -- whole slots = TreeBelief (val, bw, infoPerCoord,
 * count) -- travels point to point over RCCL (xGMI), all messages of one exchange point in ONE grouped call on the
 * library's stream: stream-ordered with the producing and consuming kernels, no host synchronisation.
 * nbp_comm_unique_id: rank 0 makes the id (NBP_COMM_ID_BYTES bytes) and hands it to every rank by whatever means the
 * host has (a broadcast over torch.distributed, MPI, a file); nbp_comm_create is collective over the ranks. */
#define NBP_COMM_ID_BYTES 128
typedef struct nbp_comm nbp_comm;
#ifndef NBP_XFER_DEFINED
#define NBP_XFER_DEFINED
typedef struct nbp_xfer { int32_t peer; int32_t slot; } nbp_xfer;
#endif
nbp_status nbp_comm_unique_id(void *id_out /* NBP_COMM_ID_BYTES */);
nbp_status nbp_comm_create(nbp_ctx *ctx, int32_t world, int32_t rank, const void *id, nbp_comm **out);
nbp_status nbp_comm_destroy(nbp_comm *comm);
/* what RCCL itself says about the communicator: ncclCommCount / ncclCommUserRank (bench.py prints them: a scaling run shows
 * how many ranks the library's own communicator had, not how many the launcher was asked for) */
nbp_status nbp_comm_info(nbp_comm *comm, int32_t *nranks_out, int32_t *rank_out);
nbp_status nbp_exchange(nbp_ctx *ctx, nbp_comm *comm, const nbp_xfer *sends, int32_t n_sends, const nbp_xfer *recvs, int32_t n_recvs);

/* per-kernel timing with HIP events on the library stream (bench.py roofline leg) */
nbp_status nbp_timing_enable(nbp_ctx *ctx, int32_t on);
/* ms[4] / launches[4] per kernel: 0 = proposal kernel, 1 = prep kernel = bandwidth fits + KD-tree builds,
 *                      2 = product kernel, 3 = plain bandwidth kernel = early flushes; reset by the call */
nbp_status nbp_timing_read(nbp_ctx *ctx, double *ms, int64_t *launches);
/* the same with n <= 5 entries: 4 = the fused update kernel (NBP_OPT_FUSED_UPDATES); all five are reset by either call */
nbp_status nbp_timing_read_n(nbp_ctx *ctx, double *ms, int64_t *launches, int32_t n);
nbp_status nbp_diag_read(nbp_ctx *ctx, nbp_diag *out, int32_t reset);

/* Self-test of the elementary functions the kernels and the CPU checker share (include/nbp_math.h), evaluated ON THE DEVICE:
 * fn 0: out0 = log(a); 1: (out0, out1) = (sin, cos)(a); 2: out0 = atan2(a, b); 3: out0 = wrap to [-pi, pi) of a;
 * 4: (out0, out1) = the Box-Muller pair of the uniforms (a, b).  Host buffers of n doubles (b, out1 may be null where unused).
 * tests/test_gpu_device_math.py compares with the same header compiled by gcc: equal to the last bit. */
nbp_status nbp_math_eval(nbp_ctx *ctx, int32_t fn, const double *a, const double *b, double *out0, double *out1, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* NBP_H */
