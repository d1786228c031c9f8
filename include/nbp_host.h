/*
 * nbp_host.h -- native (C++) host side of the clique hot path: factor graph container, sparse
 * elimination ordering, Bayes tree, clique potentials, Gibbs id lists and the compilation of a whole
 * up+down tree solve into a device-resident libnbp program (include/nbp.h).
 *
 * SURVEY.md 8(f) rank 1 ("Gibbs-schedule generator + message assembly") and rank 3 ("Bayes-tree build
 * with sparse ordering, host C++"): the integer / symbolic work the reference does in
 *   src/services/BayesNet.jl:139-189            buildBayesNet!
 *   src/services/JunctionTreeUtils.jl:435-495   newPotential / buildTree!
 *   src/services/JunctionTreeUtils.jl:1045-1083 setCliqPotentials!
 *   src/services/JunctionTreeUtils.jl:1294-1523 compCliqAssocMatrices!, setCliqMCIDs! and friends
 *   src/services/SolveTree.jl:164-239           upGibbsCliqueDensity (schedule)
 *   src/CliqueStateMachine/services/CliqStateMachineUtils.jl:424-571  down sequence / products
 *   src/services/TreeMessageUtils.jl:66-89,542-578   message factors <-> slots
 *   src/services/TreeMessageUtils.jl:126-193,279-456  joint upward messages (useMsgLikelihoods, a solver flag)
 * so that a host (the Julia shim, or the Python mirror in this repo) can hand over a graph and an
 * elimination order and get the same staged program the Python reference implementation of this repo
 * (bayestree.py / solver.TreeProgram) builds -- byte for byte, which is how it is tested.
 *
 * Variables and factors are addressed by the dense ids the add calls return (insertion order).
 */
#ifndef NBP_HOST_H
#define NBP_HOST_H
#include "nbp.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nbp_graph nbp_graph;
typedef struct nbp_tree nbp_tree;

/* the SolverParams fields the path reads (entities/SolverParams.jl:12-75) */
typedef struct nbp_solver_params {
  int32_t N;               /* particles per belief                                   */
  int32_t gibbs_iters;     /* gibbsIters (3)                                         */
  int32_t inflate_cycles;  /* inflateCycles (3)                                      */
  int32_t product_niter;   /* Niter of AMP.manifoldProduct (1)                       */
  int32_t upsolve, downsolve, limitfixeddown;
  int32_t flags;           /* 0 = the reference's defaults; enum nbp_solver_flag                */
  double spread_nh;        /* spreadNH (3.0)                                         */
  double inflation;        /* inflation (5.0): default of a factor without its own   */
  double null_surplus_add; /* nullSurplusAdd (0.3)                                   */
} nbp_solver_params;

enum nbp_solver_flag {
  NBP_SOLVER_STORED_MEASUREMENTS = 1, /* alwaysFreshMeasurements = false (SolverParams.jl:69, SolveTree.jl:119) */
  NBP_SOLVER_MSG_LIKELIHOODS = 2      /* useMsgLikelihoods = true: joint upward messages (SolverParams.jl:25,
                                         TreeMessageUtils.jl:279-456, 538-578)                                  */
};

/* one factor: addFactor!(dfg, Xi, usrfnc; multihypo, nullhypo, inflation) (FactorGraph.jl:824-875) */
typedef struct nbp_factor_spec {
  int32_t factor_kind;       /* enum nbp_factor (not NBP_F_MSGPRIOR: messages are made by the compiler) */
  int32_t nvars;
  int32_t vars[NBP_MAXV];    /* variable ids, getVariableOrder                                       */
  int32_t ncomp;             /* measurement model components (Mixture) */
  int32_t has_multihypo;     /* 0 / 1 */
  int32_t partial_mask;      /* see nbp_proposal_desc */
  double multihypo[NBP_MAXV]; /* parsed Categorical p, certain variables 0.0 (FactorGraph.jl:639-651) */
  double nullhypo;
  double inflation;          /* <= 0: use nbp_solver_params.inflation */
  double comp[NBP_MAXC][NBP_COMP_STRIDE];
} nbp_factor_spec;

nbp_status nbp_graph_create(const nbp_solver_params *params, nbp_graph **out);
nbp_status nbp_graph_destroy(nbp_graph *g);
/* return the new id (>= 0) or a negative status */
int32_t nbp_graph_add_variable(nbp_graph *g, int32_t manifold);
int32_t nbp_graph_add_factor(nbp_graph *g, const nbp_factor_spec *spec);
/* A new variable is NOT initialised (like addVariable!): run the graph-initialisation program below, or
 * write a belief yourself (nbp_slot_write) and say so here. */
nbp_status nbp_graph_set_variable_flags(nbp_graph *g, int32_t var, int32_t initialized, int32_t ismargin);
int32_t nbp_graph_num_variables(const nbp_graph *g);
int32_t nbp_graph_num_factors(const nbp_graph *g);

/* Graph initialisation -- initAll! / doautoinit! (services/GraphInit.jl:61-199): every uninitialised
 * variable gets a belief from the factors whose other variables are initialised (multihypo factors: at
 * least one hypothesis available, #427), in add order; independent initialisations share a stage.
 * nbp_graph_init_plan builds the stage descriptors and returns the slots the context needs; belief of
 * variable v lives in slot v.  nbp_graph_init_compile makes the resident program (run it once) and marks
 * the planned variables initialised in the graph. */
int32_t nbp_graph_init_plan(nbp_graph *g, uint64_t seed);
/* NBP_F_PASSTHROUGH priors (PartialPriorPassThrough, ApproxConv.jl:196-227) carry a density that the caller writes
 * into a slot of its own with nbp_belief_write before running a program, like an initial belief: density i belongs
 * to factor nbp_graph_density_factors()[i] and sits in slot (density_slot0 + i) of the plan being run — after
 * nbp_graph_init_plan for the initialisation program, after nbp_tree_plan_slots for the tree solve. */
int32_t nbp_graph_num_densities(const nbp_graph *g);
nbp_status nbp_graph_density_factors(const nbp_graph *g, int32_t *factor_ids_out);
int32_t nbp_graph_init_density_slot0(const nbp_graph *g);
int32_t nbp_tree_density_slot0(const nbp_tree *t);
int32_t nbp_graph_init_num_variables(const nbp_graph *g);
nbp_status nbp_graph_init_variables(const nbp_graph *g, int32_t *vars_out);
int32_t nbp_graph_init_num_stages(const nbp_graph *g);
nbp_status nbp_graph_init_stage(const nbp_graph *g, int32_t s, int32_t *kind, int32_t *n, void *descs_out, int64_t cap_bytes);
nbp_status nbp_graph_init_compile(nbp_graph *g, nbp_ctx *ctx, nbp_program **out);

/* Nested-dissection elimination order (recursive bisection on BFS level structures; dense nodes such as
 * landmarks seen from many poses are set aside and eliminated last, as AMD / COLAMD do with dense rows).
 * order_out[nvars].  The reference's default (dense column-pivoted QR, BayesNet.jl:40-44) stays with the
 * caller: any order can be passed to nbp_tree_build, as solveTree!(...; eliminationOrder) allows. */
nbp_status nbp_graph_order_nested_dissection(const nbp_graph *g, int32_t *order_out);

/* buildTreeReset!(dfg, eliminationOrder) + buildCliquePotentials + the Gibbs schedules */
nbp_status nbp_tree_build(const nbp_graph *g, const int32_t *order, int32_t n, nbp_tree **out);
nbp_status nbp_tree_destroy(nbp_tree *t);
int32_t nbp_tree_num_cliques(const nbp_tree *t);

/* clique k (1-based like the reference's CliqueId): every out pointer may be NULL; the arrays must hold
 * nbp_graph_num_variables / nbp_graph_num_factors entries; schedules up to nbp_tree_max_schedule(). */
typedef struct nbp_clique_info {
  int32_t parent;   /* 0 = root */
  int32_t nfrontals, nseparators, nchildren, npotentials, nup, ndown;
} nbp_clique_info;
nbp_status nbp_tree_clique(const nbp_tree *t, int32_t k, nbp_clique_info *info, int32_t *frontals, int32_t *separators,
                           int32_t *children, int32_t *potentials, int32_t *up_schedule, int32_t *down_schedule);
int32_t nbp_tree_max_schedule(const nbp_tree *t);
/* the four Gibbs id lists of clique k (getCliqueData(cliq).directFrtlMsgIDs / msgskipIDs / itervarIDs /
 * directPriorMsgIDs; setCliqMCIDs!, JunctionTreeUtils.jl:1352-1523) as variable ids: counts[4], then one array per
 * list (each nullable, nbp_graph_num_variables entries at most) -- what nbp_clique_upsolve takes */
nbp_status nbp_tree_clique_idlists(const nbp_tree *t, int32_t k, int32_t *counts, int32_t *direct_frtl_msg, int32_t *msgskip,
                                   int32_t *itervar, int32_t *direct_prior_msg);

/* The slot plan of a whole-tree solve: main[v] | snap[v] (optional) | clique-local copies | scratch.
 * Returns the number of slots the context must have. */
int32_t nbp_tree_plan_slots(nbp_tree *t, int32_t snapshot);
nbp_status nbp_tree_main_slots(const nbp_tree *t, int32_t *main_out /* nvars */, int32_t *snap_out /* nvars or NULL */);

/* Compile the up+down solve into a resident program on `ctx` (which must have been created with at
 * least nbp_tree_plan_slots() slots and the graph's N).  The caller writes the initial beliefs into the
 * main (or snap) slots and runs / reseeds / destroys the program with the nbp_program_* calls. */
nbp_status nbp_tree_compile(nbp_tree *t, nbp_ctx *ctx, uint64_t seed, nbp_program **out);
/* the host half of nbp_tree_compile alone: build the stage descriptors (no device needed) */
nbp_status nbp_tree_schedule(nbp_tree *t, uint64_t seed);

/* ---- multi-rank compile: the cliques of a tree sharded over `world` ranks (one process per GPU) --------------------
 * nbp_tree_partition proposes an owner per clique (connected subtrees of balanced work + the cliques above the cut);
 * nbp_tree_set_owner makes the tree compile THIS rank's share: slots and stages for its own cliques only, landing
 * ("ghost") slots for the up messages of children that live elsewhere, and between the stage segments the exchanges
 * of separator beliefs on tree edges that cross ranks.  Every rank computes the same stage times for the whole tree,
 * so both ends of an exchange agree on where it sits.  Call before nbp_tree_plan_slots.  The segment list of the last
 * compile: kind 0 = run stages [first, last) (nbp_program_run), kind 1 = exchange (nbp_exchange). */
/* nbp_xfer (peer rank, slot) is declared in nbp.h */
nbp_status nbp_tree_partition(const nbp_tree *t, int32_t world, int32_t *owner_out /* [cliques] */);
nbp_status nbp_tree_set_owner(nbp_tree *t, const int32_t *owner /* [cliques]; NULL = single rank */, int32_t rank);
int32_t nbp_tree_num_segments(const nbp_tree *t);
nbp_status nbp_tree_segment(const nbp_tree *t, int32_t i, int32_t *kind, int32_t *first, int32_t *last, int32_t *nsend, int32_t *nrecv,
                            nbp_xfer *sends, nbp_xfer *recvs, int32_t cap);

/* One solve of this rank's share from C: run the stage segments of the last nbp_tree_compile and, between them, the
 * separator exchanges -- nbp_exchange on `comm` (grouped RCCL send / recv on the library's stream: stream-ordered with
 * the kernels on both sides, no host synchronisation, no host code between the launches), or the caller's transport
 * (`xchg`: e.g. host-staged point-to-point in the CPU tests).  The counterpart of the reference's task graph -- one Task
 * per clique, blocked on the Channels of its tree edges (taskSolveTree!, src/services/SolverAPI.jl:50-100;
 * CliqueStateMachine.jl:221-234, 617-629) -- with the cliques of a rank compiled into stage ranges. */
typedef nbp_status (*nbp_exchange_fn)(void *user, const nbp_xfer *sends, int32_t n_sends, const nbp_xfer *recvs, int32_t n_recvs);
nbp_status nbp_tree_run_sharded(const nbp_tree *t, nbp_program *prog, nbp_ctx *ctx, nbp_comm *comm);
nbp_status nbp_tree_run_sharded_cb(const nbp_tree *t, nbp_program *prog, nbp_exchange_fn xchg, void *user);

typedef struct nbp_tree_stats {
  int64_t stages, proposals, products, updates_up, updates_down, messages, slots;
  int64_t alg_bytes;          /* sum of B_upd over all updates, SURVEY 8(d) */
  int64_t alg_bytes_proposal, alg_bytes_prep, alg_bytes_product;
} nbp_tree_stats;
nbp_status nbp_tree_get_stats(const nbp_tree *t, nbp_tree_stats *out);

/* ---- the clique seam, one clique at a time -----------------------------------------------------------------
 * For a host that keeps the tree and the CliqueStateMachine (the north star: scheduling stays in Julia): the two
 * calls the CSM makes per clique,
 *     upGibbsCliqueDensity(dfg, cliq, solveKey, inmsgs, N, dbg, iters, logger) -> Dict{Symbol,TreeBelief}
 *         src/services/SolveTree.jl:164-239, called (remotecall_fetch'ed) from approxCliqMarginalUp!,
 *         src/CliqueStateMachine/services/CliqStateMachineUtils.jl:375-385
 *     solveCliqDownFrontalProducts!(subfg, cliq, opts, logger)
 *         src/CliqueStateMachine/services/CliqStateMachineUtils.jl:479-571, called at CliqueStateMachine.jl:838
 * as one C call each: beliefs in, the whole Gibbs schedule on the device, beliefs (val, bw, infoPerCoord) out.
 * The random streams are keyed by (seed, pass, clique_id, step, factor) exactly like nbp_tree_compile keys them,
 * so a clique solved here and the same clique inside a whole-tree program give the same particles. */

/* TreeBelief (entities/BeliefTypes.jl:47-57): val, bw, infoPerCoord; host buffers owned by the caller */
typedef struct nbp_tree_belief {
  double *pts;      /* room for N x P doubles (N of the context), packed AoS like nbp_slot_write: in, the first
                       n_pts rows are read; out, n_pts rows are written (a solved belief holds N points, a belief
                       whose only factor is a pass-through density keeps that density's count)              */
  double *bw;       /* D                                                                          */
  double *ipc;      /* D: infoPerCoord (in: may be NULL = zeros; out: written when not NULL)      */
  int32_t n_pts;    /* in: particles held (<= N; fewer than N needs bw); out: particles written             */
  int32_t handle;   /* 0: the belief lives in the host buffers above.  h > 0: it lives in resident slot h of the context
                       (nbp_ctx_reserve_resident) -- read from there where a call takes it (pts / bw / ipc are then not
                       read and may be NULL), and kept there where a call delivers it; a delivered belief is ALSO copied
                       to the host buffers when pts is not NULL.  What the reference moves through the Channels of a tree
                       edge as values (CliqueStateMachine.jl:590-593, 900-903) moves as a handle.            */
} nbp_tree_belief;

/* CliqStatus (entities/BeliefTypes.jl:8), the status a LikelihoodMessage carries */
enum nbp_cliq_status {
  NBP_CLIQ_NULL = 0, NBP_CLIQ_NO_INIT = 1, NBP_CLIQ_INITIALIZED = 2, NBP_CLIQ_UPSOLVED = 3, NBP_CLIQ_MARGINALIZED = 4,
  NBP_CLIQ_DOWNSOLVED = 5, NBP_CLIQ_UPRECYCLED = 6, NBP_CLIQ_ERROR_STATUS = 7
};

typedef struct nbp_clique_desc {
  int32_t clique_id;           /* CliqueId.value: keys the random streams                                     */
  int32_t nvars;               /* variables of the clique sub graph: the frontals first, then the separators,
                                  then (down solve only) the other variables the factors of the frontals touch
                                  (addDownVariableFactors!, CliqueStateMachine.jl:823-835)                      */
  int32_t nfrontals, nseparators;
  const int32_t *manifold;     /* [nvars] enum nbp_manifold                                                   */
  const int32_t *ismargin;     /* [nvars] or NULL: marginalized variables are never updated (SolveTree.jl:61;
                                  down solve: skipped when params.limitfixeddown)                              */
  int32_t nfactors;            /* up solve: the clique's potentials; down solve: every factor of its frontals  */
  const nbp_factor_spec *factors; /* vars[] index the variable list above                                      */
  /* up solve: the Gibbs id lists of getCliqueData(cliq), as indices into the variable list
   * (setCliqMCIDs!, JunctionTreeUtils.jl:1352-1523).  Ignored by the down solve. */
  int32_t n_direct_frtl_msg, n_msgskip, n_itervar, n_direct_prior_msg;
  const int32_t *direct_frtl_msg, *msgskip, *itervar, *direct_prior_msg;
  /* up solve: the children's upward messages, one entry per (child, separator variable): a MsgPrior{MKD} on that
   * variable (addMsgFactors!, TreeMessageUtils.jl:542-578, generateMsgPrior :86-89), in the order the caller lists
   * them.  Ignored by the down solve, whose parent message is already IN the separator beliefs
   * (updateSubFgFromDownMsgs!, TreeMessageUtils.jl:66-84). */
  int32_t nmsgs;
  const int32_t *msg_var;            /* [nmsgs] index into the variable list */
  const nbp_tree_belief *msg_belief; /* [nmsgs] */
  /* [nfactors] or NULL: the density of every NBP_F_PASSTHROUGH factor (PartialPriorPassThrough.Z, a
   * ManifoldKernelDensity; its points, bandwidth and point count become the proposal as they are,
   * ApproxConv.jl:196-227); entries of other factors are ignored */
  const nbp_tree_belief *factor_density;
  /* Joint upward messages (SolverParams.useMsgLikelihoods; TreeMessageUtils.jl:279-456), the numeric half.  WHICH
   * differential factors and common priors a clique sends up and which of them its parent keeps is symbolic work that
   * stays with the caller (addLikelihoodsDifferentialCHILD!, _generateMsgJointRelativesPriors, addMsgFactors!).
   *   receiving: a differential factor of a child's message -- LinearRelative(::MKD), CircularCircular(::MKD), an SE(2)
   *     ManifoldFactor over a KDE -- is an entry of `factors` (kind, the two separator variables) with its measurement
   *     density in factor_meas_kde[f]: points = the measurement's tangent coordinates (zdim doubles per point, packed like
   *     an Euclid(zdim) belief) + bandwidth.  [nfactors] or NULL; entries with pts == NULL use the factor's own model.
   *   sending (up solve, nbp_clique_upsolve_joint): for every pair (diff_a[i], diff_b[i]) of separator variables the
   *     approxDeconv of a default-constructed factor of kind diff_kind[i] between the solved beliefs, and manikde! of
   *     the predicted measurements (TreeMessageUtils.jl:279-335) -> diff_out[i] (zdim doubles per point, N points). */
  const nbp_tree_belief *factor_meas_kde;
  int32_t n_diff;
  int32_t reserved_;
  const int32_t *diff_a, *diff_b, *diff_kind;
} nbp_clique_desc;

/* slots a context needs for this clique (nbp_ctx_create(..., n_slots >= this)): variables + messages + pass-through
 * densities + one scratch row of the widest product per variable (Gibbs steps that commute -- different variables,
 * no common factor -- run side by side in one stage; the particles are those of the step-by-step schedule) */
int32_t nbp_clique_slots(const nbp_clique_desc *cliq);
/* beliefs_inout[nvars]: in = the beliefs of the clique sub graph (the deep copy the CSM made); out = the belief of
 * every variable the schedule updated (the others are left as they were).  status_out (nullable) = NBP_CLIQ_UPSOLVED /
 * NBP_CLIQ_DOWNSOLVED.  Hard errors return < 0 (the shim raises, the CSM monitor propagates ERROR_STATUS). */
/* CONCURRENT CALLERS.  These three calls may be made from several host threads on ONE context -- the reference's shape, a
 * task per clique (CliqueStateMachine.jl; SolverAPI.jl:59-97).  Calls that arrive while a batch is on the device are merged
 * into the next batch (the path of nbp_clique_solve_batch below): the first caller that finds a free lane leads, gives the
 * callers the last batch released a moment to come back (NBP_COMBINE_GATHER_US, 150), takes what has queued up, runs it and
 * wakes its callers.  Lanes (NBP_COMBINE_LANES, 2) are the context itself and contexts of the library's own on the same
 * device, so that two batches are on the device side by side; a request whose beliefs are resident handles runs on the
 * caller's context only.  Every caller gets its own status and message; the bytes of a result do not depend on what it was
 * batched with.  1000-variable chain, 16 callers: 171 ms a walk with a context per caller, 116 ms merged
 * (profiles/r06_clique_seam_rate.txt).  A lone caller's call is a batch of one, as it always was. */
nbp_status nbp_clique_upsolve(nbp_ctx *ctx, const nbp_solver_params *params, const nbp_clique_desc *cliq, uint64_t seed,
                              nbp_tree_belief *beliefs_inout, int32_t *status_out);
nbp_status nbp_clique_downsolve(nbp_ctx *ctx, const nbp_solver_params *params, const nbp_clique_desc *cliq, uint64_t seed,
                                nbp_tree_belief *beliefs_inout, int32_t *status_out);
/* the up solve of a clique that sends a joint message: diff_out[cliq->n_diff] receives the differential KDEs (the caller
 * provides pts with room for N x zdim doubles and bw[zdim] each); nbp_clique_upsolve is this call with n_diff == 0 */
nbp_status nbp_clique_upsolve_joint(nbp_ctx *ctx, const nbp_solver_params *params, const nbp_clique_desc *cliq, uint64_t seed,
                                    nbp_tree_belief *beliefs_inout, nbp_tree_belief *diff_out, int32_t *status_out);

/* Several clique calls in ONE: cliques that do not depend on each other -- the cliques of one tree level on the way up or
 * down, which the reference runs as concurrent tasks (CliqueStateMachine.jl) -- planned side by side in one context, with
 * one transfer of beliefs each way and one program whose k-th launches serve the k-th round of every clique.  Up and down
 * requests may be mixed.  Each request is what the single calls take (diff_out: NULL, or clique->n_diff entries as in
 * nbp_clique_upsolve_joint); `status` is written on success.  The random streams are keyed by (seed, pass, clique_id, step,
 * factor), so every belief comes out as the single calls deliver it: bit for bit where the launches of batch and single
 * call pick the same geometry (the cliques of a chain or a small tree), and to summation-order rounding (1e-9 on the
 * particles) otherwise -- helper lanes per sample, helper rows of a fit and the speculative search are chosen by the
 * number of updates in a launch, and a batch has more of them.  The context needs the sum of
 * nbp_clique_slots over the requests.  This is the entry for a host that keeps the control flow of the state machines but
 * gathers the cliques that are ready (the per-clique seam is bound by the latency of each clique's own
 * launches; batched, the cliques of a level share them). */
typedef struct nbp_clique_request {
  const nbp_solver_params *params;
  const nbp_clique_desc *clique;
  uint64_t seed;
  nbp_tree_belief *beliefs;  /* in / out, clique->nvars entries */
  nbp_tree_belief *diff_out; /* NULL or clique->n_diff entries */
  int32_t down;              /* 0: upGibbsCliqueDensity; 1: solveCliqDownFrontalProducts! */
  int32_t status;            /* out: NBP_CLIQ_UPSOLVED / NBP_CLIQ_DOWNSOLVED */
} nbp_clique_request;
nbp_status nbp_clique_solve_batch(nbp_ctx *ctx, nbp_clique_request *requests, int32_t n);

/* The same call in two halves, so that a host overlaps ITS work with the device's (review r04: "give the clique seam
 * overlap").  nbp_clique_submit_batch plans the cliques, queues the beliefs (host ones: one packed copy; resident ones:
 * device copies), the program and the copies out on the library stream and RETURNS WITHOUT WAITING; the requests, their
 * descriptors and the host buffers beliefs are delivered INTO must stay alive until nbp_clique_wait(ticket), which waits
 * for this batch only (an event behind its last copy), unpacks the delivered beliefs and writes the statuses.  Batches
 * run in submission order.  With resident beliefs (nbp_tree_belief.handle) a parent's batch can be submitted before its
 * children's has run -- the messages are slots the stream orders, not host values -- so a host can queue a whole pass and
 * wait once: the next level's planning and sub-graph assembly run under the current level's kernels, and no belief
 * crosses PCIe between levels.  nbp_clique_solve_batch = submit + wait.  The counterpart of the reference's clique tasks
 * blocking on the Channels of their tree edges (CliqStateMachineUtils.jl:375-385, CliqueStateMachine.jl:590-593). */
typedef struct nbp_clique_ticket nbp_clique_ticket;
nbp_status nbp_clique_submit_batch(nbp_ctx *ctx, nbp_clique_request *requests, int32_t n, nbp_clique_ticket **ticket_out);
nbp_status nbp_clique_wait(nbp_clique_ticket *ticket); /* consumes the ticket, whatever the status */
/* resident beliefs from the host side: written (one packed copy, queued), read back (waits), copied among themselves on
 * the device (queued; points_only: the down message -- the parent's VALUES of a separator, TreeMessageUtils.jl:66-84) */
nbp_status nbp_resident_write(nbp_ctx *ctx, int32_t n, const int32_t *handles, const int32_t *manifolds, const nbp_tree_belief *beliefs);
nbp_status nbp_resident_read(nbp_ctx *ctx, int32_t n, const int32_t *handles, const int32_t *manifolds, nbp_tree_belief *beliefs);
nbp_status nbp_resident_copy(nbp_ctx *ctx, int32_t n, const int32_t *src_handles, const int32_t *dst_handles, int32_t points_only);

/* diagnostics: host wall clock the clique calls of this process have spent, by phase (seconds): [0] planning the schedule,
 * [1] beliefs host -> device, [2] program assembly + finalize, [3] launches (mode 2: + waiting for them), [4] (waiting +)
 * beliefs device -> host, [5] number of calls.  mode 0: read; 1: read and reset; 2: read, reset, and from now on wait for
 * the device after the launches so that [3] and [4] separate (not thread-safe: a measuring tool's switch). */
nbp_status nbp_clique_seam_times(double *seconds_out6, int32_t mode);

/* test access: the descriptors of stage s of the last compile (kind = NBP_STAGE_*; bytes copied <= cap) */
int32_t nbp_tree_num_stages(const nbp_tree *t);
nbp_status nbp_tree_stage(const nbp_tree *t, int32_t s, int32_t *kind, int32_t *n, void *descs_out, int64_t cap_bytes);

#ifdef __cplusplus
}
#endif
#endif /* NBP_HOST_H */
