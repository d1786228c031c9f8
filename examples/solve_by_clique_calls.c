/*
 * solve_by_clique_calls.c -- the clique seam from plain C: a host that keeps the tree and the per-clique control
 * flow (what IncrementalInference.jl's CliqueStateMachine does in Julia) and calls libnbp once per clique,
 *     nbp_clique_upsolve    = upGibbsCliqueDensity            (SolveTree.jl:164-239)
 *     nbp_clique_downsolve  = solveCliqDownFrontalProducts!   (CliqStateMachineUtils.jl:479-571)
 * with message assembly (separator beliefs up, parent values down) done on the host.  The result is compared, byte
 * for byte, with the whole-tree resident program (nbp_tree_compile) run from the same initial beliefs and seed.
 *
 *   gcc -O2 -fopenmp -Iinclude examples/solve_by_clique_calls.c -o /tmp/clique_calls \
 *       -Lincrementalinference.jl_amd/csrc -lnbp -Wl,-rpath,$PWD/incrementalinference.jl_amd/csrc -lm
 *   /tmp/clique_calls [nvars=12] [N=128] [prior every=5] [concurrent callers=1; 0 = one batched call per tree level;
 *                                                          -1 = the same, queued: resident beliefs, submit per level, wait once;
 *                                                          -2 = queued, the requests of every level kept across walks]
 *
 * With several concurrent callers the cliques of one tree level are solved side by side, one context per caller -- the
 * C equivalent of the reference's one-task-per-clique state machines.  Same posteriors: nothing depends on which context
 * ran a clique.  The HIP runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4): set it to the number of
 * concurrent callers (up to 16), or their launches queue up behind each other.
 *
 * It also times both: the resident program (beliefs stay in HBM) and the clique-by-clique walk, where every call takes its
 * beliefs from host memory and returns them there (the PCIe-inclusive rate of the seam).
 */
#include <math.h>
#include <omp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "nbp_host.h"

#define CHK(call)                                                                  \
  do {                                                                             \
    int rc_ = (call);                                                              \
    if (rc_ < 0) {                                                                 \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, nbp_last_error());             \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

enum { D = 2 };
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
typedef struct { double *pts, bw[D], ipc[D]; } belief; /* one TreeBelief on the host */

static int N;
static belief belief_new(void) { belief b; memset(&b, 0, sizeof(b)); b.pts = calloc((size_t)N * D, sizeof(double)); return b; }
static void belief_copy(belief *dst, const belief *src) {
  memcpy(dst->pts, src->pts, sizeof(double) * N * D);
  memcpy(dst->bw, src->bw, sizeof(dst->bw));
  memcpy(dst->ipc, src->ipc, sizeof(dst->ipc));
}
static nbp_tree_belief view(belief *b) { nbp_tree_belief v = {b->pts, b->bw, b->ipc, N, 0}; return v; }

static void gaussian_factor(nbp_factor_spec *f, int kind, int nvars, int a, int b, double mx, double my, double sigma) {
  memset(f, 0, sizeof(*f));
  f->factor_kind = kind; f->nvars = nvars; f->vars[0] = a; f->vars[1] = b; f->ncomp = 1;
  f->comp[0][0] = 1.0; f->comp[0][1] = mx; f->comp[0][2] = my; f->comp[0][4] = sigma; f->comp[0][8] = sigma;
}
static int find(const int32_t *l, int n, int v) { for (int i = 0; i < n; i++) if (l[i] == v) return i; return -1; }
#define MAXCF 64 /* factors of the frontals of one clique (a chain: at most 2 per frontal + a prior) */

/* ---- the host side of the clique seam: what the CliqueStateMachine keeps -------------------------------------------- */
typedef struct {
  int nvars, nfac, ncl;
  const nbp_factor_spec *fac;
  nbp_tree *tree;
  nbp_clique_info *info;
  int32_t **fr, **se, **ch, **po, *depth;
  belief **sub;          /* sub[c][i]: belief of the i-th variable (frontals, then separators) of clique c's sub graph */
  belief *graph, *post;  /* the graph's beliefs (read-only during the passes, but for the roots' frontals) and the posteriors */
  const nbp_solver_params *sp;
  uint64_t seed;
  /* callers = -1: the beliefs stay on the device between the calls (nbp_ctx_reserve_resident): handle of every graph
   * belief and of every belief of every clique's sub graph; the copies a level needs in front of / behind its batch */
  int resident;
  int32_t *graph_h, **sub_h;
  int32_t *cp_src, *cp_dst, ncp;     /* whole beliefs: the deep copy of a sub graph, a root's result back to the graph */
  int32_t *pp_src, *pp_dst, npp;     /* points only: the down message */
  /* the factors of every variable, in graph order (what ls(dfg, v) hands the reference's host): vfac[vfac0[v] .. vfac0[v+1]) */
  int32_t *vfac, *vfac0;
} host;
static nbp_tree_belief resident_view(int32_t h) { nbp_tree_belief v = {NULL, NULL, NULL, N, h}; return v; }
typedef struct { /* one concurrent caller: its context and its scratch; `q` = the clique call it has prepared */
  nbp_ctx *ctx;
  nbp_clique_desc q;
  int32_t *vars, *mani, *lists[4], *msgv;
  nbp_factor_spec *cf;
  nbp_tree_belief *bel, *msgb;
} worker;

static int up_prepare(host *H, worker *w, int c) {
  const nbp_clique_info *info = H->info;
  const int nf = info[c].nfrontals, ns = info[c].nseparators, nv = nf + ns;
  int32_t *vars = w->vars, counts[4];
  memcpy(vars, H->fr[c], sizeof(int32_t) * nf); memcpy(vars + nf, H->se[c], sizeof(int32_t) * ns);
  if (H->resident) { /* the deep copy happens on the device, queued in front of this level's batch */
    for (int i = 0; i < nv; i++) { H->cp_src[H->ncp] = H->graph_h[vars[i]]; H->cp_dst[H->ncp++] = H->sub_h[c][i]; w->bel[i] = resident_view(H->sub_h[c][i]); }
  } else {
  H->sub[c] = malloc(sizeof(belief) * nv);
  for (int i = 0; i < nv; i++) { H->sub[c][i] = belief_new(); belief_copy(&H->sub[c][i], &H->graph[vars[i]]); w->bel[i] = view(&H->sub[c][i]); } /* deep copy */
  }
  nbp_clique_desc q;
  memset(&q, 0, sizeof(q));
  q.clique_id = c; q.nvars = nv; q.nfrontals = nf; q.nseparators = ns; q.manifold = w->mani;
  for (int i = 0; i < info[c].npotentials; i++) { /* the clique's potentials, variable ids -> positions in `vars` */
    w->cf[i] = H->fac[H->po[c][i]];
    for (int k = 0; k < w->cf[i].nvars; k++) w->cf[i].vars[k] = find(vars, nv, w->cf[i].vars[k]);
  }
  q.nfactors = info[c].npotentials; q.factors = w->cf;
  CHK(nbp_tree_clique_idlists(H->tree, c, counts, w->lists[0], w->lists[1], w->lists[2], w->lists[3]));
  for (int k = 0; k < 4; k++) for (int i = 0; i < counts[k]; i++) w->lists[k][i] = find(vars, nv, w->lists[k][i]);
  q.n_direct_frtl_msg = counts[0]; q.n_msgskip = counts[1]; q.n_itervar = counts[2]; q.n_direct_prior_msg = counts[3];
  q.direct_frtl_msg = w->lists[0]; q.msgskip = w->lists[1]; q.itervar = w->lists[2]; q.direct_prior_msg = w->lists[3];
  int nm = 0; /* the children's upward messages: their separator beliefs */
  for (int j = 0; j < info[c].nchildren; j++) {
    const int cc = H->ch[c][j];
    for (int i = 0; i < info[cc].nseparators; i++) {
      w->msgv[nm] = find(vars, nv, H->se[cc][i]);
      w->msgb[nm++] = H->resident ? resident_view(H->sub_h[cc][info[cc].nfrontals + i]) /* the message is a handle: it never visits the host */
                                  : view(&H->sub[cc][info[cc].nfrontals + i]);
    }
  }
  q.nmsgs = nm; q.msg_var = w->msgv; q.msg_belief = w->msgb;
  w->q = q;
  return 0;
}
static void up_finish(host *H, int c) {
  const int nf = H->info[c].nfrontals;
  if (H->info[c].parent == 0) /* root: the up-solved frontals are the posterior and go back to the graph */
    for (int i = 0; i < nf; i++) {
      if (H->resident) { H->cp_src[H->ncp] = H->sub_h[c][i]; H->cp_dst[H->ncp++] = H->graph_h[H->fr[c][i]]; continue; }
      belief_copy(&H->post[H->fr[c][i]], &H->sub[c][i]); belief_copy(&H->graph[H->fr[c][i]], &H->sub[c][i]);
    }
}
static int up_clique(host *H, worker *w, int c) {
  int32_t status = 0;
  if (up_prepare(H, w, c)) return 1;
  CHK(nbp_clique_upsolve(w->ctx, H->sp, &w->q, H->seed, w->bel, &status));
  if (status != NBP_CLIQ_UPSOLVED) return 1;
  up_finish(H, c);
  return 0;
}

/* every factor of the frontals of clique c, in graph order (ascending factor index), from the variables' own lists -- the
 * reference reads them off the sub graph (ls(subfg, v), CliqStateMachineUtils.jl:500-510); a scan of all factors of the graph
 * per clique was 15 ms of a 31 ms walk of the 1000-variable chain */
static int frontal_factors(const host *H, int c, int32_t *out) {
  int n = 0;
  for (int i = 0; i < H->info[c].nfrontals; i++) {
    const int v = H->fr[c][i];
    for (int q = H->vfac0[v]; q < H->vfac0[v + 1]; q++) {
      const int32_t f = H->vfac[q];
      int at = n;
      while (at > 0 && out[at - 1] > f) at--;
      if (at > 0 && out[at - 1] == f) continue; /* already there */
      if (n >= MAXCF) return -1;
      for (int k = n; k > at; k--) out[k] = out[k - 1];
      out[at] = f;
      n++;
    }
  }
  return n;
}
static int down_prepare(host *H, worker *w, int c) {
  const nbp_clique_info *info = H->info;
  const int p = info[c].parent, nf = info[c].nfrontals, ns = info[c].nseparators;
  int nv = nf + ns;
  int32_t *vars = w->vars;
  memcpy(vars, H->fr[c], sizeof(int32_t) * nf); memcpy(vars + nf, H->se[c], sizeof(int32_t) * ns);
  for (int i = 0; i < ns; i++) { /* the down message: the parent's values of the separators */
    int pi = find(H->fr[p], info[p].nfrontals, H->se[c][i]);
    pi = pi >= 0 ? pi : info[p].nfrontals + find(H->se[p], info[p].nseparators, H->se[c][i]);
    if (H->resident) { H->pp_src[H->npp] = H->sub_h[p][pi]; H->pp_dst[H->npp++] = H->sub_h[c][nf + i]; }
    else memcpy(H->sub[c][nf + i].pts, H->sub[p][pi].pts, sizeof(double) * N * D);
  }
  int32_t ff[MAXCF];
  const int ncf = frontal_factors(H, c, ff); /* every factor of the frontals, in graph order; their other variables come from the graph */
  if (ncf < 0) return 1;
  for (int i = 0; i < ncf; i++) {
    const int f = ff[i];
    w->cf[i] = H->fac[f];
    for (int k = 0; k < H->fac[f].nvars; k++) if (find(vars, nv, H->fac[f].vars[k]) < 0) vars[nv++] = H->fac[f].vars[k];
  }
  for (int i = 0; i < ncf; i++) for (int k = 0; k < w->cf[i].nvars; k++) w->cf[i].vars[k] = find(vars, nv, w->cf[i].vars[k]);
  for (int i = 0; i < nv; i++)
    w->bel[i] = H->resident ? resident_view(i < nf + ns ? H->sub_h[c][i] : H->graph_h[vars[i]]) : (i < nf + ns ? view(&H->sub[c][i]) : view(&H->graph[vars[i]]));
  nbp_clique_desc q;
  memset(&q, 0, sizeof(q));
  q.clique_id = c; q.nvars = nv; q.nfrontals = nf; q.nseparators = ns; q.manifold = w->mani; q.nfactors = ncf; q.factors = w->cf;
  w->q = q;
  return 0;
}
static void down_finish(host *H, int c) {
  if (H->resident) return; /* the posteriors are read back once, at the end of the walk */
  for (int i = 0; i < H->info[c].nfrontals; i++) belief_copy(&H->post[H->fr[c][i]], &H->sub[c][i]);
}
static int down_clique(host *H, worker *w, int c) {
  int32_t status = 0;
  if (down_prepare(H, w, c)) return 1;
  CHK(nbp_clique_downsolve(w->ctx, H->sp, &w->q, H->seed, w->bel, &status));
  if (status != NBP_CLIQ_DOWNSOLVED) return 1;
  down_finish(H, c);
  return 0;
}

/* the cliques of one tree level in ONE call (nbp_clique_solve_batch): the host still assembles every clique's sub graph and
 * messages; the library plans them side by side, moves the beliefs in one transfer each way and shares the launches */
static worker worker_new(int nvars, int nfac) {
  worker w;
  memset(&w, 0, sizeof(w));
  w.vars = malloc(sizeof(int32_t) * nvars); w.mani = malloc(sizeof(int32_t) * nvars);
  for (int k = 0; k < 4; k++) w.lists[k] = malloc(sizeof(int32_t) * nvars);
  w.cf = calloc((size_t)nfac + 1, sizeof(*w.cf));
  w.bel = malloc(sizeof(nbp_tree_belief) * nvars); w.msgb = malloc(sizeof(nbp_tree_belief) * nvars);
  w.msgv = malloc(sizeof(int32_t) * nvars);
  for (int v = 0; v < nvars; v++) w.mani[v] = NBP_EUCLID2;
  return w;
}
static void worker_free(worker *w) {
  free(w->vars); free(w->mani); free(w->cf); free(w->bel); free(w->msgb); free(w->msgv);
  for (int k = 0; k < 4; k++) free(w->lists[k]);
}
static int level_batched(host *H, nbp_ctx *ctx, int d, int down) {
  int n = 0;
  for (int c = 1; c <= H->ncl; c++) n += H->depth[c] == d;
  worker *W = calloc((size_t)n, sizeof(*W));
  nbp_clique_request *R = calloc((size_t)n, sizeof(*R));
  int *id = malloc(sizeof(int) * n), k = 0;
  for (int c = 1; c <= H->ncl; c++) {
    if (H->depth[c] != d) continue;
    int ncf = H->info[c].npotentials; /* scratch sized by the clique: its potentials (up), every factor of its frontals (down) */
    if (down) { int32_t ff[MAXCF]; ncf = frontal_factors(H, c, ff); if (ncf < 0) return 1; }
    int nmsg = 0;
    for (int j = 0; j < H->info[c].nchildren; j++) nmsg += H->info[H->ch[c][j]].nseparators;
    W[k] = worker_new(H->info[c].nfrontals + H->info[c].nseparators + 2 * ncf + nmsg + 1, ncf);
    if (down ? down_prepare(H, &W[k], c) : up_prepare(H, &W[k], c)) return 1;
    R[k].params = H->sp; R[k].clique = &W[k].q; R[k].seed = H->seed; R[k].beliefs = W[k].bel; R[k].down = down;
    id[k++] = c;
  }
  CHK(nbp_clique_solve_batch(ctx, R, n));
  for (int i = 0; i < n; i++) {
    if (R[i].status != (down ? NBP_CLIQ_DOWNSOLVED : NBP_CLIQ_UPSOLVED)) return 1;
    if (down) down_finish(H, id[i]); else up_finish(H, id[i]);
    worker_free(&W[i]);
  }
  free(W); free(R); free(id);
  return 0;
}

/* callers = -1: the same level, QUEUED (nbp_clique_submit_batch): the device copies this level needs in front of its batch,
 * the batch, the copies behind it -- and on to the next level while the device works; everything a queued batch points to
 * stays alive until its ticket has been waited for */
typedef struct {
  worker *W; nbp_clique_request *R; int n; nbp_clique_ticket *t;
  /* callers = -2: the requests of a level and its copy lists are KEPT across walks (the tree has not changed: the descriptors,
   * the id lists, the handles are the same) -- what a CliqueStateMachine that solves the same tree again does not rebuild */
  int built, ncp0, npp0, ncp1;
  int32_t *cp0_src, *cp0_dst, *pp0_src, *pp0_dst, *cp1_src, *cp1_dst;
} queued_level;
static int32_t *dup32(const int32_t *a, int n) { int32_t *r = malloc(sizeof(int32_t) * (n > 0 ? n : 1)); memcpy(r, a, sizeof(int32_t) * n); return r; }
static int level_queued(host *H, nbp_ctx *ctx, int d, int down, queued_level *Q, int keep) {
  if (!(keep && Q->built)) {
    int n = 0;
    for (int c = 1; c <= H->ncl; c++) n += H->depth[c] == d;
    worker *W = calloc((size_t)n, sizeof(*W));
    nbp_clique_request *R = calloc((size_t)n, sizeof(*R));
    int k = 0;
    H->ncp = H->npp = 0;
    for (int c = 1; c <= H->ncl; c++) {
      if (H->depth[c] != d) continue;
      int ncf = H->info[c].npotentials;
      if (down) { int32_t ff[MAXCF]; ncf = frontal_factors(H, c, ff); if (ncf < 0) return 1; }
      int nmsg = 0;
      for (int j = 0; j < H->info[c].nchildren; j++) nmsg += H->info[H->ch[c][j]].nseparators;
      W[k] = worker_new(H->info[c].nfrontals + H->info[c].nseparators + 2 * ncf + nmsg + 1, ncf);
      if (down ? down_prepare(H, &W[k], c) : up_prepare(H, &W[k], c)) return 1;
      R[k].params = H->sp; R[k].clique = &W[k].q; R[k].seed = H->seed; R[k].beliefs = W[k].bel; R[k].down = down;
      k++;
    }
    Q->W = W; Q->R = R; Q->n = n;
    Q->ncp0 = H->ncp; Q->cp0_src = dup32(H->cp_src, H->ncp); Q->cp0_dst = dup32(H->cp_dst, H->ncp);
    Q->npp0 = H->npp; Q->pp0_src = dup32(H->pp_src, H->npp); Q->pp0_dst = dup32(H->pp_dst, H->npp);
    H->ncp = 0;
    for (int c = 1; c <= H->ncl && !down; c++) if (H->depth[c] == d) up_finish(H, c);
    Q->ncp1 = H->ncp; Q->cp1_src = dup32(H->cp_src, H->ncp); Q->cp1_dst = dup32(H->cp_dst, H->ncp);
    Q->built = 1;
  }
  Q->t = NULL;
  for (int i = 0; i < Q->n; i++) Q->R[i].seed = H->seed; /* (kept requests: this walk's seed) */
  CHK(nbp_resident_copy(ctx, Q->ncp0, Q->cp0_src, Q->cp0_dst, 0)); /* up: the deep copies of this level's sub graphs */
  CHK(nbp_resident_copy(ctx, Q->npp0, Q->pp0_src, Q->pp0_dst, 1)); /* down: the parents' values of this level's separators */
  CHK(nbp_clique_submit_batch(ctx, Q->R, Q->n, &Q->t));
  CHK(nbp_resident_copy(ctx, Q->ncp1, Q->cp1_src, Q->cp1_dst, 0)); /* a root's result goes back to the graph */
  return 0;
}
static void level_free(queued_level *Q) {
  for (int i = 0; i < Q->n; i++) worker_free(&Q->W[i]);
  free(Q->W); free(Q->R); free(Q->cp0_src); free(Q->cp0_dst); free(Q->pp0_src); free(Q->pp0_dst); free(Q->cp1_src); free(Q->cp1_dst);
  memset(Q, 0, sizeof(*Q));
}
static int level_wait(queued_level *Q, int down, int keep) {
  CHK(nbp_clique_wait(Q->t));
  for (int i = 0; i < Q->n; i++)
    if (Q->R[i].status != (down ? NBP_CLIQ_DOWNSOLVED : NBP_CLIQ_UPSOLVED)) return 1;
  if (!keep) level_free(Q);
  return 0;
}

int main(int argc, char **argv) {
  const int nvars = argc > 1 ? atoi(argv[1]) : 12;
  N = argc > 2 ? atoi(argv[2]) : 128;
  const int every = argc > 3 ? atoi(argv[3]) : 5;
  const int queued = argc > 4 && atoi(argv[4]) < 0;     /* callers = -1: batched per level, resident beliefs, submit / wait */
  const int keep = argc > 4 && atoi(argv[4]) == -2;     /* callers = -2: the same, the requests of every level kept across walks */
  const int batched = argc > 4 && atoi(argv[4]) <= 0;   /* callers = 0: the cliques of a level in one batched call */
  const int threads = argc > 4 && atoi(argv[4]) > 0 ? atoi(argv[4]) : 1;
  const uint64_t seed = 2024;
  nbp_solver_params sp;
  memset(&sp, 0, sizeof(sp));
  sp.N = N; sp.gibbs_iters = 3; sp.inflate_cycles = 3; sp.product_niter = 1; sp.upsolve = sp.downsolve = 1;
  sp.spread_nh = 3.0; sp.inflation = 5.0; sp.null_surplus_add = 0.3;
  /* ---- the graph: Euclid(2) odometry chain, a prior every 5th pose --------------------------------------- */
  nbp_graph *g = NULL;
  CHK(nbp_graph_create(&sp, &g));
  for (int i = 0; i < nvars; i++) CHK(nbp_graph_add_variable(g, NBP_EUCLID2));
  nbp_factor_spec *fac = calloc(2 * (size_t)nvars, sizeof(*fac));
  int nfac = 0;
  for (int i = 0; i < nvars; i++) {
    if (i % every == 0) { gaussian_factor(&fac[nfac], NBP_F_PRIOR, 1, i, 0, i, i, 0.1); CHK(nbp_graph_add_factor(g, &fac[nfac++])); }
    if (i + 1 < nvars) { gaussian_factor(&fac[nfac], NBP_F_LINREL, 2, i, i + 1, 1.0, 1.0, 0.1); CHK(nbp_graph_add_factor(g, &fac[nfac++])); }
  }
  int32_t *order = malloc(sizeof(int32_t) * nvars), *mainslot = malloc(sizeof(int32_t) * nvars);
  CHK(nbp_graph_order_nested_dissection(g, order));
  nbp_tree *tree = NULL;
  CHK(nbp_tree_build(g, order, nvars, &tree));
  const int ncl = nbp_tree_num_cliques(tree);
  const int n_slots = nbp_tree_plan_slots(tree, 0), init_slots = nbp_graph_init_plan(g, 7);
  CHK(n_slots); CHK(init_slots);
  CHK(nbp_tree_main_slots(tree, mainslot, NULL));
  nbp_ctx *ctx = NULL;
  CHK(nbp_ctx_create(0, N, (n_slots > init_slots ? n_slots : init_slots) + 64, NULL, 0, 0, &ctx));
  /* ---- initAll!, then keep the initial beliefs on the host: the "graph" both solves start from ------------ */
  belief *graph = malloc(sizeof(belief) * nvars), *post = malloc(sizeof(belief) * nvars), *whole = malloc(sizeof(belief) * nvars);
  double one[D] = {1.0, 1.0};
  for (int v = 0; v < nvars; v++) { graph[v] = belief_new(); post[v] = belief_new(); whole[v] = belief_new(); CHK(nbp_slot_write(ctx, v, NBP_EUCLID2, graph[v].pts, one)); }
  nbp_program *prog = NULL;
  CHK(nbp_graph_init_compile(g, ctx, &prog));
  CHK(nbp_program_run(prog, 0, -1));
  CHK(nbp_program_destroy(prog));
  for (int v = 0; v < nvars; v++) CHK(nbp_belief_read(ctx, v, NBP_EUCLID2, graph[v].pts, NULL, graph[v].bw, graph[v].ipc));
  /* ---- (A) the whole tree as one resident program --------------------------------------------------------- */
  for (int v = 0; v < nvars; v++) CHK(nbp_belief_write(ctx, mainslot[v], NBP_EUCLID2, graph[v].pts, N, graph[v].bw, graph[v].ipc));
  CHK(nbp_tree_compile(tree, ctx, seed, &prog));
  CHK(nbp_synchronize(ctx));
  const double ta = now_s();
  CHK(nbp_program_run(prog, 0, -1));
  CHK(nbp_synchronize(ctx));
  const double t_resident = now_s() - ta;
  for (int v = 0; v < nvars; v++) CHK(nbp_belief_read(ctx, mainslot[v], NBP_EUCLID2, whole[v].pts, NULL, whole[v].bw, whole[v].ipc));
  int32_t *io_m = malloc(sizeof(int32_t) * nvars), *io_n = malloc(sizeof(int32_t) * nvars);
  double **io_p = malloc(sizeof(double *) * nvars), **io_b = malloc(sizeof(double *) * nvars), **io_i = malloc(sizeof(double *) * nvars);
  double t_replay = 0, t_io = 0; /* the same program again from the same beliefs: the third run replays the captured hipGraph */
  for (int r = 0; r < 2; r++) {
    const double tw = now_s();
    for (int v = 0; v < nvars; v++) { io_m[v] = NBP_EUCLID2; io_n[v] = N; io_p[v] = graph[v].pts; io_b[v] = graph[v].bw; io_i[v] = graph[v].ipc; }
    CHK(nbp_belief_write_batch(ctx, nvars, mainslot, io_m, (const double *const *)io_p, io_n, (const double *const *)io_b, (const double *const *)io_i));
    CHK(nbp_synchronize(ctx));
    const double t0 = now_s();
    CHK(nbp_program_run(prog, 0, -1));
    CHK(nbp_synchronize(ctx));
    const double t1 = now_s();
    for (int v = 0; v < nvars; v++) { io_p[v] = post[v].pts; io_b[v] = post[v].bw; io_i[v] = post[v].ipc; }
    CHK(nbp_belief_read_batch(ctx, nvars, mainslot, io_m, io_p, io_n, io_b, io_i));
    t_replay = t1 - t0;
    t_io = (t0 - tw) + (now_s() - t1); /* every belief of the graph written to and read from the device: one call each way */
  }
  CHK(nbp_program_destroy(prog));
  /* ---- (B) one C call per clique ----------------------------------------------------------------------------- */
  host H;
  memset(&H, 0, sizeof(H));
  H.nvars = nvars; H.nfac = nfac; H.ncl = ncl; H.fac = fac; H.graph = graph; H.post = post; H.seed = seed; H.sp = &sp;
  H.vfac0 = calloc((size_t)nvars + 1, sizeof(int32_t));
  for (int f = 0; f < nfac; f++) for (int k = 0; k < fac[f].nvars; k++) H.vfac0[fac[f].vars[k] + 1]++;
  for (int v = 0; v < nvars; v++) H.vfac0[v + 1] += H.vfac0[v];
  H.vfac = malloc(sizeof(int32_t) * (size_t)(H.vfac0[nvars] + 1));
  { int32_t *fill = calloc((size_t)nvars, sizeof(int32_t));
    for (int f = 0; f < nfac; f++) for (int k = 0; k < fac[f].nvars; k++) { const int v = fac[f].vars[k]; H.vfac[H.vfac0[v] + fill[v]++] = f; }
    free(fill); }
  H.info = calloc((size_t)ncl + 1, sizeof(*H.info));
  H.fr = calloc((size_t)ncl + 1, sizeof(*H.fr)); H.se = calloc((size_t)ncl + 1, sizeof(*H.se));
  H.ch = calloc((size_t)ncl + 1, sizeof(*H.ch)); H.po = calloc((size_t)ncl + 1, sizeof(*H.po));
  H.depth = calloc((size_t)ncl + 1, sizeof(*H.depth));
  H.sub = calloc((size_t)ncl + 1, sizeof(*H.sub));
  H.tree = tree;
  int maxdepth = 0;
  for (int c = 1; c <= ncl; c++) {
    nbp_clique_info ci;
    CHK(nbp_tree_clique(tree, c, &ci, NULL, NULL, NULL, NULL, NULL, NULL)); /* sizes first */
    H.fr[c] = malloc(sizeof(int32_t) * (ci.nfrontals + 1)); H.se[c] = malloc(sizeof(int32_t) * (ci.nseparators + 1));
    H.ch[c] = malloc(sizeof(int32_t) * (ci.nchildren + 1)); H.po[c] = malloc(sizeof(int32_t) * (ci.npotentials + 1));
    CHK(nbp_tree_clique(tree, c, &H.info[c], H.fr[c], H.se[c], H.ch[c], H.po[c], NULL, NULL));
  }
  for (int c = 1; c <= ncl; c++) { int d = 0; for (int p = H.info[c].parent; p; p = H.info[p].parent) d++; H.depth[c] = d; if (d > maxdepth) maxdepth = d; }
  /* one context per concurrent caller: the cliques of a tree level are independent of each other (the reference runs them
   * as concurrent tasks, CliqueStateMachine.jl), and contexts share nothing */
  /* NBP_SHARED_CTX=1: every caller on the SAME context -- the library merges the single-clique calls that arrive while a batch
   * is on the device into the next batch (nbp_host.cpp, "single-clique calls ... merged"): the callers keep the reference's shape,
   * one task and one call per clique, and the device sees a launch per round of a dozen cliques instead of a dozen launches */
  const int shared = getenv("NBP_SHARED_CTX") && atoi(getenv("NBP_SHARED_CTX")) && !batched;
  /* (NBP_SHARED_SLOTS=n: the shared context is one of its own with n belief slots -- room for a few cliques only, so that a
   *  merged batch has to be cut to what the context holds) */
  nbp_ctx *sctx = ctx;
  if (shared && getenv("NBP_SHARED_SLOTS")) CHK(nbp_ctx_create(0, N, atoi(getenv("NBP_SHARED_SLOTS")), NULL, 0, 0, &sctx));
  worker *W = calloc((size_t)threads, sizeof(*W));
  for (int t = 0; t < threads; t++) {
    W[t] = worker_new(nvars, nfac);
    W[t].ctx = shared ? sctx : (t == 0 ? ctx : NULL);
    if (!W[t].ctx) CHK(nbp_ctx_create(0, N, 256, NULL, 0, 0, &W[t].ctx));
  }
  nbp_ctx *bctx = NULL; /* batched mode: a context with room for the widest level */
  if (batched) {
    int widest = 0;
    for (int d = 0; d <= maxdepth; d++) {
      int cnt = 0;
      for (int c = 1; c <= ncl; c++) cnt += H.depth[c] == d;
      if (cnt > widest) widest = cnt;
    }
    int nres = 0; /* callers = -1: a resident slot for every graph belief and every belief of every sub graph */
    if (queued) {
      H.resident = 1;
      H.graph_h = malloc(sizeof(int32_t) * nvars);
      H.sub_h = calloc((size_t)ncl + 1, sizeof(*H.sub_h));
      for (int v = 0; v < nvars; v++) H.graph_h[v] = ++nres;
      for (int c = 1; c <= ncl; c++) {
        const int nv = H.info[c].nfrontals + H.info[c].nseparators;
        H.sub_h[c] = malloc(sizeof(int32_t) * nv);
        for (int i = 0; i < nv; i++) H.sub_h[c][i] = ++nres;
      }
      H.cp_src = malloc(sizeof(int32_t) * (nres + nvars)); H.cp_dst = malloc(sizeof(int32_t) * (nres + nvars));
      H.pp_src = malloc(sizeof(int32_t) * nres); H.pp_dst = malloc(sizeof(int32_t) * nres);
    }
    CHK(nbp_ctx_create(0, N, widest * 40 + 64 + nres, NULL, 0, 0, &bctx)); /* generous: nbp_clique_slots(desc) is the exact need of a clique */
    CHK(nbp_ctx_reserve_resident(bctx, nres));
  }
  int failed = 0;
  double t_calls = 0;
  belief *graph0 = malloc(sizeof(belief) * nvars); /* the walk writes the roots' frontals back to the graph: keep the start */
  for (int v = 0; v < nvars; v++) { graph0[v] = belief_new(); belief_copy(&graph0[v], &graph[v]); }
  double t_first = 0;
  const int seam_timing = getenv("NBP_SEAM_TIMES") != NULL; /* a third walk with the library's phase clock on */
  double t_timed = 0, ph[6] = {0}, t_queued = 0, t_queued_calls = 0, t_walk[16] = {0};
  /* walks: the second runs with every buffer of the library at its final size; a queued walk of an unchanged tree is, from its
   * second submission on, a cached program per level with new seeds (the library's plan cache), and from the third a hipGraph
   * launch per level -- so the queued modes walk four times and report the last (NBP_WALKS overrides) */
  int nwalks = getenv("NBP_WALKS") ? atoi(getenv("NBP_WALKS")) : (queued ? 4 : 2);
  if (nwalks < 2) nwalks = 2;
  if (nwalks > 15) nwalks = 15;
  for (int pass = 0; pass < nwalks + seam_timing && !failed; pass++) {
  for (int v = 0; v < nvars; v++) belief_copy(&graph[v], &graph0[v]);
  /* NBP_WALK_SEEDS=1: every walk but the last two with a seed of its own (a cached level program is re-seeded, nbp_program_set_seeds);
   * the last walks use the whole-tree program's seed again, and their posteriors must be its bytes */
  H.seed = seed + ((getenv("NBP_WALK_SEEDS") && pass + 1 < nwalks) ? 1000u * (unsigned)(pass + 1) : 0u);
  for (int c = 1; c <= ncl && pass && !queued; c++) { for (int i = 0; i < H.info[c].nfrontals + H.info[c].nseparators; i++) free(H.sub[c][i].pts); free(H.sub[c]); }
  if (pass == nwalks) nbp_clique_seam_times(NULL, 2);
  const double tb = now_s();
  if (queued) {
    /* the graph goes to the device once, every level of both passes is queued behind it, ONE wait, the posteriors come back once */
    nbp_tree_belief *gb = malloc(sizeof(*gb) * nvars);
    int32_t *gm = malloc(sizeof(int32_t) * nvars), *ph_ = malloc(sizeof(int32_t) * nvars);
    for (int v = 0; v < nvars; v++) { gb[v] = view(&graph[v]); gm[v] = NBP_EUCLID2; }
    CHK(nbp_resident_write(bctx, nvars, H.graph_h, gm, gb));
    static queued_level *Q = NULL;
    if (!Q) Q = calloc(2 * ((size_t)maxdepth + 1), sizeof(*Q));
    int nq = 0;
    for (int d = maxdepth; d >= 0 && !failed; d--) failed |= level_queued(&H, bctx, d, 0, &Q[nq++], keep);
    for (int d = 1; d <= maxdepth && !failed; d++) failed |= level_queued(&H, bctx, d, 1, &Q[nq++], keep);
    t_queued = now_s() - tb; /* everything is queued: from here on the host only waits */
    for (int i = 0; i < nq && !failed; i++) failed |= level_wait(&Q[i], i > maxdepth, keep);
    for (int c = 1; c <= ncl; c++) for (int i = 0; i < H.info[c].nfrontals; i++) ph_[H.fr[c][i]] = H.sub_h[c][i]; /* a variable's posterior: its frontal clique's copy */
    for (int v = 0; v < nvars; v++) gb[v] = view(&post[v]);
    if (!failed) CHK(nbp_resident_read(bctx, nvars, ph_, gm, gb));
    free(gb); free(gm); free(ph_);
  }
  for (int d = maxdepth; d >= 0 && !failed && batched && !queued; d--) failed |= level_batched(&H, bctx, d, 0);
  for (int d = 1; d <= maxdepth && !failed && batched && !queued; d++) failed |= level_batched(&H, bctx, d, 1);
  for (int d = maxdepth; d >= 0 && !failed && !batched; d--) { /* up pass: children before parents, the cliques of a level side by side */
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1) reduction(| : failed)
    for (int c = 1; c <= ncl; c++)
      if (H.depth[c] == d) failed |= up_clique(&H, &W[omp_get_thread_num()], c);
  }
  for (int d = 1; d <= maxdepth && !failed && !batched; d++) { /* down pass: parents before children */
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1) reduction(| : failed)
    for (int c = 1; c <= ncl; c++)
      if (H.depth[c] == d) failed |= down_clique(&H, &W[omp_get_thread_num()], c);
  }
  t_walk[pass] = now_s() - tb;
  if (!pass) t_first = now_s() - tb; else if (pass < nwalks) { t_calls = now_s() - tb; t_queued_calls = t_queued; } else { t_timed = now_s() - tb; nbp_clique_seam_times(ph, 1); }
  }
  if (failed) return 4;
  for (int t = 1; t < threads && !shared; t++) nbp_ctx_destroy(W[t].ctx);
  if (sctx != ctx) nbp_ctx_destroy(sctx);
  /* ---- compare -------------------------------------------------------------------------------------------------- */
  /* (the walk's launches and the program's differ in size and so, from some size on, in geometry -- helper lanes per sample, rows of
   *  a fit, one wave per proposal: summation-order rounding, nbp_host.h -- and a Gibbs chain turns one flipped label into other
   *  draws of the same posterior: what is not the same bytes is reported as the distance between the posterior means) */
  int same = 0;
  double worst = 0, worst_dm = 0;
  for (int v = 0; v < nvars; v++) {
    same += memcmp(post[v].pts, whole[v].pts, sizeof(double) * N * D) == 0 && memcmp(post[v].bw, whole[v].bw, sizeof(post[v].bw)) == 0;
    double mx = 0, mw = 0;
    for (int n = 0; n < N; n++) { mx += post[v].pts[2 * n]; mw += whole[v].pts[2 * n]; }
    if (fabs(mx / N - v) > worst) worst = fabs(mx / N - v);
    if (fabs(mx - mw) / N > worst_dm) worst_dm = fabs(mx - mw) / N;
  }
  printf("solve_by_clique_calls: %d variables, %d cliques: %d of %d posteriors byte-identical to the whole-tree program (means of the others within %.3f); "
         "infoPerCoord of x0 = (%.0f, %.0f); worst posterior mean error %.3f; %s%d concurrent caller(s), GPU_MAX_HW_QUEUES=%s\n", nvars, ncl, same, nvars,
         worst_dm, post[0].ipc[0], post[0].ipc[1], worst, queued ? (keep ? "one QUEUED batch per tree level (resident beliefs, submit / wait), the requests KEPT across walks, " : "one QUEUED batch per tree level (resident beliefs, submit / wait), ") : (batched ? "one batched call per tree level, " : (shared ? "callers on ONE context (their calls merged by the library), " : "")), threads, getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(unset: 4)");
  const int msgs = 2 * (ncl - 1);
  printf("  resident whole-tree program: first run %.1f ms, replayed %.1f ms = %.0f clique messages/s (+ %.1f ms to write and read every belief "
         "of the graph over PCIe, one batched call each way: %.0f messages/s);  one C call per clique, beliefs from and to host memory: %.1f ms = %.0f clique messages/s "
         "(walk %d of %d; the first, while the library's buffers grow: %.1f ms)\n",
         t_resident * 1e3, t_replay * 1e3, msgs / t_replay, t_io * 1e3, msgs / (t_replay + t_io), t_calls * 1e3, msgs / t_calls, nwalks, nwalks, t_first * 1e3);
  if (nwalks > 2) {
    printf("  walks in order:");
    for (int i = 0; i < nwalks; i++) printf(" %.1f", t_walk[i] * 1e3);
    printf(" ms\n");
  }
  if (queued)
    printf("  queued walk: all %d batches submitted after %.1f ms of host work (planning, assembly, enqueueing), the rest of the %.1f ms is waiting for the device\n",
           2 * maxdepth + 1, t_queued_calls * 1e3, t_calls * 1e3);
  if (seam_timing)
    printf("  phases of a walk with the device waited for after the launches (%.1f ms, %.0f calls): planning %.2f ms, beliefs in %.2f, program assembly %.2f, "
           "launches + device %.2f, beliefs out %.2f; the caller's own sub-graph assembly and bookkeeping %.2f\n", t_timed * 1e3, ph[5], ph[0] * 1e3, ph[1] * 1e3,
           ph[2] * 1e3, ph[3] * 1e3, ph[4] * 1e3, (t_timed - ph[0] - ph[1] - ph[2] - ph[3] - ph[4]) * 1e3);
  if (bctx) nbp_ctx_destroy(bctx);
  nbp_ctx_destroy(ctx);
  nbp_tree_destroy(tree);
  nbp_graph_destroy(g);
  return same == nvars && worst < 1.5 ? 0 : 3;
}
