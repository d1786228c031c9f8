/*
 * solve_chain.c -- the whole path through the C ABI alone (include/nbp.h + include/nbp_host.h), no Python:
 * build a ContinuousEuclid(2) odometry chain with periodic priors (the shape of BASELINE config 2),
 * let the native host initialise it (initAll!), order it, build the Bayes tree and compile the up+down
 * solve, run everything on the GPU and read the posteriors back.
 *
 *   gcc -O2 -Iinclude examples/solve_chain.c -o /tmp/solve_chain \
 *       -Lincrementalinference.jl_amd/csrc -lnbp -Wl,-rpath,$PWD/incrementalinference.jl_amd/csrc -lm
 *   /tmp/solve_chain [nvars=200] [N=200]
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nbp_host.h"

#define CHK(call)                                                                  \
  do {                                                                             \
    int rc_ = (call);                                                              \
    if (rc_ < 0) {                                                                 \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, nbp_last_error());             \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

static void gaussian_factor(nbp_factor_spec *f, int kind, int nvars, int a, int b, double mx, double my, double sigma) {
  memset(f, 0, sizeof(*f));
  f->factor_kind = kind;
  f->nvars = nvars;
  f->vars[0] = a;
  f->vars[1] = b;
  f->ncomp = 1;
  f->comp[0][0] = 1.0;                    /* weight */
  f->comp[0][1] = mx; f->comp[0][2] = my; /* mean   */
  f->comp[0][4] = sigma;                  /* lower Cholesky factor, row-major 3x3 at [4..12] */
  f->comp[0][4 + 3 * 1 + 1] = sigma;
}

int main(int argc, char **argv) {
  const int nvars = argc > 1 ? atoi(argv[1]) : 200, N = argc > 2 ? atoi(argv[2]) : 200;
  nbp_solver_params sp;
  memset(&sp, 0, sizeof(sp));
  sp.N = N; sp.gibbs_iters = 3; sp.inflate_cycles = 3; sp.product_niter = 1; sp.upsolve = sp.downsolve = 1;
  sp.spread_nh = 3.0; sp.inflation = 5.0; sp.null_surplus_add = 0.3;
  nbp_graph *g = NULL;
  CHK(nbp_graph_create(&sp, &g));
  for (int i = 0; i < nvars; i++) CHK(nbp_graph_add_variable(g, NBP_EUCLID2));
  nbp_factor_spec f;
  for (int i = 0; i < nvars; i++) {
    if (i % 50 == 0) { gaussian_factor(&f, NBP_F_PRIOR, 1, i, 0, i, i, 0.1); CHK(nbp_graph_add_factor(g, &f)); }
    if (i + 1 < nvars) { gaussian_factor(&f, NBP_F_LINREL, 2, i, i + 1, 1.0, 1.0, 0.1); CHK(nbp_graph_add_factor(g, &f)); }
  }
  int32_t *order = malloc(sizeof(int32_t) * nvars), *mainslot = malloc(sizeof(int32_t) * nvars);
  CHK(nbp_graph_order_nested_dissection(g, order));
  nbp_tree *tree = NULL;
  CHK(nbp_tree_build(g, order, nvars, &tree));
  const int n_slots = nbp_tree_plan_slots(tree, 0);
  CHK(n_slots);
  CHK(nbp_tree_main_slots(tree, mainslot, NULL));
  /* graph initialisation (initAll!): variable v lives in slot v for both programs */
  const int init_slots = nbp_graph_init_plan(g, 7);
  CHK(init_slots);
  nbp_ctx *ctx = NULL;
  CHK(nbp_ctx_create(0, N, n_slots > init_slots ? n_slots : init_slots, NULL, 0, 0, &ctx));
  double *pts = calloc(2 * (size_t)N, sizeof(double)), bw[2] = {1.0, 1.0};
  for (int i = 0; i < nvars; i++) CHK(nbp_slot_write(ctx, mainslot[i], NBP_EUCLID2, pts, bw)); /* identity points */
  nbp_program *init = NULL;
  CHK(nbp_graph_init_compile(g, ctx, &init));
  CHK(nbp_program_run(init, 0, -1));
  CHK(nbp_program_destroy(init));
  nbp_program *prog = NULL;
  CHK(nbp_tree_compile(tree, ctx, 2024, &prog));
  CHK(nbp_program_run(prog, 0, -1));
  CHK(nbp_synchronize(ctx));
  nbp_tree_stats st;
  CHK(nbp_tree_get_stats(tree, &st));
  double worst = 0;
  for (int i = 0; i < nvars; i++) {
    CHK(nbp_slot_read(ctx, mainslot[i], NBP_EUCLID2, pts, bw));
    double mx = 0, my = 0;
    for (int n = 0; n < N; n++) { mx += pts[2 * n]; my += pts[2 * n + 1]; }
    const double e = fmax(fabs(mx / N - i), fabs(my / N - i));
    if (e > worst) worst = e;
    if (!(bw[0] > 0) || !(bw[1] > 0)) { fprintf(stderr, "bad bandwidth at x%d\n", i); return 2; }
  }
  nbp_diag dg;
  CHK(nbp_diag_read(ctx, &dg, 0));
  printf("solve_chain: %d variables, %d cliques, %lld messages, %lld variable updates, %lld per-particle solves; "
         "worst posterior mean error %.3f\n", nvars, nbp_tree_num_cliques(tree), (long long)st.messages,
         (long long)(st.updates_up + st.updates_down), (long long)dg.solves, worst);
  nbp_program_destroy(prog);
  nbp_ctx_destroy(ctx);
  nbp_tree_destroy(tree);
  nbp_graph_destroy(g);
  free(order); free(mainslot); free(pts);
  return worst < 1.5 ? 0 : 3;
}
