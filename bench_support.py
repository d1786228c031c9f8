"""Per-rank solve object used by bench.py (single GPU now; sharded multi-GPU in dist_solver)."""
import numpy as np


class RankSolve:
    def __init__(self, iif, nvars, N, rank, world, local, dist, python_host=False):
        self.python_host = python_host
        self.iif, self.nvars, self.N = iif, nvars, N
        self.rank, self.world, self.local, self.dist = rank, world, local, dist

    def prepare(self):
        iif = self.iif
        self.sharded = self.world > 1 or self.dist is not None
        if self.sharded:
            from iif_amd.dist_solver import ShardedTreeSolve
            self.impl = ShardedTreeSolve(iif, self.nvars * self.world, self.N, self.rank, self.world, self.local, self.dist)
            self.impl.prepare()
            self.be = self.impl.be
            self.global_messages = self.impl.global_messages
            self.stats = self.impl.stats
            return
        import time
        t0 = time.perf_counter()
        fg = iif.generateChainEuclid(self.nvars, vardims=2, priorEvery=100, N=self.N)
        t1 = time.perf_counter()
        mk = lambda n, s, side_ints=0: iif.HipBackend(n, s, side_ints=side_ints, device=self.local)
        self.fg = fg
        if self.python_host:
            order = iif.nestedDissectionOrder(fg)
            t2 = time.perf_counter()
            tree = iif.buildTreeReset(fg, order)
            t3 = time.perf_counter()
            iif.initAll(fg, backend=mk, seed=0)
            t4 = time.perf_counter()
            tp = iif.TreeProgram(fg, tree, seed=1, snapshot=True)
            self.be = mk(self.N, tp.n_slots)
            self.prog = self.be.program(tp.stages, lazy_bandwidth=True)
            t5 = time.perf_counter()
            self.main, snap, st = tp.main, tp.snap, tp.stats()
            self.global_messages = tp.n_messages
            alg = tp.alg_bytes_by_kernel()
            mirror = 0.0
        else:
            # native host (include/nbp_host.h): ordering, Bayes tree, Gibbs schedules and the stage
            # descriptors are built in C++; Python only mirrors the graph into it
            from iif_amd import native_host
            iif.initAll(fg, backend=mk, seed=0)
            t_init = time.perf_counter() - t1
            tm = time.perf_counter()
            g = native_host.NativeGraph.from_fg(fg)
            mirror = time.perf_counter() - tm
            t1 = time.perf_counter()
            order = g.order_nested_dissection()
            t2 = time.perf_counter()
            nt = g.build_tree(order)
            t3 = time.perf_counter()
            t4 = t3 + t_init  # keeps graph_init_s = t4 - t3 below
            n_slots = nt.plan_slots(True)
            self.be = mk(self.N, n_slots)
            ta = time.perf_counter()
            self.prog = nt.compile(self.be, 1)
            t5 = t4 + (time.perf_counter() - ta) + (ta - t3)
            self.main, snap, st = nt.main, nt.snap, nt.stats()
            self.global_messages = st["messages"]
            st["cliques"] = nt.n_cliques
            alg = {"nbp_proposal_kernel": st["alg_bytes_proposal"], "nbp_prep_kernel": st["alg_bytes_prep"],
                   "nbp_product_kernel": st["alg_bytes_product"], "nbp_bandwidth_kernel": 0}
            self._native = (g, nt)
        # host-side setup, outside the timed region; BASELINE.md 3 asks for the rate with and without the tree build
        self.host_setup = {"host": "python" if self.python_host else "native C++ (nbp_host.h)", "graph_s": (t1 - t0) if self.python_host else None,
                           "graph_mirror_s": mirror, "elimination_order_s": t2 - t1, "tree_build_s": t3 - t2,
                           "graph_init_s": t4 - t3, "schedule_compile_s": t5 - t4}
        for v in fg.ls():
            var = fg.getVariable(v)
            self.be.slot_write(snap[v], var.varType.manifold, var.val, var.bw)
        self.stats = {"cliques_global": st["cliques"], "updates_global": st["updates_up"] + st["updates_down"], "alg_bytes": alg}

    def step(self, k):
        if self.sharded:
            return self.impl.step(k)
        self.prog.reseed(0x9E37 + k)
        self.prog.run()

    def check_posteriors(self):
        """posteriors within the BASELINE.md tolerance of the ground truth x_i = (i, i)"""
        if self.sharded:
            self.impl.check_posteriors()
            self.posterior_max_mean_err = self.impl.posterior_max_mean_err
            return
        fg = self.fg
        worst = 0.0
        for i in range(0, self.nvars, max(1, self.nvars // 64)):
            pts, _ = self.be.slot_read(self.main[f"x{i}"], fg.getVariable(f"x{i}").varType.manifold)
            worst = max(worst, float(np.abs(pts.mean(axis=0) - i).max()))
        self.posterior_max_mean_err = worst
        # NBP posteriors carry Monte-Carlo error of a fraction of the posterior sigma (~0.5-0.7 midway
        # between priors; observed worst mean error ~0.3); the CPU oracle shows the same level
        if not worst < 1.0:
            raise RuntimeError(f"posterior means off by {worst}: result invalid")
