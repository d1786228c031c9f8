"""Per-rank solve object used by bench.py (single GPU now; sharded multi-GPU in dist_solver)."""
import numpy as np


class RankSolve:
    def __init__(self, iif, nvars, N, rank, world, local, dist):
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
        order = iif.nestedDissectionOrder(fg)
        t2 = time.perf_counter()
        tree = iif.buildTreeReset(fg, order)
        t3 = time.perf_counter()
        mk = lambda n, s, side_ints=0: iif.HipBackend(n, s, side_ints=side_ints, device=self.local)
        iif.initAll(fg, backend=mk, seed=0)
        t4 = time.perf_counter()
        self.fg, self.tree = fg, tree
        tp = iif.TreeProgram(fg, tree, seed=1, snapshot=True)
        t5 = time.perf_counter()
        # host-side (Python) setup, outside the timed region; BASELINE.md 3 asks for the rate with and
        # without the tree build
        self.host_setup = {"graph_s": t1 - t0, "elimination_order_s": t2 - t1, "tree_build_s": t3 - t2,
                           "graph_init_s": t4 - t3, "schedule_compile_s": t5 - t4}
        self.tp = tp
        self.be = mk(self.N, tp.n_slots)
        for v in fg.ls():
            var = fg.getVariable(v)
            self.be.slot_write(tp.snap[v], var.varType.manifold, var.val, var.bw)
        self.prog = self.be.program(tp.stages)
        st = tp.stats()
        self.global_messages = tp.n_messages
        self.stats = {"cliques_global": st["cliques"], "updates_global": st["updates_up"] + st["updates_down"],
                      "alg_bytes": tp.alg_bytes_by_kernel()}

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
        tp, fg = self.tp, self.fg
        worst = 0.0
        for i in range(0, self.nvars, max(1, self.nvars // 64)):
            pts, _ = self.be.slot_read(tp.main[f"x{i}"], fg.getVariable(f"x{i}").varType.manifold)
            worst = max(worst, float(np.abs(pts.mean(axis=0) - i).max()))
        self.posterior_max_mean_err = worst
        # NBP posteriors carry Monte-Carlo error of the order of the posterior sigma (~0.5 midway
        # between priors); the CPU oracle shows the same level (tests/test_gpu_tree_parity.py)
        if not worst < 1.5:
            raise RuntimeError(f"posterior means off by {worst}: result invalid")
