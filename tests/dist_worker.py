"""Worker for tests/test_dist_gloo.py: one rank of a 2-process sharded tree solve on the CPU
(gloo backend, oracle backend).  Writes the posteriors of the variables this rank owns."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import iif_amd_loader  # noqa: E402

iif = iif_amd_loader.load()
from iif_amd.dist_solver import ShardedRunner, choose_transport, partition_cliques  # noqa: E402
from oracle.oracle_backend import OracleBackend  # noqa: E402


def build(joint=False, kind="chain"):
    """kind: "chain" (config 2 shape), "lattice" (config 4: SE(2) boustrophedon lattice with loop closures), "mixture"
    (config 5: Euclid(3) chain of Mixture factors); deterministic synthetic "initialised" beliefs (no initAll needed)"""
    if kind == "chain":
        fg = iif.generateChainEuclid(24, vardims=2, priorEvery=8, N=100)
    elif kind == "lattice":
        fg = iif.generateSE2Lattice(rows=3, cols=8, N=100, closeEvery=3)
    else:
        fg = iif.generateMixtureChain(nvars=40, N=100, priorEvery=10)
    fg.solverParams.useMsgLikelihoods = joint
    for v in fg.ls():
        i = int(v[1:])
        rng = np.random.default_rng(i)
        if kind == "chain":
            iif.setValKDE(fg, v, rng.normal(size=(100, 2)) * 0.3 + i, np.array([0.1, 0.1]))
        elif kind == "lattice":
            r, c = divmod(i, 8)
            c = c if r % 2 == 0 else 7 - c
            th = (0.0 if r % 2 == 0 else np.pi) + rng.normal(size=100) * 0.05
            xy = rng.normal(size=(100, 2)) * 0.2 + np.array([c, r], dtype=float)
            iif.setValKDE(fg, v, np.stack([xy[:, 0], xy[:, 1], np.cos(th), np.sin(th), -np.sin(th), np.cos(th)], axis=1),
                          np.array([0.1, 0.1, 0.03]))
        else:
            iif.setValKDE(fg, v, rng.normal(size=(100, 3)) * 0.4 + np.array([float(i), 0.0, 0.0]), np.array([0.15, 0.15, 0.15]))
    return fg, iif.buildTreeReset(fg, iif.nestedDissectionOrder(fg))


class NativeShare:
    """this rank's share as the NATIVE host compiles it (nbp_tree_partition / nbp_tree_set_owner / nbp_tree_schedule:
    descriptors, slot plan and exchange segments from libnbp's C++ host, no device needed), in the shape ShardedRunner
    and this worker read from a solver.TreeProgram"""

    def __init__(self, fg, world, rank, seed):
        import ctypes as C
        from iif_amd import abi, native_host
        g = native_host.NativeGraph.from_fg(fg)
        nt = g.build_tree(g.order_nested_dissection())
        self.owner = nt.partition(world)
        nt.set_owner(self.owner, rank)
        self.n_slots = nt.plan_slots(False)
        nt.schedule(seed)
        ctype = {abi.STAGE_PROPOSALS: abi.ProposalDesc, abi.STAGE_PRODUCTS: abi.ProductDesc, abi.STAGE_COPIES: abi.CopyDesc,
                 abi.STAGE_DECONV: abi.ProposalDesc, abi.STAGE_COPY_POINTS: abi.CopyDesc}
        self.stages = []
        for kind, raw in nt.stages():
            n = len(raw) // C.sizeof(ctype[kind])
            self.stages.append((kind, list((ctype[kind] * n).from_buffer_copy(raw)) if n else []))
        self.segments = nt.segments()
        self.main = nt.main
        self.n_messages = nt.stats()["messages"]
        self.cliques = [k for k in range(1, nt.n_cliques + 1) if self.owner[k] == rank]
        self.frontals = {k: nt.clique(k)["frontals"] for k in self.cliques}
        self._keep = (g, nt)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    mode = sys.argv[5] if len(sys.argv) > 5 else "priors"
    kind = sys.argv[6] if len(sys.argv) > 6 else "chain"
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fg, tree = build(mode == "joint", kind)
    if mode == "native":
        tp = NativeShare(fg, world, rank, 7)
        frontals = tp.frontals
    else:
        owner = partition_cliques(tree, world)
        tp = iif.TreeProgram(fg, tree, seed=7, owner=owner, rank=rank)
        frontals = {c: tree.cliques[c].frontalIDs for c in tp.cliques}
    be = OracleBackend(100, tp.n_slots, 0, threads=2)
    for v in fg.ls():
        var = fg.getVariable(v)
        be.slot_write(tp.main[v], var.varType.manifold, var.val, var.bw)
    stride = iif.abi.slot_stride(100)
    arena_t = torch.from_numpy(be.arena)
    transport, group = choose_transport(dist, "cpu")
    assert transport == "staged" and group is None  # gloo default group, host tensors
    runner = ShardedRunner(tp, be, dist, lambda s: arena_t[s * stride:(s + 1) * stride], transport=transport, group=group)
    runner.run()
    res = {}
    for c in tp.cliques:
        for v in frontals[c]:
            pts, bw = be.slot_read(tp.main[v], fg.getVariable(v).varType.manifold)
            res[v] = pts
            res[v + "_bw"] = bw
    nx = sum(1 for s in tp.segments if s[0] == "xchg")
    # every exchange point of the solve as this rank sees it: (slots sent, slots received), empty ones included
    xc = np.array([(len(s[1]), len(s[2])) for s in tp.segments if s[0] == "xchg"], dtype=np.int64).reshape(-1, 2)
    np.savez(out, n_exchanges=nx, n_messages=tp.n_messages, exchange_counts=xc, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
