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


def build(joint=False):
    fg = iif.generateChainEuclid(24, vardims=2, priorEvery=8, N=100)
    fg.solverParams.useMsgLikelihoods = joint
    for v in fg.ls():  # deterministic synthetic "initialised" beliefs (no initAll needed here)
        i = int(v[1:])
        rng = np.random.default_rng(i)
        iif.setValKDE(fg, v, rng.normal(size=(100, 2)) * 0.3 + i, np.array([0.1, 0.1]))
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
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fg, tree = build(mode == "joint")
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
    np.savez(out, n_exchanges=nx, n_messages=tp.n_messages, **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
