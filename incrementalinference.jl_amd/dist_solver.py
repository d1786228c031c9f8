"""Multi-GPU tree solve: independent cliques of a tree level shard across ranks (one process per
GPU); only separator beliefs on tree edges that cross a rank boundary move, point to point.

This is the MI355X-native stand-in for the reference's only distributed mechanisms -- the
rendezvous `Channel{LikelihoodMessage}` per tree edge (JunctionTreeUtils.jl:943-956,
CliqueStateMachine.jl:221-234/617-629) and the optional `remotecall_fetch` of a whole clique
up-solve onto a `WorkerPool` (CliqStateMachineUtils.jl:369-385).  A message is the `TreeBelief`
payload (val N x P, bw, entities/BeliefTypes.jl:47-57) = one slot (4.9 KB at N = 200): traffic is
latency bound, so messages of one tree level are batched into a single group of point-to-point
`isend/irecv` (RCCL over xGMI with backend "nccl", gloo in the CPU tests) -- no ring collective.
"""
import sys

import numpy as np

from . import abi
from .solver import TreeProgram


def partition_cliques(tree, world, weight=None):
    """Assign every clique to a rank: cut the tree into >= world subtrees by repeatedly splitting
    the heaviest one, place subtrees largest-first on the least loaded rank, then give every
    clique above the cut to the rank of its heaviest child (one of its edges stays local)."""
    cl = tree.cliques
    w = {c: (1.0 if weight is None else float(weight(c))) for c in cl}
    sub = {}
    for c in tree.postorder():
        sub[c] = w[c] + sum(sub[ch] for ch in cl[c].children)
    roots = list(tree.roots)
    top = []
    total = sum(w.values())
    while len(roots) < 6 * world:
        splittable = [r for r in roots if cl[r].children]
        if not splittable:
            break
        r = max(splittable, key=lambda c: sub[c])
        if len(roots) >= world and sub[r] <= 0.6 * total / world:
            break  # fine enough for largest-first placement to balance the ranks
        roots.remove(r)
        top.append(r)
        roots.extend(cl[r].children)
    owner, load = {}, [0.0] * world
    for r in sorted(roots, key=lambda c: -sub[c]):
        k = int(np.argmin(load))
        load[k] += sub[r]
        stack = [r]
        while stack:
            c = stack.pop()
            owner[c] = k
            stack.extend(cl[c].children)
    for c in reversed(top):  # children before parents
        best = max(cl[c].children, key=lambda ch: sub[ch])
        owner[c] = owner[best]
        load[owner[c]] += w[c]
    return owner


def choose_transport(dist, device, log=None):
    """Agree (all ranks) on how separator slots travel.  "rccl": point-to-point on device memory, the
    design path (RCCL over xGMI).  "staged": through host memory over a gloo group -- only if a ring
    self-test of RCCL send/recv raises on ANY rank (the ranks agree through an all-reduce, so nobody
    is left waiting on a transport its peer gave up).  Returns (name, gloo_group_or_None)."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    if dist.get_backend() != "nccl":
        return "staged", None  # CPU tests: tensors already live on the host, default group is gloo
    gloo = dist.new_group(backend="gloo")  # collective: every rank creates it, used or not
    ok = 1
    try:
        a = torch.full((64,), float(rank), dtype=torch.float64, device=device)
        b = torch.empty(64, dtype=torch.float64, device=device)
        ops = [dist.P2POp(dist.isend, a, (rank + 1) % world), dist.P2POp(dist.irecv, b, (rank - 1) % world)]
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        torch.cuda.synchronize()
        if float(b[0].item()) != float((rank - 1) % world):
            ok = 0
    except Exception as e:  # noqa: BLE001 -- any failure means "do not use this transport"
        ok = 0
        if log:
            log(f"rank {rank}: RCCL point-to-point self-test failed ({type(e).__name__}: {e}); proposing host-staged exchange")
    t = torch.tensor([ok], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    if int(t.item()) == 1:
        return "rccl", None
    if log and rank == 0:
        log("separator exchange falls back to host-staged gloo send/recv on every rank")
    return "staged", gloo


class ShardedRunner:
    """Runs one rank's share of a TreeProgram: stage segments interleaved with slot exchanges."""

    def __init__(self, tp, backend, dist=None, slot_tensor=None, sync_device=None, transport="rccl", group=None):
        self.tp, self.be, self.dist = tp, backend, dist
        self.slot_tensor = slot_tensor
        self.sync_device = sync_device or (lambda: None)
        self.transport, self.group = transport, group
        self.prog = backend.program(tp.stages, lazy_bandwidth=True)  # exchanges sit behind empty copy stages (barriers)

    def run(self, salt=None):
        if salt is not None:
            self.prog.reseed(salt)
        for seg in self.tp.segments:
            if seg[0] == "run":
                if seg[2] > seg[1]:
                    self.prog.run(seg[1], seg[2])
            else:
                self._exchange(seg[1], seg[2])

    def _exchange(self, sends, recvs):
        dist = self.dist
        self.be.synchronize()  # the slots to send are complete
        ops, landing = [], []
        for peer, slot in sends:
            t = self.slot_tensor(slot)
            if t.is_cuda and self.transport == "staged":
                t = t.cpu()
            ops.append(dist.P2POp(dist.isend, t, peer, group=self.group))
        for peer, slot in recvs:
            t = self.slot_tensor(slot)
            if t.is_cuda and self.transport == "staged":
                h = t.cpu()
                landing.append((t, h))
                t = h
            ops.append(dist.P2POp(dist.irecv, t, peer, group=self.group))
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        for dev, host in landing:
            dev.copy_(host)
        self.sync_device()

    def close(self):
        self.prog.close()


class ShardedTreeSolve:
    """bench.py's multi-GPU leg: ANY graph (every BASELINE configuration), its cliques sharded over the ranks, one
    process per GPU, arena owned by torch so that RCCL can move slots.  Weak scaling is the caller's business (it
    hands over a graph grown with the number of ranks)."""

    def __init__(self, iif, fg, N, rank, world, local, dist):
        self.iif, self.fg, self.N = iif, fg, N
        self.rank, self.world, self.local, self.dist = rank, world, local, dist

    def prepare(self):
        import time

        import torch
        iif, fg = self.iif, self.fg
        t0 = time.perf_counter()
        order = iif.nestedDissectionOrder(fg)
        t1 = time.perf_counter()
        tree = iif.buildTreeReset(fg, order)
        t2 = time.perf_counter()
        mk = lambda n, s, side_ints=0: iif.HipBackend(n, s, side_ints=side_ints, device=self.local)
        iif.initAll(fg, backend=mk, seed=0)  # replicated: every rank computes the same initial beliefs
        t3 = time.perf_counter()
        self.tree = tree
        # balance by the work of a clique: the variable updates of its up schedule (wide separators iterate longer)
        from . import bayestree
        gi = fg.solverParams.gibbsIters
        owner = partition_cliques(tree, self.world, weight=lambda c: 1 + len(bayestree.upGibbsSchedule(tree.cliques[c], gi)))
        tp = TreeProgram(fg, tree, seed=1, snapshot=True, owner=owner, rank=self.rank)
        self.tp = tp
        stride = abi.slot_stride(self.N)
        self.arena = torch.zeros(tp.n_slots * stride, dtype=torch.float64, device=f"cuda:{self.local}")
        self.be = iif.HipBackend(self.N, tp.n_slots, device=self.local, arena_ptr=self.arena.data_ptr(),
                                 arena_bytes=self.arena.numel() * 8)
        for v in fg.ls():
            var = fg.getVariable(v)
            self.be.slot_write(tp.snap[v], var.varType.manifold, var.val, var.bw)
        self.transport, group = ("none", None)
        if self.dist is not None and self.world > 1:
            self.transport, group = choose_transport(self.dist, f"cuda:{self.local}", log=lambda m: print(m, file=sys.stderr, flush=True))
        self.runner = ShardedRunner(tp, self.be, self.dist, lambda s: self.arena[s * stride:(s + 1) * stride],
                                    torch.cuda.synchronize, transport=self.transport, group=group)
        t4 = time.perf_counter()
        self.host_setup = {"host": "python mirror (sharded compile)", "graph_s": None, "graph_mirror_s": 0.0,
                           "elimination_order_s": t1 - t0, "tree_build_s": t2 - t1, "graph_init_s": t3 - t2,
                           "schedule_compile_s": t4 - t3}
        st = tp.stats()
        self.global_messages = tp.n_messages
        # global totals over ranks
        keys = sorted(tp.alg)
        t = torch.tensor([float(st["updates_up"] + st["updates_down"]), float(tp.alg_bytes)] + [float(tp.alg[k]) for k in keys],
                         device=f"cuda:{self.local}", dtype=torch.float64)
        if self.dist is not None:
            self.dist.all_reduce(t)
        self.stats = {"cliques_global": len(tree.cliques), "updates_global": int(t[0].item()), "alg_bytes_total": float(t[1].item()),
                      "alg_bytes": dict(tp.alg), "alg_bytes_global": {k: float(t[2 + i].item()) for i, k in enumerate(keys)}}
        self.mine = [v for c in tp.cliques for v in tree.cliques[c].frontalIDs]

    def step(self, k):
        self.runner.run(salt=0x9E37 + k)

    def close(self):
        self.runner.close()
        self.be.close()
