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
import os
import sys

import numpy as np

from . import abi
from .solver import TreeProgram


def partition_cliques(tree, world, weight=None):
    """Assign every clique to a rank: cut the tree into >= world subtrees by repeatedly splitting
    the heaviest one, place subtrees largest-first on the least loaded rank, then give every
    clique above the cut to the rank of one of its children, level by level (siblings of a level on
    different ranks where their children allow it)."""
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
    # the cliques above the cut, level by level (level = the longest chain of such cliques below): each goes to the rank
    # of one of its children (one of its edges stays local), and the cliques of one level -- which can run side by side --
    # go to different ranks where their children allow it
    level = {}
    for c in reversed(top):  # children before parents
        level[c] = max([level[ch] + 1 for ch in cl[c].children if ch in level], default=0)
    for lv in range(max(level.values(), default=-1) + 1):
        used = set()
        for c in reversed(top):
            if level[c] != lv:
                continue
            kids = sorted(cl[c].children, key=lambda ch: -sub[ch])  # stable: first maximum first
            pick = next((owner[ch] for ch in kids if owner[ch] not in used), owner[kids[0]])
            owner[c] = pick
            used.add(pick)
            load[pick] += w[c]
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

    def __init__(self, tp, backend, dist=None, slot_tensor=None, sync_device=None, transport="rccl", group=None, prog=None, native_tree=None):
        self.tp, self.be, self.dist = tp, backend, dist
        self.native_tree = native_tree  # native_host.NativeTree: the segment loop runs in C (nbp_tree_run_sharded)
        self.slot_tensor = slot_tensor
        self.sync_device = sync_device or (lambda: None)
        self.transport, self.group = transport, group
        # exchanges sit behind empty copy stages (barriers for the deferred bandwidth fits)
        self.prog = prog if prog is not None else backend.program(tp.stages, lazy_bandwidth=True)

    def run(self, salt=None):
        if salt is not None:
            self.prog.reseed(salt)
        if self.native_tree is not None:
            # the whole solve is one C call: stage ranges and exchanges issued back to back on the library stream (RCCL
            # from C), or the C loop calling back into this object's transport
            if self.transport == "rccl-native":
                self.native_tree.run_sharded(self.prog, self.be)
            else:
                self.native_tree.run_sharded(self.prog, exchange=self._exchange)
            return
        for seg in self.tp.segments:
            if seg[0] == "run":
                if seg[2] > seg[1]:
                    self.prog.run(seg[1], seg[2])
            else:
                self._exchange(seg[1], seg[2])

    def _exchange(self, sends, recvs):
        if self.transport == "rccl-native":
            # grouped ncclSend / ncclRecv issued by libnbp on its own stream: ordered with the kernels, no host sync
            self.be.exchange(sends, recvs)
            return
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


class _NativeShard:
    """what ShardedRunner needs from a compiled share: the segment list (native_host.NativeTree.segments())"""

    def __init__(self, segments):
        self.segments = segments


def native_comm(be, dist, device, log=None):
    """RCCL communicator owned by libnbp (nbp_comm_create): rank 0 makes the id, torch.distributed carries it; then a ring
    self-test of nbp_exchange on two scratch slots.  Returns True on every rank or False on every rank."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    ok = 1
    try:
        box = [be.comm_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        be.comm_create(world, rank, box[0])
        a, b = be.n_slots - 2, be.n_slots - 1  # the two slots the caller reserved for this
        from . import abi
        be.slot_write(a, abi.EUCLID1, np.full((be.N, 1), float(rank)), np.ones(1))
        be.exchange([((rank + 1) % world, a)], [((rank - 1) % world, b)])
        be.synchronize()
        got = be.slot_read(b, abi.EUCLID1)[0]
        if not np.all(got == float((rank - 1) % world)):
            ok = 0
    except Exception as e:  # noqa: BLE001 -- any failure means "do not use this transport"
        ok = 0
        if log:
            log(f"rank {rank}: libnbp RCCL exchange self-test failed ({type(e).__name__}: {e})")
    t = torch.tensor([ok], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item()) == 1


class ShardedTreeSolve:
    """bench.py's multi-GPU leg: ANY graph (every BASELINE configuration), its cliques sharded over the ranks, one
    process per GPU.  Everything behind the C ABI: the native host partitions the tree, compiles this rank's share and
    lists the exchange points (nbp_tree_partition / nbp_tree_set_owner / nbp_tree_segment); separator slots move with
    nbp_exchange (grouped RCCL send/recv on the library stream).  The arena is owned by torch so that the fallback
    transports (torch.distributed point-to-point, host-staged gloo) can address slots too.  Weak scaling is the
    caller's business (it hands over a graph grown with the number of ranks)."""

    def __init__(self, iif, fg, N, rank, world, local, dist):
        self.iif, self.fg, self.N = iif, fg, N
        self.rank, self.world, self.local, self.dist = rank, world, local, dist

    def prepare(self):
        import time

        import torch
        from . import native_host
        iif, fg = self.iif, self.fg
        dev = f"cuda:{self.local}"
        tm = time.perf_counter()
        g = native_host.NativeGraph.from_fg(fg)
        mirror = time.perf_counter() - tm
        t0 = time.perf_counter()
        order = g.order_nested_dissection()
        t1 = time.perf_counter()
        nt = g.build_tree(order)
        owner = nt.partition(self.world)
        nt.set_owner(owner, self.rank)
        t2 = time.perf_counter()
        n_slots = nt.plan_slots(True)
        need, _ = g.init_plan(0)  # replicated: every rank computes the same initial beliefs (graph initialisation is a chain)
        total = max(n_slots, need) + 2  # + two scratch slots for the transport self-test
        stride = abi.slot_stride(self.N)
        self.arena = torch.zeros(total * stride, dtype=torch.float64, device=dev)
        self.be = iif.HipBackend(self.N, total, device=self.local, arena_ptr=self.arena.data_ptr(), arena_bytes=self.arena.numel() * 8)
        for i, v in enumerate(fg.ls()):
            var = fg.getVariable(v)
            if var.initialized:
                self.be.slot_write(i, var.varType.manifold, var.val, var.bw)
        iprog = g.init_compile(self.be)
        iprog.run()
        self.be.synchronize()
        iprog.close()
        self.be.run_copies([abi.CopyDesc(nt.main[v], nt.snap[v]) for v in fg.ls()])
        t3 = time.perf_counter()
        prog = nt.compile(self.be, 1)
        self.tp = _NativeShard(nt.segments())
        self.tp.main = nt.main
        self.main = nt.main
        t4 = time.perf_counter()
        self.transport, group = ("none", None)
        self.rccl_ranks = None
        if self.dist is not None and self.world > 1:
            log = lambda m: print(m, file=sys.stderr, flush=True)
            if self.dist.get_backend() == "nccl" and native_comm(self.be, self.dist, dev, log):
                self.transport = "rccl-native"
                self.rccl_ranks = self.be.comm_info()[0]  # ncclCommCount of the library's own communicator
            else:
                self.transport, group = choose_transport(self.dist, dev, log=log)
        self.runner = ShardedRunner(self.tp, self.be, self.dist, lambda s: self.arena[s * stride:(s + 1) * stride],
                                    torch.cuda.synchronize, transport=self.transport, group=group, prog=prog, native_tree=nt)
        self.host_setup = {"host": "native C++ (nbp_host.h), sharded compile", "graph_s": None, "graph_mirror_s": mirror,
                           "elimination_order_s": t1 - t0, "tree_build_s": t2 - t1, "graph_init_s": t3 - t2,
                           "schedule_compile_s": t4 - t3}
        st = nt.stats()
        self.global_messages = st["messages"]
        alg = {"nbp_proposal_kernel": st["alg_bytes_proposal"], "nbp_prep_kernel": st["alg_bytes_prep"],
               "nbp_product_kernel": st["alg_bytes_product"], "nbp_bandwidth_kernel": 0}
        cdev = dev if (self.dist is None or self.dist.get_backend() == "nccl") else "cpu"  # gloo (tests): host tensors
        t = torch.tensor([float(st["updates_up"] + st["updates_down"]), float(st["alg_bytes"])], device=cdev, dtype=torch.float64)
        if self.dist is not None:
            self.dist.all_reduce(t)
        self.stats = {"cliques_global": nt.n_cliques, "updates_global": int(t[0].item()), "alg_bytes_total": float(t[1].item()), "alg_bytes": alg}
        self.n_cliques = nt.n_cliques
        self.mine = [v for k in range(1, nt.n_cliques + 1) if owner[k] == self.rank for v in nt.clique(k)["frontals"]]
        self._native = (g, nt)

    def step(self, k):
        self.runner.run(salt=0x9E37 + k + 1000 * int(os.environ.get("NBP_BENCH_SEED", "0")))

    def close(self):
        self.runner.close()
        self.be.close()
