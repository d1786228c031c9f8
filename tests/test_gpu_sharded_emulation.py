"""-m gpu: two ranks of a sharded tree solve emulated on ONE GPU (two contexts, the separator slots moved
between their arenas at the exchange points exactly as dist_solver.ShardedRunner does over RCCL): the
posteriors must equal the single-rank solve bit for bit -- this covers the rank-sharded schedule, the
ghost slots, the exchange barriers and the dead-bandwidth liveness analysis across them."""
import numpy as np
import pytest

from parity_utils import abi, iif

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,nvars,joint", [(2, 48, False), (4, 48, False), (8, 128, False), (2, 48, True), (4, 64, True)])
def test_emulated_ranks_equal_single_rank(hip_backend, world, nvars, joint):
    from iif_amd.dist_solver import partition_cliques
    N = 100
    fg = iif.generateChainEuclid(nvars, vardims=2, priorEvery=8, N=N)
    fg.solverParams.useMsgLikelihoods = joint  # joint messages: the differential-factor KDEs travel too
    iif.initAll(fg, backend=hip_backend, seed=0)
    tree = iif.buildTreeReset(fg, iif.nestedDissectionOrder(fg))
    man = abi.EUCLID2

    def load(be, tp):
        for v in fg.ls():
            var = fg.getVariable(v)
            be.slot_write(tp.main[v], man, var.val, var.bw)

    # reference: one rank
    tp = iif.TreeProgram(fg, tree, seed=9)
    be = hip_backend(N, tp.n_slots)
    load(be, tp)
    prog = be.program(tp.stages, lazy_bandwidth=True)
    prog.run()
    be.synchronize()
    ref = {v: be.slot_read(tp.main[v], man) for v in fg.ls()}
    prog.close()
    be.close()

    owner = partition_cliques(tree, world)
    assert len(set(owner.values())) == world
    tps = [iif.TreeProgram(fg, tree, seed=9, owner=owner, rank=r) for r in range(world)]
    bes = [hip_backend(N, t.n_slots) for t in tps]
    progs = []
    for b, t in zip(bes, tps):
        load(b, t)
        progs.append(b.program(t.stages, lazy_bandwidth=True))
    # cooperative emulation of the ranks: a rank at an exchange posts its sends into per-pair FIFO
    # mailboxes, then waits until everything it must receive has been posted (what the batched
    # isend/irecv group does); ranks that take no part in an exchange have no such segment at all
    import collections
    mail = collections.defaultdict(collections.deque)  # (src, dst) -> payloads in send order
    pos, sent, nx = [0] * world, [False] * world, 0
    for _ in range(10000):
        progress = False
        for r, t in enumerate(tps):
            if pos[r] >= len(t.segments):
                continue
            seg = t.segments[pos[r]]
            if seg[0] == "run":
                if seg[2] > seg[1]:
                    progs[r].run(seg[1], seg[2])
                pos[r] += 1
                progress = True
                continue
            if not sent[r]:
                bes[r].synchronize()
                for peer, slot in seg[1]:
                    pts, bw = bes[r].slot_read(slot, man)
                    assert (bw > 0).all()  # the message carries a fitted bandwidth (the barrier kept it alive)
                    mail[(r, peer)].append((pts, bw))
                sent[r] = True
                progress = True
            need = collections.Counter(peer for peer, _ in seg[2])
            if all(len(mail[(q, r)]) >= n for q, n in need.items()):
                for peer, slot in seg[2]:
                    pts, bw = mail[(peer, r)].popleft()
                    bes[r].slot_write(slot, man, pts, bw)
                    nx += 1
                pos[r] += 1
                sent[r] = False
                progress = True
        if all(pos[r] >= len(t.segments) for r, t in enumerate(tps)):
            break
        assert progress, "emulated ranks deadlocked"
    assert all(not q for q in mail.values())
    assert nx > 0
    for b in bes:
        b.synchronize()
    for c, r in owner.items():
        for v in tree.cliques[c].frontalIDs:
            pts, bw = bes[r].slot_read(tps[r].main[v], man)
            np.testing.assert_array_equal(pts, ref[v][0])
            np.testing.assert_array_equal(bw, ref[v][1])
    for pr in progs:
        pr.close()
    for b in bes:
        b.close()


@pytest.mark.parametrize("kind,world", [("lattice", 8), ("mixture", 8), ("lattice", 4)])
def test_world8_exchange_lists_through_rccl_to_self(hip_backend, kind, world):
    """The exchange segments of a BASELINE-shaped graph (config 4: SE(2) lattice with loop closures; config 5: Mixture chain)
    compiled by the NATIVE host for every rank of a world of 8, replayed one exchange point at a time through the code path a
    multi-GPU run takes -- nbp_exchange: ONE ncclGroupStart / ncclGroupEnd of ncclSend / ncclRecv of whole slots on the
    library stream -- with a world of one, every peer replaced by this rank: the real slot lists, the real group sizes.  What
    a one-GPU box can check of the RCCL leg before the first multi-GPU run: every slot arrives byte for byte (points, bandwidth,
    infoPerCoord, count), in list order, with consumers on the same stream and no host synchronisation in between."""
    from iif_amd import native_host
    N = 64
    fg = (iif.generateSE2Lattice(rows=16, cols=40, N=N, closeEvery=5) if kind == "lattice"
          else iif.generateMixtureChain(nvars=800, N=N, priorEvery=400))
    for v in fg.ls():  # (the compile wants initialised variables; the values do not matter here)
        var = fg.getVariable(v)
        iif.setValKDE(fg, v, np.zeros((N, abi.MANIFOLD_P[var.varType.manifold])) + (np.array([0, 0, 1, 0, 0, 1.0]) if var.varType.manifold == abi.SE2 else 0.0),
                      np.full(var.varType.dim, 0.1))
    lists, n_slots = [], 0
    for r in range(world):
        g = native_host.NativeGraph.from_fg(fg)
        nt = g.build_tree(g.order_nested_dissection())
        owner = nt.partition(world)
        assert len(set(owner.values())) == world
        nt.set_owner(owner, r)
        n_slots = max(n_slots, nt.plan_slots(False))
        nt.schedule(7)
        lists.append([seg for seg in nt.segments() if seg[0] == "xchg"])
    # every message has two ends that agree: what rank a sends to b at its k-th exchange with b, b receives from a
    sent = sum(len(s[1]) for l in lists for s in l)
    assert sent == sum(len(s[2]) for l in lists for s in l) and sent >= world - 1
    widest = max(max(len(s[1]), len(s[2])) for l in lists for s in l)
    be = hip_backend(N, n_slots + widest)
    try:
        be.comm_create(1, 0, be.comm_unique_id())
        rng = np.random.default_rng(3)
        man = abi.EUCLID3  # reads and writes all three rows of a slot, whatever the belief in it is
        groups = 0

        def pattern(k):
            return rng.normal(size=(N - (k % 3), 3)), rng.uniform(0.1, 1.0, 3), np.array([1.0 + k, 2.0, 3.0])  # (a count below N travels too)

        def check(slot, want):
            pts, bw, ipc = be.belief_read(slot, man)
            np.testing.assert_array_equal(pts, want[0])
            np.testing.assert_array_equal(bw, want[1])
            np.testing.assert_array_equal(ipc, want[2])

        for r, segs in enumerate(lists):
            for _, sends, recvs in segs:
                assert all(0 <= p < world and p != r for p, _ in sends + recvs)
                # the sending side of this exchange point: the real send slots, in list order, into landing slots of the test's own
                if sends:
                    by_slot = {}  # (a separator belief may go to several peers: one pattern per SLOT)
                    for k, (_, s) in enumerate(sends):
                        by_slot.setdefault(s, pattern(k))
                    want = [by_slot[s] for _, s in sends]
                    for s, w in by_slot.items():
                        be.belief_write(s, man, *w)
                    be.exchange([(0, s) for _, s in sends], [(0, n_slots + k) for k in range(len(sends))])
                    be.run_copies([abi.CopyDesc(n_slots, n_slots)])  # a kernel behind it on the same stream, no synchronize
                    for k, w in enumerate(want):
                        check(n_slots + k, w)
                    groups += 1
                # the receiving side: into the real landing (ghost) slots, in list order
                if recvs:
                    want = [pattern(k + 11) for k in range(len(recvs))]
                    for k, w in enumerate(want):
                        be.belief_write(n_slots + k, man, *w)
                    be.exchange([(0, n_slots + k) for k in range(len(recvs))], [(0, s) for _, s in recvs])
                    for (_, s), w in zip(recvs, want):
                        check(s, w)
                    groups += 1
        print(f"{kind}, world {world}: {sum(len(l) for l in lists)} exchange points over the ranks, {sent} slots each way, at most {widest} in one group; "
              f"{groups} grouped self-exchanges byte-identical")
    finally:
        be.close()
