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
