"""N > 1 path on the CPU: world-size-2 gloo run of the sharded tree solve (oracle backend) must
reproduce the single-process solve exactly -- the op seeds do not depend on the rank, so only the
message plumbing (partition, ghost slots, exchange order) is under test."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import dist_worker
from iif_amd.dist_solver import partition_cliques
from oracle.oracle_backend import OracleBackend
from parity_utils import iif

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_partition_covers_tree_and_balances():
    fg = iif.generateChainEuclid(200, vardims=2, priorEvery=50, N=100)
    tree = iif.buildTreeReset(fg, iif.nestedDissectionOrder(fg))
    for world in (2, 4, 8):
        owner = partition_cliques(tree, world)
        assert set(owner) == set(tree.cliques)
        counts = np.bincount(list(owner.values()), minlength=world)
        assert counts.min() > 0 and counts.max() <= 2.0 * len(tree.cliques) / world
        cross = sum(1 for c, cl in tree.cliques.items() if cl.parent >= 0 and owner[c] != owner[cl.parent])
        assert cross <= 3 * world  # only the top of the tree crosses ranks (siblings of a level spread over ranks: a few more edges)


@pytest.mark.parametrize("mode", ["priors", "joint", "native"])  # native: this rank's share compiled by libnbp's C++ host
def test_two_rank_gloo_matches_single_process(tmp_path, mode):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    outs = [str(tmp_path / f"r{r}.npz") for r in range(2)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r), "2", port, outs[r], mode])
             for r in range(2)]
    for p in procs:
        assert p.wait(timeout=600) == 0
    # single-process reference with the same seeds
    fg, tree = dist_worker.build(mode == "joint")
    tp = iif.TreeProgram(fg, tree, seed=7)
    be = OracleBackend(100, tp.n_slots, 0, threads=4)
    for v in fg.ls():
        var = fg.getVariable(v)
        be.slot_write(tp.main[v], var.varType.manifold, var.val, var.bw)
    be.program(tp.stages).run()
    seen = set()
    nx = 0
    for o in outs:
        d = np.load(o)
        nx += int(d["n_exchanges"])
        assert int(d["n_messages"]) == tp.n_messages
        for k in d.files:
            if k.startswith("x") and not k.endswith("_bw"):
                pts, bw = be.slot_read(tp.main[k], fg.getVariable(k).varType.manifold)
                np.testing.assert_array_equal(d[k], pts)
                np.testing.assert_array_equal(d[k + "_bw"], bw)
                seen.add(k)
    assert seen == set(fg.ls())
    assert nx >= 2  # at least one up and one down exchange happened


@pytest.mark.parametrize("kind,world", [("lattice", 4), ("mixture", 4), ("chain", 8), ("lattice", 8), ("mixture", 8)])
def test_native_sharded_compile_on_more_ranks(tmp_path, kind, world):
    """4 and 8 gloo ranks on the share libnbp's C++ host compiles (partition, ghost slots, exchange segments), on the
    shapes of BASELINE configurations 4 (SE(2) lattice with loop closures) and 5 (Mixture chain): the particles of the
    single-process solve bit for bit, and SURVEY 8(e)'s bound on the exchange traffic -- per exchange point no more slots
    in flight than 2 x world (each 4.9 KB at N = 200: < 100 KB), every slot sent is received, the cut edges of the tree
    account for all of them (one slot per separator variable and direction)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = str(s.getsockname()[1])
    s.close()
    outs = [str(tmp_path / f"r{r}.npz") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker.py"), str(r), str(world), port, outs[r], "native", kind],
                              env=dict(os.environ, OMP_NUM_THREADS="1"))
             for r in range(world)]
    for p in procs:
        assert p.wait(timeout=1200) == 0
    fg, tree = dist_worker.build(False, kind)
    tp = iif.TreeProgram(fg, tree, seed=7)
    be = OracleBackend(100, tp.n_slots, 0, threads=4)
    for v in fg.ls():
        var = fg.getVariable(v)
        be.slot_write(tp.main[v], var.varType.manifold, var.val, var.bw)
    be.program(tp.stages).run()
    seen, counts = set(), []
    for o in outs:
        d = np.load(o)
        assert int(d["n_messages"]) == tp.n_messages
        counts.append(d["exchange_counts"])
        for k in d.files:
            if k.startswith("x") and not k.endswith("_bw"):
                pts, bw = be.slot_read(tp.main[k], fg.getVariable(k).varType.manifold)
                np.testing.assert_array_equal(d[k], pts)
                np.testing.assert_array_equal(d[k + "_bw"], bw)
                seen.add(k)
    assert seen == set(fg.ls())
    # a rank lists the exchange points it takes part in (both ends of a message agree on where it sits: the ranks compute
    # the stage times of the whole tree alike); over the whole solve every slot sent is received
    sent = sum(int(c[:, 0].sum()) for c in counts)
    assert sent == sum(int(c[:, 1].sum()) for c in counts) and sent >= 2
    per_point = max(int(c.sum(axis=1).max()) for c in counts if len(c))  # slots a rank moves at one exchange point
    assert per_point <= 2 * world, per_point
    assert per_point * (3 * 200 + 8) * 8 < 100 * 1024
    # all slots that move = the separator variables on the tree edges that cross a rank boundary, once up and once down
    from iif_amd import native_host
    g = native_host.NativeGraph.from_fg(fg)
    nt = g.build_tree(g.order_nested_dissection())
    owner = nt.partition(world)
    cut = 0
    for k in range(1, nt.n_cliques + 1):
        cl = nt.clique(k)
        if cl["parent"] > 0 and owner[k] != owner[cl["parent"]]:
            cut += len(cl["separators"])
    print(f"{kind}, world {world}: exchange points per rank {[len(c) for c in counts]}, at most {per_point} slots per rank and point, "
          f"{sent} slots moved in all, {cut} separator variables on cut edges")
    assert cut <= sent <= 2 * cut
