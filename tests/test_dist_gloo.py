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
