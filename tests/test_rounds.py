"""CPU: rounds of commuting Gibbs steps (solver.TreeProgram._rounds; native twin checked byte for byte in
tests/test_native_host.py).  The reference runs the steps of a clique one after the other (fmcmc!,
SolveTree.jl:97-160); steps whose variables differ and share no factor read and write disjoint beliefs, so they may
share a stage.  The equality of the particles with the step-by-step schedule is tested on random graphs in
tests/test_joint_messages.py (asap vs level compile, oracle backend, bit for bit)."""
import numpy as np
import pytest

from parity_utils import abi, iif


def program(fg, **kw):
    for v in fg.ls():
        fg.getVariable(v).initialized = True
    tree = iif.buildTreeReset(fg, iif.nestedDissectionOrder(fg))
    return tree, iif.TreeProgram(fg, tree, seed=1, **kw)


def test_chain_clique_runs_six_rounds_instead_of_nine():
    fg = iif.generateChainEuclid(200, vardims=2, priorEvery=50, N=50)
    tree, tp = program(fg)
    leaves = [c for c, cl in tree.cliques.items() if not cl.children and len(cl.frontalIDs) == 1 and len(cl.separatorIDs) == 2]
    assert leaves
    for c in leaves:  # {x_k | x_k-1, x_k+1}: the two separators do not share a factor, the frontal shares one with each
        sched, rounds = tp.upsched[c], tp.uprounds[c]
        assert len(sched) == 9 and sorted(len(r) for r in rounds) == [1, 1, 1, 2, 2, 2]
        f = tree.cliques[c].frontalIDs[0]
        for r in rounds:
            assert (len(r) == 1) == (sched[r[0]] == f)  # the frontal alone, its two neighbours side by side
    n_seq = sum(1 for k, _ in iif.TreeProgram(fg, tree, seed=1, asap=False).stages if k == abi.STAGE_PRODUCTS)
    n_rnd = sum(1 for k, _ in tp.stages if k == abi.STAGE_PRODUCTS)
    assert n_rnd < 0.7 * n_seq


@pytest.mark.parametrize("name", ["chain", "lattice", "doors", "mixture", "kaess"])
def test_rounds_respect_every_dependency(name):
    fg = {"chain": lambda: iif.generateChainEuclid(60, vardims=2, priorEvery=10, N=50),
          "lattice": lambda: iif.generateSE2Lattice(rows=4, cols=7, N=50, closeEvery=2),
          "doors": lambda: iif.generateCircularDoors(nposes=60, N=50, sightEvery=5),
          "mixture": lambda: iif.generateMixtureChain(nvars=30, N=50, priorEvery=7),
          "kaess": lambda: iif.generateGraph_Kaess(iif.SolverParams(N=50))}[name]()
    tree, tp = program(fg)
    for c in tree.cliques:
        for sched, facs, rounds in ((tp.upsched[c], tp.upfacs[c], tp.uprounds[c]), (tp.dnsched[c], tp.dnfacs[c], tp.dnrounds[c])):
            assert sorted(k for r in rounds for k in r) == list(range(len(sched)))  # every step once
            rnd = {k: i for i, r in enumerate(rounds) for k in r}
            reads = {v: {u for e in facs[v] for u in tp._entry_variables(e)} - {v} for v in set(sched)}
            for j in range(len(sched)):
                for i in range(j):
                    u, v = sched[i], sched[j]
                    if u == v or u in reads[v] or v in reads[u]:
                        assert rnd[i] < rnd[j], (c, i, j)   # conflicting steps keep their order
            for r in rounds:
                assert len({sched[k] for k in r}) == len(r)  # a variable once per round
    # scratch rows: the steps of a round write disjoint proposal slots
    for kind, descs in tp.stages:
        if kind == abi.STAGE_PROPOSALS:
            outs = [d.out_slot for d in descs]
            assert len(set(outs)) == len(outs)
