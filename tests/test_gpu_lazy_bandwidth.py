"""-m gpu: NBP_OPT_LAZY_BANDWIDTH leaves out the bandwidth fits whose result nothing reads (the reference fits on every
setBelief!, FactorGraph.jl:250-263; inside a clique's Gibbs sweeps only the last fit of a variable is ever looked at).
Dead-work elimination, not skipped work: the same program with and without the option gives the same posteriors -- points
AND bandwidths -- bit for bit, on every graph family; the share of leave-one-out evaluations it saves is recorded."""
import numpy as np
import pytest

from parity_utils import iif, record_parity

pytestmark = pytest.mark.gpu

GRAPHS = {
    "chain": lambda: iif.generateChainEuclid(60, vardims=2, priorEvery=10, N=100),
    "lattice": lambda: iif.generateSE2Lattice(rows=3, cols=6, N=100, closeEvery=2),
    "doors": lambda: iif.generateCircularDoors(nposes=50, N=100, sightEvery=5),
    "mixture": lambda: iif.generateMixtureChain(nvars=30, N=100, priorEvery=10),
}


@pytest.mark.parametrize("name", list(GRAPHS))
def test_lazy_bandwidth_changes_no_posterior(hip_backend, name):
    fg = GRAPHS[name]()
    iif.initAll(fg, backend=hip_backend, seed=0)
    tp = iif.TreeProgram(fg, iif.buildTreeReset(fg, iif.nestedDissectionOrder(fg)), seed=77)
    res, evals = [], []
    for lazy in (True, False):
        be = hip_backend(100, tp.n_slots)
        prog = be.program(tp.stages, lazy_bandwidth=lazy)
        for v in fg.ls():
            var = fg.getVariable(v)
            be.slot_write(tp.main[v], var.varType.manifold, var.val, var.bw)
        be.diag(reset=True)
        prog.run()
        be.synchronize()
        evals.append(be.diag()["lcv_evals"])
        res.append({v: be.slot_read(tp.main[v], fg.getVariable(v).varType.manifold) for v in fg.ls()})
        prog.close()
        be.close()
    for v in fg.ls():
        np.testing.assert_array_equal(res[0][v][0], res[1][v][0])
        np.testing.assert_array_equal(res[0][v][1], res[1][v][1])
    assert evals[0] < evals[1]
    line = f"lazy bandwidth, {name}: posteriors bit-identical; leave-one-out evaluations {evals[0]} instead of {evals[1]} ({1 - evals[0] / evals[1]:.0%} never read)"
    print(line)
    record_parity(line)
