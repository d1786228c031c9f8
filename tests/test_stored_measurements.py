"""alwaysFreshMeasurements = false (SolveTree.jl:119, CalcFactor.jl:492-510): Gibbs iterations after the
first reuse the measurement samples the factor drew last (`meas_seed` in nbp_proposal_desc)."""
import numpy as np

from parity_utils import abi, iif, rand_points, relative_factor_desc
from oracle.oracle_backend import OracleBackend


def _prior(seed, out, meas_seed=0, ncomp=1):
    d = relative_factor_desc(abi.F_PRIOR, abi.EUCLID2, 1, 0, [0], out, seed, [1.0, -1.0], [0.5, 0.2])
    d.meas_seed = meas_seed
    return d


def check_stored_measurements(backend):
    N = 100
    be = backend(N, 6, 0)
    be.slot_write(0, abi.EUCLID2, rand_points(np.random.default_rng(0), abi.EUCLID2, N))
    # a prior's proposal IS its measurement: same meas_seed -> same points, whatever the op's own seed
    prog = be.program([(abi.STAGE_PROPOSALS, [_prior(11, 1), _prior(12, 2), _prior(13, 3, meas_seed=11)])])
    prog.run()
    a, b, c = (be.slot_read(s, abi.EUCLID2)[0] for s in (1, 2, 3))
    assert not np.allclose(a, b) and np.array_equal(a, c)
    # re-keying the program re-keys the reference too: the third op still reuses the first one's samples
    prog.reseed(99)
    prog.run()
    a2, b2, c2 = (be.slot_read(s, abi.EUCLID2)[0] for s in (1, 2, 3))
    assert not np.allclose(a2, a) and np.array_equal(a2, c2) and not np.allclose(a2, b2)
    prog.close()
    # a relative factor with a stored measurement: the same (measurement, other point) pairs, but its own
    # entropy stream -- the results agree to the solver tolerance, not bit for bit
    be.slot_write(4, abi.EUCLID2, rand_points(np.random.default_rng(1), abi.EUCLID2, N, 3.0, 0.2))
    r1 = relative_factor_desc(abi.F_LINREL, abi.EUCLID2, 2, 1, [0, 4], 1, 21, [1.0, 1.0], [0.3, 0.3])
    r2 = relative_factor_desc(abi.F_LINREL, abi.EUCLID2, 2, 1, [0, 4], 2, 22, [1.0, 1.0], [0.3, 0.3])
    r3 = relative_factor_desc(abi.F_LINREL, abi.EUCLID2, 2, 1, [0, 4], 3, 23, [1.0, 1.0], [0.3, 0.3])
    r3.meas_seed = 21
    be.run_proposals([r1, r2, r3])
    p1, p2, p3 = (be.slot_read(s, abi.EUCLID2)[0] for s in (1, 2, 3))
    assert np.abs(p1 - p3).max() < 5e-3 < np.abs(p1 - p2).max()
    be.close()


def test_stored_measurements_oracle():
    check_stored_measurements(OracleBackend)


def test_schedule_marks_reused_measurements():
    fg = iif.generateChainEuclid(6, vardims=2, priorEvery=3, N=100)
    fg.solverParams.alwaysFreshMeasurements = False
    tree = iif.buildTreeReset(fg, iif.nestedDissectionOrder(fg))
    tp = iif.TreeProgram(fg, tree, seed=5)
    props = [d for k, ds in tp.stages if k == abi.STAGE_PROPOSALS for d in ds]
    seeds = {d.seed for d in props}
    reused = [d for d in props if d.meas_seed]
    assert reused, "iterations 2 and 3 of the itervar passes must reuse measurements"
    assert all(d.meas_seed in seeds and d.meas_seed != d.seed for d in reused)
    # per clique: 2 of 3 sweeps over the iteration variables reuse; the down pass is always fresh
    assert all(fr[0] for fr in tp.upfresh.values() if fr)
    fg.solverParams.alwaysFreshMeasurements = True
    tp2 = iif.TreeProgram(fg, tree, seed=5)
    assert not any(d.meas_seed for k, ds in tp2.stages if k == abi.STAGE_PROPOSALS for d in ds)
    assert [len(d) for _, d in tp.stages] == [len(d) for _, d in tp2.stages]


def test_chain_solves_with_stored_measurements():
    # the acceptance band of config 1 (test/testBasicGraphs.jl:351-352) also holds in this mode
    fg = iif.initfg(iif.SolverParams(N=100, alwaysFreshMeasurements=False))
    for i in range(6):
        iif.addVariable(fg, f"x{i}", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal(0.0, 1.0)))
    for i in range(5):
        iif.addFactor(fg, [f"x{i}", f"x{i + 1}"], iif.LinearRelative(iif.Normal(1.0, 0.1)))
    iif.solveTree(fg, backend=OracleBackend, seed=3)
    m0 = fg.getVal("x0")[:, 0].mean()
    for i in range(6):
        assert abs(fg.getVal(f"x{i}")[:, 0].mean() - m0 - i) < 0.1 * (1 + i / 1.5) + 0.15
