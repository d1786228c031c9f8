"""CPU: the reference-test acceptance bands of tests/band_cases.py against the oracle."""
import pytest

import band_cases


@pytest.mark.parametrize("case", band_cases.CASES, ids=lambda c: c.__name__)
def test_band(case, oracle_backend):
    case(oracle_backend)


def test_residual_values():
    """test/testApproxConv.jl:11-37: calcFactorResidualTemporary(LinearRelative{3}, z=[0,0,.5], (0, [0,0,1])) has
    |sum| 0.5 (z dimension 3); plus the closed residual set of SURVEY a10 at hand-computed points."""
    import ctypes as C
    import os

    import numpy as np

    lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "liboracle.so"))
    dp = C.POINTER(C.c_double)
    lib.orc_residual.restype = C.c_int32
    lib.orc_residual.argtypes = [C.c_int32, C.c_int32, dp, dp, dp, dp]

    def res(kind, man, z, a, b):
        z, a, b = (np.ascontiguousarray(v, dtype=np.float64) for v in (z, a, b))
        r = np.zeros(3)
        n = lib.orc_residual(kind, man, z.ctypes.data_as(dp), a.ctypes.data_as(dp), b.ctypes.data_as(dp), r.ctypes.data_as(dp))
        return r[:n]

    F_LINREL, F_CIRC, F_SE2, F_DIST = 3, 4, 5, 6
    E1, E2, E3, CIRC, SE2 = 1, 2, 3, 4, 5
    r = res(F_LINREL, E3, [0, 0, 0.5], np.zeros(3), [0, 0, 1.0])
    assert r.size == 3 and abs(np.abs(r).sum() - 0.5) < 1e-10
    # CircularCircular: wraps through +-pi
    assert abs(res(F_CIRC, CIRC, [0.2], [3.1], [-3.1])[0] - (3.3 - 2 * np.pi + 3.1)) < 1e-12
    # SE(2): x1 = x0 * exp(z) exactly -> zero residual
    a = np.array([1.0, 2.0, np.pi / 2])
    z = np.array([1.0, 0.0, np.pi / 2])
    b = np.array([1.0, 3.0, np.pi])
    np.testing.assert_allclose(res(F_SE2, SE2, z, a, b), [0, 0, 0], atol=1e-12)
    assert abs(res(F_DIST, E2, [5.0], [0, 0], [3, 4])[0]) < 1e-12


def test_marginalized_variables_are_not_updated(oracle_backend):
    """doFMCIteration skips `ismargin` variables in the up solve (SolveTree.jl:61);
    `limitfixeddown` makes the down solve skip them too (CliqStateMachineUtils.jl:498-502)."""
    import numpy as np

    from parity_utils import iif

    def build():
        fg = iif.initfg(iif.SolverParams(N=100, limitfixeddown=True))
        for i in range(4):
            iif.addVariable(fg, f"x{i}", iif.ContinuousScalar)
        iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal(0.0, 0.1)))
        for i in range(3):
            iif.addFactor(fg, [f"x{i}", f"x{i+1}"], iif.LinearRelative(iif.Normal(1.0, 0.1)))
        iif.initAll(fg, backend=oracle_backend, seed=60)
        return fg

    fg = build()
    fg.getVariable("x0").ismargin = True
    frozen = fg.getVal("x0").copy()
    iif.solveTree(fg, backend=oracle_backend, seed=61)
    np.testing.assert_array_equal(fg.getVal("x0"), frozen)
    for i in range(1, 4):
        assert abs(fg.getVal(f"x{i}").mean() - i) < 0.4
    fg2 = build()
    iif.solveTree(fg2, backend=oracle_backend, seed=61)
    assert np.abs(fg2.getVal("x0") - frozen).max() > 0  # without the flag x0 is re-estimated


def test_uninitialised_hypotheses_are_suppressed(oracle_backend):
    """ExplicitDiscreteMarginalizations.jl:161-172 + GraphInit.jl:94-105 (#427): with multihypo = [1, .5, .5]
    and only ONE of the two fractional landmarks initialised, x0 is initialised from that hypothesis alone:
    every particle draws mhidx = 2 (the suppressed one gets probability 0), certainidx stays [1]."""
    import numpy as np

    from parity_utils import abi, iif

    fg = iif.initfg(iif.SolverParams(N=100))
    iif.addVariable(fg, "x0", iif.ContinuousScalar)
    iif.addVariable(fg, "la", iif.ContinuousScalar)
    iif.addVariable(fg, "lb", iif.ContinuousScalar)
    iif.addFactor(fg, ["la"], iif.Prior(iif.Normal(10.0, 0.1)))
    iif.addFactor(fg, ["x0", "la", "lb"], iif.LinearRelative(iif.Normal(10.0, 0.1)), multihypo=[1, 0.5, 0.5])
    # lb has no prior and x0 no other factor: the reference allows the init of x0 from hypothesis `la`
    plan, _ = iif.solver._init_plan(fg)
    assert [p[0] for p in plan][:2] == ["la", "x0"] or [p[0] for p in plan][:2] == ["x0", "la"][::-1]
    iif.initAll(fg, backend=oracle_backend, seed=70)
    assert fg.isInitialized("x0") and fg.isInitialized("la")
    x0 = fg.getVal("x0")[:, 0]
    assert abs(x0.mean() - 0.0) < 0.5 and x0.std() < 1.0  # la - 10 = 0: nothing drawn towards the identity values of lb

    # the recipe itself, through the proposal op: injected flags, sampled mhidx
    N = 100
    be = oracle_backend(N, 4, N)
    rng = np.random.default_rng(0)
    for s, c in ((0, 0.0), (1, 10.0), (2, 0.0)):
        be.slot_write(s, abi.EUCLID1, rng.normal(c, 0.1, (N, 1)))
    from parity_utils import relative_factor_desc
    d = relative_factor_desc(abi.F_LINREL, abi.EUCLID1, 3, 0, [0, 1, 2], 3, 5, [10.0], [0.1], multihypo=[0.0, 0.5, 0.5], mhidx_out=0)
    d.has_multihypo = 1 | 0x80 | (1 << 9)  # only variable 1 (la) initialised
    be.run_proposals([d])
    assert set(be.side_read(0, N).tolist()) == {2}
    d.has_multihypo = 1 | 0x80 | (1 << 8) | (1 << 9)  # x0 and la initialised: nvars - 1 -> no suppression (:161)
    be.run_proposals([d])
    assert set(be.side_read(0, N).tolist()) == {2, 3}
    d.has_multihypo = 1  # no flags: every variable counts as initialised
    be.run_proposals([d])
    assert set(be.side_read(0, N).tolist()) == {2, 3}
    be.close()
