"""CPU: reference-band tests of the oracle for the multi-modal / non-Euclidean factor set."""
import numpy as np

from parity_utils import iif


def wrapdiff(a, b):
    return (a - b + np.pi) % (2 * np.pi) - np.pi


def test_circular_chain(oracle_backend):
    # test/testCircular.jl:7-29: x0..x4 with CircularCircular(Normal(1.0, 0.1)); PPE ~ rem2pi(0:4), atol 0.35
    fg = iif.initfg(iif.SolverParams(N=100))
    iif.addVariable(fg, "x0", iif.Circular)
    iif.addFactor(fg, ["x0"], iif.PriorCircular(iif.Normal(0.0, 0.1)))
    for i in range(1, 5):
        iif.addVariable(fg, f"x{i}", iif.Circular)
        iif.addFactor(fg, [f"x{i-1}", f"x{i}"], iif.CircularCircular(iif.Normal(1.0, 0.1)))
    iif.solveTree(fg, backend=oracle_backend, seed=21)
    for i in range(5):
        p = fg.getVal(f"x{i}")[:, 0]
        m = np.arctan2(np.sin(p).mean(), np.cos(p).mean())
        assert abs(wrapdiff(m, float(i))) < 0.35, (i, m)
        assert (p >= -np.pi).all() and (p < np.pi).all()


def test_se2_prior_and_odometry(oracle_backend):
    # test/testSpecialEuclidean2Mani.jl:35-77: prior at identity (sigma .01), factor mean (1, 2, pi/4)
    fg = iif.initfg(iif.SolverParams(N=100))
    iif.addVariable(fg, "x0", iif.SpecialEuclidean2)
    iif.addFactor(fg, ["x0"], iif.ManifoldPrior(np.zeros(3), iif.MvNormal(np.zeros(3), [0.01, 0.01, 0.01])))
    iif.addVariable(fg, "x1", iif.SpecialEuclidean2)
    iif.addFactor(fg, ["x0", "x1"], iif.ManifoldFactor(iif.MvNormal([1.0, 2.0, np.pi / 4], [0.01, 0.01, 0.01])))
    iif.solveTree(fg, backend=oracle_backend, seed=22)
    p0, p1 = fg.getVal("x0"), fg.getVal("x1")
    np.testing.assert_allclose(p0.mean(axis=0), [0, 0, 1, 0, 0, 1], atol=0.1)
    np.testing.assert_allclose(p1.mean(axis=0), [1, 2, 0.7071, 0.7071, -0.7071, 0.7071], atol=0.1)
    np.testing.assert_allclose(p1[:, 2] ** 2 + p1[:, 3] ** 2, 1.0, atol=1e-12)  # is_point


def test_mixture_relative_is_bimodal(oracle_backend):
    # test/testMixtureLinearConditional.jl:15-74 shape: Mixture(LinearRelative, (N(-5,.1), N(5,.1)), [.5,.5])
    fg = iif.initfg(iif.SolverParams(N=200))
    iif.addVariable(fg, "x0", iif.ContinuousScalar)
    iif.addVariable(fg, "x1", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal(0.0, 0.1)))
    iif.addFactor(fg, ["x0", "x1"], iif.Mixture(iif.LinearRelative, (iif.Normal(-5.0, 0.1), iif.Normal(5.0, 0.1)), [0.5, 0.5]))
    iif.initAll(fg, backend=oracle_backend, seed=23)
    pts = iif.approxConv(fg, "x0x1f1", "x1", backend=oracle_backend, seed=24)[:, 0]
    lo, hi = (np.abs(pts + 5) < 1).mean(), (np.abs(pts - 5) < 1).mean()
    assert lo > 0.3 and hi > 0.3 and lo + hi > 0.95


def test_nullhypo_spreads_mass(oracle_backend):
    # test/testnullhypothesis.jl:33-84: nullhypo=0.5 leaves about half of the particles un-driven
    fg = iif.initfg(iif.SolverParams(N=200))
    iif.addVariable(fg, "x0", iif.ContinuousScalar)
    iif.addVariable(fg, "x1", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal(0.0, 0.1)))
    iif.addFactor(fg, ["x0", "x1"], iif.LinearRelative(iif.Normal(10.0, 0.1)), nullhypo=0.5)
    iif.initAll(fg, backend=oracle_backend, seed=25)
    pts, bw, mh = iif.approxConvBelief(fg, "x0x1f1", "x1", backend=oracle_backend, seed=26, return_mhidx=True)
    near = np.abs(pts[:, 0] - 10) < 1
    assert 0.3 < near.mean() < 0.7
    assert near[mh == 1].all()  # every driven particle sits at the conditional
    assert 0.3 < (mh == 0).mean() < 0.7


def test_multihypo_landmark_modes(oracle_backend):
    # test/testMultiHypo3Door.jl:95-120 pattern: x0 sights one of two landmarks
    fg = iif.initfg(iif.SolverParams(N=200))
    iif.addVariable(fg, "x0", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal(0.0, 0.1)))
    for k, pos in enumerate((-10.0, 10.0)):
        iif.addVariable(fg, f"l{k}", iif.ContinuousScalar)
        iif.addFactor(fg, [f"l{k}"], iif.Prior(iif.Normal(pos, 0.1)))
    # the landmarks are initialised from their priors before the sighting exists (with the sighting in
    # the graph, a landmark's init would also use it through the one available hypothesis, #427)
    iif.initAll(fg, backend=oracle_backend, seed=27)
    iif.addFactor(fg, ["x0", "l0", "l1"], iif.LinearRelative(iif.Normal(10.0, 0.1)), multihypo=[1.0, 0.5, 0.5])
    pts, bw, mh = iif.approxConvBelief(fg, "x0l0l1f1", "x0", backend=oracle_backend, seed=28, return_mhidx=True)
    assert set(np.unique(mh)) == {2, 3}
    a, b = (np.abs(pts[:, 0] + 20) < 1).mean(), (np.abs(pts[:, 0] - 0) < 1).mean()
    assert a > 0.3 and b > 0.3  # x0 = l0 - 10 = -20 or l1 - 10 = 0
