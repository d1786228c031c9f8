"""Acceptance bands of the reference's own tests for the hot path, written once and
run against both backends: the CPU oracle (`tests/test_oracle_reference_bands2.py`) and the HIP
library on the GPU (`tests/test_gpu_reference_bands.py`).  Each case cites the reference test it
restates.  The bands are the reference's; the graphs are rebuilt from the test descriptions."""
import numpy as np

from parity_utils import iif

rng = np.random.default_rng


def se2_near(pts, xy, theta, atol):
    """count of SE(2) points (N x 6: t, R column-major) with ||p - q||_F-ish distance < atol, the role of
    isapprox(M, p, q; atol) on SpecialEuclidean(2)"""
    c, s = np.cos(theta), np.sin(theta)
    q = np.array([xy[0], xy[1], c, s, -s, c])
    return int((np.linalg.norm(pts - q, axis=1) < atol).sum())


def se2_mean(pts):
    th = np.arctan2(pts[:, 3].mean(), pts[:, 2].mean())
    return pts[:, 0].mean(), pts[:, 1].mean(), th


def case_forward_convolve(backend):
    # test/testBasicForwardConvolve.jl:16-65 (IIF issue #477): conv, product with a measurement, conv
    def forward(X0, Z, seed):
        fg = iif.initfg(iif.SolverParams(N=100))
        iif.addVariable(fg, "x0", iif.ContinuousScalar)
        iif.initVariable(fg, "x0", X0, backend=backend)
        iif.addVariable(fg, "x1", iif.ContinuousScalar)
        iif.addFactor(fg, ["x0", "x1"], iif.LinearRelative(Z))
        return iif.approxConv(fg, "x0x1f1", "x1", backend=backend, seed=seed)

    r = rng(477)
    X0 = r.normal(0, 0.1, (100, 1))
    X1_ = forward(X0, iif.Normal(11, 1.0), 1)
    # predX1 * measX1 : product of two KDEs through the variable seam
    fg = iif.initfg(iif.SolverParams(N=100))
    iif.addVariable(fg, "x1", iif.ContinuousScalar)
    be = backend(100, 3)
    try:
        man = iif.ContinuousScalar.manifold
        be.slot_write(0, man, X1_, np.ones(1))
        be.slot_write(1, man, r.normal(9.5, 0.75, (100, 1)), np.ones(1))
        be.run_bandwidth([0, 1], [man, man])
        be.run_products([iif.product_desc(man, [0, 1], 2, 12345, 1)])
        X1, _ = be.slot_read(2, man)
    finally:
        be.close()
    assert 8.5 < X1.mean() < 11.5
    X2 = forward(X1, iif.Normal(8, 2.0), 2)
    assert X2.shape == (100, 1)
    assert 15 < X2.mean() < 25


def case_five_chain_spread(backend):
    # test/testProductReproducable.jl:12-45
    fg = iif.initfg(iif.SolverParams(N=100))
    for v in "abcde":
        iif.addVariable(fg, v, iif.ContinuousScalar)
    iif.addFactor(fg, ["a"], iif.Prior(iif.Normal(0, 1)))
    for u, v in zip("abcd", "bcde"):
        iif.addFactor(fg, [u, v], iif.LinearRelative(iif.Normal(10, 1)))
    iif.initAll(fg, backend=backend, seed=3)
    iif.solveTree(fg, backend=backend, seed=4)
    for k, (v, mt, lo, hi) in enumerate(zip("abcde", (3, 4, 4, 5, 5), (0.3, 0.5, 0.9, 1.2, 1.5), (2, 4, 6, 7, 8))):
        p = fg.getVal(v)[:, 0]
        assert abs(p.mean() - 10 * k) < mt, (v, p.mean())
        assert lo < p.std(ddof=1) < hi, (v, p.std(ddof=1))


def case_back_and_forth_spreads(backend):
    # test/testProductReproducable.jl:52-99: 10 x (conv a->b, conv b->a): means stay, std grows above 3
    fg = iif.initfg(iif.SolverParams(N=100))
    iif.addVariable(fg, "a", iif.ContinuousScalar)
    iif.addVariable(fg, "b", iif.ContinuousScalar)
    iif.addFactor(fg, ["a", "b"], iif.LinearRelative(iif.Normal(10, 1)))
    r = rng(52)
    A = r.normal(0, 1, (100, 1))
    B = 10 + r.normal(0, 1, (100, 1))
    iif.initVariable(fg, "a", A, backend=backend)
    iif.initVariable(fg, "b", B, backend=backend)
    for i in range(10):
        iif.initVariable(fg, "b", iif.approxConv(fg, "abf1", "b", backend=backend, seed=100 + 2 * i), backend=backend)
        iif.initVariable(fg, "a", iif.approxConv(fg, "abf1", "a", backend=backend, seed=101 + 2 * i), backend=backend)
    A_, B_ = fg.getVal("a")[:, 0], fg.getVal("b")[:, 0]
    assert abs(A.mean()) < 1 and abs(A_.mean()) < 2
    assert abs(B.mean() - 10) < 1 and abs(B_.mean() - 10) < 2
    assert A.std(ddof=1) < 2 and 3 < A_.std(ddof=1)
    assert B.std(ddof=1) < 2 and 3 < B_.std(ddof=1)


def case_approxconv_kaess_chains(backend):
    # test/testApproxConv.jl:40-81
    fg = iif.generateGraph_Kaess(iif.SolverParams(N=100))
    iif.initAll(fg, backend=backend, seed=5)  # the reference's graphinit on addFactor!
    pts = iif.approxConv(fg, "x1f1", "x1", backend=backend, seed=6)
    assert abs(pts.mean()) < 0.4 and 0.5 < pts.std(ddof=1) < 1.5
    iif.initVariable(fg, "x1", pts, backend=backend)
    pts = iif.approxConv(fg, "x1x2f1", "x2", backend=backend, seed=7)
    assert abs(pts.mean()) < 0.7 and 0.7 < pts.std(ddof=1) < 2
    pts = iif.approxConv(fg, "x1", "x3", backend=backend, seed=8)  # along a chain of variables
    assert abs(pts.mean()) < 1.5 and 1.3 < pts.std(ddof=1) < 3.0
    pts = iif.approxConv(fg, "x1f1", "l2", backend=backend, seed=9)  # from a prior down the chain
    assert abs(pts.mean()) < 1.5 and 1.6 < pts.std(ddof=1) < 4.0


def case_ccw_forward_reverse(backend):
    # test/testCommonConvWrapper.jl:96-147: odo Normal(100, 1); forward lands in (90, 110), reverse in (-10, 10);
    # the conv never modifies the stored beliefs
    fg = iif.initfg(iif.SolverParams(N=100))
    iif.addVariable(fg, "x0", iif.ContinuousEuclid(1))
    iif.addVariable(fg, "x1", iif.ContinuousEuclid(1))
    iif.initVariable(fg, "x0", np.zeros((100, 1)) + 1e-3 * rng(1).normal(size=(100, 1)), backend=backend)
    iif.initVariable(fg, "x1", rng(2).uniform(size=(100, 1)), backend=backend)
    iif.addFactor(fg, ["x0", "x1"], iif.LinearRelative(iif.Normal(100.0, 1.0)))
    x0_before = fg.getVal("x0").copy()
    pts = iif.approxConv(fg, "x0x1f1", "x1", backend=backend, seed=10)
    assert 90.0 < pts.mean() < 110.0
    assert -10.0 < fg.getVal("x0").mean() < 10.0
    np.testing.assert_array_equal(fg.getVal("x0"), x0_before)
    iif.initVariable(fg, "x1", 100 * np.ones((100, 1)) + 1e-3 * rng(3).normal(size=(100, 1)), backend=backend)
    pts = iif.approxConv(fg, "x0x1f1", "x0", backend=backend, seed=11)
    assert -10.0 < pts.mean() < 10.0
    assert 90.0 < fg.getVal("x1").mean() < 110.0


def case_conv_95_percent(backend):
    # test/testMultithreaded.jl:14-37: 95 % of the convolved points within 5 of 10
    N = 100
    fg = iif.initfg(iif.SolverParams(N=N))
    iif.addVariable(fg, "x0", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal(0, 1)))
    iif.addVariable(fg, "x1", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0", "x1"], iif.LinearRelative(iif.Normal(10.0, 1)))
    iif.initAll(fg, backend=backend, seed=12)
    pts = iif.approxConv(fg, "x0x1f1", "x1", backend=backend, seed=13)
    assert 0.95 * N <= (np.abs(pts - 10.0) < 5.0).sum()


def case_euclid_distance_1d(backend):
    # test/testEuclidDistance.jl:9-41: bimodal at +-10, nothing in the middle
    fg = iif.initfg(iif.SolverParams(N=100))
    iif.addVariable(fg, "x0", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal(0, 1)))
    iif.addVariable(fg, "x1", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0", "x1"], iif.EuclidDistance(iif.Normal(10, 1)))
    iif.initAll(fg, backend=backend, seed=14)
    iif.solveTree(fg, backend=backend, seed=15)
    assert abs(fg.getVal("x0").mean()) < 1
    pts = fg.getVal("x1")[:, 0]
    N = pts.size
    assert 0.3 * N < (pts > 5).sum()
    assert 0.3 * N < (pts < -5).sum()
    assert ((pts > -5) & (pts < 5)).sum() < 0.1 * N


def case_euclid_distance_2d(backend):
    # test/testEuclidDistance.jl:44-70: a ring of radius 10 around the origin
    fg = iif.initfg(iif.SolverParams(N=100))
    iif.addVariable(fg, "x0", iif.ContinuousEuclid(2))
    iif.addFactor(fg, ["x0"], iif.Prior(iif.MvNormal(np.zeros(2), np.diag([1.0, 1.0]))))
    iif.addVariable(fg, "x1", iif.ContinuousEuclid(2))
    iif.addFactor(fg, ["x0", "x1"], iif.EuclidDistance(iif.Normal(10, 1)))
    iif.initAll(fg, backend=backend, seed=16)
    iif.solveTree(fg, backend=backend, seed=17)
    assert abs(fg.getVal("x0")[:, 0].mean()) < 1
    r = np.linalg.norm(fg.getVal("x1"), axis=1)
    assert 0.5 * r.size < ((r > 7) & (r < 13)).sum()


def case_se2_hex(backend):
    # test/testSpecialEuclidean2Mani.jl:164-204: hexagon of 6 odometry steps (10, 0, pi/3) from (10, 10, pi),
    # a landmark seen from x0 and x6 closes the loop
    fg = iif.initfg(iif.SolverParams(N=100))
    iif.addVariable(fg, "x0", iif.SpecialEuclidean2)
    iif.addFactor(fg, ["x0"], iif.ManifoldPrior(np.array([10.0, 10.0, np.pi]), iif.MvNormal(np.zeros(3), [0.1, 0.1, 0.01])))
    for i in range(6):
        iif.addVariable(fg, f"x{i+1}", iif.SpecialEuclidean2)
        iif.addFactor(fg, [f"x{i}", f"x{i+1}"], iif.ManifoldFactor(iif.MvNormal([10.0, 0, np.pi / 3], [0.5, 0.5, 0.05])))
    iif.addVariable(fg, "l1", iif.SpecialEuclidean2)
    iif.addFactor(fg, ["x0", "l1"], iif.ManifoldFactor(iif.MvNormal([10.0, 0, 0], [0.1, 0.1, 0.01])))
    iif.addFactor(fg, ["x6", "l1"], iif.ManifoldFactor(iif.MvNormal([10.0, 0, 0], [0.1, 0.1, 0.01])))
    iif.initAll(fg, backend=backend, seed=18)
    iif.solveTree(fg, backend=backend, seed=19)

    def dist(v, xy, th):
        x, y, t = se2_mean(fg.getVal(v))
        dth = (t - th + np.pi) % (2 * np.pi) - np.pi
        return np.sqrt((x - xy[0]) ** 2 + (y - xy[1]) ** 2 + 2 * dth * dth)  # Frobenius metric on R

    assert dist("x0", (10.0, 10.0), np.pi) < 0.2
    assert dist("x1", (0.0, 10.0), np.arctan2(-0.866, -0.5)) < 0.4
    assert dist("x6", (10.0, 10.0), np.pi) < 0.5


def case_se2_multihypo(backend):
    # test/testSpecialEuclidean2Mani.jl:605-637: ManifoldFactor on [x0, x1a, x1b] multihypo [1, .5, .5]:
    # x0 stays at the identity, more than 20 of 100 particles of each candidate land on (1, 2, pi/4)
    fg = iif.initfg(iif.SolverParams(N=100))
    iif.addVariable(fg, "x0", iif.SpecialEuclidean2)
    iif.addFactor(fg, ["x0"], iif.ManifoldPrior(np.zeros(3), iif.MvNormal(np.zeros(3), [0.01, 0.01, 0.01])))
    iif.addVariable(fg, "x1a", iif.SpecialEuclidean2)
    iif.addVariable(fg, "x1b", iif.SpecialEuclidean2)
    iif.addFactor(fg, ["x0", "x1a", "x1b"], iif.ManifoldFactor(iif.MvNormal([1, 2, np.pi / 4], [0.01, 0.01, 0.01])),
                  multihypo=[1, 0.5, 0.5])
    iif.initAll(fg, backend=backend, seed=20)
    iif.solveTree(fg, backend=backend, seed=21)
    x, y, t = se2_mean(fg.getVal("x0"))
    assert np.sqrt(x * x + y * y + 2 * t * t) < 0.1
    assert se2_near(fg.getVal("x1a"), (1.0, 2.0), np.pi / 4, 0.1) > 20
    assert se2_near(fg.getVal("x1b"), (1.0, 2.0), np.pi / 4, 0.1) > 20


def case_simple_mixture(backend):
    # test/testMixtureLinearConditional.jl:135-200: Mixture(LinearRelative, [N(-1,.1), N(1,.1)], [.5,.5])
    fg = iif.initfg(iif.SolverParams(N=100))
    iif.addVariable(fg, "x0", iif.ContinuousScalar)
    iif.addVariable(fg, "x1", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal(0.0, 0.1)))
    iif.addFactor(fg, ["x0", "x1"], iif.Mixture(iif.LinearRelative, (iif.Normal(-1.0, 0.1), iif.Normal(1.0, 0.1)), [0.5, 0.5]))
    iif.initAll(fg, backend=backend, seed=22)
    iif.solveTree(fg, backend=backend, seed=23)
    p0 = fg.getVal("x0")[:, 0]
    assert abs(p0.mean()) < 0.1 and abs(p0.std() - 0.1) < 0.05
    p1 = fg.getVal("x1")[:, 0]
    pp, pn = p1[p1 >= 0], p1[p1 < 0]
    assert pp.size > 10 and pn.size > 10
    assert abs(pp.mean() - 1.0) < 0.1 and abs(pp.std() - 0.14) < 0.05
    assert abs(pn.mean() + 1.0) < 0.1 and abs(pn.std() - 0.14) < 0.05


def _partial_graph(backend, N=100):
    # test/testpartialconstraint.jl:49-66: x1 in R^2 with a full prior N(0, 0.01 I) and a partial prior
    # N(2, 1) on coordinate 1
    fg = iif.initfg(iif.SolverParams(N=N))
    E2 = iif.ContinuousEuclid(2)
    iif.addVariable(fg, "x1", E2)
    iif.addFactor(fg, ["x1"], iif.Prior(iif.MvNormal(np.zeros(2), np.diag([0.01, 0.01]))))
    iif.addFactor(fg, ["x1"], iif.PartialPrior(E2, iif.Normal(2.0, 1.0), (1,)))
    pts, bw = iif.approxConvBelief(fg, "x1f1", "x1", backend=backend, seed=30)  # doautoinit! from the full prior
    iif.setValKDE(fg, "x1", pts, bw)
    return fg, E2


def case_partial_prior(backend):
    # test/testpartialconstraint.jl:69-115 + :118-131
    N = 100
    fg, E2 = _partial_graph(backend, N)
    pts = iif.approxConv(fg, "x1f1", "x1", backend=backend, seed=31)
    assert pts.shape == (N, 2) and abs(pts[:, 0].mean()) < 0.3
    X1 = fg.getVal("x1").copy()
    pts = iif.approxConv(fg, "x1f2", "x1", backend=backend, seed=32)
    assert pts.shape == (N, 2)
    assert abs(pts[:, 0].mean() - 2.0) < 0.75
    assert np.linalg.norm(X1[:, 0] - pts[:, 0]) > 2.0        # the partial coordinate moved
    assert np.linalg.norm(X1[:, 1] - pts[:, 1]) < 1e-10       # the other one is untouched
    np.testing.assert_array_equal(fg.getVal("x1"), X1)        # and the stored belief too
    iif.solveTree(fg, backend=backend, seed=33)
    p = fg.getVal("x1")
    assert abs(p[:, 0].mean()) < 0.4 and abs(p[:, 1].mean()) < 0.4


def case_partial_relative_and_products(backend):
    # test/testpartialconstraint.jl:135-147,185-197,250-291: x2 with a partial pairwise factor on
    # coordinate 2 (DevelopPartialPairwise) and a partial prior N(-20, 1) on coordinate 1
    N = 100
    fg, E2 = _partial_graph(backend, N)
    iif.addVariable(fg, "x2", E2)
    iif.addFactor(fg, ["x1", "x2"], iif.PartialLinearRelative(E2, iif.Normal(10.0, 1.0), (2,)))
    iif.addFactor(fg, ["x2"], iif.PartialPrior(E2, iif.Normal(-20.0, 1.0), (1,)))
    iif.initAll(fg, backend=backend, seed=34)
    X2 = fg.getVal("x2").copy()
    # the relative partial only touches coordinate 2 of its proposal
    pts = iif.approxConv(fg, "x1x2f1", "x2", backend=backend, seed=35)
    assert pts.shape == (N, 2)
    assert np.linalg.norm(X2[:, 0] - pts[:, 0]) < 1e-10
    assert abs((pts[:, 1] - fg.getVal("x1")[:, 1]).mean() - 10.0) < 0.75
    pts = iif.approxConv(fg, "x2f1", "x2", backend=backend, seed=36)
    assert abs(pts[:, 0].mean() + 20.0) < 0.75 and pts[:, 0].std() - 1.0 < 0.4
    # propagateBelief returns full-dimension points even when only partials are sent in
    (val, _), _ = iif.propagateBelief(fg, "x2", ["x2f1"], backend=backend, seed=37)
    assert np.linalg.norm(X2[:, 1] - val[:, 1]) < 1e-10 and np.linalg.norm(X2[:, 0] - val[:, 0]) > 0
    assert abs(val[:, 0].mean() + 20.0) < 0.75
    (val, _), _ = iif.propagateBelief(fg, "x2", ["x1x2f1"], backend=backend, seed=38)
    assert np.linalg.norm(X2[:, 0] - val[:, 0]) < 1e-10 and np.linalg.norm(X2[:, 1] - val[:, 1]) > 0
    # combination of partials: every coordinate informed by exactly one density
    (val, _), _ = iif.propagateBelief(fg, "x2", ["x1x2f1", "x2f1"], backend=backend, seed=39)
    assert abs(val[:, 0].mean() + 20.0) < 1 and val[:, 0].std() - 1.0 < 3.0
    assert abs(val[:, 1].mean() - 10.0) < 3.0
    iif.solveTree(fg, backend=backend, seed=40)
    p = fg.getVal("x1")
    assert abs(p[:, 0].mean()) < 0.5 and abs(p[:, 1].mean()) < 0.5
    q = fg.getVal("x2")
    assert abs(q[:, 0].mean() + 20.0) < 1.0 and abs(q[:, 1].mean() - 10.0) < 3.0


CASES = [case_forward_convolve, case_five_chain_spread, case_back_and_forth_spreads, case_approxconv_kaess_chains,
         case_ccw_forward_reverse, case_conv_95_percent, case_euclid_distance_1d, case_euclid_distance_2d,
         case_se2_hex, case_se2_multihypo, case_simple_mixture, case_partial_prior, case_partial_relative_and_products]


def case_deconv(backend):
    # test/testDefaultDeconv.jl:14-32 (LinearRelative on a 1-D line), :140-163 (EuclidDistance) and
    # test/testSpecialEuclidean2Mani.jl:225-258 (SE(2)): the predicted measurement distribution matches
    # the measured one
    fg = iif.generateGraph_LineStep(2, solverParams=iif.SolverParams(N=100))
    iif.initAll(fg, backend=backend, seed=50)
    f = [l for l in fg.lsf() if len(fg.getFactor(l).variables) == 2][0]
    pred, meas = iif.approxDeconv(fg, f, backend=backend, seed=51)
    assert pred.shape == meas.shape == (100, 1)
    a, b = fg.getFactor(f).variables
    np.testing.assert_allclose(pred[:, 0], fg.getVal(b)[:, 0] - fg.getVal(a)[:, 0], atol=1e-6)  # z* = x2 - x1
    assert abs(pred.mean() - meas.mean()) < 0.2

    fg = iif.initfg(iif.SolverParams(N=100))
    iif.addVariable(fg, "x0", iif.ContinuousScalar)
    iif.addVariable(fg, "x1", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal(0, 1)))
    iif.addFactor(fg, ["x0", "x1"], iif.EuclidDistance(iif.Normal(10, 1)))
    iif.initAll(fg, backend=backend, seed=52)
    iif.solveTree(fg, backend=backend, seed=53)
    pred, meas = iif.approxDeconv(fg, "x0x1f1", backend=backend, seed=54)
    np.testing.assert_allclose(pred[:, 0], np.abs(fg.getVal("x1")[:, 0] - fg.getVal("x0")[:, 0]), atol=1e-5)
    assert abs(pred.mean() - meas.mean()) < 1.0 and abs(pred.std() - meas.std()) < 1.0

    fg = iif.initfg(iif.SolverParams(N=100))
    iif.addVariable(fg, "x0", iif.SpecialEuclidean2)
    iif.addVariable(fg, "x1", iif.SpecialEuclidean2)
    iif.addFactor(fg, ["x0"], iif.ManifoldPrior(np.array([10.0, 10.0, np.pi]), iif.MvNormal(np.zeros(3), np.diag([0.1, 0.1, 0.01]) ** 2)))
    iif.addFactor(fg, ["x0", "x1"], iif.ManifoldFactor(iif.MvNormal([10.0, 0, 0.1], np.diag([0.5, 0.5, 0.05]) ** 2)))
    iif.initAll(fg, backend=backend, seed=55)
    pred, meas = iif.approxDeconv(fg, "x0x1f1", backend=backend, seed=56)
    assert pred.shape == (100, 3)
    assert abs(pred[:, 2].mean() - 0.1) < 0.02 and abs(pred[:, 2].std() - 0.05) < 0.02
    np.testing.assert_allclose(pred[:, :2].mean(axis=0), [10, 0], atol=0.3)
    np.testing.assert_allclose(pred[:, :2].std(axis=0), [0.5, 0.5], atol=0.3)
    assert abs(pred[:, 2].mean() - meas[:, 2].mean()) < 0.03 and abs(pred[:, 2].std() - meas[:, 2].std()) < 0.03
    np.testing.assert_allclose(pred[:, :2].mean(axis=0), meas[:, :2].mean(axis=0), atol=0.3)


CASES.append(case_deconv)


def case_four_doors(backend):
    # test/fourdoortest.jl:5-59 (the package's canonical multimodal example; the reference only checks that
    # it runs).  Mixture(Prior, four doors) sighted at x1, x3, x4 with odometry 50, 50, 200: the only
    # consistent explanation is x1 = 0, x2 = 50, x3 = 100, x4 = 300, which must hold the dominant mode.
    doors = (iif.Normal(-100, 3.0), iif.Normal(0, 3.0), iif.Normal(100, 3.0), iif.Normal(300, 3.0))
    fg = iif.initfg(iif.SolverParams(N=100))
    for i in range(1, 5):
        iif.addVariable(fg, f"x{i}", iif.ContinuousScalar)
    iif.addFactor(fg, ["x1"], iif.Mixture(iif.Prior, doors, [0.25] * 4))
    iif.addFactor(fg, ["x1", "x2"], iif.LinearRelative(iif.Normal(50.0, 2.0)))
    iif.addFactor(fg, ["x2", "x3"], iif.LinearRelative(iif.Normal(50.0, 4.0)))
    iif.addFactor(fg, ["x3"], iif.Mixture(iif.Prior, doors, [0.25] * 4))
    iif.addFactor(fg, ["x3", "x4"], iif.LinearRelative(iif.Normal(200.0, 4.0)))
    iif.addFactor(fg, ["x4"], iif.Mixture(iif.Prior, doors, [0.25] * 4))
    iif.initAll(fg, backend=backend, seed=80)
    iif.solveTree(fg, backend=backend, seed=81)
    for v, truth in (("x1", 0.0), ("x2", 50.0), ("x3", 100.0), ("x4", 300.0)):
        p = fg.getVal(v)[:, 0]
        assert (np.abs(p - truth) < 15).mean() > 0.4, (v, np.round(np.sort(p)[::10]))
    # a second solve sharpens it (the reference example solves three times while it grows)
    iif.solveTree(fg, backend=backend, seed=82)
    p = fg.getVal("x4")[:, 0]
    assert (np.abs(p - 300.0) < 15).mean() > 0.5


CASES.append(case_four_doors)


def case_partial_nullhypo_3d(backend):
    # test/testPartialNH.jl:10-53 and :56-86: partial priors on (2,3) and (1,) of Euclid(3) variables and a
    # full LinearRelative between them, without and with nullhypo = 0.2
    E3 = iif.ContinuousEuclid(3)
    for nh, tol1 in ((0.0, 1.0), (0.2, 2.0)):
        fg = iif.initfg(iif.SolverParams(N=100))
        iif.addVariable(fg, "x0", E3)
        iif.addFactor(fg, ["x0"], iif.PartialPrior(E3, iif.MvNormal(np.zeros(2), np.ones(2)), (2, 3)), nullhypo=nh)
        iif.addVariable(fg, "x1", E3)
        iif.addFactor(fg, ["x1"], iif.PartialPrior(E3, iif.Normal(10, 1), (1,)))
        iif.addFactor(fg, ["x0", "x1"], iif.LinearRelative(iif.MvNormal([10, 0, 0.0], np.ones(3))), nullhypo=nh)
        iif.initAll(fg, backend=backend, seed=90)
        iif.solveTree(fg, backend=backend, seed=91)
        np.testing.assert_allclose(fg.getVal("x0").mean(axis=0), [0, 0, 0], atol=1.0)
        np.testing.assert_allclose(fg.getVal("x1").mean(axis=0), [10, 0, 0], atol=tol1)


def case_multimodal_1d(backend):
    # test/testMultimodal1D.jl:35-107: x1 measures two landmarks whose identity is uncertain
    # (multihypo [1, 0.4, 0.6] against a new landmark lm_k and a mapped one lp_k); x1 and the lm_k start
    # uninitialised, so x1 is initialised through the one available hypothesis (#427)
    sp = iif.SolverParams(N=100, gibbsIters=6, spreadNH=0.3)
    fg = iif.initfg(sp)
    iif.addVariable(fg, "lp1", iif.ContinuousScalar)
    iif.addFactor(fg, ["lp1"], iif.Prior(iif.Normal(-30.0, 1.0)))
    iif.addVariable(fg, "lp2", iif.ContinuousScalar)
    iif.addFactor(fg, ["lp2"], iif.Prior(iif.Normal(30.0, 1.0)))
    iif.addVariable(fg, "x1", iif.ContinuousScalar)
    iif.addVariable(fg, "lm2", iif.ContinuousScalar)
    iif.addFactor(fg, ["x1", "lm2", "lp2"], iif.LinearRelative(iif.Normal(20.0, 1.0)), multihypo=[1.0, 0.4, 0.6])
    iif.addVariable(fg, "lm1", iif.ContinuousScalar)
    iif.addFactor(fg, ["x1", "lm1", "lp1"], iif.LinearRelative(iif.Normal(-20.0, 1.0)), multihypo=[1.0, 0.4, 0.6])
    iif.initAll(fg, backend=backend, seed=112)
    assert all(fg.isInitialized(v) for v in fg.ls())
    iif.solveTree(fg, eliminationOrder=["x1", "lm1", "lm2", "lp1", "lp2"], backend=backend, seed=113)
    N = 100

    def count(v, lo, hi):
        p = fg.getVal(v)[:, 0]
        return ((p > lo) & (p < hi)).sum()

    assert 0.7 * N < count("x1", -20, 0) + count("x1", 0, 20)
    assert 0.7 * N < count("lp1", -38, -28)
    assert 0.7 * N < count("lp2", 28, 38)
    assert 0.1 * N < count("lm1", -38, -25)
    assert 0.1 * N < count("lm2", 25, 38)


CASES += [case_partial_nullhypo_3d, case_multimodal_1d]


def case_multihypo_and_chain(backend):
    # test/testMultihypoAndChain.jl:7-88: two poses, two landmarks, three sightings with data association
    # 0.99 / 0.01; a single clique (prescribed order); x0 ~ 0, x1 ~ 1, l1 ~ 1, l2 keeps a mode near 2
    r = rng(42)
    fg = iif.initfg(iif.SolverParams(N=100, gibbsIters=5, spreadNH=5.0))
    iif.addVariable(fg, "x0", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal(r.normal(0.0, 0.01), 0.01)))
    iif.addVariable(fg, "l1", iif.ContinuousScalar)
    iif.addVariable(fg, "l2", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0", "l1", "l2"], iif.LinearRelative(iif.Normal(r.normal(1.0, 0.01), 0.01)), multihypo=[1, 0.99, 0.01])
    iif.addVariable(fg, "x1", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0", "x1"], iif.LinearRelative(iif.Normal(r.normal(1.0, 0.1), 0.1)))
    iif.addFactor(fg, ["x1", "l1", "l2"], iif.LinearRelative(iif.Normal(r.normal(0.0, 0.01), 0.01)), multihypo=[1, 0.99, 0.01])
    iif.addFactor(fg, ["x1", "l2", "l1"], iif.LinearRelative(iif.Normal(r.normal(1.0, 0.01), 0.01)), multihypo=[1, 0.99, 0.01])
    iif.initAll(fg, backend=backend, seed=94)
    iif.solveTree(fg, eliminationOrder=["l2", "x1", "x0", "l1"], backend=backend, seed=95)

    def ppe(v):  # MeanMaxPPE "suggested": the mode; median is a robust stand-in for these narrow beliefs
        return float(np.median(fg.getVal(v)[:, 0]))

    assert abs(ppe("x0") - 0) < 0.2 and abs(ppe("x1") - 1) < 0.2 and abs(ppe("l1") - 1) < 0.2
    l2 = fg.getVal("l2")[:, 0]  # "at least a mode present" (the reference compares with N(2, 0.1) by mmd < 1e-3)
    assert (np.abs(l2 - 2.0) < 0.5).mean() > 0.2 and abs(np.median(l2) - 2.0) < 0.8, np.round(np.percentile(l2, [10, 50, 90]), 2)


def case_mixture_prior(backend):
    # test/testMixturePrior.jl:11-68 (#605): bi-modal Mixture(Prior, (N(-5,1), N(0,1)), [.5,.5]) stays balanced
    N = 100
    fg = iif.initfg(iif.SolverParams(N=N))
    iif.addVariable(fg, "x0", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Mixture(iif.Prior, (iif.Normal(-5.0, 1.0), iif.Normal(0.0, 1.0)), [0.5, 0.5]))
    iif.initAll(fg, backend=backend, seed=96)
    s = iif.approxConv(fg, "x0f1", "x0", backend=backend, seed=97)[:, 0]
    assert abs((s < -2.5).sum() - (s > -2.5).sum()) < 0.35 * N
    iif.solveTree(fg, backend=backend, seed=98)
    m = fg.getVal("x0")[:, 0]
    assert abs((m < -2.5).sum() - (m > -2.5).sum()) < 0.35 * N


def case_skip_up_or_down(backend):
    # test/testSkipUpDown.jl:5-60: a 7-pose line, first up-solve only (frontals are handed back after the
    # up pass), then down-solve only from the stored beliefs; the estimates stay at the truth either way
    def line():
        fg = iif.generateGraph_LineStep(6, poseEvery=1, landmarkEvery=7, posePriorsAt=(0,), sightDistance=7,
                                        solverParams=iif.SolverParams(N=100))
        return fg

    fg = line()
    iif.initAll(fg, backend=backend, seed=99)
    fg.solverParams.downsolve = False
    _, tm = iif.solveTree(fg, backend=backend, seed=100, return_timing=True)
    ncl = tm["messages"]
    for v in fg.ls():
        if v.startswith("x"):
            assert abs(np.median(fg.getVal(v)[:, 0]) - int(v[1:])) < 0.25, v
    fg.solverParams.upsolve, fg.solverParams.downsolve = False, True
    _, tm2 = iif.solveTree(fg, backend=backend, seed=101, return_timing=True)
    assert tm2["messages"] == ncl  # one message per edge in either single-direction solve
    for v in fg.ls():
        if v.startswith("x"):
            assert abs(np.median(fg.getVal(v)[:, 0]) - int(v[1:])) < 0.25, v
    fg.solverParams.upsolve = fg.solverParams.downsolve = False
    try:
        iif.solveTree(fg, backend=backend, seed=102)
        raise AssertionError("expected an error")
    except ValueError:
        pass


CASES += [case_multihypo_and_chain, case_mixture_prior, case_skip_up_or_down]


def case_orphaned_forest(backend):
    # test/testSolveOrphanedFG.jl:9-70 (#518): two disconnected chains, one elimination order, a tree with two roots
    fg = iif.initfg(iif.SolverParams(N=100))
    for a, mu, sig, pr in (("x0 x1 x2", 10.0, 0.1, iif.Normal(0, 0.1)), ("x10 x11 x12", -10.0, 1.0, iif.Normal(0, 1))):
        vs = a.split()
        for v in vs:
            iif.addVariable(fg, v, iif.ContinuousScalar)
        iif.addFactor(fg, [vs[0]], iif.Prior(pr))
        iif.addFactor(fg, [vs[0], vs[1]], iif.LinearRelative(iif.Normal(mu, sig)))
        iif.addFactor(fg, [vs[1], vs[2]], iif.LinearRelative(iif.Normal(mu, sig)))
    vo = ["x12", "x2", "x0", "x11", "x1", "x10"]
    iif.initAll(fg, backend=backend, seed=110)
    tree = iif.solveTree(fg, eliminationOrder=vo, backend=backend, seed=111)
    assert len(tree.roots) == 2
    clq = {v: c for c in tree.cliques.values() for v in c.frontalIDs}
    assert clq["x1"].parent < 0 and clq["x10"].parent < 0
    assert len(clq["x1"].children) == 1 and len(clq["x10"].children) == 1
    assert len(clq["x2"].children) == 0 and len(clq["x12"].children) == 0
    for v, m, tol in (("x0", 0, 1.0), ("x1", 10, 2.0), ("x2", 20, 3.0), ("x10", 0, 2.0), ("x11", -10, 3.0), ("x12", -20, 4.0)):
        assert abs(fg.getVal(v).mean() - m) < tol, v


def case_translation_group_prior_and_factor(backend):
    # test/testTranslationMani.jl:7-38: ManifoldPrior / ManifoldFactor on TranslationGroup(2) = Euclid(2)
    E2 = iif.ContinuousEuclid(2)
    fg = iif.initfg(iif.SolverParams(N=100))
    iif.addVariable(fg, "x0", E2)
    iif.addVariable(fg, "x1", E2)
    iif.addFactor(fg, ["x0"], iif.ManifoldPrior(np.array([10.0, 20.0]), iif.MvNormal(np.zeros(2), [1.0, 1.0])))
    iif.addFactor(fg, ["x0", "x1"], iif.LinearRelative(iif.MvNormal([1.0, 2.0], [0.1, 0.1])))
    iif.initAll(fg, backend=backend, seed=112)
    iif.solveTree(fg, backend=backend, seed=113)
    np.testing.assert_allclose(fg.getVal("x0").mean(axis=0), [10, 20], atol=0.6)
    np.testing.assert_allclose(fg.getVal("x1").mean(axis=0), [11, 22], atol=0.6)


CASES += [case_orphaned_forest, case_translation_group_prior_and_factor]


def case_multihypothesis_api(backend):
    # test/testmultihypothesisapi.jl:38-108: DevelopPrior = Prior, DevelopLikelihood: r = z - (x2 - x1);
    # multihypo [1, .5, .5] parsing and the three convolution directions of the fractional factor
    N = 100
    fg = iif.initfg(iif.SolverParams(N=N))
    iif.addVariable(fg, "x1", iif.ContinuousScalar)
    iif.addFactor(fg, ["x1"], iif.Prior(iif.Normal(10.0, 1.0)))
    iif.initAll(fg, backend=backend, seed=120)
    pts = iif.approxConv(fg, "x1f1", "x1", backend=backend, seed=121)[:, 0]
    assert (np.abs(pts - 1.0) < 5).sum() < 30 and (np.abs(pts - 10.0) < 5).sum() > 30
    iif.addVariable(fg, "x2", iif.ContinuousScalar)
    iif.addFactor(fg, ["x1", "x2"], iif.LinearRelative(iif.Normal(100.0, 1.0)))
    iif.initAll(fg, backend=backend, seed=122)
    assert abs(fg.getVal("x2").mean() - 110.0) < 10.0
    iif.addVariable(fg, "x3", iif.ContinuousScalar)
    iif.addVariable(fg, "x4", iif.ContinuousScalar)
    iif.addFactor(fg, ["x2", "x3", "x4"], iif.LinearRelative(iif.Normal(90.0, 1.0)), multihypo=[1.0, 0.5, 0.5])
    p = fg.getFactor("x2x3x4f1").multihypo
    assert abs(p[0]) < 0.1 and np.abs(p[1:] - 0.5).sum() < 0.1  # 1.0 becomes 0.0 for computational convenience
    for v, c in (("x2", 1.0), ("x3", 2.0), ("x4", 3.0)):
        iif.initVariable(fg, v, c * np.ones((N, 1)) + 1e-6 * rng(int(c)).normal(size=(N, 1)), backend=backend)
    pts = iif.approxConv(fg, "x2x3x4f1", "x2", backend=backend, seed=123)[:, 0]
    assert 99 < (pts <= -70.0).sum()
    for v, s in (("x3", 124), ("x4", 125)):
        pts = iif.approxConv(fg, "x2x3x4f1", v, backend=backend, seed=s)[:, 0]
        assert 15 < ((pts > 70) & (pts < 110)).sum() < 75


CASES.append(case_multihypothesis_api)


def case_joint_messages_circular(backend):
    # test/testCircular.jl:7-29: five Circular poses, prior at 0, CircularCircular(N(1, 0.1)) odometry, solved
    # with useMsgLikelihoods = true; PPE ~ rem2pi(0:4) within 0.35
    fg = iif.initfg(iif.SolverParams(N=100, useMsgLikelihoods=True))
    for i in range(5):
        iif.addVariable(fg, f"x{i}", iif.Circular)
    iif.addFactor(fg, ["x0"], iif.PriorCircular(iif.Normal(0.0, 0.1)))
    for i in range(4):
        iif.addFactor(fg, [f"x{i}", f"x{i + 1}"], iif.CircularCircular(iif.Normal(1.0, 0.1)))
    iif.solveTree(fg, backend=backend, seed=130)
    for i in range(5):
        th = fg.getVal(f"x{i}")[:, 0]
        m = np.arctan2(np.sin(th).mean(), np.cos(th).mean())
        gt = (i + np.pi) % (2 * np.pi) - np.pi
        assert abs((m - gt + np.pi) % (2 * np.pi) - np.pi) < 0.35, (i, m, gt)


def case_joint_messages_has_priors(backend):
    # test/testHasPriors913.jl:7-50: a 5-pose line with one landmark seen from both ends, initialised from a
    # WRONG prior (x0 ~ 5) and solved with the right one (x0 ~ 0) and useMsgLikelihoods: the common message
    # priors only travel up where a prior exists below (msg.hasPriors), the differential factors carry the
    # rest; after three solves x_i ~ i within 0.7
    def line(mu0):
        fg = iif.generateGraph_LineStep(4, poseEvery=1, landmarkEvery=5, posePriorsAt=(), landmarkPriorsAt=(), sightDistance=5,
                                        solverParams=iif.SolverParams(N=100))
        for i in (1, 2, 3):
            iif.deleteFactor(fg, f"x{i}lm0f1")
        iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal(mu0, 0.01)))
        return fg

    wrong = line(5.0)
    iif.initAll(wrong, backend=backend, seed=131)
    fg = line(0.0)
    for v in fg.ls():
        iif.initVariable(fg, v, wrong.getVal(v), backend=backend)
    fg.solverParams.useMsgLikelihoods = True
    fg.solverParams.graphinit = False
    for k in range(3):
        iif.solveTree(fg, backend=backend, seed=132 + k)
    for i in range(5):
        assert abs(float(np.median(fg.getVal(f"x{i}")[:, 0])) - i) < 0.7, (i, np.median(fg.getVal(f"x{i}")[:, 0]))


def case_joint_messages_caesar_ring(backend):
    # test/testUseMsgLikelihoods.jl:10-105: the 1-D Caesar ring solved upward only with useMsgLikelihoods and the
    # test's elimination order (the structural assertions of that test live in tests/test_joint_messages.py);
    # and test/testSpecialEuclidean2Mani.jl:206-211: an SE(2) graph solves in this mode
    fg = iif.initfg(iif.SolverParams(N=100, useMsgLikelihoods=True, downsolve=False))
    for v in ["x0", "x1", "x2", "x3", "x4", "x5", "x6"]:
        iif.addVariable(fg, v, iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal()))
    for a, b in [("x0", "x1"), ("x1", "x2"), ("x2", "x3"), ("x3", "x4"), ("x4", "x5"), ("x5", "x6")]:
        iif.addFactor(fg, [a, b], iif.LinearRelative(iif.Normal()))
    iif.addVariable(fg, "l1", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0", "l1"], iif.LinearRelative(iif.Normal()))
    iif.addFactor(fg, ["x6", "l1"], iif.LinearRelative(iif.Normal()))
    iif.solveTree(fg, eliminationOrder=["x3", "x5", "l1", "x1", "x6", "x4", "x2", "x0"], backend=backend, seed=136)
    for v in fg.ls():  # every belief stays a proper, centred density (all measurements are N(0, 1))
        x = fg.getVal(v)[:, 0]
        assert np.isfinite(x).all() and abs(x.mean()) < 2.5 and 0.3 < x.std() < 6.0, (v, x.mean(), x.std())
    fg = iif.initfg(iif.SolverParams(N=100, useMsgLikelihoods=True))
    for i in range(4):
        iif.addVariable(fg, f"x{i}", iif.SpecialEuclidean2)
    iif.addFactor(fg, ["x0"], iif.ManifoldPrior([0.0, 0.0, 0.0], iif.MvNormal(np.zeros(3), np.diag([0.1, 0.1, 0.01]) ** 2)))
    for i in range(3):
        iif.addFactor(fg, [f"x{i}", f"x{i + 1}"], iif.ManifoldFactor(iif.MvNormal([1.0, 0.0, 0.0], np.diag([0.1, 0.1, 0.01]) ** 2)))
    iif.addFactor(fg, ["x0", "x3"], iif.ManifoldFactor(iif.MvNormal([3.0, 0.0, 0.0], np.diag([0.1, 0.1, 0.01]) ** 2)))
    iif.solveTree(fg, eliminationOrder=["x1", "x2", "x3", "x0"], backend=backend, seed=137)
    for i in range(4):
        x, y, th = se2_mean(fg.getVal(f"x{i}"))
        assert abs(x - i) < 0.5 and abs(y) < 0.5 and abs(th) < 0.3, (i, x, y, th)


CASES += [case_joint_messages_circular, case_joint_messages_has_priors, case_joint_messages_caesar_ring]


def _ppe(fg, v):
    return float(np.median(fg.getVal(v)[:, 0]))


def case_two_priors_tight_links(backend):
    # test/priorusetest.jl:12-123: priors N(-1, 1) and N(+1, 1) at the two ends of very tight links (sigma 0.01):
    # every mean within 1.0 (1.2 for landmarks) of 0 and all of them within 0.4 / 0.3 of their average
    fg = iif.initfg(iif.SolverParams(N=100))
    for v in ("x0", "x1", "x2"):
        iif.addVariable(fg, v, iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal(-1.0, 1.0)))
    iif.addFactor(fg, ["x2"], iif.Prior(iif.Normal(+1.0, 1.0)))
    iif.addFactor(fg, ["x0", "x1"], iif.LinearRelative(iif.Normal(0.0, 0.01)))
    iif.addFactor(fg, ["x1", "x2"], iif.LinearRelative(iif.Normal(0.0, 0.01)))
    iif.solveTree(fg, backend=backend, seed=140)
    m = np.array([fg.getVal(v)[:, 0].mean() for v in ("x0", "x1", "x2")])
    assert (np.abs(m) < 1.0).all() and (np.abs(m - m.mean()) < 0.4).all(), m
    fg = iif.initfg(iif.SolverParams(N=100))
    for v in ("x0", "l0", "l1", "x1", "x2"):
        iif.addVariable(fg, v, iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal(-1.0, 1.0)))
    iif.addFactor(fg, ["l0"], iif.Prior(iif.Normal(+1.0, 1.0)))
    for a, b in (("x0", "l0"), ("x0", "l1"), ("x0", "x1"), ("x1", "x2"), ("x2", "l0"), ("x2", "l1")):
        iif.addFactor(fg, [a, b], iif.LinearRelative(iif.Normal(0.0, 0.01)))
    iif.solveTree(fg, backend=backend, seed=141)
    m = np.array([fg.getVal(v)[:, 0].mean() for v in ("x0", "x1", "x2", "l0", "l1")])
    assert (np.abs(m[:3]) < 1.0).all() and (np.abs(m[3:]) < 1.2).all() and (np.abs(m - m.mean()) < 0.3).all(), m


def case_pose_pose_constraint(backend):
    # test/testlocalconstraintexamples.jl:8-44: a wide prior at the door (the reference builds it from a one-point
    # KDE with bandwidth 3) and LinearRelative(N(50, 2)): the convolution and the solved x2 sit at 50 +- 15
    fg = iif.initfg(iif.SolverParams(N=100))
    iif.addVariable(fg, "x1", iif.ContinuousScalar)
    iif.addFactor(fg, ["x1"], iif.Prior(iif.Normal(0.0, 3.0)))
    iif.addVariable(fg, "x2", iif.ContinuousScalar)
    iif.addFactor(fg, ["x1", "x2"], iif.LinearRelative(iif.Normal(50.0, 2.0)))
    iif.initAll(fg, backend=backend, seed=142)
    pts = iif.approxConv(fg, "x1x2f1", "x2", backend=backend, seed=143)
    assert abs(pts[:, 0].mean() - 50.0) < 15.0
    iif.solveTree(fg, backend=backend, seed=144)
    assert abs(fg.getVal("x2")[:, 0].mean() - 50.0) < 15.0


def case_multihypo_three_landmarks(backend):
    # test/TestCSMMultihypo.jl:9-73 (#427: no runaway on the up solve) and test/testCalcFactorHypos.jl:41-78
    # (#424: the multihypo vector must have one entry per variable; the fractional factor solves)
    fg = iif.initfg(iif.SolverParams(N=100))
    for v, mu in (("l1", 50.0), ("l2", -50.0)):
        iif.addVariable(fg, v, iif.ContinuousScalar)
        iif.addFactor(fg, [v], iif.Prior(iif.Normal(mu, 0.1)))
    for v in ("l1_0", "l2_0", "x1"):
        iif.addVariable(fg, v, iif.ContinuousScalar)
    iif.addFactor(fg, ["x1", "l1", "l1_0"], iif.LinearRelative(iif.Normal(40.0, 0.25)), multihypo=[1.0, 0.5, 0.5])
    iif.addVariable(fg, "x2", iif.ContinuousScalar)
    iif.addFactor(fg, ["x1", "x2"], iif.LinearRelative(iif.Normal(0.0, 0.1)))
    iif.addFactor(fg, ["x2", "l2", "l2_0"], iif.LinearRelative(iif.Normal(-40.0, 0.25)), multihypo=[1.0, 0.5, 0.5])
    iif.solveTree(fg, backend=backend, seed=145)
    for v in fg.ls():
        assert np.isfinite(fg.getVal(v)).all()
    for v, mu in (("l1", 50.0), ("l2", -50.0)):  # the landmarks with priors stay put
        assert abs(_ppe(fg, v) - mu) < 1.0
    fg = iif.initfg(iif.SolverParams(N=100))
    for v in ("x0", "x1_a", "x1_b"):
        iif.addVariable(fg, v, iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal()))
    try:
        iif.addFactor(fg, ["x0", "x1_a", "x1_b"], iif.LinearRelative(iif.Normal(10.0, 1.0)), multihypo=[0.5, 0.5])
        raise AssertionError("a multihypo vector shorter than the variable list must be rejected (#424)")
    except ValueError:
        pass
    f = iif.addFactor(fg, ["x0", "x1_a", "x1_b"], iif.LinearRelative(iif.Normal(10.0, 1.0)), multihypo=[1, 0.5, 0.5])
    assert f.isMultihypo and not fg.getFactor("x0f1").isMultihypo
    iif.solveTree(fg, backend=backend, seed=146)
    assert abs(_ppe(fg, "x0")) < 1.0


def case_joint_messages_xstroke(backend):
    # test/testExpXstroke.jl:10-128 (#754, the endless-cycle graphs), all with useMsgLikelihoods = true:
    # PPE of every variable at its index within 0.2 / 0.4 / 0.45
    def solve(fg, seed):
        fg.solverParams.useMsgLikelihoods = True
        iif.solveTree(fg, backend=backend, seed=seed)

    sp = lambda: iif.SolverParams(N=100)
    fg = iif.generateGraph_LineStep(5, poseEvery=1, landmarkEvery=5, posePriorsAt=(0, 2), sightDistance=4, solverParams=sp())
    solve(fg, 147)
    for v in fg.ls():
        assert abs(_ppe(fg, v) - int(v.lstrip("xlm"))) < 0.2, (v, _ppe(fg, v))
    N = 8
    fg = iif.generateGraph_LineStep(N, poseEvery=1, landmarkEvery=N + 1, posePriorsAt=(0,), landmarkPriorsAt=(), sightDistance=N + 1,
                                    solverParams=sp())
    for i in range(1, N):
        iif.deleteFactor(fg, f"x{i}lm0f1")
    solve(fg, 148)
    for v in fg.ls():
        assert abs(_ppe(fg, v) - int(v.lstrip("xlm"))) < 0.4, (v, _ppe(fg, v))
    fg = iif.generateGraph_LineStep(15, poseEvery=1, landmarkEvery=3, posePriorsAt=(0, 7, 12), landmarkPriorsAt=(0, 3), sightDistance=2,
                                    solverParams=sp())
    solve(fg, 149)
    for v in fg.ls():
        assert abs(_ppe(fg, v) - int(v.lstrip("xlm"))) < 0.45, (v, _ppe(fg, v))


CASES += [case_two_priors_tight_links, case_pose_pose_constraint, case_multihypo_three_landmarks, case_joint_messages_xstroke]
