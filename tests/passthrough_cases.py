"""PartialPriorPassThrough (Factors/PartialPriorPassThrough.jl; calcProposalBelief dispatch, ApproxConv.jl:196-227): a
prior whose density goes to inference as it is.  The cases follow the reference's two test sets
(test/testSpecialEuclidean2Mani.jl:331-451 "w Priors", :456-527 "w Relative"); written once, run on the oracle
(tests/test_passthrough_prior.py) and on the GPU (tests/test_gpu_passthrough_prior.py).

The reference builds the density from a LevelSetGridNormal over a random image (N = 120 points); here the density is
120 seeded points on the first two coordinates of an SE(2) pose -- what inference sees is the ManifoldKernelDensity
either way."""
import numpy as np

from parity_utils import abi, iif

NDENS = 120


def density(seed=0, n=NDENS):
    rng = np.random.default_rng(seed)
    # a ring: the level set of a bowl, far from Gaussian
    a = rng.uniform(-np.pi, np.pi, n)
    r = 6.0 + 0.3 * rng.normal(size=n)
    return np.stack([r * np.cos(a), r * np.sin(a)], axis=1), np.array([0.35, 0.35])


def graph_w_priors(N=150, nullhypo=0.0, second_prior=False):
    """N = 150 > 120: a slot holds at most the context's N points, so the density's own count survives only below it
    (the reference runs this with N = 100 < 120; case_init_with_more_points_than_n covers that side: the first N)"""
    sp = iif.SolverParams(N=N)
    fg = iif.initfg(sp)
    iif.addVariable(fg, "x0", iif.SpecialEuclidean2)
    pts, bw = density()
    iif.addFactor(fg, ["x0"], iif.PartialPriorPassThrough(iif.SpecialEuclidean2, pts, bw, (1, 2)), nullhypo=nullhypo, label="x0f1")
    if second_prior:
        iif.addFactor(fg, ["x0"], iif.ManifoldPrior(np.zeros(3), iif.MvNormal(np.zeros(3), np.diag([0.01, 0.01, 0.01]) ** 2)), label="x0f2")
    return fg


def graph_w_relative(N=150):
    """x0 --pass-through (1,2)--   x0 --ManifoldFactor--> x1 <-- ManifoldPrior"""
    fg = graph_w_priors(N)
    iif.addVariable(fg, "x1", iif.SpecialEuclidean2)
    iif.addFactor(fg, ["x1"], iif.ManifoldPrior(np.zeros(3), iif.MvNormal(np.zeros(3), np.diag([0.01, 0.01, 0.01]) ** 2)), label="x1f1")
    iif.addFactor(fg, ["x0", "x1"], iif.ManifoldFactor(iif.MvNormal([1.0, 2.0, np.pi / 4], np.diag([0.01, 0.01, 0.01]) ** 2)), label="x0x1f1")
    return fg


def se2_coords(pts):
    return np.stack([pts[:, 0], pts[:, 1], np.arctan2(pts[:, 3], pts[:, 2])], axis=1)


def case_alone_keeps_the_density(backend, nullhypo=0.0):
    """propagateBelief(fg, x0, [f0]): the belief IS the density -- its 120 points on the partial coordinates, its
    bandwidth (:366-369, with nullhypo :378-383: evalFactor is bypassed)"""
    fg = graph_w_priors(nullhypo=nullhypo)
    (pts, bw), ipc = iif.propagateBelief(fg, "x0", ["x0f1"], backend=backend, seed=5)
    assert pts.shape == (NDENS, 6)
    d, h = density()
    c = se2_coords(pts)
    np.testing.assert_allclose(c[:, :2], d, atol=1e-12)
    np.testing.assert_allclose(c[:, 2], 0.0, atol=1e-12)   # the coordinate the density says nothing about: the variable's
    np.testing.assert_allclose(bw, [h[0], h[1], 0.0])
    np.testing.assert_allclose(ipc, [1.0, 1.0, 1.0])  # fct_ipc = ones(vardim), partial or not (ApproxConv.jl:273)
    return pts


def case_conv_is_the_density(backend):
    """approxConvBelief through the factor: the same density, nothing sampled"""
    fg = graph_w_priors()
    pts, bw = iif.approxConvBelief(fg, "x0f1", "x0", backend=backend, seed=6)
    assert pts.shape[0] == NDENS
    np.testing.assert_allclose(se2_coords(pts)[:, :2], density()[0], atol=1e-12)
    return pts


def case_product_with_a_prior_has_n_points(backend):
    """propagateBelief(fg, x0, [f0; f1]) (:423-425): a product is calculated, the belief has N points; the prior pins
    theta, (x, y) is the product of the ring with the prior at the origin"""
    fg = graph_w_priors(second_prior=True)
    (pts, bw), ipc = iif.propagateBelief(fg, "x0", ["x0f1", "x0f2"], backend=backend, seed=7)
    N = fg.solverParams.N
    assert pts.shape == (N, 6)
    c = se2_coords(pts)
    assert np.all(np.isfinite(c)) and np.abs(c[:, 2]).max() < 0.1
    np.testing.assert_allclose(ipc, [2.0, 2.0, 2.0])
    assert np.all(bw > 0)
    return pts


def case_product_with_a_relative_is_full(backend):
    """propagateBelief(fg, x0, [f0; f2]) (:491-494): not partial, N points"""
    fg = graph_w_relative()
    (p1, b1), _ = iif.propagateBelief(fg, "x1", ["x1f1"], backend=backend, seed=3)  # doautoinit!(fg, :x1)
    iif.setValKDE(fg, "x1", p1, b1, True)
    (pts, bw), ipc = iif.propagateBelief(fg, "x0", ["x0f1", "x0x1f1"], backend=backend, seed=8)
    N = fg.solverParams.N
    assert pts.shape == (N, 6)
    np.testing.assert_allclose(ipc, [2.0, 2.0, 2.0])
    assert np.all(bw > 0)  # every coordinate informed
    return pts


def case_init_restricts_the_graph_to_n(backend, nullhypo=0.2):
    """doautoinit!(fg, :x0) (:387-389): "while the propagate step might allow large point counts, the graph should stay
    restricted to N" (GraphInit.jl:174-177) -- N = 100 points, the first ... of them the density's own"""
    fg = graph_w_priors(N=150, nullhypo=nullhypo)
    assert iif.initAll(fg, backend=backend, seed=9) == 1
    v = fg.getVariable("x0")
    assert v.initialized and v.val.shape == (150, 6)
    c = se2_coords(v.val)
    d, h = density()
    np.testing.assert_allclose(c[:NDENS, :2], d, atol=1e-12)
    # the 30 drawn from the KDE: each within a few bandwidths of a point of the density, none on top of one
    dist = np.linalg.norm(c[NDENS:, None, :2] - d[None, :, :], axis=2).min(axis=1)
    assert dist.max() < 6 * h[0] and dist.min() > 1e-6
    np.testing.assert_allclose(c[:, 2], 0.0, atol=1e-12)
    return v.val


def case_init_with_more_points_than_n(backend):
    """N = 100 < 120: the graph takes the first N"""
    fg = graph_w_priors(N=100)
    iif.initAll(fg, backend=backend, seed=10)
    v = fg.getVariable("x0")
    assert v.val.shape == (100, 6)
    np.testing.assert_allclose(se2_coords(v.val)[:, :2], density()[0][:100], atol=1e-12)
    return v.val


def case_solve(backend, native=None, N=150):
    """solveTree! / initAll! on the graph with the relative (:498-523): N points on both variables; x1 is pinned at the
    origin by its prior, so x0 = x1 (-) z sits at the pose that maps onto the origin through z = (1, 2, pi/4), if the
    ring allows -- the ring (radius 6) does not contain that pose, the relative wins on (x, y) up to the ring's pull"""
    fg = graph_w_relative(N)
    iif.initAll(fg, backend=backend, seed=11)
    for v in ("x0", "x1"):
        assert fg.getVariable(v).val.shape == (N, 6)
    iif.solveTree(fg, backend=backend, seed=12, native=native)
    out = {}
    for v in ("x0", "x1"):
        val = fg.getVariable(v).val
        assert val.shape == (N, 6) and np.all(np.isfinite(val))
        out[v] = val
    c1 = se2_coords(out["x1"])
    assert np.abs(c1.mean(axis=0)).max() < 0.1
    return out
