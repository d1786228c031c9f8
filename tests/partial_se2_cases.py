"""Partial relative factors on SE(2) -- the reference's `.partial` mechanism (EvalFactor.jl:184-198, NumericalCalculations.jl:
424-446) applied to ManifoldFactor's residual: only the residual components in `.partial` count, entropy goes on those
coordinates only, the per-particle search is BFGS over the whole point.  Run on the oracle (test_partial_se2.py) and on
the GPU (test_gpu_partial_se2.py).  Known answers that depend on no random stream: with a noise-free measurement the
constrained components of the residual vanish at the solution and the unconstrained coordinates keep the target's values."""
import numpy as np

from parity_utils import abi, iif, rand_points, relative_factor_desc


def _theta(p):
    return np.arctan2(p[:, 3], p[:, 2])


def _wrap(a):
    return (a + np.pi) % (2 * np.pi) - np.pi


def _residual(a, b, z):
    """ManifoldFactor(SE2) residual (GenericFunctions.jl:39-44) of points a -> b (n x 6 host layout) for measurement z"""
    ta, tb = _theta(a), _theta(b)
    c, s = np.cos(ta), np.sin(ta)
    r0 = a[:, 0] + c * z[0] - s * z[1] - b[:, 0]
    r1 = a[:, 1] + s * z[0] + c * z[1] - b[:, 1]
    r2 = _wrap(ta + z[2] - tb)
    return np.stack([r0, r1, r2], axis=1)


def _setup(make, N=100):
    be = make(N, 6, 0)
    rng = np.random.default_rng(4)
    a, b = rand_points(rng, abi.SE2, N, 0.5, 0.4), rand_points(rng, abi.SE2, N, 1.5, 0.4)
    be.slot_write(0, abi.SE2, a)
    be.slot_write(1, abi.SE2, b)
    return be, a, b


def case_translation_only(make):
    """partial = (1, 2): the translation of the second pose is solved, its heading keeps the target's values"""
    z = [1.0, -0.5, 0.3]
    be, a, b = _setup(make)
    d = relative_factor_desc(abi.F_SE2, abi.SE2, 2, 1, [0, 1], 2, 21, z, [1e-9, 1e-9, 1e-9], partial_mask=3, inflate_cycles=3)
    be.run_proposals([d])
    out, bw = be.slot_read(2, abi.SE2)
    r = _residual(a, out, z)
    assert np.abs(r[:, :2]).max() < 1e-5, np.abs(r[:, :2]).max()
    np.testing.assert_allclose(_theta(out), _theta(b), atol=1e-12)   # heading untouched: no entropy, zero gradient
    assert np.abs(r[:, 2]).max() > 0.1                                # ... and NOT at the factor's full root
    be.close()
    return out, bw


def case_heading_only(make):
    """partial = (3,): the heading is solved, the translation keeps the target's values"""
    z = [1.0, -0.5, 0.3]
    be, a, b = _setup(make)
    d = relative_factor_desc(abi.F_SE2, abi.SE2, 2, 1, [0, 1], 2, 22, z, [1e-9, 1e-9, 1e-9], partial_mask=4, inflate_cycles=3)
    be.run_proposals([d])
    out, bw = be.slot_read(2, abi.SE2)
    r = _residual(a, out, z)
    assert np.abs(r[:, 2]).max() < 1e-5
    np.testing.assert_allclose(out[:, :2], b[:, :2], atol=1e-12)
    be.close()
    return out, bw


def case_first_pose_translation(make):
    """solving for the FIRST pose with partial = (1, 2): two residual components, three decision variables (the heading
    rotates the measurement): BFGS finds a point where the translation residual vanishes"""
    z = [1.0, -0.5, 0.3]
    be, a, b = _setup(make)
    d = relative_factor_desc(abi.F_SE2, abi.SE2, 2, 0, [0, 1], 2, 23, z, [1e-9, 1e-9, 1e-9], partial_mask=3, inflate_cycles=3)
    be.run_proposals([d])
    out, bw = be.slot_read(2, abi.SE2)
    r = _residual(out, b, z)
    assert np.abs(r[:, :2]).max() < 1e-4, np.abs(r[:, :2]).max()
    be.close()
    return out, bw


def case_in_a_graph(backend):
    """a pose with a full odometry factor and a heading-only constraint to a third pose: the product takes the heading
    from both, the translation from the odometry alone"""
    SE2 = iif.SpecialEuclidean2
    fg = iif.initfg(iif.SolverParams(N=100))
    for v in ("x0", "x1", "x2"):
        iif.addVariable(fg, v, SE2)
    iif.addFactor(fg, ["x0"], iif.ManifoldPrior([0.0, 0.0, 0.0], iif.MvNormal(np.zeros(3), [0.05, 0.05, 0.02])))
    iif.addFactor(fg, ["x0", "x1"], iif.ManifoldFactor(iif.MvNormal([1.0, 0.0, 0.4], [0.05, 0.05, 0.2])))
    iif.addFactor(fg, ["x2"], iif.ManifoldPrior([5.0, 5.0, 1.0], iif.MvNormal(np.zeros(3), [0.05, 0.05, 0.02])))
    iif.addFactor(fg, ["x2", "x1"], iif.PartialManifoldFactor(SE2, iif.MvNormal([0.0, 0.0, -0.5], [1.0, 1.0, 0.02]), (3,)))
    iif.initAll(fg, backend=backend, seed=5)
    iif.solveTree(fg, backend=backend, seed=6)
    p = fg.getVal("x1")
    th = _theta(p)
    assert abs(p[:, 0].mean() - 1.0) < 0.15 and abs(p[:, 1].mean()) < 0.15       # translation: odometry only
    assert abs(_wrap(th - 0.5).mean()) < 0.08 and th.std() < 0.1                   # heading: pinned by the partial factor (1.0 - 0.5)
    return p


CASES = [case_translation_only, case_heading_only, case_first_pose_translation]
