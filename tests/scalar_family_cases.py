"""Scalar measurement families beyond the Gaussian (enum nbp_dist): Uniform and Rayleigh, as the reference's tests use
them -- Mixture(Prior, (Normal(-5, 1), Uniform(0, 1)), (0.5, 0.5)) (test/testMixturePrior.jl:30),
LinearRelative(Rayleigh()) (test/testCompareVariablesFactors.jl:106).  Known answers are the distributions' own
moments and supports; written once, run on the oracle and on the GPU."""
import numpy as np

from parity_utils import abi, iif

N = 400


def graph(prior, rel=None):
    fg = iif.initfg(iif.SolverParams(N=N))
    iif.addVariable(fg, "x0", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], prior, label="x0f1")
    if rel is not None:
        iif.addVariable(fg, "x1", iif.ContinuousScalar)
        iif.addFactor(fg, ["x0", "x1"], rel, label="x0x1f1")
    return fg


def case_uniform_prior(backend):
    fg = graph(iif.Prior(iif.Uniform(2.0, 5.0)))
    pts, _ = iif.approxConvBelief(fg, "x0f1", "x0", backend=backend, seed=1)
    x = pts[:, 0]
    assert x.min() >= 2.0 and x.max() <= 5.0
    assert abs(x.mean() - 3.5) < 4 * (3.0 / np.sqrt(12)) / np.sqrt(N)
    assert abs(x.std() - 3.0 / np.sqrt(12)) < 0.1
    assert abs(np.mean(x < 3.0) - 1 / 3) < 0.08  # flat, not bell shaped
    return pts


def case_rayleigh_prior(backend):
    s = 1.7
    fg = graph(iif.Prior(iif.Rayleigh(s)))
    pts, _ = iif.approxConvBelief(fg, "x0f1", "x0", backend=backend, seed=2)
    x = pts[:, 0]
    assert x.min() > 0.0
    assert abs(x.mean() - s * np.sqrt(np.pi / 2)) < 0.15
    assert abs(np.median(x) - s * np.sqrt(2 * np.log(2))) < 0.15
    assert abs(x.var() - (4 - np.pi) / 2 * s * s) < 0.35
    return pts


def case_mixture_with_a_uniform_component(backend):
    """testMixturePrior.jl:53: "should be a balance of particles" around -2.5"""
    fg = graph(iif.Mixture(iif.Prior, [iif.Normal(-5.0, 1.0), iif.Uniform(0.0, 1.0)], [0.5, 0.5]))
    pts, _ = iif.approxConvBelief(fg, "x0f1", "x0", backend=backend, seed=3)
    x = pts[:, 0]
    assert abs(np.sum(x < -2.5) - np.sum(x > -2.5)) < 0.35 * N
    right = x[x > -2.5]
    assert right.min() >= 0.0 and right.max() <= 1.0  # the uniform component, nothing Gaussian about it
    iif.solveTree(fg, backend=backend, seed=4)
    m = fg.getVariable("x0").val[:, 0]
    assert abs(np.sum(m < -2.5) - np.sum(m > -2.5)) < 0.35 * N
    return pts


def case_rayleigh_relative(backend):
    """x1 = x0 + z, z ~ Rayleigh(1), x0 ~ N(0, 0.01): the forward convolution is the Rayleigh shifted by x0"""
    fg = graph(iif.Prior(iif.Normal(0.0, 0.01)), iif.LinearRelative(iif.Rayleigh(1.0)))
    iif.initAll(fg, backend=backend, seed=5)
    x1 = fg.getVariable("x1").val[:, 0]
    assert x1.min() > -0.1
    assert abs(x1.mean() - np.sqrt(np.pi / 2)) < 0.15
    pts, _ = iif.approxConvBelief(fg, "x0x1f1", "x0", backend=backend, seed=6)  # and back: x0 = x1 - z'
    assert abs(pts[:, 0].mean()) < 0.25
    return x1
