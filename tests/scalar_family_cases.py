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


def case_alias_sampler_prior(backend):
    """Prior(AliasingScalarSampler(domain, weights)) (entities/AliasScalarSampling.jl:13-74): every particle sits ON a domain
    value, the values come up with their weights (chi-square against the pmf), a zero-weight value never does; the
    constructor's SNRfloor conditioning is the reference's"""
    rng = np.random.default_rng(11)
    dom = np.arange(1.0, 31.0)
    w = rng.uniform(size=30)
    w[9] = 0.0
    w[19:25] += 4.0
    bss = iif.AliasingScalarSampler(dom, w)
    assert abs(bss.weights.sum() - 1.0) < 1e-12 and bss.weights[9] == 0.0
    fg = graph(iif.Prior(bss))
    pts, _ = iif.approxConvBelief(fg, "x0f1", "x0", backend=backend, seed=7)
    x = pts[:, 0]
    assert np.isin(x, dom).all()
    cnt = np.array([(x == d).sum() for d in dom])
    assert cnt[9] == 0
    exp = bss.weights * N
    big = exp > 5
    chi2 = float(((cnt[big] - exp[big]) ** 2 / exp[big]).sum())
    assert chi2 < 3.0 * big.sum(), (chi2, big.sum())
    assert abs(x.mean() - float((dom * bss.weights).sum())) < 4 * np.sqrt(float(((dom - (dom * bss.weights).sum()) ** 2 * bss.weights).sum()) / N)
    # SNRfloor: the lowest half of the pmf is removed before sampling
    f2 = iif.AliasingScalarSampler(dom, w, SNRfloor=0.5)
    assert (f2.weights > 0).sum() <= 15 and abs(f2.weights.sum() - 1.0) < 1e-12
    return pts


def case_mixture_with_an_alias_sampler(backend):
    """test/testMixturePrior.jl:22-63 at its own size: Mixture(Prior, (Normal(-5, 1), bss), Categorical([.5, .5])) with bss an
    AliasingScalarSampler over 1:50 -- "should be a balance of particles" either side of -2.5, after approxConv and after
    solveTree!"""
    rng = np.random.default_rng(12)
    v = rng.uniform(size=50)
    v[19:29] += 5 * rng.uniform(size=10)
    bss = iif.AliasingScalarSampler(np.arange(1.0, 51.0), v / v.sum())
    fg = iif.initfg(iif.SolverParams(N=100))
    iif.addVariable(fg, "x0", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Mixture(iif.Prior, (iif.Normal(-5.0, 1.0), bss), [0.5, 0.5]), label="x0f1")
    pts, _ = iif.approxConvBelief(fg, "x0f1", "x0", backend=backend, seed=8)
    x = pts[:, 0]
    assert abs(np.sum(x < -2.5) - np.sum(x > -2.5)) < 0.35 * 100
    right = x[x > -2.5]
    assert np.isin(right, np.arange(1.0, 51.0)).all()   # the sampler's component: domain values only
    iif.solveTree(fg, backend=backend, seed=9)
    m = fg.getVariable("x0").val[:, 0]
    assert abs(np.sum(m < -2.5) - np.sum(m > -2.5)) < 0.35 * 100
    return pts


def case_alias_sampler_relative(backend):
    """a relative factor whose measurement is tabulated: x1 = x0 + z with z from a three-valued table.  The constructor
    removes the SNRfloor QUANTILE of the pmf, and the 0-quantile is its minimum (AliasScalarSampling.jl:34-38): weights
    (.1, .3, .6) become (0, 2/7, 5/7) -- the reference's behaviour, kept"""
    bss = iif.AliasingScalarSampler([2.0, 7.0, 12.0], [0.1, 0.3, 0.6])
    assert np.allclose(bss.weights, [0.0, 2 / 7, 5 / 7])
    fg = graph(iif.Prior(iif.Normal(0.0, 0.01)), iif.LinearRelative(bss))
    iif.initAll(fg, backend=backend, seed=10)
    x1 = fg.getVariable("x1").val[:, 0]
    near7, near12 = np.abs(x1 - 7.0) < 0.2, np.abs(x1 - 12.0) < 0.2
    assert (near7 | near12).all()
    assert abs(near12.mean() - 5 / 7) < 0.08
    iif.solveTree(fg, backend=backend, seed=11)   # the whole-tree program carries the table slot too
    x1 = fg.getVariable("x1").val[:, 0]
    assert np.isfinite(x1).all() and ((np.abs(x1 - 7.0) < 0.5) | (np.abs(x1 - 12.0) < 0.5)).mean() > 0.9
    return x1
