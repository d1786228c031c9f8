"""-m gpu: the BASELINE.md 5 symmetric-KL figure, GPU solve against oracle solve, on every BASELINE configuration at
reduced size and on the exact Gaussian chain.  See tests/kl_parity.py for the criterion."""
import numpy as np
import pytest

import kl_tools
from kl_parity import compare_solves
from parity_utils import abi, iif

pytestmark = pytest.mark.gpu


def config1():
    fg = iif.initfg(iif.SolverParams(N=100))
    for i in range(6):
        iif.addVariable(fg, f"x{i}", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal(0.0, 1.0)))
    for i in range(5):
        iif.addFactor(fg, [f"x{i}", f"x{i+1}"], iif.LinearRelative(iif.Normal(1.0, 0.1)))
    return fg


CONFIGS = {
    "config1_scalar_chain": config1,
    "config2_euclid2_chain": lambda: iif.generateChainEuclid(40, vardims=2, priorEvery=10, N=200),
    "config3_circular_doors": lambda: iif.generateCircularDoors(nposes=25, N=200, sightEvery=10),
    "config4_se2_lattice": lambda: iif.generateSE2Lattice(rows=3, cols=5, N=200, closeEvery=2),
    "config5_mixture_chain": lambda: iif.generateMixtureChain(nvars=24, N=300, priorEvery=8),
}


@pytest.mark.parametrize("name", list(CONFIGS))
def test_symmetric_kl_gpu_vs_oracle(oracle_backend, hip_backend, name):
    build = CONFIGS[name]
    fo, fg_, fo2 = build(), build(), build()
    order = iif.nestedDissectionOrder(fo)
    iif.solveTree(fo, eliminationOrder=order, backend=oracle_backend, seed=31)
    iif.solveTree(fg_, eliminationOrder=order, backend=hip_backend, seed=31)
    iif.solveTree(fo2, eliminationOrder=order, backend=oracle_backend, seed=32)
    share, kl = compare_solves(fo, fg_, fo2)
    print(f"{name}: {share:.0%} of the variables agree particle by particle; symKL max {max(kl.values()):.3f} median {np.median(list(kl.values())):.3f}")


@pytest.mark.parametrize("seed", [3, 17])
def test_symmetric_kl_against_exact_gaussian_chain(hip_backend, seed):
    """The exact posterior of the 41-pose chain is known; the reference algorithm is over-confident by construction
    (testBasicGraphs.jl:89,110), so its KL to the exact posterior is NOT small -- the figure is
    recorded and bounded by what the documented width range (0.35x ... 3.5x the exact sigma, mean error < 1.2 sigma)
    implies, not by 0.05."""
    from exact_gaussian import chain_with_end_priors
    fg, mu, sig = chain_with_end_priors(41)
    iif.solveTree(fg, eliminationOrder=iif.nestedDissectionOrder(fg), backend=hip_backend, seed=seed)
    kl = np.array([kl_tools.symmetric_kl_to_gaussian(fg.getVal(f"x{i}")[:, 0], mu[i], sig[i]) for i in range(41)])
    print(f"seed {seed}: symKL to the exact posterior: median {np.median(kl):.3f} max {kl.max():.3f}")
    assert np.median(kl) < 0.6 and kl.max() < 4.0, (np.median(kl), kl.max())
