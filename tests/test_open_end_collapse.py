"""The open end of a chain on the ORACLE (no GPU): what bench_support's acceptance of config 5 has to allow for.

fmcmc! updates a variable with the product of the proposals of ALL its factors in the clique's sub graph (propagateBelief,
GraphProductOperations.jl:16-64), the neighbours' CURRENT beliefs as operands.  Where the frontals of a clique see each other
through a factor -- the last cliques of a chain: {x_{n-1} | x_{n-2}}, {x_{n-2}, x_{n-3} | ...} -- a belief is multiplied, iteration
after iteration, with a proposal that was made from itself: the beliefs of the dangling end narrow far below the exact posterior
(whose std at pose d of a one-prior Mixture chain is sqrt(0.208 d)) and sit on a point that is spread like a draw from it.  The
device reproduces the oracle stage by stage (tests/test_gpu_stagewise_parity.py); this pins the behaviour itself on the
restatement, at a size the oracle solves in a second: the numbers at BASELINE's size are in profiles/r05_open_end_of_a_chain.txt
(std of x397 .. x399 after one solve 0.16-0.34 of the exact sigma; the mean of the end pose spread with std ~0.25 sigma over
graph initialisations, 0.87 sigma the largest seen)."""
import numpy as np

import iif_amd_loader

iif = iif_amd_loader.load()


def test_beliefs_at_the_dangling_end_of_a_chain_are_narrower_than_the_exact_posterior():
    from oracle.oracle_backend import OracleBackend
    nv, N = 40, 150
    ratios_end, ratios_quarter, zs = [], [], []
    for seed in (0, 1, 2):
        fg = iif.generateMixtureChain(nvars=nv, N=N, priorEvery=500)  # one prior, on x0
        order = iif.nestedDissectionOrder(fg)
        iif.solveTree(fg, eliminationOrder=order, backend=lambda n, k, side_ints=0: OracleBackend(n, k, side_ints, threads=4), seed=seed)
        sig = lambda i: np.sqrt(0.208 * i)
        ratios_end.append(np.mean([fg.getVal(f"x{i}").std(axis=0).mean() / sig(i) for i in (nv - 3, nv - 2, nv - 1)]))
        ratios_quarter.append(fg.getVal(f"x{nv // 4}").std(axis=0).mean() / sig(nv // 4))
        zs.append(np.abs(fg.getVal(f"x{nv - 1}").mean(axis=0) - np.array([nv - 1.0, 0.0, 0.0])).max() / sig(nv - 1))
    # the end is narrower than the exact posterior by more than a factor of two, and narrower than the chain's first quarter
    assert max(ratios_end) < 0.55, ratios_end
    assert np.mean(ratios_end) < np.mean(ratios_quarter) - 0.1, (ratios_end, ratios_quarter)
    # ... and its mean is where a draw of the exact posterior could be (what bench_support accepts: 1 + 1.5 sigma)
    assert max(zs) < 1.5, zs
