"""-m gpu: whole solves of random graphs (random manifold, loop closures, multihypo, mixtures, nullhypo,
marginalized variables) -- HIP backend with the native host's schedule against the CPU oracle with the
Python mirror's schedule, identical seeds."""
import numpy as np
import pytest

from kl_parity import compare_solves
from parity_utils import abi, iif, record_parity
from test_native_host import random_graph

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", range(12))
def test_random_graph_solve_matches_oracle(oracle_backend, hip_backend, seed):
    fa, fb = random_graph(seed), random_graph(seed)
    order = iif.nestedDissectionOrder(fa)
    try:
        iif.solveTree(fa, eliminationOrder=order, backend=oracle_backend, seed=seed)
    except ValueError:
        pytest.skip("a product wider than NBP_MAXF")
    iif.solveTree(fb, eliminationOrder=order, backend=hip_backend, seed=seed)

    def another_oracle_solve():
        fc = random_graph(seed)
        iif.solveTree(fc, eliminationOrder=order, backend=oracle_backend, seed=seed + 1000)
        return fc

    # Identical streams give identical particles until a data-dependent branch (a Nelder-Mead comparison, a
    # golden-section step) resolves differently on the two sides; from there on the two runs are independent draws of
    # the same sampler, and these graphs -- inconsistent multihypo loops, nullhypo -- have multi-modal posteriors whose
    # independent draws differ by whole modes (two ORACLE solves of graph 9 with different seeds put v19 at 3.4 and at
    # 0.5 with a spread of 0.1).  Criterion (tests/kl_parity.py): particle-identical, or no further from the oracle
    # solve in symmetric KL than a second oracle solve with another seed is.
    share, kl = compare_solves(fa, fb, another_oracle_solve)
    line = f"random graph {seed}: {share:.0%} of {len(fa.ls())} variables particle-identical (1e-6) to the oracle solve; symKL max {max(kl.values()):.3f} nats"
    print(line)
    record_parity(line)
    # (no floor here: eight of the twelve graphs come out particle-identical, the ones with three-dimensional searches or
    #  inconsistent multihypo loops part ways early -- profiles/r04_whole_solve_parity.txt -- and are held to the
    #  two-sample criterion above)
