"""-m gpu: whole solves of random graphs (random manifold, loop closures, multihypo, mixtures, nullhypo,
marginalized variables) -- HIP backend with the native host's schedule against the CPU oracle with the
Python mirror's schedule, identical seeds."""
import numpy as np
import pytest

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

    # Identical streams give identical particles -- to the last bit since round 6 (one arithmetic for the values that travel:
    # DESIGN.md section 5); through round 5 the graphs with three-dimensional searches or inconsistent multihypo loops parted
    # from the oracle at a last-bit branch flip and were held to a two-sample KL criterion (tests/kl_parity.py, retired).
    differ = [v for v in fa.ls() if not (np.array_equal(fa.getVal(v), fb.getVal(v)) and
                                         np.array_equal(np.asarray(fa.getVariable(v).bw), np.asarray(fb.getVariable(v).bw)))]
    line = f"random graph {seed}: {len(fa.ls()) - len(differ)} of {len(fa.ls())} variables BIT-identical to the oracle solve"
    print(line)
    record_parity(line)
    assert not differ, (seed, differ, {v: float(np.abs(fa.getVal(v) - fb.getVal(v)).max()) for v in differ[:5]})
