"""-m gpu: a differential fuzz of the op kernels against the oracle, bit for bit (tests/fuzz_proposals.py, fuzz_products.py, fuzz_graphs.py; runners of the same names in tools/exp/).

Random descriptors over everything a descriptor can say -- factor kinds x manifolds x the variable solved for x nullhypo x
mixtures x door-sighting multihypo x partial masks x inflation cycles and spread x measurement noise 1e-3 .. 3 x beliefs at
0 / 100 / -1e4 with spreads 1e-3 .. 3 (on the circle: all the way round); products of 1 .. 7 (now and then 20 / 60) densities,
Niter 1 .. 3, partial inputs with old points, labels -- in mixed launches and in uniform ones of every geometry (a lone op,
dozens, hundreds: the latency kernels, the single-manifold ones, one wave per proposal, throughput rows), N = 64 / 200 / 257 / 300.
The builder's run of 36 000 ops (profiles/r06_fuzz_ops.txt) found what four rounds of parity tests on well-scaled inputs had
not: a search that STALLS with an objective of ~1e8 (a pose 1e4 from its start) compares EQUAL vertex values, and Optim orders
those by their slot (nm_cswap in csrc/nbp_device.h).  That launch is the first case here."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    """a fresh copy of tests/<name>.py (the tests set its module-level switches)"""
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tests", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_the_launch_in_which_a_stalled_search_compared_equal_vertex_values():
    fz = load("fuzz_proposals")
    fz.SHORT = False  # (the generator's stream of the run that found it: every belief with N points)
    n, bad = fz.run_launch(1000 * 19 + 400, 300, 400, fz.KINDS[4], True)  # seed 19's 400 plain SE(2) proposals: op 358
    assert n == 400 and not bad, bad[:3]


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_proposal_launches_are_the_oracles(seed):
    fz = load("fuzz_proposals")
    N = [64, 200, 257, 300][seed % 4]
    for B, which, simple in ((60, None, False), (40, fz.KINDS[1 + seed % 4], False), (1200, fz.KINDS[1 + seed % 2], True), (1, None, False)):
        n, bad = fz.run_launch(500000 + 1000 * seed + B, N, B, which, simple)
        assert n == B and not bad, (B, which, bad[:3])


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_product_launches_are_the_oracles(seed):
    fz = load("fuzz_products")
    N = [64, 200, 257, 300][seed % 4]
    for B, man in ((90, None), (1, None), (90, fz.MANS[seed % 5]), (400, fz.MANS[1 + seed % 4])):
        n, bad = fz.run_launch(900000 + 7000 * seed + B, N, B, man)
        assert n == B and not bad, (B, man, bad[:3])


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_random_deconvolutions_are_the_oracles(seed):
    fz = load("fuzz_proposals")
    N = [64, 200, 257, 300][seed % 4]
    for B in (1, 120):
        n, bad = fz.run_deconv_launch(300000 + 10 * seed + B, N, B)
        assert n == B and not bad, (B, bad[:3])


@pytest.mark.parametrize("seed", range(16))
def test_whole_solves_of_random_graphs_on_every_manifold_are_the_oracles(seed):
    """tests/fuzz_graphs.py: all five manifolds (tests/test_gpu_random_graphs.py draws from three), odometry steps up to
    1000 and priors at 1e4, joint messages on every third graph; 660 such solves in profiles/r06_fuzz_graphs.txt"""
    fz = load("fuzz_graphs")
    info, res, why = fz.solve_pair(seed)
    assert res is not None, why
    nv, differ, worst, finite = res
    assert finite and not differ, (info, differ[:5], worst)


@pytest.mark.parametrize("seed", [0, 2, 3, 4, 5, 6, 7, 8])
def test_random_graphs_sharded_over_emulated_ranks_are_the_one_rank_program(seed):
    """row (e) on graphs nobody drew by hand: 2 .. 4 ranks emulated on one GPU (tests/fuzz_graphs.py sharded_pair; the machinery
    of tests/test_gpu_sharded_emulation.py on random graphs of every manifold, joint messages on a third of them)"""
    fz = load("fuzz_graphs")
    info, res, why = fz.sharded_pair(seed)
    assert res is not None, why
    nv, differ, worst, _ = res
    assert not differ, (info, differ[:5], worst)


@pytest.mark.parametrize("seed", [1, 4, 5, 6, 10, 11])
def test_random_graphs_walked_by_concurrent_single_clique_calls(seed):
    """the per-clique entry points on random graphs (Euclid(1/2/3), circle; joint messages on seeds 1, 4, 10): one call per clique,
    and the cliques of a level as CONCURRENT single calls from eight host threads on one context -- merged by the library --
    against the whole-tree program: the same bytes (tests/fuzz_graphs.py seam_pair, FUZZ_SEAM_THREADS)"""
    fz = load("fuzz_graphs")
    fz.THREADS = 8
    info, res, why = fz.seam_pair(seed)
    assert res is not None, why
    assert info["kind"] != 4  # (SE(2) crosses the host boundary as (t, R): held as distributions in the builder's run, not here)
    nv, differ, worst, _ = res
    assert not differ, (info, differ[:5], worst)


@pytest.mark.parametrize("count", [64, 2, 17])
@pytest.mark.parametrize("man", [1, 2, 3, 4, 5])
def test_degenerate_beliefs_are_the_oracles_too(man, count):
    """identical points, clusters 1e6 apart, an offset of 1e8, a spread of 1e-12, three distinct values, 2 or 17 points in a slot
    of 64 -- through the fit, proposals from and onto the belief, a prior with nullhypo, products of two and three densities
    (tests/fuzz_degenerate.py; 390 such cases in profiles/r06_fuzz_ops.txt): finite, and the oracle's bits"""
    fd = load("fuzz_degenerate")
    for name in ("identical", "two far clusters", "huge offset", "tiny spread", "three values"):
        bad, finite = fd.run_case(man, 64, count, name, 7 * man + count)
        assert finite and not bad, (name, bad[:2])
