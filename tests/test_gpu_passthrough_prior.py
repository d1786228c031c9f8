"""-m gpu: PartialPriorPassThrough on the HIP library (tests/passthrough_cases.py), each case also compared with the
oracle on identical streams; the tree solve through the native compile and through the per-clique entry."""
import numpy as np
import pytest

import passthrough_cases as pc
from parity_utils import abi, iif

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nullhypo", [0.0, 0.2])
def test_alone_keeps_the_density(hip_backend, oracle_backend, nullhypo):
    np.testing.assert_array_equal(pc.case_alone_keeps_the_density(hip_backend, nullhypo), pc.case_alone_keeps_the_density(oracle_backend, nullhypo))


def test_conv_is_the_density(hip_backend, oracle_backend):
    np.testing.assert_array_equal(pc.case_conv_is_the_density(hip_backend), pc.case_conv_is_the_density(oracle_backend))


def test_product_with_a_prior_has_n_points(hip_backend, oracle_backend):
    np.testing.assert_allclose(pc.case_product_with_a_prior_has_n_points(hip_backend), pc.case_product_with_a_prior_has_n_points(oracle_backend),
                               rtol=0, atol=0)


def test_product_with_a_relative_is_full(hip_backend, oracle_backend):
    np.testing.assert_allclose(pc.case_product_with_a_relative_is_full(hip_backend), pc.case_product_with_a_relative_is_full(oracle_backend),
                               rtol=0, atol=0)  # after a Nelder-Mead search per particle


def test_init_restricts_the_graph_to_n(hip_backend, oracle_backend):
    np.testing.assert_allclose(pc.case_init_restricts_the_graph_to_n(hip_backend), pc.case_init_restricts_the_graph_to_n(oracle_backend),
                               rtol=0, atol=0)


def test_init_with_more_points_than_n(hip_backend, oracle_backend):
    np.testing.assert_array_equal(pc.case_init_with_more_points_than_n(hip_backend), pc.case_init_with_more_points_than_n(oracle_backend))


@pytest.mark.parametrize("native", [True, False])
def test_solve(hip_backend, oracle_backend, native):
    """the whole solve, compiled by the native host or by the Python mirror: the same descriptors, so the same particles
    as the oracle up to the per-particle searches"""
    a, b = pc.case_solve(hip_backend, native=native), pc.case_solve(oracle_backend, native=False)
    for v in a:
        ca, cb = pc.se2_coords(a[v]), pc.se2_coords(b[v])
        assert np.abs(np.median(ca, axis=0) - np.median(cb, axis=0)).max() < 0.15, v


def test_clique_entry_takes_the_density(hip_backend):
    """nbp_clique_upsolve with factor_density: one clique {x0 | } holding the pass-through prior alone hands the density
    back as the belief of x0 (its 120 points); with the ManifoldPrior next to it, N points"""
    from iif_amd.native_host import Belief, clique_solve
    for second in (False, True):
        fg = pc.graph_w_priors(second_prior=second)
        sp, N = fg.solverParams, fg.solverParams.N
        be = hip_backend(N, 8)
        try:
            bel = {"x0": Belief(abi.SE2, fg.getVariable("x0").val, np.ones(3))}
            facs = [fg.getFactor(f) for f in fg.lsf()]
            st = clique_solve(be, sp, 1, ["x0"], 1, 0, [abi.SE2], facs, bel, 77, lists={"directPriorMsg": ["x0"]})
            assert st == 3
            pts = bel["x0"].pts
            if second:
                assert pts.shape == (N, 6) and np.abs(pc.se2_coords(pts)[:, 2]).max() < 0.1
            else:
                assert pts.shape == (pc.NDENS, 6)
                np.testing.assert_allclose(pc.se2_coords(pts)[:, :2], pc.density()[0], atol=0)
                np.testing.assert_allclose(bel["x0"].bw, [0.35, 0.35, 0.0])
        finally:
            be.close()


def test_clique_entry_needs_the_density(hip_backend):
    import ctypes as C
    from iif_amd import native_host as nh
    fg = pc.graph_w_priors()
    lib = nh._lib()
    q = nh.CliqueDescC()
    q.clique_id, q.nvars, q.nfrontals, q.nseparators = 1, 1, 1, 0
    man = (C.c_int32 * 1)(abi.SE2)
    q.manifold = man
    specs = (nh.FactorSpec * 1)(nh.factor_spec(fg.getFactor("x0f1"), {"x0": 0}))
    q.nfactors, q.factors = 1, specs
    assert lib.nbp_clique_slots(C.byref(q)) == 1 + 0 + 1 + 1   # variable | messages | density | scratch
    be = hip_backend(fg.solverParams.N, 8)
    try:
        b = nh.Belief(abi.SE2, fg.getVariable("x0").val, np.ones(3))
        bel = (nh.TreeBeliefC * 1)(b.c(capacity=be.N))
        p = nh.solver_params_c(fg.solverParams)
        rc = lib.nbp_clique_upsolve(be._ctx, C.byref(p), C.byref(q), C.c_uint64(1), bel, None)
        assert rc < 0 and b"density" in lib.nbp_last_error()
    finally:
        be.close()
