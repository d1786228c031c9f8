"""-m gpu: beliefs with a particle count other than N on the HIP library (tests/unequal_n_cases.py), each case also
compared with the oracle on identical streams (1e-9, 1e-8 after a Nelder-Mead search)."""
import numpy as np
import pytest

import unequal_n_cases as uc
from parity_utils import abi

pytestmark = pytest.mark.gpu


def test_count_round_trip(hip_backend):
    uc.case_count_round_trip(hip_backend)


def test_shorter_operand(hip_backend, oracle_backend):
    np.testing.assert_allclose(uc.case_shorter_operand_is_read_at_a_random_element(hip_backend),
                               uc.case_shorter_operand_is_read_at_a_random_element(oracle_backend), rtol=0, atol=0)


def test_shorter_target(hip_backend, oracle_backend):
    np.testing.assert_allclose(uc.case_shorter_target_is_filled_with_the_point_default(hip_backend),
                               uc.case_shorter_target_is_filled_with_the_point_default(oracle_backend), rtol=0, atol=0)


def test_message_with_fewer_points(hip_backend, oracle_backend):
    (a, ba), (b, bb) = uc.case_message_with_fewer_points(hip_backend), uc.case_message_with_fewer_points(oracle_backend)
    np.testing.assert_allclose(a, b, rtol=0, atol=0)
    np.testing.assert_allclose(ba, bb, rtol=0)


def test_bandwidth_of_a_shorter_belief(hip_backend, oracle_backend):
    np.testing.assert_allclose(uc.case_bandwidth_of_a_shorter_belief(hip_backend), uc.case_bandwidth_of_a_shorter_belief(oracle_backend), rtol=0)


def test_resample(hip_backend):
    uc.case_resample_tops_up_to_n(hip_backend)


def test_old_points_of_a_partial_product(hip_backend, oracle_backend):
    np.testing.assert_allclose(uc.case_old_points_of_a_partial_product(hip_backend), uc.case_old_points_of_a_partial_product(oracle_backend),
                               rtol=0, atol=0)


def test_clique_call_with_a_short_message(hip_backend):
    """the clique seam with a child message of 40 points and a separator belief of 64 in a context of N = 128"""
    from iif_amd.native_host import Belief, clique_solve
    from parity_utils import iif
    rng = np.random.default_rng(8)
    fg = iif.generateChainEuclid(3, vardims=2, priorEvery=100, N=128)
    f01 = [f for f in fg.lsf() if fg.getFactor(f).variables == ["x0", "x1"]][0]
    bel = {"x1": Belief(abi.EUCLID2, rng.normal(1.0, 0.3, (128, 2)), np.full(2, 0.1)),
           "x0": Belief(abi.EUCLID2, rng.normal(0.0, 0.3, (64, 2)), np.full(2, 0.1))}
    msg = Belief(abi.EUCLID2, rng.normal(0.0, 0.1, (40, 2)), np.full(2, 0.05), np.ones(2))
    be = hip_backend(128, 16)
    try:
        st = clique_solve(be, fg.solverParams, 1, ["x1", "x0"], 1, 1, [abi.EUCLID2] * 2, [fg.getFactor(f01)], bel, 5,
                          lists={"itervar": ["x1", "x0"]}, msgs=[("x0", msg)])
    finally:
        be.close()
    assert st == 3
    assert bel["x0"].pts.shape == (128, 2) and bel["x1"].pts.shape == (128, 2)
    assert np.abs(bel["x0"].pts.mean(axis=0)).max() < 0.3 and np.abs(bel["x1"].pts.mean(axis=0) - 1.0).max() < 0.4
    assert np.all(bel["x0"].ipc == 2.0) and np.all(bel["x1"].ipc == 1.0)
