"""CPU (oracle): PartialPriorPassThrough -- tests/passthrough_cases.py; plus the host side: the native compile of a
graph with a pass-through prior is byte-identical to the Python mirror's."""
import numpy as np
import pytest

import passthrough_cases as pc
from parity_utils import abi, iif


@pytest.mark.parametrize("nullhypo", [0.0, 0.2])
def test_alone_keeps_the_density(oracle_backend, nullhypo):
    pc.case_alone_keeps_the_density(oracle_backend, nullhypo)


def test_conv_is_the_density(oracle_backend):
    pc.case_conv_is_the_density(oracle_backend)


def test_product_with_a_prior_has_n_points(oracle_backend):
    pc.case_product_with_a_prior_has_n_points(oracle_backend)


def test_product_with_a_relative_is_full(oracle_backend):
    pc.case_product_with_a_relative_is_full(oracle_backend)


def test_init_restricts_the_graph_to_n(oracle_backend):
    pc.case_init_restricts_the_graph_to_n(oracle_backend)


def test_init_with_more_points_than_n(oracle_backend):
    pc.case_init_with_more_points_than_n(oracle_backend)


def test_solve(oracle_backend):
    pc.case_solve(oracle_backend, native=False)


def test_bad_descriptions_are_refused():
    with pytest.raises(ValueError):
        iif.PartialPriorPassThrough(iif.SpecialEuclidean2, np.zeros((5, 3)), [0.1, 0.1], (1, 2))
    fg = pc.graph_w_priors()
    with pytest.raises(ValueError):  # a prior is unary
        iif.addVariable(fg, "x1", iif.SpecialEuclidean2)
        iif.addFactor(fg, ["x0", "x1"], iif.PartialPriorPassThrough(iif.SpecialEuclidean2, np.zeros((5, 2)), [0.1, 0.1], (1, 2)))
