"""CPU: partial relative factors on SE(2) on the oracle (tests/partial_se2_cases.py)."""
import pytest

import partial_se2_cases as cases


@pytest.mark.parametrize("case", cases.CASES, ids=lambda c: c.__name__)
def test_partial_se2_known_answers_oracle(oracle_backend, case):
    case(oracle_backend)


def test_partial_se2_in_a_graph_oracle(oracle_backend):
    cases.case_in_a_graph(oracle_backend)
