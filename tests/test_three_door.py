"""CPU: test/testMultiHypo3Door.jl at its own size on the oracle (tests/three_door_cases.py)"""
import pytest

from three_door_cases import case_three_doors


@pytest.mark.parametrize("seed", [40, 41])
def test_three_doors_oracle(oracle_backend, seed):
    print(case_three_doors(oracle_backend, seed))
