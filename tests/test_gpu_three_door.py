"""-m gpu: test/testMultiHypo3Door.jl at its own size through the C ABI (tests/three_door_cases.py)"""
import pytest

from three_door_cases import case_three_doors

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [40, 41])
def test_three_doors_hip(hip_backend, seed):
    print(case_three_doors(hip_backend, seed))
