"""GPU: the reference-test acceptance bands of tests/band_cases.py through the C ABI (libnbp)."""
import pytest

import band_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", band_cases.CASES, ids=lambda c: c.__name__)
def test_band(case, hip_backend):
    case(hip_backend)
