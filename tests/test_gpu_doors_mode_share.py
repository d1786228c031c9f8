"""-m gpu: tests/doors_cases.py on the HIP library"""
import pytest

import doors_cases as dc

pytestmark = pytest.mark.gpu


def test_sighting_is_resolved_by_odometry(hip_backend):
    dc.case_sighting_is_resolved_by_odometry(hip_backend)


def test_short_chains_keep_the_true_mode(hip_backend):
    dc.case_short_chains_keep_the_true_mode(hip_backend)
