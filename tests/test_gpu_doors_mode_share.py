"""-m gpu: tests/doors_cases.py on the HIP library"""
import pytest

import doors_cases as dc

pytestmark = pytest.mark.gpu


def test_sighting_is_resolved_by_odometry(hip_backend):
    dc.case_sighting_is_resolved_by_odometry(hip_backend)


def test_short_chains_keep_the_true_mode(hip_backend):
    dc.case_short_chains_keep_the_true_mode(hip_backend)


def test_true_mode_survives_without_null_surplus_and_with_a_mixed_product(hip_backend):
    print(dc.case_true_mode_survives_without_null_surplus_and_with_a_mixed_product(hip_backend))


def test_config3_mechanisms_at_2000_poses(hip_backend):
    """review r04 item 3, at BASELINE's size: (i) nullSurplusAdd = 0, (ii) Niter = 6, (iii) what the reference's own parameters reach"""
    from parity_utils import record_parity
    res = dc.case_full_size_mechanisms(hip_backend)
    for k, v in res.items():
        record_parity(f"config 3 at 2000 poses, {k}: (median, min, share of poses above 0.8, first 200 poses at >= 0.6) after 1 / 2 / 3 solves: {v}")
    print(res)
