"""CPU (oracle): tests/doors_cases.py"""
import doors_cases as dc


def test_sighting_is_resolved_by_odometry(oracle_backend):
    dc.case_sighting_is_resolved_by_odometry(oracle_backend)


def test_short_chains_keep_the_true_mode(oracle_backend):
    dc.case_short_chains_keep_the_true_mode(oracle_backend)


def test_true_mode_survives_without_null_surplus_and_with_a_mixed_product(oracle_backend):
    print(dc.case_true_mode_survives_without_null_surplus_and_with_a_mixed_product(oracle_backend))
