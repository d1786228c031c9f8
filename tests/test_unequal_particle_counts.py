"""CPU (oracle): beliefs with a particle count other than N -- tests/unequal_n_cases.py"""
import unequal_n_cases as uc


def test_count_round_trip(oracle_backend):
    uc.case_count_round_trip(oracle_backend)


def test_shorter_operand(oracle_backend):
    uc.case_shorter_operand_is_read_at_a_random_element(oracle_backend)


def test_shorter_target(oracle_backend):
    uc.case_shorter_target_is_filled_with_the_point_default(oracle_backend)


def test_message_with_fewer_points(oracle_backend):
    uc.case_message_with_fewer_points(oracle_backend)


def test_bandwidth_of_a_shorter_belief(oracle_backend):
    uc.case_bandwidth_of_a_shorter_belief(oracle_backend)


def test_resample(oracle_backend):
    uc.case_resample_tops_up_to_n(oracle_backend)


def test_old_points_of_a_partial_product(oracle_backend):
    uc.case_old_points_of_a_partial_product(oracle_backend)
