"""-m gpu: the stream-independent known answers of tests/analytic_cases.py on the HIP library (the same cases run on
the oracle in tests/test_analytic_known_answers.py).  Nothing here compares the kernels with the oracle."""
import pytest

import analytic_cases as ac

pytestmark = pytest.mark.gpu


def test_lcv_maximises_loo_likelihood(hip_backend):
    ac.case_lcv_maximises_loo_likelihood(hip_backend)


def test_product_of_two_matches_exact_mixture(hip_backend):
    ac.case_product_of_two_matches_exact_mixture(hip_backend)


def test_partial_product_matches_exact_mixture(hip_backend):
    ac.case_partial_product_matches_exact_mixture(hip_backend)


def test_bimodal_mode_masses(hip_backend):
    ac.case_bimodal_mode_masses(hip_backend)


def test_product_of_many_densities(hip_backend):
    ac.case_product_of_many_densities(hip_backend)


def test_solver_finds_the_residual_root(hip_backend):
    ac.case_solver_finds_the_residual_root(hip_backend)


def test_euclid_distance_ring(hip_backend):
    ac.case_euclid_distance_ring(hip_backend)


def test_partial_relative_over_two_coordinates(hip_backend):
    ac.case_partial_relative_over_two_coordinates(hip_backend)


def test_product_labels_match_enumeration(hip_backend):
    print(ac.case_product_labels_match_enumeration(hip_backend))
