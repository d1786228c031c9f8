"""CPU: the stream-independent known answers of tests/analytic_cases.py on the oracle (the same cases run on the
GPU in tests/test_gpu_analytic_known_answers.py).  These -- not GPU-vs-oracle agreement -- are what ties the
restated third-party algorithms (bandwidth selection, product sampler, optimisers) to their definitions."""
import analytic_cases as ac


def test_lcv_maximises_loo_likelihood(oracle_backend):
    ac.case_lcv_maximises_loo_likelihood(oracle_backend)


def test_product_of_two_matches_exact_mixture(oracle_backend):
    ac.case_product_of_two_matches_exact_mixture(oracle_backend)


def test_partial_product_matches_exact_mixture(oracle_backend):
    ac.case_partial_product_matches_exact_mixture(oracle_backend)


def test_bimodal_mode_masses(oracle_backend):
    ac.case_bimodal_mode_masses(oracle_backend)


def test_product_of_many_densities(oracle_backend):
    ac.case_product_of_many_densities(oracle_backend)


def test_solver_finds_the_residual_root(oracle_backend):
    ac.case_solver_finds_the_residual_root(oracle_backend)


def test_euclid_distance_ring(oracle_backend):
    ac.case_euclid_distance_ring(oracle_backend)


def test_partial_relative_over_two_coordinates(oracle_backend):
    ac.case_partial_relative_over_two_coordinates(oracle_backend)


def test_product_labels_match_enumeration(oracle_backend):
    print(ac.case_product_labels_match_enumeration(oracle_backend))
