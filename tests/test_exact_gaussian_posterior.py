"""CPU: the oracle's whole solve against the exact posterior of a linear-Gaussian chain (an analytic known
answer, independent of the reference's random streams)."""
import pytest

from exact_gaussian import check_against_exact
from oracle.oracle_backend import OracleBackend


@pytest.mark.parametrize("seed", [3, 17])
def test_oracle_solve_matches_exact_gaussian_posterior(seed):
    check_against_exact(lambda N, s, side_ints=0: OracleBackend(N, s, side_ints, threads=8), seed)
