"""CPU: degenerate inputs through the oracle (tests/degenerate_inputs.py)."""
from degenerate_inputs import run_degenerate
from oracle.oracle_backend import OracleBackend


def test_oracle_degenerate_inputs():
    run_degenerate(OracleBackend)
