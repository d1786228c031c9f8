"""CPU (oracle): Uniform / Rayleigh scalar measurements -- tests/scalar_family_cases.py"""
import numpy as np
import pytest

import scalar_family_cases as sc
from parity_utils import iif


def test_uniform_prior(oracle_backend):
    sc.case_uniform_prior(oracle_backend)


def test_rayleigh_prior(oracle_backend):
    sc.case_rayleigh_prior(oracle_backend)


def test_mixture_with_a_uniform_component(oracle_backend):
    sc.case_mixture_with_a_uniform_component(oracle_backend)


def test_rayleigh_relative(oracle_backend):
    sc.case_rayleigh_relative(oracle_backend)


def test_alias_sampler_prior(oracle_backend):
    sc.case_alias_sampler_prior(oracle_backend)


def test_mixture_with_an_alias_sampler(oracle_backend):
    sc.case_mixture_with_an_alias_sampler(oracle_backend)


def test_alias_sampler_relative(oracle_backend):
    sc.case_alias_sampler_relative(oracle_backend)


def test_families_are_scalar_only():
    with pytest.raises(ValueError):
        iif.Uniform(1.0, 1.0)
    with pytest.raises(ValueError):
        iif.Rayleigh(0.0)


def test_a_table_longer_than_the_particle_count_is_refused():
    """a sampler table lives in a belief slot of N rows; one with more entries used to be cut silently (the tail's
    mass then landed on entry N - 1).  The hosts refuse it before anything is written."""
    import numpy as np
    import pytest
    from parity_utils import iif
    tb = iif.AliasingScalarSampler(np.linspace(0.0, 1.0, 150), np.ones(150))
    pts, _ = tb.table_belief(200)                     # fits
    assert pts.shape == (150, 2) and pts[-1, 1] == 1.0
    with pytest.raises(ValueError, match="does not fit"):
        tb.table_belief(100)
    fg = iif.initfg(iif.SolverParams(N=100))
    iif.addVariable(fg, "x0", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Prior(tb))
    with pytest.raises(ValueError, match="does not fit"):
        iif.initAll(fg, backend=lambda N, n, side_ints=0: __import__("oracle.oracle_backend", fromlist=["OracleBackend"]).OracleBackend(N, n, side_ints), seed=1)
