"""-m gpu: Uniform / Rayleigh scalar measurements on the HIP library (tests/scalar_family_cases.py), each case also
compared with the oracle on identical streams."""
import numpy as np
import pytest

import scalar_family_cases as sc

pytestmark = pytest.mark.gpu


def test_uniform_prior(hip_backend, oracle_backend):
    np.testing.assert_allclose(sc.case_uniform_prior(hip_backend), sc.case_uniform_prior(oracle_backend), rtol=0, atol=0)


def test_rayleigh_prior(hip_backend, oracle_backend):
    np.testing.assert_allclose(sc.case_rayleigh_prior(hip_backend), sc.case_rayleigh_prior(oracle_backend), rtol=0, atol=0)


def test_mixture_with_a_uniform_component(hip_backend, oracle_backend):
    np.testing.assert_allclose(sc.case_mixture_with_a_uniform_component(hip_backend), sc.case_mixture_with_a_uniform_component(oracle_backend),
                               rtol=0, atol=0)


def test_rayleigh_relative(hip_backend, oracle_backend):
    np.testing.assert_allclose(sc.case_rayleigh_relative(hip_backend), sc.case_rayleigh_relative(oracle_backend), rtol=0, atol=0)


def test_alias_sampler_prior(hip_backend, oracle_backend):
    np.testing.assert_array_equal(sc.case_alias_sampler_prior(hip_backend), sc.case_alias_sampler_prior(oracle_backend))


def test_mixture_with_an_alias_sampler(hip_backend, oracle_backend):
    np.testing.assert_allclose(sc.case_mixture_with_an_alias_sampler(hip_backend), sc.case_mixture_with_an_alias_sampler(oracle_backend),
                               rtol=0, atol=0)


def test_alias_sampler_relative(hip_backend, oracle_backend):
    np.testing.assert_allclose(sc.case_alias_sampler_relative(hip_backend), sc.case_alias_sampler_relative(oracle_backend), rtol=0, atol=0)
