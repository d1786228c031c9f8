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
