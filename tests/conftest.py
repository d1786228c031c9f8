import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import iif_amd_loader  # noqa: E402

iif_amd_loader.load()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def iif():
    return iif_amd_loader.load()


@pytest.fixture(scope="session")
def oracle_backend():
    from oracle.oracle_backend import OracleBackend

    def make(N, n_slots, side_ints=0):
        return OracleBackend(N, n_slots, side_ints, threads=8)

    return make


@pytest.fixture(scope="session")
def hip_backend(iif):
    def make(N, n_slots, side_ints=0):
        return iif.HipBackend(N, n_slots, side_ints=side_ints)

    make.is_hip = True  # solveTree compiles the schedule with the native host for libnbp backends
    return make
