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


# Order of the `-m gpu` run (the driver runs it with -x): the parity evidence first -- op-level golden fixtures, op parity,
# clique / tree / whole-solve parity, the BASELINE configurations, properties -- and everything that starts other
# processes (torchrun legs of bench.py, the plain-C examples) LAST, so that an infrastructure failure can never again keep
# the parity tests from running.  Files not listed keep their alphabetical place in the middle group.
_GPU_ORDER = [
    "test_golden", "test_gpu_parity_ops", "test_gpu_analytic_known_answers", "test_gpu_tree_parity", "test_gpu_clique_entry",
    "test_gpu_kl_parity", "test_gpu_stagewise_parity", "test_gpu_configs", "test_gpu_fullsize_configs", "test_gpu_unequal_particle_counts",
    "test_gpu_reference_bands", "test_gpu_random_graphs", "test_gpu_properties",
]
_GPU_LAST = ["test_gpu_concurrent_contexts", "test_gpu_sharded_emulation", "test_gpu_native_host", "test_gpu_bench"]


def pytest_collection_modifyitems(session, config, items):
    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        if mod in _GPU_ORDER:
            return (0, _GPU_ORDER.index(mod))
        if mod in _GPU_LAST:
            return (2, _GPU_LAST.index(mod))
        return (1, 0)
    items.sort(key=key)  # stable: the order inside a file, and of unlisted files among themselves, stays


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
