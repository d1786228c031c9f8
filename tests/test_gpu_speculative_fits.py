"""-m gpu: the speculative golden-section search (3 or 7 workgroups per fit, lcv_bandwidth_1d_spec) selects the
bit-identical bandwidth of the sequential search: every likelihood value is computed by the same code at the same
point, only the order in which the points are visited differs."""
import os

import numpy as np
import pytest

from parity_utils import abi, iif, rand_points

pytestmark = pytest.mark.gpu


def fit(N, manifold, nfits, env):
    old = {k: os.environ.get(k) for k in ("NBP_NO_SPECULATIVE_FITS", "NBP_SPEC_DEPTH3", "NBP_FIT_F64")}
    for k in old:
        os.environ.pop(k, None)
    os.environ.update(env)
    try:
        be = iif.HipBackend(N, nfits, 0)  # the switches are read when the context is created
        try:
            rng = np.random.default_rng(7)
            for s in range(nfits):
                be.slot_write(s, manifold, rand_points(rng, manifold, N, float(s), 0.2 + 0.3 * s))
            be.run_bandwidth(list(range(nfits)), [manifold] * nfits)
            out = np.array([be.slot_read(s, manifold)[1] for s in range(nfits)])
            evals = be.diag()["lcv_evals"]
        finally:
            be.close()
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    return out, evals


@pytest.mark.parametrize("manifold", [abi.EUCLID1, abi.EUCLID2, abi.EUCLID3, abi.CIRCULAR, abi.SE2])
@pytest.mark.parametrize("N,nfits", [(100, 1), (200, 3), (256, 8), (37, 2)])
def test_speculative_search_is_bit_identical(manifold, N, nfits):
    # (the sequential search in double precision throughout: with its bracketing evaluations in single precision it counts
    #  fewer double-precision evaluations -- tests/test_gpu_fit_bracketing.py -- for the same bandwidths)
    seq, ev0 = fit(N, manifold, nfits, {"NBP_NO_SPECULATIVE_FITS": "1", "NBP_FIT_F64": "1"})
    brk, _ = fit(N, manifold, nfits, {"NBP_NO_SPECULATIVE_FITS": "1"})
    np.testing.assert_array_equal(brk, seq)
    k3, ev3 = fit(N, manifold, nfits, {"NBP_SPEC_DEPTH3": "0"})
    k7, ev7 = fit(N, manifold, nfits, {})
    assert np.all(seq > 0)
    np.testing.assert_array_equal(k3, seq)
    np.testing.assert_array_equal(k7, seq)
    assert ev3 == ev0 and ev7 == ev0  # the same iterations, advanced two or three per rendezvous


def test_many_fits_are_reproducible_and_equal_the_sequential_search():
    """a regression guard: 240 fits of 64-point beliefs in launches of 8 (three workgroups per fit, 72 workgroups per
    launch), twice -- an earlier build selected a different bandwidth in a few fits per thousand, differently each run"""
    N, man = 64, abi.EUCLID2
    rng = np.random.default_rng(5)
    data = [rng.normal(0, rng.uniform(0.01, 5), (N, 2)) for _ in range(240)]

    def run(env, group):
        old = {k: os.environ.pop(k, None) for k in ("NBP_NO_SPECULATIVE_FITS", "NBP_SPEC_DEPTH3")}
        os.environ.update(env)
        try:
            be = iif.HipBackend(N, group, 0)
            out = []
            try:
                for g0 in range(0, len(data), group):
                    chunk = data[g0:g0 + group]
                    for s, pts in enumerate(chunk):
                        be.slot_write(s, man, pts)
                    be.run_bandwidth(list(range(len(chunk))), [man] * len(chunk))
                    out += [be.slot_read(s, man)[1].copy() for s in range(len(chunk))]
            finally:
                be.close()
        finally:
            for k in ("NBP_NO_SPECULATIVE_FITS", "NBP_SPEC_DEPTH3"):
                os.environ.pop(k, None)
                if old[k] is not None:
                    os.environ[k] = old[k]
        return np.array(out)

    seq = run({"NBP_NO_SPECULATIVE_FITS": "1"}, 8)
    np.testing.assert_array_equal(run({"NBP_NO_SPECULATIVE_FITS": "1"}, 240), seq)  # one chip-filling launch: other helper geometry
    for env, group in (({"NBP_SPEC_DEPTH3": "0"}, 8), ({"NBP_SPEC_DEPTH3": "0"}, 8), ({}, 3), ({}, 3)):
        np.testing.assert_array_equal(run(env, group), seq)
