"""-m gpu: bandwidth fits of rows of 4k + 1 waves (N = 257 .. 320) in THROUGHPUT mode -- the launches where the full waves
take the last wave's points between them and the five-waves-per-SIMD instances of the fit kernels run
(nbp_device.h neg_loo_ll "fold", nbp_bandwidth_kernel_w5; profiles/r04_lcv_five_wave_rows.txt).  The op-level parity tests
fit a handful of beliefs at a time, which is latency mode: another geometry of the same search.  Here the same beliefs are
fitted 700 at a time (one helper row per fit) and a few at a time, and against the oracle."""
import numpy as np
import pytest

from parity_utils import abi, both, iif, rand_points

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N", [257, 288, 300, 320])
@pytest.mark.parametrize("manifold", [abi.EUCLID3, abi.CIRCULAR])
def test_throughput_fits_of_five_wave_rows(oracle_backend, hip_backend, N, manifold):
    rng = np.random.default_rng(N * 7 + manifold)
    K, B = 12, 700  # distinct beliefs; slots fitted in one launch (2 B blocks and more: one helper row per fit)
    pts = [rand_points(rng, manifold, N, center=0.3 * k, spread=0.2 + 0.15 * k) for k in range(K)]
    be = hip_backend(N, B + 2)
    try:
        for k in range(K):
            be.slot_write(k, manifold, pts[k])
        be.run_copies([abi.CopyDesc(s % K, s) for s in range(K, B)])
        be.run_bandwidth(list(range(B)), [manifold] * B)          # throughput mode
        thr = [be.slot_read(s, manifold)[1] for s in range(B)]
        for k in range(K):
            be.slot_write(k, manifold, pts[k])
        lat = []
        for k in range(K):                                        # latency mode: one belief per launch
            be.run_bandwidth([k], [manifold])
            lat.append(be.slot_read(k, manifold)[1])
    finally:
        be.close()
    for s in range(B):  # every copy of a belief gets the bandwidth of the original, bit for bit
        np.testing.assert_array_equal(thr[s], thr[s % K])
    for k in range(K):  # the two geometries sum in different orders: equal to rounding
        np.testing.assert_allclose(thr[k], lat[k], rtol=0)
    ob = oracle_backend(N, K + 1)
    try:
        for k in range(K):
            ob.slot_write(k, manifold, pts[k])
        ob.run_bandwidth(list(range(K)), [manifold] * K)
        for k in range(K):
            np.testing.assert_allclose(thr[k], ob.slot_read(k, manifold)[1], rtol=0)
    finally:
        ob.close()
