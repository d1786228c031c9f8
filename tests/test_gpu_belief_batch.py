"""-m gpu: nbp_belief_write_batch / nbp_belief_read_batch move what nbp_belief_write / nbp_belief_read move -- every
manifold, beliefs with fewer than N points, gaps in the slot list (several copies), a consumer launched right behind the
write with no synchronisation in between -- and error behaviour (range, null, short belief without bandwidth)."""
import numpy as np
import pytest

from parity_utils import abi, iif, rand_points

pytestmark = pytest.mark.gpu


def test_batch_equals_one_by_one(hip_backend):
    N = 96
    be = hip_backend(N, 24)
    rng = np.random.default_rng(4)
    manis = [abi.EUCLID1, abi.EUCLID2, abi.EUCLID3, abi.CIRCULAR, abi.SE2, abi.EUCLID2, abi.SE2]
    slots = [2, 3, 4, 7, 8, 9, 15]          # runs 2-4, 7-9, 15: three copies
    counts = [N, N, 40, N, N, 17, N]        # two beliefs hold fewer than N points (densities: bandwidth required)
    bel = []
    for m, n in zip(manis, counts):
        D = abi.MANIFOLD_DIM[m]
        bel.append((rand_points(rng, m, n, 0.5, 0.7), rng.uniform(0.1, 0.5, D), rng.uniform(0.0, 3.0, D)))
    be.beliefs_write(slots, manis, bel)
    be.run_copies([abi.CopyDesc(s, s + 1) for s in (4, 9, 15)])  # consumers behind the queued copies, same stream
    one = [be.belief_read(s, m) for s, m in zip(slots, manis)]
    many = be.beliefs_read(slots, manis)
    for (p0, b0, i0), (p1, b1, i1), (pw, bw, iw), n in zip(one, many, bel, counts):
        assert p0.shape[0] == n and p1.shape[0] == n
        np.testing.assert_array_equal(p0, p1)
        np.testing.assert_array_equal(b0, b1)
        np.testing.assert_array_equal(i0, i1)
        np.testing.assert_allclose(p1, pw[:n], atol=1e-15)   # SE(2) goes through atan2 / cos / sin at the host boundary (numpy's on the way in): not bit-exact
        np.testing.assert_array_equal(b1, bw)
        np.testing.assert_array_equal(i1, iw)
    for s, m in ((5, abi.EUCLID3), (10, abi.EUCLID2), (16, abi.SE2)):  # the copies saw the batch
        src = be.belief_read(s - 1, m)
        dst = be.belief_read(s, m)
        np.testing.assert_array_equal(src[0], dst[0])
    # written one by one into other slots: the same bytes
    for s, m, (p, b, i) in zip(slots, manis, bel):
        be.belief_write(s, m, p, b, i)
    again = be.beliefs_read(slots, manis)
    for (p0, b0, i0), (p1, b1, i1) in zip(many, again):
        np.testing.assert_array_equal(p0, p1)
        np.testing.assert_array_equal(b0, b1)
    be.close()


def test_batch_errors(hip_backend):
    N = 32
    be = hip_backend(N, 4)
    p = np.zeros((N, 2))
    with pytest.raises(iif.NbpError):
        be.beliefs_write([0, 4], [abi.EUCLID2, abi.EUCLID2], [(p, None, None), (p, None, None)])  # slot out of range
    with pytest.raises(iif.NbpError):
        be.beliefs_write([0], [abi.EUCLID2], [(p[:10], None, None)])  # fewer than N points and no bandwidth
    with pytest.raises(iif.NbpError):
        be.beliefs_read([7], [abi.EUCLID2])
    be.beliefs_write([], [], [])
    assert be.beliefs_read([], []) == []
    be.close()
