"""A batch of proposals whose relative factors are all LinearRelative on Euclid(2) (or Euclid(3)) runs in a kernel
instance compiled for that case (nbp_proposal_kernel_lin2 / _lin3, launch_proposals in nbp_api.hip); a batch that also
holds a proposal on another manifold runs in the generic kernel.  The particles must not depend on which one ran: same
random streams, same algorithm; the instances fold the manifold into their spread statistics at compile time, so the
compiler contracts a few sums differently (observed: 1e-13 .. 2e-11 on the particles, no particle moved) -- compared to
1e-9 like every other pair of geometries."""
import numpy as np
import pytest

from parity_utils import abi, iif, rand_points, relative_factor_desc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("man,dim", [(abi.EUCLID2, 2), (abi.EUCLID3, 3)])
def test_uniform_batch_equals_generic_batch(man, dim):
    N = 150
    be = iif.HipBackend(N, 12, 0)
    rng = np.random.default_rng(5)
    for j in range(4):
        be.slot_write(j, man, rand_points(rng, man, N, 2.0 * j, 0.4))
    be.slot_write(4, abi.CIRCULAR, rand_points(rng, abi.CIRCULAR, N, 0.3, 0.2))
    mean, sig = [1.0, -0.5, 0.25][:dim], [0.1, 0.2, 0.15][:dim]
    descs = [
        relative_factor_desc(abi.F_LINREL, man, 2, 1, [0, 1], 6, 11, mean, sig),
        relative_factor_desc(abi.F_LINREL, man, 2, 0, [2, 3], 7, 12, mean, sig, nullhypo=0.2),
        relative_factor_desc(abi.F_LINREL, man, 3, 0, [1, 2, 3], 8, 13, mean, sig, multihypo=[0.0, 0.6, 0.4]),
        relative_factor_desc(abi.F_PRIOR, man, 1, 0, [0], 9, 14, mean, sig),
    ]
    be.run_proposals(descs)  # uniform: the compiled-for-the-case instance
    uniform = [be.slot_read(s, man) for s in (6, 7, 8, 9)]
    other = relative_factor_desc(abi.F_PRIOR, abi.CIRCULAR, 1, 0, [4], 10, 15, [0.3], [0.05])
    be.run_proposals(descs + [other])  # mixed manifolds: the generic kernel
    generic = [be.slot_read(s, man) for s in (6, 7, 8, 9)]
    for u, g in zip(uniform, generic):
        np.testing.assert_allclose(u[0], g[0], rtol=0, atol=0)  # points
        np.testing.assert_allclose(u[1], g[1], rtol=0)  # bandwidths
    be.close()
