"""-m gpu: a product LAUNCH that mixes density counts, where the node statistics do not fit the LDS and live in the launch's
global scratch ("big": 3-D manifolds from ~260 particles on with four densities, or many densities at fewer particles).
The scratch of a workgroup is addressed by the LAUNCH's largest density count -- addressed by the product's own, as it was
through round 5, the workgroups wrote over each other's statistics and products came out wrong or non-finite (found by the
whole-solve differential check of round 5 on an SE(2) lattice at N = 300: tools/exp/whole_solve_sha_se2.sh).  Every product of
the launch that is compared must be the oracle's."""
import numpy as np
import pytest

from parity_utils import abi, iif, product_desc, rand_points

pytestmark = pytest.mark.gpu


def run(make, N, man, Fs, keep=None, nsrc=16):
    nprod = len(Fs)
    be = make(N, nsrc + nprod)
    try:
        rng = np.random.default_rng(1)
        for j in range(nsrc):
            be.slot_write(j, man, rand_points(rng, man, N, 0.2 * j, 0.3))
        be.run_bandwidth(list(range(nsrc)), [man] * nsrc)
        descs = [product_desc(man, [(3 * i + j) % nsrc for j in range(Fs[i])], nsrc + i, 5 + i) for i in range(nprod)]
        if keep is not None:  # the oracle: only the products that are compared (each is independent of the others)
            descs = [descs[i] for i in keep]
        be.run_products(descs)
        return {i: be.slot_read(nsrc + i, man)[0] for i in (keep if keep is not None else range(nprod))}
    finally:
        be.close()


@pytest.mark.parametrize("man,N,nprod,counts", [
    (abi.SE2, 300, 332, (2, 3, 4)),      # the launch that failed: a tree level of an SE(2) lattice, throughput geometry asked for
    (abi.SE2, 320, 120, (2, 3, 4)),
    (abi.EUCLID3, 300, 332, (2, 3, 4)),
    (abi.EUCLID3, 320, 40, (2, 3, 4)),
    (abi.SE2, 200, 60, (2, 5, 9)),       # many densities at BASELINE's particle count
    (abi.EUCLID2, 300, 250, (2, 4, 12)),
    (abi.CIRCULAR, 200, 30, (2, 7, 20)),
])
def test_products_of_a_launch_that_mixes_density_counts_are_the_oracles(oracle_backend, hip_backend, man, N, nprod, counts):
    Fs = [counts[2] if i % 9 == 8 else (counts[1] if i % 17 == 3 else counts[0]) for i in range(nprod)]
    keep = [i for i in range(nprod) if Fs[i] != counts[0]][:6] + list(range(4))
    d = run(hip_backend, N, man, Fs)
    o = run(oracle_backend, N, man, Fs, keep=keep)
    assert all(np.isfinite(v).all() for v in d.values())
    for i in keep:
        np.testing.assert_allclose(d[i], o[i], rtol=0, atol=0, err_msg=f"product {i} ({Fs[i]} densities)")
