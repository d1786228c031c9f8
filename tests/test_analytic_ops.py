"""CPU: op-level analytic known answers for the restated third-party algorithms (a12 bandwidth, a13 product)
and for the convolution -- Gaussian inputs, closed-form outputs, Monte-Carlo tolerances over many seeds."""
import numpy as np

from parity_utils import abi, product_desc, relative_factor_desc
from oracle.oracle_backend import OracleBackend


def test_lcv_bandwidth_tracks_the_gaussian_reference_rule():
    # leave-one-out likelihood CV on Gaussian data lands near the normal-reference bandwidth 1.06 sigma N^(-1/5)
    # (the KDE literature's benchmark; LCV is noisier and a little larger on average)
    rng = np.random.default_rng(0)
    for N in (100, 200):
        bws = []
        for _ in range(24):
            be = OracleBackend(N, 1, 0)
            be.slot_write(0, abi.EUCLID1, rng.normal(0.0, 2.0, (N, 1)), np.ones(1))
            be.run_bandwidth([0], [abi.EUCLID1])
            bws.append(be.slot_read(0, abi.EUCLID1)[1][0])
            be.close()
        ref = 1.06 * 2.0 * N ** -0.2
        assert 0.8 * ref < np.mean(bws) < 1.35 * ref, (N, np.mean(bws), ref)
        assert min(bws) > 0.15 * ref and max(bws) < 3.0 * ref  # LCV is a high-variance selector: it undersmooths now and then


def test_convolution_moments():
    # x_b = x_a + z: mean mu_a + mu_z, variance var_a + var_z; the reverse direction subtracts
    N, man = 200, abi.EUCLID2
    rng = np.random.default_rng(1)
    mf, vf, mr, vr = [], [], [], []
    for seed in range(20):
        be = OracleBackend(N, 4, 0)
        a = rng.normal([1.0, -2.0], [0.3, 0.5], (N, 2))
        be.slot_write(0, man, a, np.ones(2))
        be.slot_write(1, man, a, np.ones(2))
        fwd = relative_factor_desc(abi.F_LINREL, man, 2, 1, [0, 1], 2, 100 + seed, [2.0, 0.5], [0.4, 0.2])
        rev = relative_factor_desc(abi.F_LINREL, man, 2, 0, [0, 1], 3, 200 + seed, [2.0, 0.5], [0.4, 0.2])
        be.run_proposals([fwd, rev])
        f, r = be.slot_read(2, man)[0], be.slot_read(3, man)[0]
        mf.append(f.mean(axis=0) - a.mean(axis=0)); vf.append(f.var(axis=0) - a.var(axis=0))
        mr.append(r.mean(axis=0) - a.mean(axis=0)); vr.append(r.var(axis=0) - a.var(axis=0))
        be.close()
    assert np.allclose(np.mean(mf, axis=0), [2.0, 0.5], atol=0.03) and np.allclose(np.mean(mr, axis=0), [-2.0, -0.5], atol=0.03)
    assert np.allclose(np.mean(vf, axis=0), [0.16, 0.04], rtol=0.2) and np.allclose(np.mean(vr, axis=0), [0.16, 0.04], rtol=0.2)


def test_product_moments_of_gaussian_kdes():
    # the product of F KDEs of N(mu_f, s^2) samples: each KDE is a Gaussian of variance s^2 + h^2 to second order,
    # so the product has the precision-weighted mean and variance (s^2 + h^2) / F
    N, man, s = 200, abi.EUCLID1, 1.0
    rng = np.random.default_rng(2)
    for F, mus in ((2, [-0.5, 0.5]), (3, [-1.0, 0.0, 1.6])):
        means, vars_, h2 = [], [], []
        for seed in range(30):
            be = OracleBackend(N, F + 1, 0)
            for j, m in enumerate(mus):
                x = rng.normal(0.0, s, (N, 1))
                be.slot_write(j, man, (x - x.mean()) / x.std() * s + m, np.ones(1))
            be.run_bandwidth(list(range(F)), [man] * F)
            h2.append(np.mean([be.slot_read(j, man)[1][0] ** 2 for j in range(F)]))
            be.run_products([product_desc(man, list(range(F)), F, 300 + seed)])
            p = be.slot_read(F, man)[0][:, 0]
            means.append(p.mean()); vars_.append(p.var())
            be.close()
        assert abs(np.mean(means) - np.mean(mus)) < 0.05, (F, np.mean(means))
        expect = (s * s + np.mean(h2)) / F
        assert 0.85 * expect < np.mean(vars_) < 1.15 * expect, (F, np.mean(vars_), expect)
