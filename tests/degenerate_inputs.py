"""Degenerate beliefs through every op: identical points (zero spread), far-apart clusters, huge offsets, a NaN
particle.  The ops must terminate, keep finite inputs finite and confine a NaN to the particles it touches
(the reference warns and carries on: NumericalCalculations.jl:128-131, :348-351)."""
import numpy as np

from parity_utils import abi, product_desc, relative_factor_desc


def run_degenerate(backend):
    N, man = 64, abi.EUCLID2
    r = np.random.default_rng(0)
    cases = {
        "identical": np.tile([[1.5, -2.0]], (N, 1)),
        "two far clusters": np.concatenate([r.normal(0, 1e-3, (N // 2, 2)), r.normal(1e6, 1e-3, (N - N // 2, 2))]),
        "huge offset": r.normal(0, 0.1, (N, 2)) + 1e12,
        "tiny spread": r.normal(0, 1e-12, (N, 2)),
    }
    for name, pts in cases.items():
        be = backend(N, 6, 0)
        be.slot_write(0, man, pts, np.ones(2))
        be.slot_write(1, man, pts + 1.0, np.ones(2))
        be.run_bandwidth([0, 1], [man, man])
        bw = be.slot_read(0, man)[1]
        assert np.isfinite(bw).all() and (bw > 0).all(), (name, bw)
        conv = relative_factor_desc(abi.F_LINREL, man, 2, 1, [0, 1], 2, 11, [1.0, 1.0], [0.1, 0.1])
        prior = relative_factor_desc(abi.F_PRIOR, man, 1, 0, [0], 3, 12, [0.0, 0.0], [1.0, 1.0])
        be.run_proposals([conv, prior])
        be.run_products([product_desc(man, [2, 1], 4, 13), product_desc(man, [2, 3, 0], 5, 14)])
        for s in (2, 3, 4, 5):
            p, b = be.slot_read(s, man)
            assert np.isfinite(p).all() and np.isfinite(b).all(), (name, s)
        be.close()
    # one NaN particle: the convolution leaves that particle alone (counted), everything else is solved
    be = backend(N, 4, 0)
    a = r.normal(0, 0.3, (N, 2))
    a[5] = np.nan
    be.slot_write(0, man, a, np.ones(2))
    be.slot_write(1, man, r.normal(1, 0.3, (N, 2)), np.ones(2))
    conv = relative_factor_desc(abi.F_LINREL, man, 2, 1, [0, 1], 2, 21, [1.0, 1.0], [0.1, 0.1])
    conv.skip_bandwidth = 1
    be.run_proposals([conv])
    out = be.slot_read(2, man)[0]
    bad = ~np.isfinite(out).all(axis=1)
    assert bad.sum() <= 1 and np.isfinite(np.delete(out, 5, axis=0)).all()
    be.close()
