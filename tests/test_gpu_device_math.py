"""-m gpu: the device's own elementary functions against numpy -- sincos_fast (Cody-Waite + fdlibm kernels) through an
observable: a CircularCircular / SE(2) reverse solve uses it in every residual evaluation, the product uses it for the
node means of circular coordinates; here directly through the SE(2) relative residual root (closed form)."""
import numpy as np
import pytest

import analytic_cases as ac
from parity_utils import abi, relative_factor_desc

pytestmark = pytest.mark.gpu


def test_shared_math_header_gives_the_same_bits_on_the_device_and_on_the_host(hip_backend):
    """include/nbp_math.h compiled by hipcc for gfx950 (nbp_math_eval runs it in a kernel) and by gcc for the host (the CPU
    checker's copy): log, sin / cos, atan2, the wrap and the Box-Muller pair come out EQUAL TO THE LAST BIT on ~10^6 arguments
    each -- the premise of every bit-for-bit comparison of the suite"""
    from oracle import oracle_backend as ob
    from test_nbp_math import math_arguments
    u, ang, y, x = math_arguments(seed=5)
    rng = np.random.default_rng(6)
    be = hip_backend(64, 2)
    try:
        for fn, a, b in ((0, u, None), (1, ang, None), (2, y, x), (3, np.concatenate([ang, ang * 50.0, rng.uniform(-1e9, 1e9, 1000)]), None),
                         (4, u, rng.uniform(0, 1, u.size))):
            d, h = be.math_eval(fn, a, b), ob.math_eval(fn, a, b)
            assert np.array_equal(d[0], h[0]), (fn, int((d[0] != h[0]).sum()), a.size)
            if fn in (1, 4):
                assert np.array_equal(d[1], h[1]), (fn, int((d[1] != h[1]).sum()), a.size)
    finally:
        be.close()


def test_se2_reverse_root_over_the_whole_circle(hip_backend):
    """reverse SE(2) solves with headings all around the circle (and beyond +-pi before wrapping): the root of the residual
    involves sin / cos of the solved heading"""
    N = 256
    rng = np.random.default_rng(0)
    o = np.stack([rng.normal(0, 3, N), rng.normal(0, 3, N), rng.uniform(-np.pi, np.pi, N)], axis=1)
    x0 = np.stack([rng.normal(0, 3, N), rng.normal(0, 3, N), rng.uniform(-np.pi, np.pi, N)], axis=1)
    for z in ([1.0, 0.5, 0.8], [-3.0, 2.0, 3.0], [0.3, -0.2, -3.1]):
        be = hip_backend(N, 3)
        try:
            be.slot_write(0, abi.SE2, ac.to_points(abi.SE2, x0), np.ones(3))
            be.slot_write(1, abi.SE2, ac.to_points(abi.SE2, o), np.ones(3))
            d = relative_factor_desc(abi.F_SE2, abi.SE2, 2, 0, [0, 1], 2, 99, z, [0.0, 0.0, 0.0])
            d.skip_bandwidth = 1
            be.run_proposals([d])
            got = ac.to_coords(abi.SE2, be.slot_read(2, abi.SE2)[0])
        finally:
            be.close()
        want = ac.root_of(abi.F_SE2, abi.SE2, np.tile(z, (N, 1)), o, 0)
        err = got - want
        err[:, 2] = ac.wrap(err[:, 2])
        assert np.abs(err).max() < 5e-3 and np.sqrt((err ** 2).mean()) < 5e-4, (z, np.abs(err).max())


@pytest.mark.parametrize("N", [37, 100, 200, 256, 300])
@pytest.mark.parametrize("shape", ["uniform", "doors", "near_pi", "one_mode"])
def test_geodesic_mean_of_beliefs_spread_over_the_circle(hip_backend, oracle_backend, N, shape):
    """calcStdBasicSpread of a circular belief: the oracle walks the running geodesic mean point by point (what
    Manifolds.jl does); the kernel iterates lifts and prefix means to the same trajectory.  Observable: a prior proposal
    with nullhypo -- the null particles get entropy scaled by spreadNH * that spread (EvalFactor.jl:464-476)"""
    rng = np.random.default_rng(N)
    x = {"uniform": rng.uniform(-np.pi, np.pi, N),
         "doors": rng.choice([-2.5, -0.8, 0.9, 2.6], N) + rng.normal(0, 0.1, N),
         "near_pi": rng.normal(3.1, 0.4, N),
         "one_mode": rng.normal(0.4, 0.3, N)}[shape]
    x = ((x + np.pi) % (2 * np.pi) - np.pi).reshape(N, 1)
    out = []
    for fac in (hip_backend, oracle_backend):
        be = fac(N, 2)
        try:
            be.slot_write(0, abi.CIRCULAR, x, np.ones(1))
            d = relative_factor_desc(abi.F_PRIOR, abi.CIRCULAR, 1, 0, [0], 1, 321, [0.3], [0.05], nullhypo=0.5)
            d.skip_bandwidth = 1
            be.run_proposals([d])
            out.append(be.slot_read(1, abi.CIRCULAR)[0])
        finally:
            be.close()
    d = (out[0] - out[1] + np.pi) % (2 * np.pi) - np.pi
    assert np.abs(d).max() < 1e-9
