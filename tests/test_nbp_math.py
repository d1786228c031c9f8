"""The elementary functions of the continuous data path (include/nbp_math.h: one definition for the kernels and the CPU
checker) against the host libm, and the checker's device-order spread statistics against the reference's definition.

CPU leg: accuracy -- nothing here compares one implementation of the product path with another.  The GPU leg
(tests/test_gpu_device_math.py) evaluates the same header on the device and asserts the same BITS."""
import numpy as np
import pytest

from oracle import oracle_backend as ob


def _ulps(got, want):
    return np.abs(got - want) / np.spacing(np.abs(want))


def math_arguments(seed=0, n=400_000):
    rng = np.random.default_rng(seed)
    u = np.concatenate([(rng.integers(0, 2 ** 53, n).astype(np.float64) + 0.5) / 2.0 ** 53,     # the uniforms Box-Muller sees
                        np.ldexp(rng.uniform(0.5, 1.0, n // 4), -rng.integers(0, 54, n // 4)),  # down to 2^-54
                        1.0 - np.ldexp(rng.uniform(0.5, 1.0, n // 4), -rng.integers(1, 53, n // 4))])
    ang = np.concatenate([rng.uniform(-7, 7, n), rng.uniform(-2e4, 2e4, n // 4), np.arange(-64, 65) * (np.pi / 4),
                          np.arange(-64, 65) * (np.pi / 4) + rng.normal(0, 1e-9, 129)])
    y = rng.normal(size=n) * np.ldexp(1.0, rng.integers(-20, 20, n))
    x = rng.normal(size=n) * np.ldexp(1.0, rng.integers(-20, 20, n))
    y[:8] = [0.0, 0.0, 1.0, -1.0, 0.0, -0.0, 1e-300, 1e300]
    x[:8] = [1.0, -1.0, 0.0, 0.0, 0.0, -1.0, 1e300, 1e-300]
    return u, ang, y, x


def test_log_sincos_atan2_within_an_ulp_of_libm():
    u, ang, y, x = math_arguments()
    assert _ulps(ob.math_eval(0, u)[0], np.log(u)).max() <= 1.0
    # sin / cos: within an ulp, plus the reduction's absolute error next to a zero of the function (pi/2 is carried to 86 bits:
    # |a| * 1e-26 -- at a = 14 pi, where sin is 1.7e-15, that is the tenth digit of a number that stands for zero)
    s, c = ob.math_eval(1, ang)
    assert np.all(np.abs(s - np.sin(ang)) <= np.spacing(np.abs(np.sin(ang))) + np.abs(ang) * 1e-26)
    assert np.all(np.abs(c - np.cos(ang)) <= np.spacing(np.abs(np.cos(ang))) + np.abs(ang) * 1e-26)
    a = ob.math_eval(2, y, x)[0]
    want = np.arctan2(y, x)
    nz = want != 0
    assert _ulps(a[nz], want[nz]).max() <= 1.5 and np.all(a[~nz] == 0)


def test_wrap_is_the_exact_remainder():
    rng = np.random.default_rng(1)
    a = np.concatenate([rng.uniform(-np.pi, np.pi, 1000), rng.uniform(-9, 9, 100000), rng.uniform(-1e6, 1e6, 100000),
                        [np.pi, -np.pi, 3 * np.pi, -3 * np.pi, 0.0]])
    w = ob.math_eval(3, a)[0]
    assert np.all((w >= -np.pi) & (w < np.pi))
    inside = (a >= -np.pi) & (a < np.pi)
    assert np.array_equal(w[inside], a[inside])  # the identity on the principal interval
    want = np.fmod(a + np.pi, 2 * np.pi)
    want = np.where(want < 0, want + 2 * np.pi, want) - np.pi
    assert np.array_equal(w[~inside], want[~inside])


def test_box_muller_pair():
    rng = np.random.default_rng(2)
    ua, ub = rng.uniform(1e-16, 1, 200000), rng.uniform(0, 1, 200000)
    na, nb = ob.math_eval(4, ua, ub)
    r = np.sqrt(-2 * np.log(ua))
    np.testing.assert_allclose(na, r * np.cos(2 * np.pi * ub), rtol=0, atol=1e-14)
    np.testing.assert_allclose(nb, r * np.sin(2 * np.pi * ub), rtol=0, atol=1e-14)
    assert abs(na.mean()) < 0.01 and abs(na.std() - 1) < 0.01 and abs(np.mean(na * nb)) < 0.01


@pytest.mark.parametrize("N", [37, 64, 100, 200, 256, 300, 500])
def test_device_order_mean_is_the_reference_walk_up_to_rounding(N):
    """the checker sums a belief's spread statistics in the order the kernels reduce them (chunks of 64 by a butterfly, the
    lifts of a circular coordinate iterated with prefix sums): the reference's own definition -- Manifolds.jl's running
    geodesic mean, point by point -- must come out of it up to the rounding of the sums, on every shape of belief"""
    import ctypes as C
    L = ob.lib()
    rng = np.random.default_rng(N)
    dp = C.POINTER(C.c_double)
    worst = 0.0
    for shape in range(6):
        for _ in range(10):
            x = {0: rng.uniform(-np.pi, np.pi, N), 1: rng.choice([-2.5, -0.8, 0.9, 2.6], N) + rng.normal(0, 0.1, N),
                 2: rng.normal(3.1, 0.4, N), 3: rng.normal(0.4, 0.3, N), 4: rng.normal(0, 2.0, N), 5: rng.normal(-3.0, 1.0, N)}[shape]
            x = np.ascontiguousarray((x + np.pi) % (2 * np.pi) - np.pi)
            a = L.orc_mean_geodesic_device_order(x.ctypes.data_as(dp), N, 1)
            b = L.orc_mean_geodesic_walk(x.ctypes.data_as(dp), N, 1)
            worst = max(worst, abs((a - b + np.pi) % (2 * np.pi) - np.pi))
            a = L.orc_mean_geodesic_device_order(x.ctypes.data_as(dp), N, 0)
            b = L.orc_mean_geodesic_walk(x.ctypes.data_as(dp), N, 0)
            worst = max(worst, abs(a - b))
    assert worst < 1e-13, worst
