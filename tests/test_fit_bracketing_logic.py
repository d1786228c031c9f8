"""The search that brackets in single precision (nbp_device.h lcv_bandwidth_1d), restated in Python, against the plain
golden-section search of the oracle (oracle/nbp_oracle.c orc_lcv_bandwidth_1d) -- on the CPU, with the single-precision
value replaced by the exact one plus ANY perturbation inside the stated bound (adversarial signs included): the escalation
rule must give the all-double search's bandwidth, bit for bit, whatever the perturbation does.  (The device's evaluations
are checked against the bound in tools/exp/lcv_f32_values.hip and its bandwidths in tests/test_gpu_fit_bracketing.py; this
file is about the rule.)"""
import math

import numpy as np
import pytest

U = 2.0 ** -24
R, C, TOL = 0.61803399, 1.0 - 0.61803399, 1e-2


def loo(x, h):
    n = len(x)
    d = x[:, None] - x[None, :]
    e = np.exp(-d * d / (2 * h * h))
    np.fill_diagonal(e, 0.0)
    s = np.maximum(e.sum(1), 1e-300)
    return -float(np.mean(np.log(s) - (math.log(h) + 0.5 * math.log(2 * math.pi) + math.log(n - 1))))


def bracket(x):
    lo, hi = x.min(), x.max()
    minm = np.abs(np.diff(x)).min()
    maxm = hi - lo
    minm = max(minm, 1e-6 * maxm)
    sc = 0.5 * (minm + maxm)
    return minm / sc, 1.0, maxm / sc, sc, 0.5 * (hi - lo)


def plain_search(x):
    ax, bx, cx, sc, _ = bracket(x)
    x0, x3 = ax, cx
    if abs(cx - bx) > abs(bx - ax):
        x1, x2 = bx, bx + C * (cx - bx)
    else:
        x2, x1 = bx, bx - C * (bx - ax)
    f1, f2 = loo(x, x1 * sc), loo(x, x2 * sc)
    n = 2
    while abs(x3 - x0) > TOL * (abs(x1) + abs(x2)):
        n += 1
        if f2 < f1:
            x0, x1, x2 = x1, x2, R * x2 + C * x3
            f1, f2 = f2, loo(x, x2 * sc)
        else:
            x3, x2, x1 = x2, x1, R * x1 + C * x0
            f2, f1 = f1, loo(x, x1 * sc)
    return (x1 if f1 < f2 else x2) * sc, n


def bound(xmax, h, n):  # lcv_f32_bound
    return U * (24.0 * (xmax * (0.84932180028801907 / h)) + 314.0 + 0.25 * n)


def bracketed_search(x, noise):
    """lcv_bandwidth_1d: `noise(k, b)` = what the k-th single-precision evaluation is off by (|.| <= its bound b)"""
    ax, bx, cx, sc, xmax = bracket(x)
    x0, x3 = ax, cx
    if abs(cx - bx) > abs(bx - ax):
        x1, x2 = bx, bx + C * (cx - bx)
    else:
        x2, x1 = bx, bx - C * (bx - ax)
    m32, f1, f2, e1, e2, pt, todo, c = True, 0.0, 0.0, 0.0, 0.0, x1, 0, False
    nd = ns = 0
    while True:
        ev = 0.0
        if m32 and todo <= 2:
            ev = bound(xmax, pt * sc, len(x))
            v = loo(x, pt * sc) + noise(ns, ev)
            ns += 1
        else:
            v = loo(x, pt * sc)
            nd += 1
        if todo == 0:
            f1, e1, todo, pt = v, ev, 1, x2
            continue
        if todo == 1 or (todo == 2 and c) or todo == 4:
            f2, e2 = v, ev
        else:
            f1, e1 = v, ev
        if (e1 > 0 or e2 > 0) and not abs(f1 - f2) > e1 + e2:
            m32 = False
            if e1 > 0:
                todo, pt = 3, x1
            else:
                todo, pt = 4, x2
            continue
        if not abs(x3 - x0) > TOL * (abs(x1) + abs(x2)):
            break
        c = f2 < f1
        if c:
            x0, x1, x2 = x1, x2, R * x2 + C * x3
            f1, e1, pt = f2, e2, x2
        else:
            x3, x2, x1 = x2, x1, R * x1 + C * x0
            f2, e2, pt = f1, e1, x1
        todo = 2
    return (x1 if f1 < f2 else x2) * sc, nd, ns


def clouds():
    rng = np.random.default_rng(12)
    for _ in range(6):
        yield rng.normal(size=120) * rng.uniform(0.1, 5) + rng.normal() * 50
    for _ in range(3):
        yield np.concatenate([rng.normal(0, 0.2, 70), rng.normal(4, 0.5, 50)])
    for _ in range(3):
        yield rng.standard_cauchy(100)
    yield np.repeat(rng.normal(size=40), 3) + rng.normal(size=120) * 1e-9
    yield rng.uniform(-1, 1, 33)


@pytest.mark.parametrize("mode", ["zero", "plus", "minus", "alternating", "against_the_gap", "random"])
def test_any_perturbation_inside_the_bound_gives_the_plain_searchs_bandwidth(mode):
    rng = np.random.default_rng(5)
    singles = 0
    for x in clouds():
        x = rng.permutation(x)
        want, nplain = plain_search(x)
        noise = {"zero": lambda k, b: 0.0, "plus": lambda k, b: b, "minus": lambda k, b: -b,
                 "alternating": lambda k, b: b if k % 2 else -b,
                 # (pushes consecutive values towards each other: the perturbation most likely to flip a comparison)
                 "against_the_gap": lambda k, b: (-b if k % 2 else b) * 0.999,
                 "random": lambda k, b: float(rng.uniform(-b, b))}[mode]
        got, nd, ns = bracketed_search(x, noise)
        assert got == want  # the same floating-point number
        singles += ns
        assert nd <= nplain + 2  # at most the two re-evaluations on top of what the plain search takes in double precision
    assert singles > 0


def test_the_bracketing_phase_takes_over_most_evaluations_of_an_ordinary_cloud():
    rng = np.random.default_rng(9)
    tot_d = tot_s = tot_p = 0
    for _ in range(10):
        x = rng.normal(size=200)
        want, nplain = plain_search(x)
        got, nd, ns = bracketed_search(x, lambda k, b: 0.0)
        assert got == want
        tot_d, tot_s, tot_p = tot_d + nd, tot_s + ns, tot_p + nplain
    assert tot_s > tot_d and tot_d < 0.6 * tot_p, (tot_d, tot_s, tot_p)
