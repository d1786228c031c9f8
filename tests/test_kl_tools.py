"""CPU: the symmetric-KL estimator of tests/kl_tools.py against closed forms, and its noise floor."""
import numpy as np

import kl_tools
from parity_utils import abi, iif


def test_identical_sets_read_zero():
    x = np.random.default_rng(0).normal(size=(200, 2))
    assert kl_tools.symmetric_kl_coords(x, x.copy(), [False, False]) < 1e-12


def test_shifted_and_scaled_gaussians_read_the_closed_form():
    # symKL(N(0,1), N(d, s^2)) = 1/4 [ (1 + d^2)/s^2 + s^2 + d^2 - 2 ];  the kernel smoothing shrinks it (data processing)
    rng = np.random.default_rng(1)
    for d, s in ((0.5, 1.0), (1.0, 1.0), (0.0, 1.5), (0.7, 0.8)):
        exact = 0.25 * ((1 + d * d) / (s * s) + s * s + d * d - 2)
        est = np.mean([kl_tools.symmetric_kl_coords(rng.normal(0, 1, (400, 1)), rng.normal(d, s, (400, 1)), [False]) for _ in range(8)])
        assert 0.55 * exact < est < 1.25 * exact + 0.01, (d, s, exact, est)


def test_circular_seam_is_invisible():
    rng = np.random.default_rng(2)
    a, b = rng.normal(0, 0.2, (200, 1)), rng.normal(0.05, 0.2, (200, 1))
    k0 = kl_tools.symmetric_kl_coords(a, b, [True])
    k1 = kl_tools.symmetric_kl_coords(kl_tools.wrap(a + 3.1), kl_tools.wrap(b + 3.1), [True])
    assert abs(k0 - k1) < 1e-9


def test_noise_floor():
    """two independent N = 200 samples of the SAME density: what "equal in distribution" reads on this estimator"""
    rng = np.random.default_rng(3)
    for D in (1, 2, 3):
        est = [kl_tools.symmetric_kl_coords(rng.normal(size=(200, D)), rng.normal(size=(200, D)), [False] * D, return_raw=True) for _ in range(12)]
        assert np.mean([e[0] for e in est]) < 0.02 and np.max([e[0] for e in est]) < 0.05, (D, est)
        # the plug-in figure alone sits at or above the 0.05 bound for D >= 2: that is why the baseline is subtracted
        assert np.mean([e[1] for e in est]) > (0.01, 0.04, 0.1)[D - 1]


def test_against_exact_gaussian():
    rng = np.random.default_rng(4)
    assert kl_tools.symmetric_kl_to_gaussian(rng.normal(2.0, 0.5, 400), 2.0, 0.5) < 0.02
    far = kl_tools.symmetric_kl_to_gaussian(rng.normal(2.5, 0.5, 400), 2.0, 0.5)
    assert 0.3 < far < 0.6, far  # closed form 0.5


def test_independent_solves_differ_by_far_more_than_the_bound(oracle_backend):
    """Why the 0.05-nat figure is a common-random-numbers figure: two oracle solves of the config-1 chain with
    INDEPENDENT streams read 0.5-1.9 nats per variable (the posteriors of this algorithm are narrower than the
    spread of their means from seed to seed -- sigma 0.3-0.4 against mean offsets of +-0.3), while the same seed gives
    the same particles.  tests/kl_parity.py built the whole-solve criterion of rounds 2-5 on exactly this (retired in round 6: whole solves are bit-identical)."""
    def chain():
        fg = iif.initfg(iif.SolverParams(N=100))
        for i in range(6):
            iif.addVariable(fg, f"x{i}", iif.ContinuousScalar)
        iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal(0.0, 1.0)))
        for i in range(5):
            iif.addFactor(fg, [f"x{i}", f"x{i+1}"], iif.LinearRelative(iif.Normal(1.0, 0.1)))
        return fg
    fa, fb, fc = chain(), chain(), chain()
    iif.solveTree(fa, backend=oracle_backend, seed=11)
    iif.solveTree(fb, backend=oracle_backend, seed=12)
    iif.solveTree(fc, backend=oracle_backend, seed=11)
    assert max(kl_tools.kl_table(abi, fa, fc).values()) < 1e-12  # same seed: the same particles
    kl = kl_tools.kl_table(abi, fa, fb)
    assert np.median(list(kl.values())) > 0.05, kl
