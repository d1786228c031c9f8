"""-m gpu: the bracketing evaluations of the bandwidth searches in single precision (neg_loo_ll_f32, nbp_device.h).  A
golden-section search uses a likelihood value only in the comparison f2 < f1; the search takes a single-precision value
where the two sides are further apart than the sum of their error bounds and re-evaluates in double precision what it
cannot decide -- so the selected bandwidth is the all-double search's (NBP_FIT_F64=1), bit for bit, on every manifold, for
clouds with separated modes, isolated points, offsets far from the origin, duplicates, and for beliefs that hold fewer
points than the slot."""
import os

import numpy as np
import pytest

from parity_utils import abi, iif

pytestmark = pytest.mark.gpu


def clouds(rng, manifold, N, kind):
    D = abi.MANIFOLD_DIM[manifold]
    eucl = manifold not in (abi.CIRCULAR, abi.SE2)
    if kind == "gauss":
        c = rng.normal(size=(N, D)) * rng.uniform(0.05, 3.0)
    elif kind == "modes":
        c = rng.normal(size=(N, D)) * 0.05 + np.array([-2.5, -0.7, 0.3, 2.1])[np.arange(N) % 4][:, None]
    elif kind == "outliers":
        c = rng.normal(size=(N, D)) * 0.5 + (40.0 if eucl else 0.0)
        c[0] += 30.0 if eucl else 2.0
        c[1] -= 55.0 if eucl else 1.5
    elif kind == "offset":
        c = rng.normal(size=(N, D)) * 0.01
        c[:2] += 1000.0 if eucl else 3.0
    elif kind == "duplicates":
        c = np.repeat(rng.normal(size=(max(N // 4, 2), D)), 4, axis=0)[:N]
        c = np.concatenate([c, rng.normal(size=(N - c.shape[0], D))]) if c.shape[0] < N else c
    else:  # "heavy": Cauchy tails
        c = rng.standard_cauchy(size=(N, D)) * (1.0 if eucl else 0.2)
    if manifold == abi.SE2:
        th = c[:, 2]
        return np.stack([c[:, 0], c[:, 1], np.cos(th), np.sin(th), -np.sin(th), np.cos(th)], axis=1)
    if manifold == abi.CIRCULAR:
        return (c + np.pi) % (2 * np.pi) - np.pi
    return c


def fit(N, manifold, beliefs, f64, copies=1):
    old = {k: os.environ.get(k) for k in ("NBP_FIT_F64", "NBP_NO_SPECULATIVE_FITS")}
    os.environ["NBP_FIT_F64"] = "1" if f64 else "0"
    os.environ["NBP_NO_SPECULATIVE_FITS"] = "1"  # the sequential search is the one that brackets
    try:
        n = len(beliefs) * copies
        be = iif.HipBackend(N, n, 0)
        try:
            for s, b in enumerate(beliefs):
                if len(b) == N:
                    be.slot_write(s, manifold, b)
                else:  # a belief that holds fewer points than the slot (the fit is of the points it holds)
                    be.belief_write(s, manifold, b, np.ones(abi.MANIFOLD_DIM[manifold]))
            if copies > 1:
                be.run_copies([abi.CopyDesc(s % len(beliefs), s) for s in range(len(beliefs), n)])
            be.diag(reset=True)
            be.run_bandwidth(list(range(n)), [manifold] * n)
            bw = np.array([be.slot_read(s, manifold)[1] for s in range(len(beliefs))])
            d = be.diag()
        finally:
            be.close()
    finally:
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    return bw, d


KINDS = ["gauss", "modes", "outliers", "offset", "duplicates", "heavy"]


@pytest.mark.parametrize("manifold", [abi.EUCLID1, abi.EUCLID2, abi.EUCLID3, abi.CIRCULAR, abi.SE2])
@pytest.mark.parametrize("N", [200, 300, 64, 37, 256, 257])
def test_bracketed_search_selects_the_all_double_bandwidth(manifold, N):
    rng = np.random.default_rng(1000 * manifold + N)
    beliefs = [clouds(rng, manifold, N, k) for k in KINDS for _ in range(3)]
    a, da = fit(N, manifold, beliefs, True)
    b, db = fit(N, manifold, beliefs, False)
    assert np.all(a > 0) and np.all(np.isfinite(a))
    np.testing.assert_array_equal(b, a)
    assert da["lcv_evals_f32"] == 0 and db["lcv_evals_f32"] > 0
    assert db["lcv_evals"] < da["lcv_evals"]  # fewer double-precision evaluations than the all-double search makes


def test_beliefs_with_fewer_points_than_the_slot():
    N, man = 200, abi.EUCLID2
    rng = np.random.default_rng(3)
    beliefs = [rng.normal(size=(n, 2)) * 0.7 for n in (200, 150, 65, 64, 33, 9, 3)]
    a, _ = fit(N, man, beliefs, True)
    b, db = fit(N, man, beliefs, False)
    np.testing.assert_array_equal(b, a)
    assert db["lcv_evals_f32"] > 0


@pytest.mark.parametrize("manifold", [abi.EUCLID2, abi.CIRCULAR, abi.SE2])
@pytest.mark.parametrize("N,counts", [(200, (150, 129, 65, 40, 9, 3)), (300, (257, 200, 130, 70, 20))])
def test_fit_of_a_belief_with_fewer_points_is_the_oracles(oracle_backend, manifold, N, counts):
    """manikde! of the points a belief HOLDS (a12): the waves of the slot's row that are left without a point take no part in
    the likelihood (through round 4 they repeated the last wave's pairs: bandwidths off by up to 15 % for counts whose last
    wave holds at most 32 points -- the first case below read 0.1214 where the oracle fits 0.1448)"""
    rng = np.random.default_rng(17 * manifold + N)
    beliefs = [clouds(rng, manifold, n, "gauss") for n in counts]
    D = abi.MANIFOLD_DIM[manifold]
    ob = oracle_backend(N, len(beliefs))
    try:
        for s, b in enumerate(beliefs):
            ob.belief_write(s, manifold, b, np.ones(D))
        ob.run_bandwidth(list(range(len(beliefs))), [manifold] * len(beliefs))
        want = np.array([ob.belief_read(s, manifold)[1] for s in range(len(beliefs))])
    finally:
        ob.close()
    for f64 in (True, False):
        got, _ = fit(N, manifold, beliefs, f64)
        np.testing.assert_allclose(got, want, rtol=0)


def test_chip_filling_launch_and_the_share_of_single_precision_evaluations():
    """2048 fits of 200-point Euclid(2) beliefs (a tree level of the 10 000-variable chain): identical bandwidths, and the
    share of the evaluations the bracketing takes over -- the reason it exists -- stays where it was measured (~0.6)"""
    N, man = 200, abi.EUCLID2
    rng = np.random.default_rng(11)
    beliefs = [rng.normal(size=(N, 2)) * rng.uniform(0.1, 2.0) + rng.normal(size=2) * 10 for _ in range(64)]
    a, da = fit(N, man, beliefs, True, copies=32)
    b, db = fit(N, man, beliefs, False, copies=32)
    np.testing.assert_array_equal(b, a)
    share = db["lcv_evals_f32"] / (db["lcv_evals_f32"] + db["lcv_evals"])
    assert share > 0.5, share
    assert db["lcv_evals"] < 0.55 * da["lcv_evals"], (db["lcv_evals"], da["lcv_evals"])


def extreme_cloud(rng, man, n):
    """scales from 1e-6 to 1e6, offsets to 1e6, duplicates, heavy tails, lattices, a few points three decades out"""
    D = abi.MANIFOLD_DIM[man]
    eucl = man not in (abi.CIRCULAR, abi.SE2)
    kind = rng.integers(0, 7)
    scale = 10.0 ** rng.uniform(-6, 6) if eucl else 10.0 ** rng.uniform(-4, 0.5)
    off = rng.normal(size=D) * 10.0 ** rng.uniform(-2, 6) if eucl else rng.normal(size=D)
    if kind == 0:
        c = rng.normal(size=(n, D))
    elif kind == 1:
        c = rng.standard_cauchy(size=(n, D))
    elif kind == 2:
        c = rng.normal(size=(n, D)) * 0.02 + rng.integers(0, 5, size=(n, 1)) * 1.0
    elif kind == 3:
        c = np.repeat(rng.normal(size=((n + 2) // 3, D)), 3, axis=0)[:n] + rng.normal(size=(n, D)) * 1e-9
    elif kind == 4:
        c = rng.uniform(-1, 1, size=(n, D))
    elif kind == 5:
        c = np.round(rng.normal(size=(n, D)) * 4) / 4 + rng.normal(size=(n, D)) * 1e-3
    else:
        c = rng.normal(size=(n, D))
        c[: max(1, n // 50)] *= 1e3
    c = c * scale + off
    if man == abi.SE2:
        th = c[:, 2]
        return np.stack([c[:, 0], c[:, 1], np.cos(th), np.sin(th), -np.sin(th), np.cos(th)], axis=1)
    if man == abi.CIRCULAR:
        return (c + np.pi) % (2 * np.pi) - np.pi
    return c


@pytest.mark.parametrize("seed", range(10))
def test_soak_of_extreme_clouds(seed):
    """(tools/exp/fit_soak.py is the long form: 13 492 bandwidths in profiles/r05_fit_bracketing.txt)"""
    rng = np.random.default_rng(4000 + seed)
    N = int(rng.choice([200, 300, 64, 100, 256, 257, 37, 320, 128, 500]))
    man = [abi.EUCLID1, abi.EUCLID2, abi.EUCLID3, abi.CIRCULAR, abi.SE2][seed % 5]
    nb = int(rng.choice([3, 12, 70]))
    beliefs = [extreme_cloud(rng, man, N if rng.random() < 0.6 else int(rng.integers(2, N + 1))) for _ in range(nb)]
    a, _ = fit(N, man, beliefs, True)
    b, _ = fit(N, man, beliefs, False)
    assert np.all(np.isfinite(a)) and np.all(a > 0)
    np.testing.assert_array_equal(b, a)


@pytest.mark.parametrize("config,nvars", [("2", None), ("3", 400), ("5", 600)])
def test_whole_solve_posteriors_do_not_depend_on_the_bracketing(config, nvars):
    """bench.py's posterior sha (every particle of every posterior of a full up + down solve) with the bracketed fits and with
    every evaluation in double precision: the bandwidths are bit-identical, so is everything computed from them
    (tools/exp/bracketing_whole_solve_sha.sh: the five BASELINE configurations at full size)"""
    import re
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    shas = []
    for f64 in ("0", "1"):
        env = dict(os.environ, NBP_BENCH_SHA="1", NBP_FIT_F64=f64)
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--config", config, "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-10k",
               "--no-profile-pass"] + (["--nvars", str(nvars)] if nvars else [])
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
        assert out.returncode == 0, out.stderr[-2000:]
        m = re.search(r"sha=([0-9a-f]+)", out.stderr)
        assert m, out.stderr[-2000:]
        shas.append(m.group(1))
    assert shas[0] == shas[1], shas
