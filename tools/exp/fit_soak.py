"""Soak of the bracketed bandwidth search against the all-double one (NBP_FIT_F64=1): random particle counts, slot sizes,
manifolds, cloud shapes (scales from 1e-6 to 1e6, offsets to 1e6, duplicates, heavy tails, lattices), sequential and
speculative geometry -- every bandwidth must be bit-equal.  Usage (GPU box): python tools/exp/fit_soak.py [rounds]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif

MANIS = [abi.EUCLID1, abi.EUCLID2, abi.EUCLID3, abi.CIRCULAR, abi.SE2]


def cloud(rng, man, n):
    D = abi.MANIFOLD_DIM[man]
    kind = rng.integers(0, 7)
    scale = 10.0 ** rng.uniform(-6, 6) if man not in (abi.CIRCULAR, abi.SE2) else 10.0 ** rng.uniform(-4, 0.5)
    off = rng.normal(size=D) * 10.0 ** rng.uniform(-2, 6) if man not in (abi.CIRCULAR, abi.SE2) else rng.normal(size=D)
    if kind == 0: c = rng.normal(size=(n, D))
    elif kind == 1: c = rng.standard_cauchy(size=(n, D))
    elif kind == 2: c = rng.normal(size=(n, D)) * 0.02 + rng.integers(0, 5, size=(n, 1)) * 1.0
    elif kind == 3: c = np.repeat(rng.normal(size=((n + 2) // 3, D)), 3, axis=0)[:n] + rng.normal(size=(n, D)) * 1e-9
    elif kind == 4: c = rng.uniform(-1, 1, size=(n, D))
    elif kind == 5: c = np.round(rng.normal(size=(n, D)) * 4) / 4 + rng.normal(size=(n, D)) * 1e-3
    else:
        c = rng.normal(size=(n, D)); c[: max(1, n // 50)] *= 1e3
    c = c * scale + off
    if man == abi.SE2:
        th = c[:, 2]
        return np.stack([c[:, 0], c[:, 1], np.cos(th), np.sin(th), -np.sin(th), np.cos(th)], axis=1)
    if man == abi.CIRCULAR:
        return (c + np.pi) % (2 * np.pi) - np.pi
    return c


def fit(N, man, beliefs, env):
    for k in ("NBP_FIT_F64", "NBP_NO_SPECULATIVE_FITS"): os.environ.pop(k, None)
    os.environ.update(env)
    be = iif.HipBackend(N, len(beliefs), 0)
    for s, b in enumerate(beliefs):
        if len(b) == N: be.slot_write(s, man, b)
        else: be.belief_write(s, man, b, np.ones(abi.MANIFOLD_DIM[man]))
    be.run_bandwidth(list(range(len(beliefs))), [man] * len(beliefs))
    bw = np.array([be.slot_read(s, man)[1] for s in range(len(beliefs))])
    d = be.diag()
    be.close()
    return bw, d


rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(2026)
bad = tot = 0
for r in range(rounds):
    N = int(rng.choice([200, 300, 64, 100, 256, 257, 37, 320, 128, 500]))
    man = MANIS[r % 5]
    nb = int(rng.choice([1, 3, 12, 70, 400]))
    beliefs = [cloud(rng, man, N if rng.random() < 0.6 else int(rng.integers(2, N + 1))) for _ in range(nb)]
    a, da = fit(N, man, beliefs, {"NBP_FIT_F64": "1", "NBP_NO_SPECULATIVE_FITS": "1"})
    b, db = fit(N, man, beliefs, {"NBP_NO_SPECULATIVE_FITS": "1"})
    c, dc = fit(N, man, beliefs, {})
    ok = np.array_equal(a, b) and np.array_equal(a, c) and np.all(np.isfinite(a))
    tot += a.size
    if not ok:
        bad += 1
        w = np.argwhere(~((a == b) & (a == c)))
        print(f"round {r}: N={N} manifold {man} {nb} beliefs: MISMATCH at {w[:5].tolist()} all-double {a[tuple(w[0])]} bracketed {b[tuple(w[0])]} default {c[tuple(w[0])]} count {len(beliefs[w[0][0]])}", flush=True)
    else:
        print(f"round {r}: N={N} manifold {man} {nb:3d} beliefs ok   evals all-double {da['lcv_evals']}, bracketed {db['lcv_evals']} + {db['lcv_evals_f32']} single", flush=True)
print(f"{tot} bandwidths, {bad} rounds with a mismatch")
sys.exit(1 if bad else 0)
