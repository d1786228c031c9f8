"""many random fits, sequential vs speculative search (3 and 7 workgroups): count the bandwidths that differ"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif

def fits(N, man, data, group, env):
    for k in ("NBP_NO_SPECULATIVE_FITS", "NBP_SPEC_DEPTH3"):
        os.environ.pop(k, None)
    os.environ.update(env)
    be = iif.HipBackend(N, group, 0)
    out = []
    for g0 in range(0, len(data), group):
        chunk = data[g0:g0 + group]
        for s, pts in enumerate(chunk):
            be.slot_write(s, man, pts)
        be.run_bandwidth(list(range(len(chunk))), [man] * len(chunk))
        out += [be.slot_read(s, man)[1].copy() for s in range(len(chunk))]
    be.close()
    return np.array(out)

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = 0
for N in (64, 64, 32, 60, 70):
    man = abi.EUCLID2
    data = []
    for i in range(96):
        kind = i % 4
        if kind == 0: p = rng.normal(0, rng.uniform(0.01, 5), (N, 2))
        elif kind == 1: p = np.concatenate([rng.normal(-3, 0.3, (N // 2, 2)), rng.normal(4, 1.0, (N - N // 2, 2))])
        elif kind == 2: p = rng.uniform(-1, 1, (N, 2)) * rng.uniform(0.1, 100)
        else: p = rng.normal(0, 1, (N, 2)) + np.arange(N)[:, None] * 0.05
        data.append(p)
    seq = fits(N, man, data, 3, {"NBP_NO_SPECULATIVE_FITS": "1"})
    for grp, env, name in ((3, {}, "K=7"), (8, {"NBP_SPEC_DEPTH3": "0"}, "K=3")):
        got = fits(N, man, data, grp, env)
        d = np.argwhere(got != seq)
        bad += len(d)
        print(N, name, "differing bandwidths:", len(d), [tuple(x) for x in d[:6]], (got[got != seq][:3], seq[got != seq][:3]) if len(d) else "")
print("TOTAL", bad)
