"""speculative fits from several contexts at once (their kernels overlap on the device): compare with the sequential search"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif

def run(N, man, data, group, nctx, env):
    for k in ("NBP_NO_SPECULATIVE_FITS", "NBP_SPEC_DEPTH3"):
        os.environ.pop(k, None)
    os.environ.update(env)
    bes = [iif.HipBackend(N, group, 0) for _ in range(nctx)]
    out = [None] * len(data)
    per = group * nctx
    for g0 in range(0, len(data), per):
        chunks = [data[g0 + c * group: g0 + (c + 1) * group] for c in range(nctx)]
        for be, ch in zip(bes, chunks):
            for s, pts in enumerate(ch):
                be.slot_write(s, man, pts)
        for be, ch in zip(bes, chunks):
            if ch:
                be.run_bandwidth(list(range(len(ch))), [man] * len(ch))   # asynchronous: the launches of all contexts overlap
        for c, (be, ch) in enumerate(zip(bes, chunks)):
            for s in range(len(ch)):
                out[g0 + c * group + s] = be.slot_read(s, man)[1].copy()
    for be in bes:
        be.close()
    return np.array(out)

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
tot = 0
for N in (64, 128, 200):
    data = [rng.normal(0, rng.uniform(0.01, 5), (N, 2)) for _ in range(192)]
    seq = run(N, abi.EUCLID2, data, 8, 1, {"NBP_NO_SPECULATIVE_FITS": "1"})
    for nctx, grp, env, name in ((8, 3, {}, "K=7 x8ctx"), (8, 8, {"NBP_SPEC_DEPTH3": "0"}, "K=3 x8ctx"), (1, 8, {"NBP_SPEC_DEPTH3": "0"}, "K=3 x1ctx")):
        got = run(N, abi.EUCLID2, data, grp, nctx, env)
        n = int((got != seq).sum()); tot += n
        print(N, name, "differing:", n)
print("TOTAL", tot)
