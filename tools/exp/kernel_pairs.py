"""the same fits through every kernel that can run them: bandwidth kernel / prep kernel, sequential / 3 / 7 workgroups.
Programs of nprod products of two prior proposals each: the prep launch fits 2*nprod proposals."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, relative_factor_desc
from iif_amd.solver import product_desc

def prep_fits(N, nprod, env, seed0=100):
    for k in ("NBP_NO_SPECULATIVE_FITS", "NBP_SPEC_DEPTH3"):
        os.environ.pop(k, None)
    os.environ.update(env)
    man = abi.EUCLID2
    nslots = 3 * nprod
    be = iif.HipBackend(N, nslots, 0)
    props, prods = [], []
    for i in range(nprod):
        for j in range(2):
            d = relative_factor_desc(abi.F_PRIOR, man, 1, 0, [3 * i + 2], 3 * i + j, seed0 + 7 * i + j, [float(i), -1.0 + j], [0.3 + 0.1 * j, 0.5])
            props.append(d)
        prods.append(product_desc(man, [3 * i, 3 * i + 1], 3 * i + 2, seed0 + 1000 + i))
    rng = np.random.default_rng(0)
    for i in range(nprod):
        be.slot_write(3 * i + 2, man, rng.normal(size=(N, 2)), np.ones(2))
    prog = be.program([(abi.STAGE_PROPOSALS, props), (abi.STAGE_PRODUCTS, prods)], lazy_bandwidth=True)
    prog.run(); be.synchronize()
    out = np.array([be.slot_read(3 * i + j, man)[1] for i in range(nprod) for j in range(2)])
    pts = np.array([be.slot_read(3 * i + j, man)[0] for i in range(nprod) for j in range(2)])
    prog.close(); be.close()
    return out, pts

def bw_kernel(N, pts, group, env):
    for k in ("NBP_NO_SPECULATIVE_FITS", "NBP_SPEC_DEPTH3"):
        os.environ.pop(k, None)
    os.environ.update(env)
    be = iif.HipBackend(N, group, 0)
    out = []
    for g0 in range(0, len(pts), group):
        ch = pts[g0:g0 + group]
        for s, p in enumerate(ch):
            be.slot_write(s, abi.EUCLID2, p)
        be.run_bandwidth(list(range(len(ch))), [abi.EUCLID2] * len(ch))
        out += [be.slot_read(s, abi.EUCLID2)[1].copy() for s in range(len(ch))]
    be.close()
    return np.array(out)

tot = 0
for N in (64, 100, 128, 200):
    for nprod in (1, 3, 6, 12):
        ref, pts = prep_fits(N, nprod, {"NBP_NO_SPECULATIVE_FITS": "1"})
        res = {"prep spec (auto depth)": prep_fits(N, nprod, {})[0], "prep K=3": prep_fits(N, nprod, {"NBP_SPEC_DEPTH3": "0"})[0],
               "bw seq": bw_kernel(N, pts, 8, {"NBP_NO_SPECULATIVE_FITS": "1"}), "bw K=7": bw_kernel(N, pts, 3, {}),
               "bw K=3": bw_kernel(N, pts, 8, {"NBP_SPEC_DEPTH3": "0"})}
        for k, v in res.items():
            n = int((v != ref).sum()); tot += n
            if n:
                print(N, nprod, k, "differs from prep seq in", n, "of", v.size)
print("TOTAL", tot)
