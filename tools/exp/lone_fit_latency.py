"""Latency of small launches of fits (N = 200, Euclid(2)): the speculative search (3 / 7 workgroups per fit, double precision)
against the sequential search with its bracketing evaluations in single precision, and all-double."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, rand_points

def run(N, nfits, env):
    keys = ("NBP_NO_SPECULATIVE_FITS", "NBP_SPEC_DEPTH3", "NBP_FIT_F64")
    for k in keys: os.environ.pop(k, None)
    os.environ.update(env)
    be = iif.HipBackend(N, nfits, 0)
    rng = np.random.default_rng(0)
    for s in range(nfits): be.slot_write(s, abi.EUCLID2, rand_points(rng, abi.EUCLID2, N, 0.0, 0.5))
    sl, ma = list(range(nfits)), [abi.EUCLID2] * nfits
    be.run_bandwidth(sl, ma)
    be.timing_enable(True); be.timing_read()
    for _ in range(5): be.run_bandwidth(sl, ma)
    t = be.timing_read()["nbp_bandwidth_kernel"][0] / 5
    bw = np.array([be.slot_read(s, abi.EUCLID2)[1] for s in range(nfits)])
    be.close()
    return t * 1e3, bw

for N in (200, 300):
    for nfits in (1, 2, 4, 8, 16, 45, 91, 180, 330):
        a, ba = run(N, nfits, {})
        b, bb = run(N, nfits, {"NBP_SPEC_DEPTH3": "0"})
        c, bc = run(N, nfits, {"NBP_NO_SPECULATIVE_FITS": "1"})
        d, bd = run(N, nfits, {"NBP_NO_SPECULATIVE_FITS": "1", "NBP_FIT_F64": "1"})
        print(f"N={N} fits={nfits:4d}: default {a:7.1f} us | depth 2 at most {b:7.1f} | sequential bracketed {c:7.1f} | sequential all-double {d:7.1f} | same {np.array_equal(ba, bc) and np.array_equal(ba, bd) and np.array_equal(ba, bb)}", flush=True)
