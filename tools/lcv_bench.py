"""Throughput of the bandwidth fit alone (nbp_bandwidth_kernel through nbp_run_bandwidth): with every evaluation in double
precision (NBP_FIT_F64=1) ps per kernel pair and the fraction of the FP64 vector peak; as shipped (the bracketing
evaluations in single precision) the time and the evaluations of either kind -- for a few particle counts and batch sizes."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from parity_utils import abi, iif, rand_points


def run(N, nfits, manifold=abi.EUCLID2, f64=True):
    os.environ["NBP_FIT_F64"] = "1" if f64 else "0"  # read when the context is created
    be = iif.HipBackend(N, nfits, 0)
    rng = np.random.default_rng(0)
    for s in range(min(nfits, 64)):
        be.slot_write(s, manifold, rand_points(rng, manifold, N, 0.0, 0.5))
    if nfits > 64:
        be.run_copies([abi.CopyDesc(s % 64, s) for s in range(64, nfits)])
    slots, manis = list(range(nfits)), [manifold] * nfits
    be.run_bandwidth(slots, manis)
    be.timing_enable(True)
    be.timing_read()
    be.diag(reset=True)
    for _ in range(3):
        be.run_bandwidth(slots, manis)
    t = be.timing_read()["nbp_bandwidth_kernel"][0] / 3
    ev = be.diag()["lcv_evals"] / 3
    global LAST_F32
    LAST_F32 = be.diag()["lcv_evals_f32"] / 3
    global LAST_SHA  # the fitted bandwidths of the distinct beliefs, bit for bit (kernel experiments must not move them)
    LAST_SHA = hashlib.sha1(b"".join(be.slot_read(s, manifold)[1].tobytes() for s in range(min(nfits, 64)))).hexdigest()[:12]
    be.close()
    pairs = ev * N * (N - 1) / 2
    return t, ev, t * 1e6 / pairs * 1e3, pairs * 25 / (t * 1e-3) / 78.6e12


if __name__ == "__main__":
    def line(N, nfits):
        t, ev, ps, frac = run(N, nfits, f64=True)
        sha = LAST_SHA
        tb, evb, _, _ = run(N, nfits, f64=False)
        print(f"N={N:4d} fits={nfits:5d} x2 coords: all-double {t:9.3f} ms, {ev:9.0f} evals, {ps:7.3f} ps/pair, FP64 frac {frac:.3f} | "
              f"as shipped {tb:9.3f} ms, {evb:9.0f} double + {LAST_F32:9.0f} single evals, x{t / tb:.2f}  bw sha {sha} "
              f"{'= ' if LAST_SHA == sha else '!= '}{LAST_SHA}", flush=True)

    if len(sys.argv) >= 3:  # one geometry (for a rocprofv3 --pmc pass): N nfits [f64]
        N, nfits = int(sys.argv[1]), int(sys.argv[2])
        if len(sys.argv) > 3:
            t, ev, ps, frac = run(N, nfits, f64=sys.argv[3] == "f64")
            print(f"N={N:4d} fits={nfits:5d} x2 coords ({sys.argv[3]}): {t:9.3f} ms, {ev:9.0f} double + {LAST_F32:9.0f} single evals  bw sha {LAST_SHA}", flush=True)
        else:
            line(N, nfits)
        sys.exit(0)
    for N in (192, 200, 256, 300, 128, 100):
        for nfits in (1, 64, 2048, 8192):
            line(N, nfits)
