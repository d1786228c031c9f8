"""Throughput of the bandwidth fit alone (nbp_bandwidth_kernel through nbp_run_bandwidth): ns per kernel
pair and the fraction of the FP64 vector peak, for a few particle counts and batch sizes."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from parity_utils import abi, iif, rand_points


def run(N, nfits, manifold=abi.EUCLID2):
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
    global LAST_SHA  # the fitted bandwidths of the distinct beliefs, bit for bit (kernel experiments must not move them)
    LAST_SHA = hashlib.sha1(b"".join(be.slot_read(s, manifold)[1].tobytes() for s in range(min(nfits, 64)))).hexdigest()[:12]
    be.close()
    pairs = ev * N * (N - 1) / 2
    return t, ev, t * 1e6 / pairs * 1e3, pairs * 25 / (t * 1e-3) / 78.6e12


if __name__ == "__main__":
    if len(sys.argv) == 3:  # one geometry (for a rocprofv3 --pmc pass): N nfits
        N, nfits = int(sys.argv[1]), int(sys.argv[2])
        t, ev, ps, frac = run(N, nfits)
        print(f"N={N:4d} fits={nfits:5d} x2 coords: {t:9.3f} ms, {ev:9.0f} evals, {ps:7.3f} ps/pair, FP64 frac {frac:.3f}  bw sha {LAST_SHA}", flush=True)
        sys.exit(0)
    for N in (192, 200, 256, 128, 100):
        for nfits in (1, 64, 2048, 8192):
            t, ev, ps, frac = run(N, nfits)
            print(f"N={N:4d} fits={nfits:5d} x2 coords: {t:9.3f} ms, {ev:9.0f} evals, {ps:7.3f} ps/pair, FP64 frac {frac:.3f}  bw sha {LAST_SHA}", flush=True)
