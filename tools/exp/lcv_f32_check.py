"""Single-precision bracketing of the bandwidth searches (neg_loo_ll_f32): same bandwidths bit for bit as the all-double search
(NBP_FIT_F64=1), how many evaluations of each kind a fit takes, and what a chip-filling launch of fits costs either way.
Usage (GPU box): python tools/exp/lcv_f32_check.py"""
import hashlib
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, rand_points


def clouds(rng, manifold, N, kind):
    D = abi.MANIFOLD_DIM[manifold]
    eucl = manifold not in (abi.CIRCULAR, abi.SE2)
    if kind == "gauss":
        c = rng.normal(size=(N, D)) * 0.5
    elif kind == "modes":  # four well separated modes
        c = rng.normal(size=(N, D)) * 0.05 + np.array([-2.5, -0.7, 0.3, 2.1])[np.arange(N) % 4][:, None]
    elif kind == "outliers":  # a cloud far from the origin and a few isolated points
        c = rng.normal(size=(N, D)) * 0.5 + (40.0 if eucl else 0.0)
        c[0] += 30.0 if eucl else 2.0
        c[1] -= 55.0 if eucl else 1.5
    else:  # "offset": a tight cloud and two points far away
        c = rng.normal(size=(N, D)) * 0.01
        c[:2] += 1000.0 if eucl else 3.0
    if manifold == abi.SE2:
        th = c[:, 2]
        return np.stack([c[:, 0], c[:, 1], np.cos(th), np.sin(th), -np.sin(th), np.cos(th)], axis=1)
    if manifold == abi.CIRCULAR:
        return (c + np.pi) % (2 * np.pi) - np.pi
    return c


def run(N, nfits, manifold, kind, f64):
    os.environ["NBP_FIT_F64"] = "1" if f64 else "0"
    be = iif.HipBackend(N, nfits, 0)
    rng = np.random.default_rng(7)
    nd = min(nfits, 64)
    for s in range(nd):
        be.slot_write(s, manifold, clouds(rng, manifold, N, kind))
    if nfits > nd:
        be.run_copies([abi.CopyDesc(s % nd, s) for s in range(nd, nfits)])
    slots, manis = list(range(nfits)), [manifold] * nfits
    be.run_bandwidth(slots, manis)
    be.timing_enable(True)
    be.timing_read()
    be.diag(reset=True)
    for _ in range(3):
        be.run_bandwidth(slots, manis)
    t = be.timing_read()["nbp_bandwidth_kernel"][0] / 3
    d = be.diag()
    bw = np.concatenate([be.slot_read(s, manifold)[1] for s in range(nd)])
    be.close()
    return t, d["lcv_evals"] / 3 / nfits, d["lcv_evals_f32"] / 3 / nfits, hashlib.sha1(bw.tobytes()).hexdigest()[:12], bw


if __name__ == "__main__":
    names = {abi.EUCLID1: "Euclid(1)", abi.EUCLID2: "Euclid(2)", abi.EUCLID3: "Euclid(3)", abi.CIRCULAR: "Circular", abi.SE2: "SE(2)"}
    bad = 0
    for N, nfits in ((200, 8192), (200, 256), (200, 1), (300, 4096), (100, 4096), (257, 512), (64, 512), (37, 64)):
        for manifold in (abi.EUCLID2, abi.CIRCULAR, abi.SE2, abi.EUCLID3):
            for kind in ("gauss", "modes", "outliers", "offset"):
                if nfits > 256 and (manifold, kind) not in ((abi.EUCLID2, "gauss"), (abi.CIRCULAR, "modes"), (abi.EUCLID3, "gauss"), (abi.SE2, "gauss")):
                    continue
                a = run(N, nfits, manifold, kind, True)
                b = run(N, nfits, manifold, kind, False)
                same = a[3] == b[3]
                bad += 0 if same else 1
                print(f"N={N:4d} fits={nfits:5d} {names[manifold]:10s} {kind:9s}: all-double {a[0]:8.3f} ms ({a[1]:5.1f} evals per coordinate fit) | "
                      f"bracketed {b[0]:8.3f} ms ({b[1]:5.1f} double + {b[2]:5.1f} single) x{a[0] / b[0]:.2f} | bandwidths "
                      f"{'identical ' + a[3] if same else 'DIFFER: max rel ' + str(np.max(np.abs(a[4] - b[4]) / np.maximum(np.abs(a[4]), 1e-300)))}", flush=True)
    print("bandwidth sets that differ:", bad)
    sys.exit(1 if bad else 0)
