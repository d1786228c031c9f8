"""Differential check of DEGENERATE beliefs, device against oracle, bit for bit (tests/degenerate_inputs.py asks each side only for
finite results): identical points, two clusters 1e6 apart, an offset of 1e8, a spread of 1e-12, three distinct values, and beliefs
that hold 2 / 3 / 5 / 17 / 63 / 65 points in a slot of N -- on every manifold, through the fit, a relative proposal from and to the
belief, a prior proposal, and products of two and three densities.  (FUZZ_NAN=1 adds a belief with one NaN particle: NOT part of the claim -- the bandwidth fit of
data with a NaN in it is garbage on both sides, different garbage: each side terminates, and a NaN particle stays where it was,
tests/degenerate_inputs.py.)
usage (GPU box): python tools/exp/fuzz_degenerate.py   (test infrastructure: drives the oracle)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, rand_points, relative_factor_desc, product_desc
from oracle.oracle_backend import OracleBackend

NAN = os.environ.get("FUZZ_NAN", "0") != "0"  # FUZZ_NAN=1: also a belief with one NaN particle (compared NaN for NaN; finiteness not asked)
REL = {abi.EUCLID1: (abi.F_LINREL, [1.0], [0.1]), abi.EUCLID2: (abi.F_LINREL, [1.0, -0.5], [0.1, 0.2]), abi.EUCLID3: (abi.F_LINREL, [1.0, 0.0, 0.3], [0.1, 0.1, 0.1]),
       abi.CIRCULAR: (abi.F_CIRCULAR, [0.4], [0.05]), abi.SE2: (abi.F_SE2, [1.0, 0.2, 0.3], [0.1, 0.1, 0.01])}


def coords_to_points(man, c):
    if man == abi.SE2:
        th = c[:, 2]
        return np.stack([c[:, 0], c[:, 1], np.cos(th), np.sin(th), -np.sin(th), np.cos(th)], axis=1)
    if man == abi.CIRCULAR:
        return (c + np.pi) % (2 * np.pi) - np.pi
    return c


def shapes(rng, man, n):
    D = abi.MANIFOLD_DIM[man]
    circ = man in (abi.CIRCULAR, abi.SE2)
    base = rng.normal(size=(n, D)) * 0.3
    out = {"identical": np.tile(rng.normal(size=(1, D)), (n, 1)),
           "two far clusters": base * 1e-3 + np.where(np.arange(n)[:, None] % 2 == 0, 0.0, 1e6 if not circ else 2.0),
           "huge offset": base * 0.1 + (1e8 if not circ else 3.0),
           "tiny spread": base * 1e-12 + 0.5,
           "three values": np.array([[0.1] * D, [0.7] * D, [-1.3] * D])[np.arange(n) % 3],
           "plain": base,
           "one NaN particle": np.where((np.arange(n) == min(5, n - 1))[:, None], np.nan, base)}
    if man == abi.SE2:  # keep the offsets off the heading
        for k in ("two far clusters", "huge offset"):
            out[k][:, 2] = base[:, 2]
    return {k: coords_to_points(man, v) for k, v in out.items()}


def run_case(man, N, count, shape_name, seed):
    rng = np.random.default_rng(seed)
    pts = shapes(rng, man, count)[shape_name]
    other = rand_points(rng, man, N, 0.5, 0.3)
    kind, mean, sig = REL[man]
    D = abi.MANIFOLD_DIM[man]
    res = []
    for make in (lambda: OracleBackend(N, 10, 2 * N, threads=4), lambda: iif.HipBackend(N, 10, side_ints=2 * N)):
        be = make()
        try:
            if count == N:
                be.slot_write(0, man, pts)
            else:
                be.belief_write(0, man, pts, np.full(D, 0.2))
            be.slot_write(1, man, other)
            be.run_bandwidth([0, 1], [man, man])
            descs = [relative_factor_desc(kind, man, 2, 1, [0, 1], 2, 11 + seed, mean, sig),      # from the degenerate belief
                     relative_factor_desc(kind, man, 2, 0, [0, 1], 3, 12 + seed, mean, sig),      # onto it
                     relative_factor_desc(abi.F_PRIOR, man, 1, 0, [0], 4, 13 + seed, [0.2] * D, [0.5] * D, nullhypo=0.3, mhidx_out=0)]
            be.run_proposals(descs)
            prods = [product_desc(man, [2, 1], 5, 21 + seed), product_desc(man, [3, 4, 1], 6, 22 + seed, niter=2)]
            if count == N:
                prods.append(product_desc(man, [0, 1], 7, 23 + seed))
            be.run_products(prods)
            res.append([be.slot_read(s, abi.EUCLID3) for s in range(8)] + [be.side_read(0, N)])
        finally:
            be.close()
    o, h = res
    bad = []
    for s in range(8):
        if not (np.array_equal(o[s][0], h[s][0], equal_nan=True) and np.array_equal(np.asarray(o[s][1]), np.asarray(h[s][1]), equal_nan=True)):
            bad.append(f"slot {s}: points differ by {np.nanmax(np.abs(o[s][0] - h[s][0])):.3e}, bandwidths {np.asarray(o[s][1])} vs {np.asarray(h[s][1])}")
    if not np.array_equal(o[8], h[8]):
        bad.append("hypothesis indices differ")
    finite = all(np.isfinite(h[s][0]).all() and np.isfinite(np.asarray(h[s][1])).all() for s in range(8))
    return bad, finite


def main():
    total = nbad = nonfinite = 0
    for man in (abi.EUCLID1, abi.EUCLID2, abi.EUCLID3, abi.CIRCULAR, abi.SE2):
        for N in (64, 200):
            for count in (N, 2, 3, 5, 17, 63, 65):
                if count > N:
                    continue
                for name in ("identical", "two far clusters", "huge offset", "tiny spread", "three values", "plain") + (("one NaN particle",) if NAN else ()):
                    bad, finite = run_case(man, N, count, name, 7 * man + count)
                    total += 1; nbad += bool(bad); nonfinite += not finite
                    if bad or not finite:
                        print(f"manifold {man} N={N} points held {count} '{name}': {'finite' if finite else 'NON-FINITE on the device'}; " + "; ".join(bad[:3])[:400], flush=True)
    print(f"fuzz_degenerate: {total - nbad} of {total} cases bit-identical to the oracle ({nbad} differ; {nonfinite} with non-finite values on the device)")


if __name__ == "__main__":
    main()
