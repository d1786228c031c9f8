"""Differential fuzz of the product kernels against the oracle, BIT FOR BIT: launches that mix manifolds (the any-manifold
kernels) or hold one (the single-manifold ones), density counts 1 .. 7 (and a 20 / 60 now and then), Niter 1 .. 3, partial input
densities with old points, input beliefs that are tight / wide / far from the origin / all the way round the circle, with
bandwidths from their own fits; launches of 1 / 12 / 90 / 700 products (every geometry: helper lanes per sample, throughput rows),
N = 64 / 200 / 257 / 300.  Points (raw rows), bandwidths, infoPerCoord and -- where asked for -- the labels compared with np.array_equal.
usage (GPU box): fuzz_products.py [seeds=4] [first seed=0]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, rand_points
from oracle.oracle_backend import OracleBackend

DUP = os.environ.get("FUZZ_DUP", "0") != "0"  # FUZZ_DUP=1: four source beliefs in ten hold repeated points
MANS = [abi.EUCLID1, abi.EUCLID2, abi.EUCLID3, abi.CIRCULAR, abi.SE2]


def run_launch(seed, N, B, man_fixed):
    rng = np.random.default_rng(seed)
    nsrc = 24
    src = {}   # manifold -> slots of its source beliefs
    writes, s = [], 0
    mans = MANS if man_fixed is None else [man_fixed]
    for man in mans:
        src[man] = list(range(s, s + nsrc))
        base = float(rng.choice([0.0, 50.0, -1e3]))
        for j in range(nsrc):
            spread = float(rng.choice([0.02, 0.3, 1.5]))
            c = (base + 0.2 * j) if man != abi.CIRCULAR else float(rng.uniform(-3, 3))
            pts = rand_points(rng, man, N, c, spread)
            if DUP and rng.random() < 0.4:  # a belief with REPEATED points (a multinomial resampling without noise): equal keys in the KD sort
                k = int(rng.integers(1, max(2, N // 2)))
                pts[rng.choice(N, size=k, replace=False)] = pts[rng.choice(N, size=k)]
            writes.append((s + j, man, pts))
        s += nsrc
    n_src = s
    descs, outs, side = [], [], 0
    for i in range(B):
        man = mans[int(rng.integers(0, len(mans)))]
        D = abi.MANIFOLD_DIM[man]
        F = int(rng.choice([1, 2, 2, 2, 3, 3, 4, 5, 7])) if rng.random() > 0.03 else int(rng.choice([20, 60]))
        ins = [int(x) for x in rng.choice(src[man], size=F, replace=F > nsrc)]
        partials, old = None, -1
        if D > 1 and F >= 2 and rng.random() < 0.2:
            partials = [int(rng.integers(1, 1 << D)) if rng.random() < 0.5 else 0 for _ in range(F)]
            old = int(rng.choice(src[man]))
        want_labels = F <= 4 and rng.random() < 0.3
        d = iif.solver.product_desc(man, ins, n_src + i, int(rng.integers(1, 2**31)), int(rng.choice([1, 1, 2, 3])),
                                    side if want_labels else -1, partials, old)
        if want_labels:
            side += N * F
        descs.append(d); outs.append((n_src + i, man, F, d, (d.labels_out, N * F) if want_labels else None))
    res = []
    for make in (lambda: OracleBackend(N, n_src + B, max(side, 1), threads=48), lambda: iif.HipBackend(N, n_src + B, side_ints=max(side, 1))):
        be = make()
        try:
            for sl, man, pts in writes:
                be.slot_write(sl, man, pts)
            for man in mans:
                be.run_bandwidth(src[man], [man] * nsrc)
            be.run_products(descs)
            res.append(([be.slot_read(o, abi.EUCLID3) for o, _, _, _, _ in outs], be.side_read(0, max(side, 1)), [be.belief_read(o, m) for o, m, _, _, _ in outs] if hasattr(be, "belief_read") else None))
        finally:
            be.close()
    (po, so, _), (ph, sh, _) = res
    bad = []
    for j, (o, man, F, d, lab) in enumerate(outs):
        same = np.array_equal(po[j][0], ph[j][0]) and np.array_equal(np.asarray(po[j][1]), np.asarray(ph[j][1]))
        if lab is not None:
            same = same and np.array_equal(so[lab[0]:lab[0] + lab[1]], sh[lab[0]:lab[0] + lab[1]])
        if not same:
            dp = np.abs(po[j][0] - ph[j][0])
            bad.append(f"    product {j} manifold {man} F {F} niter {d.niter} partial {'yes' if d.old_slot >= 0 else 'no'}: {int((dp > 0).any(axis=1).sum())} of {N} samples differ "
                       f"(max {np.nanmax(dp):.3e}), bandwidths by {np.abs(np.asarray(po[j][1]) - np.asarray(ph[j][1])).max():.3e}"
                       f"{'' if lab is None else ', labels differ: %d' % int((so[lab[0]:lab[0] + lab[1]] != sh[lab[0]:lab[0] + lab[1]]).sum())}"
                       f"{', non-finite on the device' if not np.isfinite(ph[j][0]).all() else ''}")
    return len(outs), bad


NS = [int(x) for x in os.environ.get("FUZZ_NS", "64,200,257,300").split(",")]  # particle counts by seed (FUZZ_NS=37,129,333,512: others)


def main():
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    total = nbad = 0
    for seed in range(first, first + nseeds):
        N = NS[seed % len(NS)]
        plans = [("mixed manifolds, 90 products", 90, None), ("mixed manifolds, 12 products", 12, None), ("a lone product", 1, None)]
        plans += [(f"manifold {m}, 90 products", 90, m) for m in MANS]
        plans.append((f"manifold {MANS[1 + seed % 2]}, 700 products", 700, MANS[1 + seed % 2]))
        plans.append((f"manifold {MANS[4 - (seed % 2) * 1]}, 400 products", 400, MANS[4 - (seed % 2)]))
        for what, B, man in plans:
            n, bad = run_launch(7000 * seed + B + (0 if man is None else man), N, B, man)
            total += n; nbad += len(bad)
            print(f"seed {seed} N={N}: {what}: {n - len(bad)} of {n} outputs bit-identical", flush=True)
            for b in bad[:6]:
                print(b, flush=True)
    print(f"fuzz_products: {total - nbad} of {total} product outputs bit-identical to the oracle ({nbad} differ)")


if __name__ == "__main__":
    main()
