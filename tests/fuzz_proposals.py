"""Differential fuzz of the proposal kernels against the oracle, BIT FOR BIT: launches of random proposal descriptors -- factor
kinds x manifolds x the variable solved for x nullhypo x mixtures x multihypo (door sightings) x partial masks x inflation cycles
and spread x measurement noise from 1e-3 to 3 x beliefs centred at 0 / 100 / -1e4 with spreads from 1e-3 to 3 (on the circle: all
the way round, the lifted and the walked geodesic means), one input belief in seven with fewer points than the slot -- in MIXED launches (the generic kernel), in uniform ones of 1 / 40 /
1200 proposals (the single-manifold workgroup kernels, the one-wave-per-proposal kernels), at N = 64 / 200 / 257 / 300; every
output's points (raw rows), bandwidths and hypothesis indices compared with np.array_equal.
usage (GPU box): fuzz_proposals.py [seeds=6] [first seed=0]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, rand_points, relative_factor_desc
from oracle.oracle_backend import OracleBackend

SHORT = os.environ.get("FUZZ_SHORT", "1") != "0"  # some input beliefs with fewer points than N (FUZZ_SHORT=0: none)
KINDS = [("lin1", abi.F_LINREL, abi.EUCLID1), ("lin2", abi.F_LINREL, abi.EUCLID2), ("lin3", abi.F_LINREL, abi.EUCLID3),
         ("circ", abi.F_CIRCULAR, abi.CIRCULAR), ("se2", abi.F_SE2, abi.SE2), ("dist2", abi.F_EUCLIDDIST, abi.EUCLID2),
         ("dist3", abi.F_EUCLIDDIST, abi.EUCLID3), ("prior1", abi.F_PRIOR, abi.EUCLID1), ("prior2", abi.F_PRIOR, abi.EUCLID2),
         ("prior3", abi.F_PRIOR, abi.EUCLID3), ("priorc", abi.F_PRIOR, abi.CIRCULAR), ("priorse", abi.F_PRIOR, abi.SE2)]


def belief(rng, man, N):
    center = float(rng.choice([0.0, 1.0, 100.0, -1e4]))
    spread = float(rng.choice([1e-3, 0.05, 0.3, 3.0]))
    if man in (abi.CIRCULAR, abi.SE2):
        center = float(rng.uniform(-3, 3)) if man == abi.CIRCULAR else center
    pts = rand_points(rng, man, N, center, spread)
    if SHORT and rng.random() < 0.15:  # a belief that holds fewer points than the slot (a density of its own count: its bandwidth rides along)
        pts = pts[:int(rng.integers(max(8, N // 3), N))]
    return pts


def make_case(rng, name, kind, man, slots, out_slot, side_off, N, simple):
    """one random descriptor reading beliefs in `slots` (a list of free slot numbers it may use: returns those it used)"""
    D = abi.MANIFOLD_DIM[man]
    zd = {abi.F_LINREL: D, abi.F_CIRCULAR: 1, abi.F_SE2: 3, abi.F_EUCLIDDIST: 1, abi.F_PRIOR: D}[kind]
    sig = [float(rng.choice([1e-3, 0.05, 0.3, 3.0])) for _ in range(zd)]
    mean = [float(rng.normal()) for _ in range(zd)]
    if kind == abi.F_EUCLIDDIST:
        mean = [abs(mean[0]) + 0.5]
    kw = dict(cycles=int(rng.choice([1, 3, 5])), inflation=float(rng.choice([1.0, 5.0, 20.0])))
    if not simple:
        kw["nullhypo"] = float(rng.choice([0.0, 0.0, 0.1, 0.5]))
        if rng.random() < 0.3 and kind != abi.F_EUCLIDDIST:
            kw.update(ncomp=2, comps=[(0.7, mean, sig), (0.3, [m + 1.0 for m in mean], [s * 3 for s in sig])])
        if kind == abi.F_PRIOR and D > 1 and rng.random() < 0.3:
            kw["partial_mask"] = int(rng.integers(1, 1 << D))
        if kind == abi.F_LINREL and D > 1 and rng.random() < 0.25:
            kw["partial_mask"] = int(rng.choice([1, 2] if D == 2 else [1, 2, 4, 3, 5, 6]))
        if kind == abi.F_SE2 and rng.random() < 0.2:
            kw["partial_mask"] = int(rng.choice([3, 4, 1, 6]))
    kw["mhidx_out"] = side_off
    if kind == abi.F_PRIOR:
        pm = kw.get("partial_mask", 0)
        nz = bin(pm).count("1") if pm else D
        return relative_factor_desc(kind, man, 1, 0, [slots[0]], out_slot, int(rng.integers(1, 2**31)), mean[:nz], sig[:nz], **kw), 1
    if kw.get("partial_mask") and kind == abi.F_LINREL:
        nz = bin(kw["partial_mask"]).count("1")
        mean, sig = mean[:nz], sig[:nz]
    if not simple and kind in (abi.F_LINREL, abi.F_CIRCULAR) and not kw.get("partial_mask") and rng.random() < 0.25:
        # a sighting of one of two or three landmarks: [pose, l0, l1(, l2)], the pose certain
        nl = int(rng.choice([2, 3]))
        sf = int(rng.integers(0, nl + 1))
        return relative_factor_desc(kind, man, nl + 1, sf, slots[:nl + 1], out_slot, int(rng.integers(1, 2**31)), mean, sig,
                                    multihypo=[0.0] + [1.0 / nl] * nl, **kw), nl + 1
    return relative_factor_desc(kind, man, 2, int(rng.integers(0, 2)), slots[:2], out_slot, int(rng.integers(1, 2**31)), mean, sig, **kw), 2


def run_launch(seed, N, B, which, simple):
    rng = np.random.default_rng(seed)
    nslots = 5 * B + 4
    descs, writes, outs = [], [], []
    s = 0
    for j in range(B):
        name, kind, man = KINDS[int(rng.integers(0, len(KINDS)))] if which is None else which
        ins = list(range(s, s + 4))
        d, used = make_case(rng, name, kind, man, ins, s + 4, j * N, N, simple)
        for k in range(used):
            writes.append((ins[k], man, belief(rng, man, N)))
        descs.append(d); outs.append((s + 4, man, name, d))
        s += 5
    res = []
    for make in (lambda: OracleBackend(N, nslots, B * N, threads=32), lambda: iif.HipBackend(N, nslots, side_ints=B * N)):
        be = make()
        try:
            for sl, man, pts in writes:
                if pts.shape[0] == N:
                    be.slot_write(sl, man, pts)
                else:
                    be.belief_write(sl, man, pts, np.full(abi.MANIFOLD_DIM[man], 0.1))
            be.run_proposals(descs)
            res.append(([be.slot_read(o, abi.EUCLID3) for o, _, _, _ in outs], be.side_read(0, B * N)))
        finally:
            be.close()
    bad = []
    (po, so), (ph, sh) = res
    for j, (o, man, name, d) in enumerate(outs):
        same = np.array_equal(po[j][0], ph[j][0]) and np.array_equal(np.asarray(po[j][1]), np.asarray(ph[j][1])) and np.array_equal(so[j * N:(j + 1) * N], sh[j * N:(j + 1) * N])
        if not same:
            dp = np.abs(po[j][0] - ph[j][0])
            bad.append(f"    op {j} {name} nvars {d.nvars} sfidx {d.sfidx} nullhypo {d.nullhypo} ncomp {d.ncomp} partial {d.partial_mask} multihypo {d.has_multihypo} cycles {d.inflate_cycles} "
                       f"inflation {d.inflation}: {int((dp > 0).any(axis=1).sum())} of {N} particles differ (max {np.nanmax(dp):.3e}), bandwidths by "
                       f"{np.abs(np.asarray(po[j][1]) - np.asarray(ph[j][1])).max():.3e}, hypothesis indices {int((so[j * N:(j + 1) * N] != sh[j * N:(j + 1) * N]).sum())} differ"
                       f"{', NaN in the oracle' if not np.isfinite(po[j][0]).all() else ''}")
    return len(outs), bad


def run_deconv_launch(seed, N, B):
    """approxDeconv of B random relative factors (plain: the closed set of DeconvUtils.jl) + manikde! of the predicted
    measurements on the variable's manifold: predicted and sampled measurements (raw rows) and bandwidths, bit for bit"""
    rng = np.random.default_rng(seed)
    rel = [k for k in KINDS if k[1] != abi.F_PRIOR]
    descs, writes, outs, s = [], [], [], 0
    for j in range(B):
        name, kind, man = rel[int(rng.integers(0, len(rel)))]
        d, used = make_case(rng, name, kind, man, [s, s + 1, s + 2, s + 3], s + 4, -1, N, True)
        d.mhidx_out = -1
        for k in range(2):
            writes.append((s + k, man, belief(rng, man, N)))
        descs.append(d); outs.append((s + 4, s + 3, man, name, d))
        s += 5
    res = []
    for make in (lambda: OracleBackend(N, s + 1, 0, threads=32), lambda: iif.HipBackend(N, s + 1, side_ints=0)):
        be = make()
        try:
            for sl, man, pts in writes:
                if pts.shape[0] == N:
                    be.slot_write(sl, man, pts)
                else:
                    be.belief_write(sl, man, pts, np.full(abi.MANIFOLD_DIM[man], 0.1))
            be.run_deconv(descs, [m for _, m, _, _, _ in outs])
            be.run_bandwidth([o for o, _, _, _, _ in outs], [man for _, _, man, _, _ in outs])
            res.append([(be.slot_read(o, abi.EUCLID3), be.slot_read(m, abi.EUCLID3)[0]) for o, m, _, _, _ in outs])
        finally:
            be.close()
    bad = []
    for j, (o, m, man, name, d) in enumerate(outs):
        (po, bo), mo = res[0][j]
        (ph, bh), mh = res[1][j]
        if not (np.array_equal(po, ph) and np.array_equal(np.asarray(bo), np.asarray(bh)) and np.array_equal(mo, mh)):
            dp = np.abs(po - ph)
            bad.append(f"    deconv {j} {name} sfidx {d.sfidx}: {int((dp > 0).any(axis=1).sum())} of {N} predicted measurements differ (max {np.nanmax(dp):.3e}), "
                       f"bandwidths by {np.abs(np.asarray(bo) - np.asarray(bh)).max():.3e}, sampled measurements {'differ' if not np.array_equal(mo, mh) else 'equal'}")
    return len(outs), bad


NS = [int(x) for x in os.environ.get("FUZZ_NS", "64,200,257,300").split(",")]  # particle counts by seed (FUZZ_NS=37,129,333,512: others)


def main():
    nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    total = nbad = 0
    for seed in range(first, first + nseeds):
        N = NS[seed % len(NS)]
        plans = [("mixed launch of 60 (every kind, every option)", 60, None, False)]
        for w in (KINDS[1], KINDS[2], KINDS[3], KINDS[4]):
            plans.append((f"uniform {w[0]}, 40, every option", 40, w, False))
        plans.append((f"uniform {KINDS[1 + seed % 2][0]}, 1200 plain proposals (one wave per proposal)", 1200, KINDS[1 + seed % 2], True))
        plans.append((f"uniform se2, 400 plain proposals", 400, KINDS[4], True))
        plans.append(("a lone proposal", 1, None, False))
        for what, B, which, simple in plans:
            n, bad = run_launch(1000 * seed + B, N, B, which, simple)
            total += n; nbad += len(bad)
            print(f"seed {seed} N={N}: {what}: {n - len(bad)} of {n} outputs bit-identical", flush=True)
            for b in bad[:6]:
                print(b, flush=True)
        for B in (1, 120):
            n, bad = run_deconv_launch(77000 + 10 * seed + B, N, B)
            total += n; nbad += len(bad)
            print(f"seed {seed} N={N}: approxDeconv of {B} random relative factors + the fit of the predicted measurements: {n - len(bad)} of {n} outputs bit-identical", flush=True)
            for b in bad[:6]:
                print(b, flush=True)
    print(f"fuzz_proposals: {total - nbad} of {total} proposal outputs bit-identical to the oracle ({nbad} differ)")


if __name__ == "__main__":
    main()
