"""launches that mix partial and full products, density counts and (Euclid(2) / Euclid(3) / SE(2)) particle counts -- incl. the launches whose
node statistics live in global memory -- against the oracle; labels too"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, rand_points
from oracle.oracle_backend import OracleBackend

def run(make, N, man, specs, keep=None, nsrc=24):
    nprod = len(specs)
    be = make(N, nsrc + 1 + nprod, nprod * N * 12 + 16)
    rng = np.random.default_rng(2)
    for j in range(nsrc + 1):
        be.slot_write(j, man, rand_points(rng, man, N, 0.15 * j, 0.3))
    be.run_bandwidth(list(range(nsrc + 1)), [man] * (nsrc + 1))
    descs = []
    for i, (F, parts, lab) in enumerate(specs):
        descs.append(iif.solver.product_desc(man, [(5 * i + j) % nsrc for j in range(F)], nsrc + 1 + i, 77 + i, 1, (i * N * 12) if lab else -1,
                                             partials=parts, old_slot=nsrc))
    idx = list(range(nprod)) if keep is None else keep
    be.run_products([descs[i] for i in idx])
    out = {i: be.slot_read(nsrc + 1 + i, man)[0] for i in idx}
    labs = {i: np.array(be.side_read(i * N * 12, N * specs[i][0])) for i in idx if specs[i][2]}
    be.close()
    return out, labs

bad = 0
D = {abi.EUCLID2: 2, abi.EUCLID3: 3, abi.SE2: 3}
for man, name in ((abi.SE2, "SE(2)"), (abi.EUCLID3, "Euclid(3)"), (abi.EUCLID2, "Euclid(2)")):
    full = (1 << D[man]) - 1
    for N in (200, 300, 320, 100):
        for nprod in (250, 90, 12):
            specs = []
            for i in range(nprod):
                F = 4 if i % 9 == 8 else (3 if i % 5 == 3 else (9 if i % 31 == 30 else 2))
                parts = None
                if i % 4 == 1:  # a partial input among full ones; every coordinate informed by some density or by oldPoints
                    parts = [0] * F
                    parts[0] = 1 if i % 8 == 1 else (full & ~1)
                if i % 12 == 7:  # every input partial on the same coordinates: the others keep the old points
                    parts = [1] * F
                specs.append((F, parts, i % 3 == 0))
            keep = sorted(set(list(range(8)) + [i for i, s in enumerate(specs) if s[0] > 3][:5] + [i for i, s in enumerate(specs) if s[1]][:6]))
            try:
                d, dl = run(lambda n, s, si: iif.HipBackend(n, s, si), N, man, specs)
                o, ol = run(lambda n, s, si: OracleBackend(n, s, si, threads=32), N, man, specs, keep=keep)
                w = max(float(np.nanmax(np.abs(d[i] - o[i]))) for i in keep)
                lab_ok = all(np.array_equal(dl[i], ol[i]) for i in ol)
                fin = all(np.isfinite(v).all() for v in d.values())
                flag = "" if fin and w < 1e-8 and lab_ok else "   <-- DIFFERS"
            except Exception as e:  # noqa: BLE001
                w, fin, lab_ok, flag = float("nan"), False, False, f"   <-- ERROR {str(e)[:200]}"
            bad += bool(flag)
            print(f"{name} N={N} {nprod} products (2 / 3 / 4 / 9 densities, partial and full, labels on a third): finite {fin}, labels equal {lab_ok}, max |device - oracle| {w:.2e}{flag}", flush=True)
print("launches that differ:", bad)
