"""products of a MIXED launch (two-, three- and four-density products side by side, as a tree level of an SE(2) lattice has them)
against the oracle, in the throughput geometries, by manifold and particle count"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, product_desc, rand_points
from oracle.oracle_backend import OracleBackend

def run(make, N, man, Fs, nsrc=16, keep=None):
    nprod = len(Fs)
    be = make(N, nsrc + nprod)
    rng = np.random.default_rng(1)
    for j in range(nsrc):
        be.slot_write(j, man, rand_points(rng, man, N, 0.2 * j, 0.3))
    be.run_bandwidth(list(range(nsrc)), [man] * nsrc)
    descs = [product_desc(man, [(3 * i + j) % nsrc for j in range(Fs[i])], nsrc + i, 5 + i) for i in range(nprod)]
    if keep is not None:  # the oracle: only the products that are compared (each is independent of the others)
        descs = [descs[i] for i in keep]
    be.run_products(descs)
    idx = keep if keep is not None else range(nprod)
    out = {i: be.slot_read(nsrc + i, man)[0] for i in idx}
    be.close()
    return out

bad = 0
for man, name in ((abi.SE2, "SE(2)"), (abi.EUCLID3, "Euclid(3)"), (abi.EUCLID2, "Euclid(2)"), (abi.CIRCULAR, "Circular")):
    for N in (300, 200, 256, 320):
        for nprod in (332, 120):
            Fs = [4 if i % 9 == 8 else (3 if i % 17 == 3 else 2) for i in range(nprod)]
            keep = [i for i in range(nprod) if Fs[i] > 2][:8] + list(range(6))
            d = run(lambda n, s: iif.HipBackend(n, s, 0), N, man, Fs)
            o = run(lambda n, s: OracleBackend(n, s, 0, threads=16), N, man, Fs, keep=keep)
            worst = {F: max([float(np.nanmax(np.abs(d[i] - o[i]))) for i in keep if Fs[i] == F] or [0]) for F in (2, 3, 4)}
            fin = all(np.isfinite(d[i]).all() for i in d)
            flag = "" if fin and max(worst.values()) < 1e-8 else "   <-- DIFFERS"
            bad += bool(flag)
            print(f"{name} N={N} {nprod} products (2/3/4 densities mixed): finite {fin}, max |device - oracle| by density count {worst}{flag}", flush=True)
print("launches that differ:", bad)
