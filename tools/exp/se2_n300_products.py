"""SE(2) products at N = 300 in the throughput geometries: finite? equal to the oracle's?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, product_desc, rand_points
from oracle.oracle_backend import OracleBackend

def run(make, N, man, F, nprod, nsrc=16):
    be = make(N, nsrc + nprod)
    rng = np.random.default_rng(1)
    for j in range(nsrc):
        be.slot_write(j, man, rand_points(rng, man, N, 0.2 * j, 0.3))
    be.run_bandwidth(list(range(nsrc)), [man] * nsrc)
    descs = [product_desc(man, [(3 * i + j) % nsrc for j in range(F)], nsrc + i, 5 + i) for i in range(nprod)]
    be.run_products(descs)
    out = [be.slot_read(nsrc + i, man)[0] for i in range(min(nprod, 24))]
    be.close()
    return np.array(out)

for man, name in ((abi.SE2, "SE(2)"), (abi.EUCLID3, "Euclid(3)"), (abi.CIRCULAR, "Circular")):
    for N in (300, 257, 320, 256, 200):
        for F in (2, 3, 4):
            for nprod in (332, 200, 100, 20):
                for env in ({}, {"NBP_PRODUCT_HL2_MIN": "100000"}):
                    for k in ("NBP_PRODUCT_HL2_MIN",): os.environ.pop(k, None)
                    os.environ.update(env)
                    d = run(lambda n, s: iif.HipBackend(n, s, 0), N, man, F, nprod)
                    fin = np.isfinite(d).all()
                    line = f"{name} N={N} F={F} {nprod} products {'HL4 forced' if env else 'default   '}: finite {fin}"
                    if not env:
                        o = run(lambda n, s: OracleBackend(n, s, 0, threads=16), N, man, F, min(nprod, 24))
                        if o is not None:
                            line += f"  max |device - oracle| {np.nanmax(np.abs(d - o)):.2e}"
                    print(line, flush=True)
