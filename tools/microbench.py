"""GPU micro-benchmark of the two kernels in isolation (HIP events inside libnbp)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from parity_utils import abi, iif, rand_points, relative_factor_desc, product_desc

def run(N, manifold, nops, F, what):
    be = iif.HipBackend(N, 4 + F + 2, 0)
    rng = np.random.default_rng(0)
    D = abi.MANIFOLD_DIM[manifold]
    be.slot_write(0, manifold, rand_points(rng, manifold, N, 0.0, 0.5))
    be.slot_write(1, manifold, rand_points(rng, manifold, N, 1.0, 0.5))
    for j in range(F):
        be.slot_write(2 + j, manifold, rand_points(rng, manifold, N, 0.1 * j, 0.5), np.full(D, 0.15))
    kind = {abi.EUCLID1: abi.F_LINREL, abi.EUCLID2: abi.F_LINREL, abi.EUCLID3: abi.F_LINREL, abi.CIRCULAR: abi.F_CIRCULAR, abi.SE2: abi.F_SE2}[manifold]
    out = 2 + F
    if what.startswith("prop"):
        descs = []
        for i in range(nops):
            d = relative_factor_desc(kind, manifold, 2, 1, [0, 1], out, 100 + i, [1.0, 0.5, 0.1][:D] if kind != abi.F_CIRCULAR else [0.3], [0.1] * (D if kind == abi.F_LINREL else (3 if kind == abi.F_SE2 else 1)))
            d.skip_bandwidth = 1 if what == "prop_nobw" else 0
            if what == "prop_nosolve":
                d.inflate_cycles = 0
            descs.append(d)
        stage = (abi.STAGE_PROPOSALS, descs)
    else:
        stage = (abi.STAGE_PRODUCTS, [product_desc(manifold, list(range(2, 2 + F)), out, 7 + i) for i in range(nops)])
    prog = be.program([stage])
    prog.run(); be.synchronize()
    be.timing_enable(True); be.timing_read()
    for _ in range(3):
        prog.run()
    be.synchronize()
    t = be.timing_read()
    ms = sum(v[0] for v in t.values()) / 3
    be.close()
    return ms

if __name__ == "__main__":
    for (N, man) in [(200, abi.EUCLID2), (200, abi.SE2), (200, abi.CIRCULAR), (300, abi.EUCLID3), (100, abi.EUCLID1)]:
        for nops in (1, 256, 2048):
            r = {w: run(N, man, nops, 2, w) for w in ("prop", "prop_nobw", "prop_nosolve")}
            p2 = run(N, man, nops, 2, "prod")
            p3 = run(N, man, nops, 3, "prod")
            print(f"N={N} man={man} nops={nops}: proposal {r['prop']:.3f} ms (no-bw {r['prop_nobw']:.3f}, no-solve {r['prop_nosolve']:.3f}) | product F=2 {p2:.3f} F=3 {p3:.3f}", flush=True)
