"""shader-clock split of a LONE manifold product on the circle (config 3's critical path near the root); needs tools/libnbp_dbg.so"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, product_desc, rand_points
lib = abi.load_library(os.path.join(R, "tools", "libnbp_dbg.so")); abi._lib = lib
names = {40: "staging + bookkeeping", 41: "node statistics", 42: "final draw", 43: "uniforms of the pass", 44: "conditional moments", 45: "shuffle combine",
         47: "samplePoint normals", 48: "point moments", 56: "pass 1 sweep", 57: "pass 1 on the point", 58: "pass 1 sweep (leaf)", 59: "pass 1 on the point (leaf)",
         60: "pass 2 sweep", 61: "pass 2 on the point", 62: "pass 2 sweep (leaf)", 63: "pass 2 on the point (leaf)"}
for man in (abi.CIRCULAR, abi.EUCLID2):
    for F in (2, 3, 4):
        N = 200
        be = iif.HipBackend(N, F + 1, 0)
        rng = np.random.default_rng(0)
        for j in range(F): be.slot_write(j, man, rand_points(rng, man, N, 0.3 * j, 0.3))
        be.run_bandwidth(list(range(F)), [man] * F)
        descs = [product_desc(man, list(range(F)), F, 5)]
        out = (C.c_longlong * 64)()
        be.run_products(descs); lib.nbp_debug_phase_read(out, 64, 1)
        be.run_products(descs); lib.nbp_debug_phase_read(out, 64, 1)
        tot = sum(out[k] for k in names)
        print(f"manifold {man} F={F} lone: {tot} cycles | " + ", ".join(f"{n} {out[k]} ({100 * out[k] // max(tot, 1)}%)" for k, n in names.items() if out[k]))
        be.close()
