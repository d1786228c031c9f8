"""shader-clock split of one manifold product (last workgroup of the launch); needs tools/libnbp_dbg.so built
with -DNBP_PHASE_TIMING"""
import ctypes as C
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, product_desc, rand_points

lib = abi.load_library(os.path.join(R, "tools", "libnbp_dbg.so"))
abi._lib = lib
names = ["staging + level bookkeeping", "node statistics", "final draw", "draw: conditional moments + uniform", "draw: pass 1 node weights",
         "draw: shuffle combine", "draw: pass 2 rescan + broadcast"]
for man, F in ((abi.EUCLID2, 2), (abi.EUCLID2, 3), (abi.SE2, 3)):
    for nops in (1, 2048):
        N = 200
        be = iif.HipBackend(N, F + 1, 0)
        rng = np.random.default_rng(0)
        for j in range(F):
            be.slot_write(j, man, rand_points(rng, man, N, 1.0 + 0.1 * j, 0.3))
        be.run_bandwidth(list(range(F)), [man] * F)
        descs = [product_desc(man, list(range(F)), F, 5 + i) for i in range(nops)]
        out = (C.c_longlong * 64)()
        be.run_products(descs)
        lib.nbp_debug_phase_read(out, 64, 1)
        be.run_products(descs)
        lib.nbp_debug_phase_read(out, 64, 1)
        extra = {47: "samplePoint normals", 48: "point moments", 56: "pass 1 sweep", 57: "pass 1 on the point", 58: "pass 1 sweep (leaf)",
                 59: "pass 1 on the point (leaf)", 60: "pass 2 sweep", 61: "pass 2 on the point", 62: "pass 2 sweep (leaf)", 63: "pass 2 on the point (leaf)"}
        tot = sum(out[40:49]) + sum(out[56:64])
        print(f"manifold {man} F={F} batch {nops}: {tot} cycles | " + ", ".join(f"{n} {out[40 + i]}" for i, n in enumerate(names))
              + " | " + ", ".join(f"{n} {out[i]}" for i, n in extra.items()))
        be.close()
