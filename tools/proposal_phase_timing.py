"""shader-clock split of one relative proposal (last workgroup of the launch); needs tools/libnbp_dbg.so
built with -DNBP_PHASE_TIMING"""
import ctypes as C
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, rand_points, relative_factor_desc

lib = abi.load_library(os.path.join(R, "tools", "libnbp_dbg.so"))
abi._lib = lib
names = ["setup / between cycles", "spread statistics", "entropy + solve (lane 0)", "wait for slowest lane"]
for man, kind, mean, sig in ((abi.EUCLID2, abi.F_LINREL, [1.0, 1.0], [0.1, 0.1]), (abi.SE2, abi.F_SE2, [1.0, 0.0, 0.1], [0.1, 0.1, 0.01])):
    for nops in (1, 2048):
        N = 200
        be = iif.HipBackend(N, 3, 0)
        rng = np.random.default_rng(0)
        be.slot_write(0, man, rand_points(rng, man, N, 10.0, 0.3))
        be.slot_write(1, man, rand_points(rng, man, N, 11.0, 0.3))
        descs = [relative_factor_desc(kind, man, 2, 1, [0, 1], 2, 5 + i, mean, sig) for i in range(nops)]
        for d in descs:
            d.skip_bandwidth = 1
        be.run_proposals(descs)
        out = (C.c_longlong * 64)()
        lib.nbp_debug_phase_read(out, 64, 1)
        be.run_proposals(descs)
        lib.nbp_debug_phase_read(out, 64, 1)
        tot = sum(out[30:34])
        print(f"manifold {man} batch {nops}: {tot} cycles for 3 inflate cycles | " + ", ".join(f"{n} {out[30 + i]}" for i, n in enumerate(names)))
        be.close()
