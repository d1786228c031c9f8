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
CASES = ((abi.EUCLID2, abi.F_LINREL, [1.0, 1.0], [0.1, 0.1], 0.3), (abi.SE2, abi.F_SE2, [1.0, 0.0, 0.1], [0.1, 0.1, 0.01], 0.3),
         (abi.CIRCULAR, abi.F_CIRCULAR, [0.5], [0.05], 0.3),   # one mode: the half-circle shortcut of the geodesic mean applies
         (abi.CIRCULAR, abi.F_CIRCULAR, [0.5], [0.05], 2.5))   # spread over the whole circle (the doors graph): the sequential walk
for man, kind, mean, sig, spread in CASES:
    for nops in (1, 2048):
        N = 200
        be = iif.HipBackend(N, 3, 0)
        rng = np.random.default_rng(0)
        be.slot_write(0, man, rand_points(rng, man, N, 1.0, spread))
        be.slot_write(1, man, rand_points(rng, man, N, 1.5, spread))
        descs = [relative_factor_desc(kind, man, 2, 1, [0, 1], 2, 5 + i, mean, sig) for i in range(nops)]
        for d in descs:
            d.skip_bandwidth = 1
        be.run_proposals(descs)
        out = (C.c_longlong * 64)()
        lib.nbp_debug_phase_read(out, 64, 1)
        be.run_proposals(descs)
        lib.nbp_debug_phase_read(out, 64, 1)
        tot = sum(out[30:34])
        print(f"manifold {man} spread {spread} batch {nops}: {tot} cycles for 3 inflate cycles | " + ", ".join(f"{n} {out[30 + i]}" for i, n in enumerate(names)))
        be.close()
