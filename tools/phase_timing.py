"""per-phase wall clock (100 MHz) of ONE product block; needs tools/libnbp_dbg.so (-DNBP_PHASE_TIMING)"""
import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, rand_points, product_desc
lib = abi.load_library(os.path.join(R, "tools", "libnbp_dbg.so"))
abi._lib = lib
names = ["-", "-", "stage ws->LDS", "level stats", "pass1 (to barrier)", "barrier after pass2", "final draw", "combine(w0)", "barrier after combine", "pass2(w0)"]
for (N, man, F) in [(200, abi.EUCLID2, 2), (200, abi.EUCLID2, 3), (200, abi.SE2, 3), (300, abi.EUCLID3, 2)]:
    be = iif.HipBackend(N, F + 2, 0)
    rng = np.random.default_rng(0); D = abi.MANIFOLD_DIM[man]
    for j in range(F):
        be.slot_write(j, man, rand_points(rng, man, N, 0.1 * j, 0.5), np.full(D, 0.15))
    d = product_desc(man, list(range(F)), F, 7)
    be.run_products([d])
    out = (C.c_longlong * 64)()
    lib.nbp_debug_phase_read(out, 64, 1)
    for _ in range(5):
        be.run_products([d])
    lib.nbp_debug_phase_read(out, 64, 1)
    tot = sum(out[:10])
    print(f"N={N} man={man} F={F}: total {tot / 5 / 100:.1f} us | " + ", ".join(f"{n} {out[i] / 5 / 100:.1f}" for i, n in enumerate(names)))
    be.close()
