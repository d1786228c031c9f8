"""shader-clock split of one LCV evaluation (last workgroup of the launch); needs tools/libnbp_dbg.so
built with -DNBP_PHASE_TIMING"""
import ctypes as C
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, rand_points

lib = abi.load_library(os.path.join(R, "tools", "libnbp_dbg.so"))
abi._lib = lib
names = ["pair loop", "barrier 1", "combine", "barrier 2", "log+reduce", "barrier 3"]
for N in (200, 256):
    for nfits in (1, 4096):
        be = iif.HipBackend(N, nfits, 0)
        rng = np.random.default_rng(0)
        for s in range(min(nfits, 64)):
            be.slot_write(s, abi.EUCLID2, rand_points(rng, abi.EUCLID2, N, 0.0, 0.5))
        if nfits > 64:
            be.run_copies([abi.CopyDesc(s % 64, s) for s in range(64, nfits)])
        sl, mn = list(range(nfits)), [abi.EUCLID2] * nfits
        be.run_bandwidth(sl, mn)
        out = (C.c_longlong * 64)()
        lib.nbp_debug_phase_read(out, 64, 1)
        be.diag(reset=True)
        be.run_bandwidth(sl, mn)
        lib.nbp_debug_phase_read(out, 64, 1)
        ev = be.diag()["lcv_evals"] / (2 * nfits)  # evaluations of one coordinate fit
        tot = sum(out[20:26])
        print(f"N={N} fits={nfits}: {ev:.1f} evals/fit, {tot / ev:.0f} cycles per evaluation | " +
              ", ".join(f"{n} {out[20 + i] / ev:.0f}" for i, n in enumerate(names)))
        be.close()
