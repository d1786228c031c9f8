"""wall-clock split of ONE workgroup of the fused update kernel inside a launch of `n` updates (debug build:
tools/build_lib.py --single -DNBP_PHASE_TIMING -mllvm -disable-machine-licm -o tools/libnbp_dbg.so)
usage: fused_phase.py [n=488] [F=2]"""
import ctypes as C
import os
import sys

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
os.environ["NBP_LIB_OVERRIDE"] = os.path.join(R, "tools", "libnbp_dbg.so")
os.environ["NBP_FUSED_MIN"] = "1"
import iif_amd_loader  # noqa: E402
iif = iif_amd_loader.load()
from iif_amd import abi  # noqa: E402
from parity_utils import rand_points  # noqa: E402
import test_gpu_fused_update as t  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 488
F = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lib = abi.load_library()
lib.nbp_debug_phase_read.argtypes = [C.POINTER(C.c_longlong), C.c_int, C.c_int]
rng = np.random.default_rng(5)
props, prods, stride = t._round(n, F)
for fused in (True, False):
    be = iif.HipBackend(200, 4 + stride * n, side_ints=0)
    for s, c in enumerate((0.0, 2.0, 1.0)):
        be.slot_write(s, abi.EUCLID2, rand_points(rng, abi.EUCLID2, 200, c, 0.4))
    prog = be.program([(abi.STAGE_PROPOSALS, props), (abi.STAGE_PRODUCTS, prods)], lazy_bandwidth=False, fused_updates=fused)
    prog.run(); be.synchronize()
    buf = (C.c_longlong * 64)()
    lib.nbp_debug_phase_read(buf, 64, 1)
    be.timing_enable(True); be.timing_read()
    prog.run(); be.synchronize()
    tim = be.timing_read()
    lib.nbp_debug_phase_read(buf, 64, 1)
    print(f"n = {n}, F = {F}, fused = {fused}: " + ", ".join(f"{k.replace('nbp_', '').replace('_kernel', '')} {v[0] * 1e3:.0f} us" for k, v in tim.items() if v[1]))
    if fused:
        names = ["proposals", "proposal fits", "KD builds", "product", "result fit"]
        print("   workgroup 0 (us): " + ", ".join(f"{nm} {buf[50 + i] / 100:.0f}" for i, nm in enumerate(names)))
    prog.close(); be.close()
