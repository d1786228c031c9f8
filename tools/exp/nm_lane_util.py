"""lane utilisation of the Nelder-Mead / BFGS searches of whole solves: sum over lanes of residual evaluations against
64 x the slowest lane of each wave (debug build, -DNBP_PHASE_TIMING).  usage: nm_lane_util.py [config ...]"""
import ctypes as C
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
os.environ["NBP_LIB_OVERRIDE"] = os.path.join(R, "tools", "libnbp_dbg.so")
import iif_amd_loader  # noqa: E402
iif = iif_amd_loader.load()
from iif_amd import abi  # noqa: E402
from bench_support import RankSolve, workloads  # noqa: E402

lib = abi.load_library()
lib.nbp_debug_phase_read.argtypes = [C.POINTER(C.c_longlong), C.c_int, C.c_int]
buf = (C.c_longlong * 64)()
for cfg in (sys.argv[1:] or ["2", "3", "4", "5"]):
    wl = workloads(iif)[cfg]
    size = {"2": 1000, "3": 2000, "4": 20, "5": 3000}[cfg]
    rs = RankSolve(iif, wl, size, wl.N, 0, 1, 0, None)
    rs.prepare()
    lib.nbp_debug_phase_read(buf, 64, 1)
    rs.step(0)
    rs.be.synchronize()
    lib.nbp_debug_phase_read(buf, 64, 1)
    print(f"config {cfg} (size {size}): lanes busy {buf[60] / max(buf[61], 1):.1%} of the wave-time of the searches of one solve")
    rs.close()
