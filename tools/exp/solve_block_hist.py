"""durations of the proposal workgroups of config-2 solves, by grid-size class (debug build: tools/libnbp_dbg.so)"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["NBP_LIB_OVERRIDE"] = os.path.join(R, "tools", "libnbp_dbg.so")
sys.path.insert(0, R)
import numpy as np
import iif_amd_loader
iif = iif_amd_loader.load()
from bench_support import workloads, RankSolve
lib = iif.abi.load_library()
cfg = sys.argv[1] if len(sys.argv) > 1 else "2"
wl = workloads(iif)[cfg]
run = RankSolve(iif, wl, wl.size, wl.N, 0, 1, 0, None)
run.prepare()
for k in range(2): run.step(k)
out = (C.c_uint * 256)()
lib.nbp_debug_block_hist(out, 1)
for k in range(3): run.step(2 + k)
run.be.synchronize()
lib.nbp_debug_block_hist(out, 1)
h = np.array(out[:], dtype=np.int64).reshape(4, 64)
for g, name in enumerate(("<=8", "<=64", "<=300", ">300")):
    n = h[g].sum()
    if not n: continue
    nz = np.nonzero(h[g])[0]
    print(f"grid {name}: {n // 3} workgroups per solve; duration bins of 16 us:", {int(16 * b): int(h[g][b]) for b in nz})
