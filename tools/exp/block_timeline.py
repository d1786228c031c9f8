"""begin / end of every workgroup of one chip-filling proposal launch (needs tools/libnbp_dbg.so, -DNBP_PHASE_TIMING)"""
import ctypes as C, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, rand_points, relative_factor_desc
lib = abi.load_library(os.path.join(R, "tools", "libnbp_dbg.so")); abi._lib = lib
N, B = 200, int(sys.argv[1]) if len(sys.argv) > 1 else 975
be = iif.HipBackend(N, 2 * B + 2, 0)
rng = np.random.default_rng(0)
for j in range(B + 1):
    be.slot_write(j, abi.EUCLID2, rand_points(rng, abi.EUCLID2, N, 3.0 + j, 0.3))
descs = []
for j in range(B):
    d = relative_factor_desc(abi.F_LINREL, abi.EUCLID2, 2, 1, [j, j + 1], B + 1 + j, 5 + j, [1.0, 0.0], [0.1, 0.1]); d.skip_bandwidth = 1
    descs.append(d)
for _ in range(3): be.run_proposals(descs)
buf = (C.c_longlong * (3 * B))()
lib.nbp_debug_block_read(buf, B)
a = np.array(buf[:], dtype=np.int64).reshape(B, 3)
t0 = a[:, 0].min()
beg, end = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0   # us (100 MHz)
dur = end - beg
print(f"{B} workgroups: launch span {end.max():.1f} us; begin: median {np.median(beg):.1f} p90 {np.percentile(beg, 90):.1f} max {beg.max():.1f}")
print(f"duration: min {dur.min():.1f} median {np.median(dur):.1f} p90 {np.percentile(dur, 90):.1f} p99 {np.percentile(dur, 99):.1f} max {dur.max():.1f} us")
hw = a[:, 2]
cu = (hw >> 8) & 0xF; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
print("begin histogram (20 us bins):", np.histogram(beg, bins=np.arange(0, end.max() + 20, 20))[0].tolist())
print("end histogram (20 us bins):  ", np.histogram(end, bins=np.arange(0, end.max() + 20, 20))[0].tolist())
late = np.argsort(-end)[:8]
print("last to finish (block, begin, dur):", [(int(i), round(float(beg[i]), 1), round(float(dur[i]), 1)) for i in late])
