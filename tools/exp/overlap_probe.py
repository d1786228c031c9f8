"""do a chip-filling proposal launch and a chip-filling fit + product launch share the chip well?  Two contexts (two
streams); each program alone, then both enqueued together.  usage: overlap_probe.py [n=488] [F=2] [reps=20]"""
import os
import sys
import time

import numpy as np

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
import iif_amd_loader  # noqa: E402
iif = iif_amd_loader.load()
from iif_amd import abi  # noqa: E402
from parity_utils import rand_points  # noqa: E402
import test_gpu_fused_update as t  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 488
F = int(sys.argv[2]) if len(sys.argv) > 2 else 2
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rng = np.random.default_rng(5)
props, prods, stride = t._round(n, F)


def mk():
    be = iif.HipBackend(200, 4 + stride * n, side_ints=0)
    for s, c in enumerate((0.0, 2.0, 1.0)):
        be.slot_write(s, abi.EUCLID2, rand_points(rng, abi.EUCLID2, 200, c, 0.4))
    return be


a, b = mk(), mk()
pa = a.program([(abi.STAGE_PROPOSALS, props)] * reps, lazy_bandwidth=False)
b.program([(abi.STAGE_PROPOSALS, props)], lazy_bandwidth=False).run()
b.synchronize()
pb = b.program([(abi.STAGE_PRODUCTS, prods)] * reps, lazy_bandwidth=False)


def timed(fs):
    for be in (a, b):
        be.synchronize()
    t0 = time.perf_counter()
    for f in fs:
        f()
    for be in (a, b):
        be.synchronize()
    return (time.perf_counter() - t0) * 1e3


for _ in range(2):
    timed([pa.run, pb.run])
ta = min(timed([pa.run]) for _ in range(3))
tb = min(timed([pb.run]) for _ in range(3))
tab = min(timed([pa.run, pb.run]) for _ in range(3))
print(f"n = {n} updates, F = {F}, {reps} launches each: proposals alone {ta:.2f} ms, fits + products alone {tb:.2f} ms, "
      f"sum {ta + tb:.2f} ms, both streams together {tab:.2f} ms ({tab / (ta + tb):.2f} of the sum)")
