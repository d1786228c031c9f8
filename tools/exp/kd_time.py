"""how long the KD builds of a chip-filling product stage take on their own (prep launch without pending fits)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, product_desc, rand_points
N, F = 200, 3
for nprod in (1, 16, 256, 975, 2000):
    man = abi.EUCLID2
    be = iif.HipBackend(N, 64 + nprod, 0)
    rng = np.random.default_rng(0)
    for j in range(64):
        be.slot_write(j, man, rand_points(rng, man, N, 1.0 + 0.1 * j, 0.3))
    be.run_bandwidth(list(range(64)), [man] * 64)
    descs = [product_desc(man, [(3 * i + j) % 64 for j in range(F)], 64 + i, 5 + i) for i in range(nprod)]
    be.run_products(descs)
    be.timing_enable(True); be.timing_read()
    for _ in range(3): be.run_products(descs)
    t = be.timing_read()
    print(nprod, "products F=3:", {k: round(v[0] / 3 * 1e3, 1) for k, v in t.items() if v[0] > 0}, "us")
    be.close()
