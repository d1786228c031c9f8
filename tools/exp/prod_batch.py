"""a chip-filling batch of Euclid(2) products of F densities (for counter passes)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, product_desc, rand_points
N = 200
nprod = int(sys.argv[1]) if len(sys.argv) > 1 else 975
F = int(sys.argv[2]) if len(sys.argv) > 2 else 2
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
print(nprod, f"products F={F}:", {k: round(v[0] / 3 * 1e3, 1) for k, v in t.items() if v[0] > 0}, "us")
be.close()
