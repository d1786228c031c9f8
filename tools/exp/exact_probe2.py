import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, rand_points, relative_factor_desc, both
from oracle.oracle_backend import OracleBackend
for threads in (1, 8, 16):
    ob = lambda n, s, side_ints=0: OracleBackend(n, s, side_ints, threads=threads)
    hb = lambda n, s, side_ints=0: iif.HipBackend(n, s, side_ints=side_ints)
    N = 200
    kind, manifold, mean, sig = abi.F_LINREL, abi.EUCLID1, [1.0], [0.1]
    rng = np.random.default_rng(4000 + 10 * kind + manifold)
    a = rand_points(rng, manifold, N, center=0.0, spread=0.3)
    b = rand_points(rng, manifold, N, center=1.0, spread=0.3)
    d = relative_factor_desc(kind, manifold, 2, 1, [0, 1], 2, 777 + kind, mean, sig)
    def setup(be):
        be.slot_write(0, manifold, a); be.slot_write(1, manifold, b)
    def read(be):
        return be.slot_read(2, abi.EUCLID1)[0], be.slot_read(3, abi.EUCLID1)[0], be.diag(reset=True)
    o, h = both(ob, hb, N, 4, 0, setup, lambda be: be.run_deconv([d], [3]), read)
    print(threads, "predicted differ", int((o[0] != h[0]).sum()), "sampled differ", int((o[1] != h[1]).sum()), np.abs(o[0]-h[0]).max(), np.abs(o[1]-h[1]).max())
    o2, _ = both(ob, ob, N, 4, 0, setup, lambda be: be.run_deconv([d], [3]), read)
    print("   oracle twice:", int((o[0] != o2[0]).sum()))
