import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from parity_utils import abi, iif
os.environ["NBP_NO_SPECULATIVE_FITS"] = "1"
N, man = 200, abi.EUCLID2
rng = np.random.default_rng(3)
beliefs = [rng.normal(size=(n, 2)) * 0.7 for n in (200, 150, 65, 64, 33, 9, 3)]
for one in range(len(beliefs)):
    res = []
    for f64 in ("1", "0"):
        os.environ["NBP_FIT_F64"] = f64
        be = iif.HipBackend(N, 2, 0)
        b = beliefs[one]
        if len(b) == N: be.slot_write(0, man, b)
        else: be.belief_write(0, man, b, np.ones(2))
        be.diag(reset=True)
        be.run_bandwidth([0], [man])
        res.append((be.slot_read(0, man)[1], be.diag()))
        be.close()
    print(len(beliefs[one]), res[0][0], res[1][0], "SAME" if np.array_equal(res[0][0], res[1][0]) else "DIFF", res[0][1]["lcv_evals"], res[1][1]["lcv_evals"], res[1][1]["lcv_evals_f32"])
