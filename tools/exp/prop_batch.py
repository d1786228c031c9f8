"""a chip-filling batch of LinearRelative Euclid(2) proposals (for counter passes)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, rand_points, relative_factor_desc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 975
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
be = iif.HipBackend(N, 2 * B + 2, 0)
rng = np.random.default_rng(0)
for j in range(B + 1):
    be.slot_write(j, abi.EUCLID2, rand_points(rng, abi.EUCLID2, N, 3.0 + j, 0.3))
descs = []
for j in range(B):
    d = relative_factor_desc(abi.F_LINREL, abi.EUCLID2, 2, 1, [j, j + 1], B + 1 + j, 5 + j, [1.0, 0.0], [0.1, 0.1]); d.skip_bandwidth = 1
    descs.append(d)
be.run_proposals(descs); be.timing_enable(True); be.timing_read(); be.diag(reset=True)
for _ in range(4): be.run_proposals(descs)
t = be.timing_read()["nbp_proposal_kernel"][0] / 4
dg = be.diag()
print(N, "particles", B, "proposals:", round(t * 1e3, 1), "us", dg["residual_evals"] / 4 / (B * N * 3), "evals per solve", dg["nonconverged"])
