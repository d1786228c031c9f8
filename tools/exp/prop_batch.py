"""a chip-filling batch of relative-factor proposals (for counter passes): prop_batch.py [B=975] [N=200] [lin2|lin3|se2|circ] [sfidx=1: the variable solved for]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, rand_points, relative_factor_desc
B = int(sys.argv[1]) if len(sys.argv) > 1 else 975
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
KIND = sys.argv[3] if len(sys.argv) > 3 else "lin2"
SF = int(sys.argv[4]) if len(sys.argv) > 4 else 1
FK, MAN, MEAN, SIG = {"lin2": (abi.F_LINREL, abi.EUCLID2, [1.0, 0.0], [0.1, 0.1]), "lin3": (abi.F_LINREL, abi.EUCLID3, [1.0, 0.0, -0.5], [0.1, 0.1, 0.1]),
                      "se2": (abi.F_SE2, abi.SE2, [1.0, 0.2, 0.3], [0.1, 0.1, 0.01]), "circ": (abi.F_CIRCULAR, abi.CIRCULAR, [0.4], [0.05])}[KIND]
be = iif.HipBackend(N, 2 * B + 2, 0)
rng = np.random.default_rng(0)
for j in range(B + 1):
    be.slot_write(j, MAN, rand_points(rng, MAN, N, (3.0 + j) if KIND != "circ" else 0.3 * j, 0.3))
descs = []
for j in range(B):
    d = relative_factor_desc(FK, MAN, 2, SF, [j, j + 1], B + 1 + j, 5 + j, MEAN, SIG); d.skip_bandwidth = 1
    descs.append(d)
be.run_proposals(descs); be.timing_enable(True); be.timing_read(); be.diag(reset=True)
for _ in range(4): be.run_proposals(descs)
t = be.timing_read()["nbp_proposal_kernel"][0] / 4
dg = be.diag()
print(KIND, "sfidx", SF, N, "particles", B, "proposals:", round(t * 1e3, 1), "us", dg["residual_evals"] / 4 / (B * N * 3), "evals per solve", dg["nonconverged"])
