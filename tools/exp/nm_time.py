import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, rand_points, relative_factor_desc
for rep in range(2):
  for centre in (1.0, 10.0, 0.0, 100.0, 1.0):
    N = 200
    be = iif.HipBackend(N, 3, 0)
    rng = np.random.default_rng(0)
    be.slot_write(0, abi.EUCLID2, rand_points(rng, abi.EUCLID2, N, centre, 0.3))
    be.slot_write(1, abi.EUCLID2, rand_points(rng, abi.EUCLID2, N, centre + 0.5, 0.3))
    d = relative_factor_desc(abi.F_LINREL, abi.EUCLID2, 2, 1, [0, 1], 2, 5, [1.0, 1.0], [0.1, 0.1]); d.skip_bandwidth = 1
    be.run_proposals([d]); be.timing_enable(True); be.timing_read(); be.diag(reset=True)
    for _ in range(5): be.run_proposals([d])
    t = be.timing_read()["nbp_proposal_kernel"][0] / 5
    print(rep, centre, round(t * 1e3, 1), "us", be.diag()["residual_evals"] / 5 / 600, "evals per solve", be.diag()["nonconverged"])
    be.close()
