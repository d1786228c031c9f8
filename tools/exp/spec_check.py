import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import iif_amd_loader
iif = iif_amd_loader.load(); abi = iif.abi
rng = np.random.default_rng(0)
out = []
for trial in range(40):
    N = 200
    be = iif.HipBackend(N, 8)
    n = 1 + trial % 5
    for s in range(n):
        be.slot_write(s, abi.EUCLID2, rng.normal(size=(N, 2)) * (1 + trial), np.ones(2))
    be.diag(reset=True)
    be.run_bandwidth(list(range(n)), [abi.EUCLID2] * n)
    out.append([be.slot_read(s, abi.EUCLID2)[1] for s in range(n)] + [[be.diag()["lcv_evals"], 0]])
    be.close()
np.save(sys.argv[1], np.concatenate([np.ravel(o) for o in out]))
