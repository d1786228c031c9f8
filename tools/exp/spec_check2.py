import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import iif_amd_loader
iif = iif_amd_loader.load(); abi = iif.abi
rng = np.random.default_rng(0)
for trial in range(6):
    N = 200
    n = 1 + trial % 3
    data = [rng.normal(size=(N, 1)) * (1 + trial) for _ in range(n)]
    res = []
    for mode in ("spec", "seq"):
        if mode == "seq": os.environ["NBP_NO_SPECULATIVE_FITS"] = "1"
        else: os.environ.pop("NBP_NO_SPECULATIVE_FITS", None)
        be = iif.HipBackend(N, 8)
        for s in range(n): be.slot_write(s, abi.EUCLID1, data[s], np.ones(1))
        be.diag(reset=True)
        be.run_bandwidth(list(range(n)), [abi.EUCLID1] * n)
        res.append(([be.slot_read(s, abi.EUCLID1)[1][0].hex() for s in range(n)], be.diag()["lcv_evals"]))
        be.close()
    print(trial, n, res[0], res[1])
