"""products of MANY densities (AMP.manifoldProduct takes any number; the descriptor holds up to 128) against the oracle, alone and in
launches that mix them with small ones"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif
from test_gpu_mixed_product_launches import run
from oracle.oracle_backend import OracleBackend
bad = 0
for man, name in ((abi.EUCLID2, "Euclid(2)"), (abi.SE2, "SE(2)"), (abi.CIRCULAR, "Circular"), (abi.EUCLID3, "Euclid(3)"), (abi.EUCLID1, "Euclid(1)")):
    for N in (200, 100, 300):
        for Fs in ([32] * 6, [64] * 4, [128] * 3, [2, 128, 2, 64, 3, 2, 2, 32, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2], [2] * 30 + [128] + [2] * 30 + [16] + [3] * 40):
            keep = sorted(set([i for i, f in enumerate(Fs) if f > 3][:4] + [0, 1, len(Fs) - 1]))
            try:
                d = run(lambda n, s: iif.HipBackend(n, s, 0), N, man, Fs, nsrc=40)
                o = run(lambda n, s: OracleBackend(n, s, 0, threads=32), N, man, Fs, keep=keep, nsrc=40)
                w = max(float(np.nanmax(np.abs(d[i] - o[i]))) for i in keep)
                fin = all(np.isfinite(v).all() for v in d.values())
                flag = "" if fin and w < 1e-8 else "   <-- DIFFERS"
            except Exception as e:  # noqa: BLE001
                w, fin, flag = float("nan"), False, f"   <-- ERROR {str(e)[:150]}"
            bad += bool(flag)
            print(f"{name} N={N} launch of {len(Fs)} products, density counts {sorted(set(Fs))}: finite {fin}, max |device - oracle| {w:.2e}{flag}", flush=True)
print("launches that differ:", bad)
