import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
sys.argv = [sys.argv[0]]
from parity_utils import abi, iif
import importlib.util
spec = importlib.util.spec_from_file_location("ss", os.path.join(R, "tools/exp/spec_stress.py"))
# reuse fits()
src = open(os.path.join(R, "tools/exp/spec_stress.py")).read().split("rng = np.random")[0]
exec(src)
rng = np.random.default_rng(5)
N = 64
data = [rng.normal(0, rng.uniform(0.01, 5), (N, 2)) for _ in range(240)]
a = fits(N, abi.EUCLID2, data, 3, {"NBP_NO_SPECULATIVE_FITS": "1"})
b = fits(N, abi.EUCLID2, data, 8, {"NBP_NO_SPECULATIVE_FITS": "1"})
c = fits(N, abi.EUCLID2, data, 240, {"NBP_NO_SPECULATIVE_FITS": "1"})
print("seq groups of 3 vs 8:", int((a != b).sum()), " vs one launch of 240:", int((a != c).sum()))
d = fits(N, abi.EUCLID2, data, 8, {"NBP_SPEC_DEPTH3": "0"})
e = fits(N, abi.EUCLID2, data, 8, {"NBP_SPEC_DEPTH3": "0"})
print("K=3 vs seq:", int((d != a).sum()), " K=3 run twice:", int((d != e).sum()))
