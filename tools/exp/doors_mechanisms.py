"""Config 3 at 2000 poses on the device: what happens to the true mode with (i) nullSurplusAdd = 0, (ii) Niter = 6, (iii) further solves.
Review r04 item 3: turn DESIGN 5's argument ("mechanism 2: nullSurplusAdd leaks 12 % per sighting") into measurements."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import iif
import doors_cases as dc

nposes = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
def run(tag, nsa, niter, solves=3, seed=1):
    fg = iif.generateCircularDoors(nposes=nposes, N=200, sightEvery=25)
    fg.solverParams.nullSurplusAdd = nsa
    fg.solverParams.productNiter = niter
    order = iif.nestedDissectionOrder(fg)
    rows = []
    for k in range(solves):
        if k > 0:
            fg.solverParams.graphinit = False
        iif.solveTree(fg, eliminationOrder=order, backend=iif.HipBackend, seed=seed + k)
        s = np.array([dc.share(fg, i) for i in range(nposes)])
        blocks = [float(np.median(s[a:a + 200])) for a in range(0, nposes, 200)]
        rows.append((float(np.median(s)), float(s.min()), float((s > 0.8).mean()), float((s[:200] >= 0.6).mean()), blocks))
        print(f"{tag:34s} solve {k + 1}: median {rows[-1][0]:.3f} min {rows[-1][1]:.3f} poses>0.8 {rows[-1][2]:.3f} | first 200 poses >= 0.6: {rows[-1][3]:.3f} | block medians " +
              " ".join(f"{b:.2f}" for b in blocks), flush=True)
    return rows

run("reference (nsa 0.3, Niter 1)", 0.3, 1)
run("nullSurplusAdd 0, Niter 1", 0.0, 1)
run("nsa 0.3, Niter 6", 0.3, 6)
run("nullSurplusAdd 0, Niter 6", 0.0, 6)
