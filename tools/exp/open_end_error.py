"""config 5's chain (Mixture(LinearRelative, [0.8, 0.2]) links on Euclid(3), N = 300) with ONE prior (x0): the error of the posterior mean along the
chain after 1 / 2 / 3 solveTree calls, over seeds, in units of the exact posterior's sigma = sqrt(0.208 d) -- what a tolerance for the open end of
a chain can be read from.   open_end_error.py [nvars=400] [seeds=24]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import iif_amd_loader
iif = iif_amd_loader.load()
nvars = int(sys.argv[1]) if len(sys.argv) > 1 else 400
seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 24
probe = [nvars // 8, nvars // 4, nvars // 2, 3 * nvars // 4, nvars - 1]
rows = {k: [] for k in (1, 2, 3)}
for s in range(seeds):
    fg = iif.generateMixtureChain(nvars=nvars, N=300, priorEvery=500)
    order = iif.nestedDissectionOrder(fg)
    for r in (1, 2, 3):
        iif.solveTree(fg, eliminationOrder=order, backend=iif.HipBackend, seed=1000 * s + r)
        z = []
        for i in probe:
            m = fg.getVal(f"x{i}").mean(axis=0) - np.array([i, 0.0, 0.0])
            z.append(m / np.sqrt(0.208 * i))
        rows[r].append(z)
print(f"{nvars}-variable mixture chain, one prior at x0, N = 300, {seeds} seeds; |posterior mean - truth| along x, in exact sigmas sqrt(0.208 d)")
print("probe poses d =", probe, " exact sigma =", [round(float(np.sqrt(0.208 * i)), 2) for i in probe])
for r in (1, 2, 3):
    a = np.array(rows[r])  # (seeds, probes, 3)
    for k, nm in enumerate("xyz"):
        print(f"after {r} solve(s), {nm}: std over seeds", np.round(a[:, :, k].std(axis=0), 2), " max |.|", np.round(np.abs(a[:, :, k]).max(axis=0), 2))
a = np.array(rows[1])
print("end pose after one solve, per seed (x y z):")
for s_ in range(seeds): print("  seed", s_, np.round(a[s_, -1], 2))
