"""per-pose error of the posterior mean (x coordinate, in exact sigmas) on config 5's chain with one prior, two ways: bench.py's own path
(bench_support.RankSolve: native graph initialisation with seed 0, then a resident program replayed from the snapshot of the initial beliefs)
and iif.solveTree (graph initialisation + one solve)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import iif_amd_loader
iif = iif_amd_loader.load()
import bench_support
nvars = int(sys.argv[1]) if len(sys.argv) > 1 else 400
wl = bench_support.workloads(iif)["5"]
poses = list(range(0, nvars, 20)) + [nvars - 1]
sig = np.sqrt(0.208 * np.maximum(1, np.array(poses)))
def row(tag, get):
    e = np.array([get(i) for i in poses])  # (poses, 3)
    for k, nm in enumerate("xyz"):
        print(tag, nm, "err/sigma:", " ".join(f"{v:+.2f}" for v in e[:, k] / sig))
rs = bench_support.RankSolve(iif, wl, nvars, 300, 0, 1, 0, None)
rs.prepare()
man = rs.fg.getVariable("x0").varType.manifold
row("bench path, initial beliefs     ", lambda i: rs.be.slot_read(rs.main[f"x{i}"], man)[0].mean(axis=0) - np.array([i, 0.0, 0.0]))
for k in range(2):
    rs.step(k)
    rs.be.synchronize()
    row(f"bench path, step seed {k}          ", lambda i: rs.be.slot_read(rs.main[f"x{i}"], man)[0].mean(axis=0) - np.array([i, 0.0, 0.0]))
for s in range(2):
    fg = iif.generateMixtureChain(nvars=nvars, N=300, priorEvery=500)
    order = iif.nestedDissectionOrder(fg)
    iif.initAll(fg, backend=iif.HipBackend, seed=s)
    row(f"solveTree path seed {s}, initial   ", lambda i: fg.getVal(f"x{i}").mean(axis=0) - np.array([i, 0.0, 0.0]))
    iif.solveTree(fg, eliminationOrder=order, backend=iif.HipBackend, seed=77 + s)
    row(f"solveTree path seed {s}, one solve ", lambda i: fg.getVal(f"x{i}").mean(axis=0) - np.array([i, 0.0, 0.0]))
