"""the last poses of config 5's one-prior chain one by one: y error of the mean (exact sigmas) and the sample std of y (exact sigmas), init vs after one solve"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import iif_amd_loader
iif = iif_amd_loader.load()
nvars = 400
for s in (0, 1):
    fg = iif.generateMixtureChain(nvars=nvars, N=300, priorEvery=500)
    order = iif.nestedDissectionOrder(fg)
    iif.initAll(fg, backend=iif.HipBackend, seed=s)
    poses = list(range(340, 400, 3)) + [397, 398, 399]
    def show(tag):
        m = np.array([fg.getVal(f"x{i}")[:, 1].mean() / np.sqrt(0.208 * i) for i in poses])
        sd = np.array([fg.getVal(f"x{i}")[:, 1].std() / np.sqrt(0.208 * i) for i in poses])
        print(f"init seed {s} {tag} mean y/sigma:", " ".join(f"{v:+.2f}" for v in m))
        print(f"init seed {s} {tag}  std y/sigma:", " ".join(f"{v:5.2f}" for v in sd))
    print("poses:", poses)
    show("initial  ")
    tree = iif.solveTree(fg, eliminationOrder=order, backend=iif.HipBackend, seed=77)
    show("one solve")
    if s == 0:
        d = tree.depths()
        for c in tree.cliques.values() if isinstance(tree.cliques, dict) else tree.cliques:
            names = [str(x) for x in list(c.frontalIDs) + list(c.separatorIDs)]
            if any(n in ("x399", "x398", "x397", "x396", "x395", "x390", "x385") for n in [str(x) for x in c.frontalIDs]):
                print("clique", c.id, "depth", d[c.id], "frontals", [str(x) for x in c.frontalIDs], "separators", [str(x) for x in c.separatorIDs])
