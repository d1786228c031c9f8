"""whole-tree programs from several contexts at once (kernels of different contexts overlap): every solve must equal its
own sequential-search result bit for bit"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif

def build(nvars, N, seed):
    fg = iif.generateChainEuclid(nvars, vardims=2, priorEvery=8, N=N)
    rng = np.random.default_rng(seed)
    for v in fg.ls():
        var = fg.getVariable(v)
        var.val = rng.normal(float(v[1:]), 0.5, (N, 2)); var.bw = np.array([0.2, 0.2]); var.initialized = True
    tree = iif.buildTreeReset(fg, iif.nestedDissectionOrder(fg))
    return fg, iif.TreeProgram(fg, tree, seed=seed)

def run_all(items, N, env, reps=1):
    for k in ("NBP_NO_SPECULATIVE_FITS", "NBP_SPEC_DEPTH3"):
        os.environ.pop(k, None)
    os.environ.update(env)
    bes, progs = [], []
    for fg, tp in items:
        be = iif.HipBackend(N, tp.n_slots)
        bes.append(be); progs.append(be.program(tp.stages, lazy_bandwidth=True))
    outs = []
    for _ in range(reps):
        for (fg, tp), be in zip(items, bes):
            for v in fg.ls():
                var = fg.getVariable(v)
                be.slot_write(tp.main[v], abi.EUCLID2, var.val, var.bw)
        for p in progs:
            p.run()                     # asynchronous: all contexts in flight together
        for be in bes:
            be.synchronize()
        outs.append([np.concatenate([np.concatenate([x.ravel() for x in be.slot_read(tp.main[v], abi.EUCLID2)]) for v in fg.ls()])
                     for (fg, tp), be in zip(items, bes)])
    for p in progs: p.close()
    for be in bes: be.close()
    return outs

N = 100
items = [build(nv, N, 10 + i) for i, nv in enumerate((48, 24, 64, 16, 128, 32, 48, 12))]
ref = run_all(items, N, {"NBP_NO_SPECULATIVE_FITS": "1"})[0]
bad = 0
for rep, out in enumerate(run_all(items, N, {}, reps=int(sys.argv[1]) if len(sys.argv) > 1 else 6)):
    d = [int((a != b).sum()) for a, b in zip(out, ref)]
    bad += sum(1 for x in d if x)
    print("rep", rep, "differing values per context:", d)
print("TOTAL contexts differing", bad)
