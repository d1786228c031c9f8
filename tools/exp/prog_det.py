"""whole-tree program, single context: speculative fits on vs off, and run-to-run"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif

def solve(nvars, N, env, seed=9):
    for k in ("NBP_NO_SPECULATIVE_FITS", "NBP_SPEC_DEPTH3"):
        os.environ.pop(k, None)
    os.environ.update(env)
    fg = iif.generateChainEuclid(nvars, vardims=2, priorEvery=8, N=N)
    rng = np.random.default_rng(1)
    for v in fg.ls():
        var = fg.getVariable(v)
        var.val = rng.normal(float(v[1:]), 0.5, (N, 2)); var.bw = np.array([0.2, 0.2]); var.initialized = True
    tree = iif.buildTreeReset(fg, iif.nestedDissectionOrder(fg))
    tp = iif.TreeProgram(fg, tree, seed=seed)
    be = iif.HipBackend(N, tp.n_slots)
    for v in fg.ls():
        var = fg.getVariable(v)
        be.slot_write(tp.main[v], abi.EUCLID2, var.val, var.bw)
    prog = be.program(tp.stages, lazy_bandwidth=True)
    prog.run(); be.synchronize()
    out = np.concatenate([np.concatenate([x.ravel() for x in be.slot_read(tp.main[v], abi.EUCLID2)]) for v in fg.ls()])
    prog.close(); be.close()
    return out

for nvars, N in ((48, 100), (128, 100), (64, 64)):
    a = solve(nvars, N, {"NBP_NO_SPECULATIVE_FITS": "1"})
    b = solve(nvars, N, {"NBP_NO_SPECULATIVE_FITS": "1"})
    c = solve(nvars, N, {})
    d = solve(nvars, N, {})
    e = solve(nvars, N, {"NBP_SPEC_DEPTH3": "0"})
    print(nvars, N, "seq twice:", int((a != b).sum()), " spec vs seq:", int((c != a).sum()), " spec twice:", int((c != d).sum()), " K=3 vs seq:", int((e != a).sum()))
