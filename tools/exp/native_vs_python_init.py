"""graph initialisation by the native C++ host against the Python mirror's initAll, belief by belief, on the four shapes"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import iif
from iif_amd import native_host
hb = lambda n, s, side_ints=0: iif.HipBackend(n, s, side_ints=side_ints)
shapes = {"chain2": lambda: iif.generateChainEuclid(120, vardims=2, priorEvery=40, N=100),
          "doors": lambda: iif.generateCircularDoors(nposes=120, N=100, sightEvery=25),
          "se2 lattice 3x20": lambda: iif.generateSE2Lattice(rows=3, cols=20, N=100, closeEvery=5),
          "se2 lattice 6x100": lambda: iif.generateSE2Lattice(rows=6, cols=100, N=200, closeEvery=5),
          "mixture3": lambda: iif.generateMixtureChain(nvars=120, N=100, priorEvery=40)}
for name, fresh in shapes.items():
    fa = fresh(); N = fa.solverParams.N
    iif.initAll(fa, backend=hb, seed=31)
    fb = fresh()
    g = native_host.NativeGraph.from_fg(fb)
    need, planned = g.init_plan(31)
    be = hb(N, need)
    for i, v in enumerate(fb.ls()):
        var = fb.getVariable(v)
        be.slot_write(i, var.varType.manifold, var.val, var.bw)
    prog = g.init_compile(be)
    prog.run(); be.synchronize()
    diff = []
    for i, v in enumerate(fb.ls()):
        pts, bw = be.slot_read(i, fb.getVariable(v).varType.manifold)
        if not (np.array_equal(pts, fa.getVal(v)) and np.array_equal(bw, fa.getVariable(v).bw)):
            diff.append((v, float(np.abs(pts - fa.getVal(v)).max()), float(np.abs(bw - fa.getVariable(v).bw).max())))
    print(f"{name}: {len(fb.ls())} variables, planned {len(planned)}, {len(diff)} beliefs differ", diff[:4], flush=True)
    prog.close(); be.close()
