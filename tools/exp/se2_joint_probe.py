"""SE(2) lattice with joint upward messages (useMsgLikelihoods): the proposal stage that parts from the oracle -- which factor kind, how
many particles, did the searches converge?"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif
from oracle.oracle_backend import OracleBackend
N, rows = 200, 3
fg = iif.generateSE2Lattice(rows=rows, cols=100, N=N, closeEvery=5)
fg.solverParams.useMsgLikelihoods = True
order = iif.nestedDissectionOrder(fg)
hb = lambda n, s, side_ints=0: iif.HipBackend(n, s, side_ints=side_ints)
iif.initAll(fg, backend=hb, seed=31)
tree = iif.buildTreeReset(fg, order)
tp = iif.TreeProgram(fg, tree, seed=31)
bes = [OracleBackend(N, tp.n_slots, 0, threads=48), iif.HipBackend(N, tp.n_slots, 0)]
progs = []
for be in bes:
    for v in fg.ls():
        var = fg.getVariable(v)
        be.belief_write(tp.main[v], var.varType.manifold, var.val, var.bw)
    iif.solver.write_densities(fg, be)
    progs.append(be.program(tp.stages, lazy_bandwidth=False))
names = {getattr(abi, k): k for k in dir(abi) if k.startswith("F_")}
for s, (kind, descs) in enumerate(tp.stages):
    d0 = [be.diag(reset=True) for be in bes]
    for p in progs: p.run(s, s + 1)
    d1 = [be.diag() for be in bes]
    if kind in (abi.STAGE_COPIES, abi.STAGE_COPY_POINTS): continue
    for i, d in enumerate(descs):
        (po, bo), (ph, bh) = bes[0].slot_read(d.out_slot, d.manifold), bes[1].slot_read(d.out_slot, d.manifold)
        pp = np.abs(po - ph).reshape(po.shape[0], -1).max(axis=1) / max(1.0, np.abs(po).max())
        if pp.max() > 1e-5:
            fk = names.get(getattr(d, "factor_kind", -1), "?") if kind == abi.STAGE_PROPOSALS else "product"
            print(f"stage {s} kind {kind} op {i} {fk} manifold {d.manifold}: worst {pp.max():.2e}, particles beyond 1e-7 / 1e-5 / 1e-4: {(pp > 1e-7).sum()} / {(pp > 1e-5).sum()} / {(pp > 1e-4).sum()}; "
                  f"stage nonconverged oracle {d1[0]['nonconverged']} device {d1[1]['nonconverged']}; meas_kde {getattr(d, 'meas_kde', None)} nvars {getattr(d, 'nvars', None)}", flush=True)
        bes[1].slot_write(d.out_slot, d.manifold, po, np.asarray(bo, dtype=float))
print("done")
