#!/usr/bin/env python
"""Where a GPU solve and an oracle solve with identical random streams part: the tree program of a reduced BASELINE
configuration run stage by stage on both backends, the output slots of every stage compared.  Prints the first stages
whose outputs differ by more than 1e-12 (relative), with the op that produced the difference.

    python tools/exp/first_divergence.py config5_mixture_chain [max_reports]
    RESYNC=1 ...: the device continues from the oracle's outputs after every stage (differences injected per stage,
    not propagated ones)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import iif_amd_loader  # noqa: E402
iif = iif_amd_loader.load()
abi = iif.abi
from iif_amd.backend import HipBackend  # noqa: E402
from oracle.oracle_backend import OracleBackend  # noqa: E402

CONFIGS = {
    "config2_euclid2_chain": lambda: iif.generateChainEuclid(40, vardims=2, priorEvery=10, N=200),
    "config4_se2_lattice": lambda: iif.generateSE2Lattice(rows=3, cols=5, N=200, closeEvery=2),
    "config5_mixture_chain": lambda: iif.generateMixtureChain(nvars=24, N=300, priorEvery=8),
}
KIND = {abi.STAGE_PROPOSALS: "proposals", abi.STAGE_PRODUCTS: "products", abi.STAGE_COPIES: "copies", abi.STAGE_DECONV: "deconv",
        abi.STAGE_COPY_POINTS: "copy_points"}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "config5_mixture_chain"
    max_reports = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    fg = CONFIGS[name]()
    order = iif.nestedDissectionOrder(fg)
    iif.initAll(fg, backend=OracleBackend, seed=31)
    tree = iif.buildTreeReset(fg, order)
    tp = iif.TreeProgram(fg, tree, seed=31)
    N = fg.solverParams.N
    bes = [OracleBackend(N, tp.n_slots, 0), HipBackend(N, tp.n_slots, 0)]
    progs = []
    for be in bes:
        for v in fg.ls():
            var = fg.getVariable(v)
            be.belief_write(tp.main[v], var.varType.manifold, var.val, var.bw)
        iif.solver.write_densities(fg, be)
        progs.append(be.program(tp.stages, lazy_bandwidth=False))
    reports = 0
    for s, (kind, descs) in enumerate(tp.stages):
        for p in progs:
            p.run(s, s + 1)
        bes[1].synchronize()
        if kind in (abi.STAGE_COPIES, abi.STAGE_COPY_POINTS):
            continue
        worst = (0.0, None)
        nbad = 0
        for i, d in enumerate(descs):
            (po, bo), (ph, bh) = bes[0].slot_read(d.out_slot, d.manifold), bes[1].slot_read(d.out_slot, d.manifold)
            scale = max(1.0, np.abs(po).max())
            e = np.abs(po - ph).max() / scale
            eb = np.abs(np.asarray(bo) - np.asarray(bh)).max() / max(1e-300, np.abs(bo).max())
            if e > 1e-12:
                nbad += 1
            if os.environ.get("RESYNC"):  # every stage starts from the oracle's state: what is reported is injected in that stage
                bes[1].slot_write(d.out_slot, d.manifold, po, bo)
            if max(e, 0) > worst[0]:
                worst = (e, (i, d, int((np.abs(po - ph).max(axis=tuple(range(1, po.ndim))) > 1e-12 * scale).sum()), eb))
        if worst[0] > 1e-12:
            i, d, npart, eb = worst[1]
            extra = (f"factor_kind {d.factor_kind} sfidx {d.sfidx} nullhypo {d.nullhypo:.2f} ncomp {d.ncomp} inflate_cycles {d.inflate_cycles}"
                     if kind != abi.STAGE_PRODUCTS else f"nfactors {d.nfactors}")
            print(f"stage {s:4d} {KIND[kind]:10s}: {nbad} of {len(descs)} outputs differ; worst {worst[0]:.3e} (op {i}, manifold {d.manifold}, "
                  f"{npart} of {N} particles beyond 1e-12, bandwidth rel diff {eb:.2e}; {extra})")
            reports += 1
            if reports >= max_reports:
                break
    if not reports:
        print(f"{name}: all {len(tp.stages)} stages agree to 1e-12")
    for p in progs:
        p.close()
    for be in bes:
        be.close()


if __name__ == "__main__":
    main()
