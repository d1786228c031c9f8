#!/usr/bin/env python3
"""Every stage of a whole-tree program on the oracle's state, counted in BITS (round 6): how many of the ops of each kind
come out of the device bit-identical to the CPU checker's on identical inputs, and how far the others are.

    python tools/exp/stagewise_bits.py [name ...] [--whole]     (GPU box; default: the five reduced BASELINE configurations
                                                                 and four mid-size ones)
    --whole: also one whole solve per configuration on both sides (no hand-over of state): share of bit-identical variables

Kinds: proposals by (factor kind, manifold), bandwidths of proposals, products of several densities, their bandwidths.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import iif_amd_loader  # noqa: E402

iif = iif_amd_loader.load()
abi = iif.abi
from oracle.oracle_backend import OracleBackend  # noqa: E402

NTHREADS = max(8, min(64, os.cpu_count() or 8))


def config1():
    fg = iif.initfg(iif.SolverParams(N=100))
    for i in range(6):
        iif.addVariable(fg, f"x{i}", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal(0.0, 1.0)))
    for i in range(5):
        iif.addFactor(fg, [f"x{i}", f"x{i+1}"], iif.LinearRelative(iif.Normal(1.0, 0.1)))
    return fg


CONFIGS = {
    "config1_scalar_chain": config1,
    "config2_euclid2_chain": lambda: iif.generateChainEuclid(40, vardims=2, priorEvery=10, N=200),
    "config3_circular_doors": lambda: iif.generateCircularDoors(nposes=25, N=200, sightEvery=10),
    "config4_se2_lattice": lambda: iif.generateSE2Lattice(rows=3, cols=5, N=200, closeEvery=2),
    "config5_mixture_chain": lambda: iif.generateMixtureChain(nvars=24, N=300, priorEvery=8),
    "config2_1000": lambda: iif.generateChainEuclid(1000, vardims=2, priorEvery=100, N=200),
    "config3_1000": lambda: iif.generateCircularDoors(nposes=1000, N=200, sightEvery=25),
    "config4_16x40": lambda: iif.generateSE2Lattice(rows=16, cols=40, N=200, closeEvery=5),
    "config5_800": lambda: iif.generateMixtureChain(nvars=800, N=300, priorEvery=400),
    "config5_2500": lambda: iif.generateMixtureChain(nvars=2500, N=300, priorEvery=500),
    "config4_50x100": lambda: iif.generateSE2Lattice(rows=50, cols=100, N=200, closeEvery=5),
    "config5_10000": lambda: iif.generateMixtureChain(nvars=10000, N=300, priorEvery=500),
    "config3_2000": lambda: iif.generateCircularDoors(nposes=2000, N=200, sightEvery=25),
}
DEFAULT = ["config1_scalar_chain", "config2_euclid2_chain", "config3_circular_doors", "config4_se2_lattice", "config5_mixture_chain",
           "config2_1000", "config3_1000", "config4_16x40", "config5_800"]
KIND = {abi.F_PRIOR: "prior", abi.F_MSGPRIOR: "msgprior", abi.F_LINREL: "linrel", abi.F_CIRCULAR: "circ", abi.F_SE2: "se2",
        abi.F_EUCLIDDIST: "dist", abi.F_PASSTHROUGH: "passthrough"}


def oracle_be(N, n, side_ints=0):
    return OracleBackend(N, n, side_ints, threads=NTHREADS)


def stagewise(name):
    fg = CONFIGS[name]()
    order = iif.nestedDissectionOrder(fg)
    big = len(fg.ls()) > 100
    iif.initAll(fg, backend=iif.HipBackend if big else oracle_be, seed=31)
    tree = iif.buildTreeReset(fg, order)
    tp = iif.TreeProgram(fg, tree, seed=31)
    N = fg.solverParams.N
    bes = [oracle_be(N, tp.n_slots), iif.HipBackend(N, tp.n_slots)]
    progs = []
    stat = {}

    def rec(key, same, err):
        s = stat.setdefault(key, [0, 0, 0.0])
        s[0] += 1
        s[1] += int(same)
        s[2] = max(s[2], err)

    t0 = time.time()
    try:
        for be in bes:
            for v in fg.ls():
                var = fg.getVariable(v)
                be.belief_write(tp.main[v], var.varType.manifold, var.val, var.bw)
            iif.solver.write_densities(fg, be)
            progs.append(be.program(tp.stages, lazy_bandwidth=False))
        for s, (kind, descs) in enumerate(tp.stages):
            for p in progs:
                p.run(s, s + 1)
            if kind in (abi.STAGE_COPIES, abi.STAGE_COPY_POINTS):
                continue
            prod = kind == abi.STAGE_PRODUCTS
            for d in descs:
                if prod and d.nfactors == 1:
                    continue
                # the stored coordinates themselves ((x, y, theta) of an SE(2) slot read as three Euclidean rows): what the next
                # op reads, and what the device is handed back -- through rotation matrices a heading would come back an ulp off
                rm = abi.EUCLID3 if d.manifold == abi.SE2 else d.manifold
                (po, bo), (ph, bh) = bes[0].slot_read(d.out_slot, rm), bes[1].slot_read(d.out_slot, rm)
                bo, bh = np.asarray(bo, dtype=float), np.asarray(bh, dtype=float)
                co, ch = po, ph
                diff = np.abs(co - ch)
                if d.manifold in (abi.CIRCULAR, abi.SE2):
                    diff = np.minimum(diff, np.abs(2 * np.pi - diff))
                e = float(diff.max() / max(1.0, np.abs(co).max()))
                key = (f"product F={min(d.nfactors, 9)}{'+' if d.nfactors >= 9 else ''}" if prod else f"proposal {KIND.get(d.factor_kind, d.factor_kind)}") + f" m{d.manifold}"
                rec(key + " points", np.array_equal(po, ph), e)
                eb = float(np.abs(bo - bh).max() / max(1e-300, np.abs(bo).max())) if np.abs(bo).max() > 0 else 0.0
                rec(key + " bandwidth", np.array_equal(bo, bh), eb)
                bes[1].slot_write(d.out_slot, rm, po, bo)
        print(f"== {name}: {len(tp.stages)} stages, N = {N}, {time.time() - t0:.0f} s", flush=True)
        for k in sorted(stat):
            n, same, worst = stat[k]
            print(f"   {k:42s} {same:7d} of {n:7d} bit-identical   worst of the others {worst:.2e}", flush=True)
    finally:
        for p in progs:
            p.close()
        for be in bes:
            be.close()


def whole(name):
    fa, fb = CONFIGS[name](), CONFIGS[name]()
    order = iif.nestedDissectionOrder(fa)
    iif.solveTree(fa, eliminationOrder=order, backend=oracle_be, seed=31)
    iif.solveTree(fb, eliminationOrder=order, backend=iif.HipBackend, seed=31)
    same = close = 0
    worst = 0.0
    for v in fa.ls():
        a, b = fa.getVal(v), fb.getVal(v)
        same += int(np.array_equal(a, b))
        e = float(np.abs(a - b).max())
        close += int(e <= 1e-6)
        worst = max(worst, e)
    print(f"== {name}: whole solve (graph initialisation + up + down), identical streams: {same} of {len(fa.ls())} variables bit-identical, "
          f"{close} within 1e-6, worst {worst:.2e}", flush=True)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    for nm in (args or DEFAULT):
        stagewise(nm)
    if "--whole" in sys.argv:
        for nm in (args or DEFAULT[:5]):
            whole(nm)
