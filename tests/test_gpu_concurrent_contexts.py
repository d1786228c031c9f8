"""-m gpu: several contexts in flight at once (the situation of a multi-threaded host with pooled contexts, one per clique
Task): every solve equals its own sequential-search, one-context-at-a-time result bit for bit.  A regression guard: with
wave-uniform constants moved into SGPRs by an inline-asm `v_readfirstlane` placed right behind the VALU instruction that
produced the value -- no wait state, because the compiler's hazard recognizer does not look inside inline asm -- a fit
occasionally read the constants of its PREVIOUS evaluation, and only kernels of other contexts interleaving on the same
SIMDs made it visible (single-context runs, and every comparison of kernels against each other, were clean)."""
import os

import numpy as np
import pytest

from parity_utils import abi, iif

pytestmark = pytest.mark.gpu


def build(nvars, N, seed):
    fg = iif.generateChainEuclid(nvars, vardims=2, priorEvery=8, N=N)
    rng = np.random.default_rng(seed)
    for v in fg.ls():
        var = fg.getVariable(v)
        var.val, var.bw, var.initialized = rng.normal(float(v[1:]), 0.5, (N, 2)), np.array([0.2, 0.2]), True
    tree = iif.buildTreeReset(fg, iif.nestedDissectionOrder(fg))
    return fg, iif.TreeProgram(fg, tree, seed=seed)


def run_all(items, N, env, reps):
    old = {k: os.environ.pop(k, None) for k in ("NBP_NO_SPECULATIVE_FITS", "NBP_SPEC_DEPTH3")}
    os.environ.update(env)
    bes, progs, outs = [], [], []
    try:
        for fg, tp in items:
            be = iif.HipBackend(N, tp.n_slots)
            bes.append(be)
            progs.append(be.program(tp.stages, lazy_bandwidth=True))
        for _ in range(reps):
            for (fg, tp), be in zip(items, bes):
                for v in fg.ls():
                    var = fg.getVariable(v)
                    be.slot_write(tp.main[v], abi.EUCLID2, var.val, var.bw)
            for p in progs:
                p.run()  # asynchronous: the programs of all contexts are in flight together
            for be in bes:
                be.synchronize()
            outs.append([np.concatenate([np.concatenate([x.ravel() for x in be.slot_read(tp.main[v], abi.EUCLID2)]) for v in fg.ls()])
                         for (fg, tp), be in zip(items, bes)])
    finally:
        for p in progs:
            p.close()
        for be in bes:
            be.close()
        for k, v in old.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
    return outs


def test_concurrent_contexts_are_reproducible():
    N = 100
    items = [build(nv, N, 10 + i) for i, nv in enumerate((48, 24, 64, 16, 128, 32))]
    ref = [run_all([it], N, {"NBP_NO_SPECULATIVE_FITS": "1"}, 1)[0][0] for it in items]  # one context at a time, sequential search
    for out in run_all(items, N, {}, 4):  # first run: plain launches; later runs: hipGraph replays
        for got, want in zip(out, ref):
            np.testing.assert_array_equal(got, want)
