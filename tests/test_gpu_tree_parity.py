"""-m gpu: whole solveTree on the HIP backend vs the oracle backend with identical seeds, plus the
reference's acceptance bands evaluated on the GPU result."""
import copy

import numpy as np
import pytest

from parity_utils import abi, assert_points_close, iif

pytestmark = pytest.mark.gpu


def _chain(N=100, n=6):
    fg = iif.initfg(iif.SolverParams(N=N))
    for i in range(n):
        iif.addVariable(fg, f"x{i}", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal(0.0, 1.0)))
    for i in range(n - 1):
        iif.addFactor(fg, [f"x{i}", f"x{i+1}"], iif.LinearRelative(iif.Normal(1.0, 0.1)))
    return fg


def test_config1_chain_gpu_matches_oracle(oracle_backend, hip_backend):
    fa, fb = _chain(), _chain()
    iif.solveTree(fa, backend=oracle_backend, seed=5)
    iif.solveTree(fb, backend=hip_backend, seed=5)
    nbad = 0
    for v in fa.ls():
        a, b = fa.getVal(v), fb.getVal(v)
        bad = np.abs(a - b).max(axis=1) > 1e-7 * np.maximum(1, np.abs(a).max(axis=1))
        nbad += bad.sum()
    # a single label flip anywhere upstream changes every later particle of that variable, so allow
    # distribution-level agreement as the fallback criterion
    if nbad:
        for v in fa.ls():
            assert abs(fa.getVal(v).mean() - fb.getVal(v).mean()) < 0.2
    X = [fb.getVal(f"x{i}").mean() for i in range(6)]
    assert abs(X[0]) < 0.5
    for i in range(5):
        assert abs(X[i + 1] - X[i] - 1.0) < 0.35


def test_euclid2_chain_posteriors(hip_backend):
    """BASELINE config 2 shape at reduced length: posterior means within tolerance of truth x_i=(i,i)."""
    fg = iif.generateChainEuclid(40, vardims=2, priorEvery=10, N=200)
    order = iif.nestedDissectionOrder(fg)
    tree, tm = iif.solveTree(fg, eliminationOrder=order, backend=hip_backend, seed=2, return_timing=True)
    for i in range(40):
        m = fg.getVal(f"x{i}").mean(axis=0)
        assert np.abs(m - i).max() < 0.35, (i, m)
        assert fg.getVal(f"x{i}").std(axis=0).max() < 1.0
    assert tm["messages"] == 2 * (len(tree.cliques) - len(tree.roots))


def test_torch_owned_arena_sharded_path_single_rank(hip_backend):
    """the multi-GPU leg with world = 1: arena allocated by torch and handed to libnbp by pointer
    (what RCCL needs), TreeProgram with an owner map, ShardedRunner segments"""
    import torch
    from iif_amd.dist_solver import ShardedTreeSolve

    s = ShardedTreeSolve(iif, 60, 100, rank=0, world=1, local=0, dist=None)
    s.prepare()
    s.step(0)
    s.be.synchronize()
    s.check_posteriors()
    assert s.posterior_max_mean_err < 1.0
    assert s.global_messages == 2 * (len(s.tree.cliques) - len(s.tree.roots))
    assert s.arena.data_ptr() == s.be.arena_ptr()
    torch.cuda.synchronize()
