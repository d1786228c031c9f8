"""-m gpu: two-stream rounds (nbp_api.hip plan_pipeline; environment NBP_PIPELINE_MIN).  A round with many variable
updates is cut into two halves that run their proposal / fit / product launches on two streams, one launch apart.  No
particle and no bandwidth may depend on that: the same tree program with and without the split, every slot compared
bit for bit, on a chain (components = cliques), on a program run several times (hipGraph replay with a fork and a
join inside) and on a hand-made round whose updates read each other's outputs (they must end up in one half)."""
import os

import numpy as np
import pytest

from parity_utils import abi, iif, product_desc, rand_points, relative_factor_desc

pytestmark = pytest.mark.gpu


@pytest.fixture
def pipeline_env():
    def set_(v):
        if v is None:
            os.environ.pop("NBP_PIPELINE_MIN", None)
        else:
            os.environ["NBP_PIPELINE_MIN"] = str(v)
    yield set_
    os.environ.pop("NBP_PIPELINE_MIN", None)


def _solve(fg_build, pipe_min, set_env, runs=1):
    from iif_amd import native_host
    set_env(pipe_min)
    fg = fg_build()
    N = fg.solverParams.N
    g = native_host.NativeGraph.from_fg(fg)
    nt = g.build_tree(g.order_nested_dissection())
    n_slots = nt.plan_slots(True)
    need, _ = g.init_plan(0)
    be = iif.HipBackend(N, max(n_slots, need))
    ip = g.init_compile(be)
    ip.run()
    be.synchronize()
    ip.close()
    be.run_copies([abi.CopyDesc(nt.main[v], nt.snap[v]) for v in fg.ls()])
    prog = nt.compile(be, 1)
    n2 = prog.num_two_stream()
    for k in range(runs):
        prog.reseed(77 + k)
        prog.run()
    be.synchronize()
    out = {v: be.slot_read(nt.main[v], fg.getVariable(v).varType.manifold) for v in fg.ls()}
    prog.close()
    be.close()
    return n2, out


@pytest.mark.parametrize("build,runs", [
    (lambda: iif.generateChainEuclid(600, vardims=2, priorEvery=50, N=100), 1),
    (lambda: iif.generateChainEuclid(600, vardims=2, priorEvery=50, N=100), 3),   # the third run replays the captured graph
    (lambda: iif.generateSE2Lattice(rows=20, cols=40, N=100, closeEvery=2), 3),
])
def test_two_stream_rounds_change_nothing(pipeline_env, build, runs):
    n0, ref = _solve(build, None, pipeline_env, runs)
    n2, got = _solve(build, 128, pipeline_env, runs)
    assert n0 == 0 and n2 > 0, (n0, n2)
    for v in ref:
        np.testing.assert_array_equal(ref[v][0], got[v][0], err_msg=v)
        np.testing.assert_array_equal(ref[v][1], got[v][1], err_msg=v)


def test_updates_that_read_each_others_outputs_share_a_half(pipeline_env):
    """update i's relative proposal reads the belief slot update i+1 writes (within a pair): whichever order the halves run
    in, the proposal must see the OLD points -- the planner keeps such updates in the same half"""
    N, n = 100, 256
    rng = np.random.default_rng(3)
    res = []
    for pipe in (None, 128):
        pipeline_env(pipe)
        be = iif.HipBackend(N, 2 + 4 * n, 0)
        be.slot_write(0, abi.EUCLID2, rand_points(rng if pipe is None else np.random.default_rng(3), abi.EUCLID2, N, 0.0, 0.4))
        for i in range(n):  # belief slots 2 + i
            be.slot_write(2 + i, abi.EUCLID2, rand_points(np.random.default_rng(100 + i), abi.EUCLID2, N, float(i % 7), 0.5))
        props, prods = [], []
        for i in range(n):
            partner = i ^ 1  # the other update of the pair
            a, b = 2 + n + 2 * i, 2 + n + 2 * i + 1
            props.append(relative_factor_desc(abi.F_LINREL, abi.EUCLID2, 2, 1, [2 + partner, 2 + i], a, 500 + 2 * i, [1.0, 0.5], [0.1, 0.1]))
            props.append(relative_factor_desc(abi.F_PRIOR, abi.EUCLID2, 1, 0, [2 + i], b, 501 + 2 * i, [float(i % 7), 0.0], [0.3, 0.3]))
            prods.append(product_desc(abi.EUCLID2, [a, b], 2 + i, 9000 + i))
        prog = be.program([(abi.STAGE_PROPOSALS, props), (abi.STAGE_PRODUCTS, prods)] * 2 + [(abi.STAGE_COPIES, [])], lazy_bandwidth=True)
        n2 = prog.num_two_stream()
        assert n2 == (2 if pipe else 0)
        for _ in range(3):
            for i in range(n):
                be.slot_write(2 + i, abi.EUCLID2, rand_points(np.random.default_rng(100 + i), abi.EUCLID2, N, float(i % 7), 0.5))
            prog.run()
            be.synchronize()
        res.append([be.slot_read(2 + i, abi.EUCLID2) for i in range(n)])
        prog.close()
        be.close()
    for (p0, b0), (p1, b1) in zip(*res):
        np.testing.assert_array_equal(p0, p1)
        np.testing.assert_array_equal(b0, b1)
