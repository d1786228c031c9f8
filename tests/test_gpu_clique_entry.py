"""-m gpu: the per-clique C entry points (nbp_clique_upsolve / nbp_clique_downsolve, include/nbp_host.h): a host-side
stand-in for the CliqueStateMachine drives the tree one clique call at a time (tests/clique_csm.py) and must arrive at
the particles the whole-tree resident program (nbp_tree_compile) produces for the same seed -- bit for bit where the
launch geometries coincide (trees with fewer than 16 concurrent updates), to rounding otherwise."""
import numpy as np
import pytest

from clique_csm import solve_tree_by_clique_calls, solve_tree_by_clique_calls_joint, solve_tree_by_level_batches
from parity_utils import abi, assert_points_close, iif

pytestmark = pytest.mark.gpu


def _graphs():
    def kaess():
        return iif.generateGraph_Kaess(iif.SolverParams(N=100))

    def chain():
        return iif.generateChainEuclid(12, vardims=2, priorEvery=5, N=200)

    def doors():
        return iif.generateCircularDoors(nposes=8, N=128, sightEvery=4)

    def lattice():
        return iif.generateSE2Lattice(rows=2, cols=4, N=128, closeEvery=2)

    def alias():
        # tabulated measurements (AliasingScalarSampler: the table rides in nbp_clique_desc.factor_density[f])
        fg = iif.initfg(iif.SolverParams(N=128))
        bss = iif.AliasingScalarSampler([1.0, 1.5, 2.0, 6.0], [0.05, 0.4, 0.4, 0.15])
        for i in range(5):
            iif.addVariable(fg, f"x{i}", iif.ContinuousScalar)
        iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal(0.0, 0.1)))
        iif.addFactor(fg, ["x4"], iif.Mixture(iif.Prior, (iif.Normal(7.0, 0.5), iif.AliasingScalarSampler([6.0, 7.0, 8.0, 30.0], [1.0, 3.0, 2.0, 1.5])), [0.5, 0.5]))
        for i in range(4):
            iif.addFactor(fg, [f"x{i}", f"x{i + 1}"], iif.LinearRelative(bss if i % 2 == 0 else iif.Normal(1.5, 0.2)))
        return fg

    return {"kaess": kaess, "euclid2_chain": chain, "circular_doors_multihypo": doors, "se2_lattice": lattice, "alias_sampler_tables": alias}


@pytest.mark.parametrize("name", list(_graphs()))
def test_clique_calls_equal_whole_tree_program(hip_backend, name):
    build = _graphs()[name]
    fa, fb = build(), build()
    iif.initAll(fa, backend=hip_backend, seed=0)
    iif.initAll(fb, backend=hip_backend, seed=0)
    order = iif.nestedDissectionOrder(fa)
    tree = iif.buildTreeReset(fa, order)
    fa.solverParams.graphinit = fb.solverParams.graphinit = False
    iif.solveTree(fa, tree=iif.buildTreeReset(fa, order), backend=hip_backend, seed=77)
    be = hip_backend(fb.solverParams.N, 64)
    try:
        post, status = solve_tree_by_clique_calls(fb, tree, be, 77)
    finally:
        be.close()
    assert set(post) == set(fb.ls())
    roots = set(tree.roots)
    assert all(s == (3 if c in roots else 5) for c, s in status.items())  # UPSOLVED at the roots, DOWNSOLVED elsewhere
    nbit = 0
    for v in fa.ls():
        man = fa.getVariable(v).varType.manifold
        # (between clique calls a belief travels in its HOST form -- rotation matrices on SE(2), whose heading comes back from
        #  atan2(sin, cos) an ulp off now and then -- so this comparison keeps a tolerance; everywhere else it holds bit for bit)
        assert_points_close(man, fa.getVal(v), post[v].pts, rtol=1e-9 if man == abi.SE2 else 0, what=f"{name}:{v}")
        np.testing.assert_allclose(post[v].bw, fa.getVariable(v).bw, rtol=1e-9 if man == abi.SE2 else 0)
        nbit += int(np.array_equal(fa.getVal(v), post[v].pts))
        # infoPerCoord: the number of densities of the variable's last update, on every coordinate (ApproxConv.jl:277,298-303)
        D = abi.MANIFOLD_DIM[man]
        assert post[v].ipc.shape == (D,) and np.all(post[v].ipc >= 1.0) and np.all(post[v].ipc == post[v].ipc[0])
    # SE(2) beliefs cross the boundary as rotation matrices (ArrayPartition(t, R), what Julia holds): theta -> (cos, sin) ->
    # atan2 is not a bitwise round trip, so the clique-by-clique solve agrees to rounding (1e-9 above) there
    if name != "se2_lattice":
        assert nbit == len(fa.ls()), f"{nbit} of {len(fa.ls())} variables bit-identical"


@pytest.mark.parametrize("name", list(_graphs()) + ["euclid2_chain_60"])
def test_level_batches_equal_single_clique_calls(hip_backend, name):
    """nbp_clique_solve_batch: the cliques of a tree level in one call -- one transfer each way, one program -- deliver what
    one call per clique delivers.  Same ops, same seeds; only the size of the launches differs."""
    build = _graphs().get(name) or (lambda: iif.generateChainEuclid(60, vardims=2, priorEvery=10, N=100))
    fa = build()
    iif.initAll(fa, backend=hip_backend, seed=0)
    tree = iif.buildTreeReset(fa, iif.nestedDissectionOrder(fa))
    be = hip_backend(fa.solverParams.N, 1024)
    from iif_amd.native_host import clique_seam_times
    try:
        clique_seam_times(1)
        one, st1 = solve_tree_by_clique_calls(fa, tree, be, 55)
        t_one = clique_seam_times(2)  # from here on the calls wait for the device after their launches
        many, st2 = solve_tree_by_level_batches(fa, tree, be, 55)
        t_many = clique_seam_times(1)
    finally:
        be.close()
    # the seam's phase clock (nbp_clique_seam_times): one call per clique and direction against one per level and direction
    assert t_one["calls"] == 2 * len(tree.cliques) - len(tree.roots) and 0 < t_many["calls"] <= t_one["calls"]
    assert all(t_many[k] > 0 for k in ("planning_s", "beliefs_in_s", "assembly_s", "launches_s", "beliefs_out_s"))
    assert st1 == st2 and set(one) == set(many) == set(fa.ls())
    for v in fa.ls():
        man = fa.getVariable(v).varType.manifold
        assert_points_close(man, one[v].pts, many[v].pts, rtol=0, what=f"{name}:{v}")
        np.testing.assert_allclose(one[v].bw, many[v].bw, rtol=0)
        np.testing.assert_array_equal(one[v].ipc, many[v].ipc)
        if name != "se2_lattice":
            np.testing.assert_array_equal(one[v].pts, many[v].pts)


@pytest.mark.parametrize("name", ["euclid2_chain", "kaess", "circular_chain"])
def test_joint_messages_through_the_clique_entry(hip_backend, name):
    """SolverParams.useMsgLikelihoods: the differential factors of the children's messages arrive as measurement KDEs
    (nbp_clique_desc.factor_meas_kde), the clique's own are made by nbp_clique_upsolve_joint (approxDeconv + manikde!) --
    clique call by clique call the posteriors are those of the whole-tree program of the same mode, bit for bit"""
    build = {"euclid2_chain": lambda: iif.generateChainEuclid(14, vardims=2, priorEvery=6, N=128),
             "kaess": lambda: iif.generateGraph_Kaess(iif.SolverParams(N=100)),
             "circular_chain": lambda: iif.generateCircularDoors(nposes=10, N=128, sightEvery=100)}[name]
    fa, fb = build(), build()
    for f in (fa, fb):
        f.solverParams.useMsgLikelihoods = True
        iif.initAll(f, backend=hip_backend, seed=0)
        f.solverParams.graphinit = False
    order = iif.nestedDissectionOrder(fa)
    tree = iif.buildTreeReset(fa, order)
    iif.solveTree(fa, tree=iif.buildTreeReset(fa, order), backend=hip_backend, seed=91)
    be = hip_backend(fb.solverParams.N, 512)
    try:
        post, status = solve_tree_by_clique_calls_joint(fb, tree, be, 91)
        # ... and with the cliques of a level in one nbp_clique_solve_batch call (diff_out of every request)
        post2, status2 = solve_tree_by_clique_calls_joint(fb, tree, be, 91, batched=True)
    finally:
        be.close()
    assert set(post) == set(fb.ls()) and status2 == status
    for v in fb.ls():
        np.testing.assert_array_equal(post[v].pts, post2[v].pts, err_msg=f"batched {name}:{v}")
        np.testing.assert_array_equal(post[v].bw, post2[v].bw)
    from iif_amd import jointmsg
    ndiff = sum(len(j.relatives) for j in jointmsg.plan_joint_messages(fb, tree).values())
    assert ndiff > 0 or name == "kaess"  # differentials did travel (the Kaess graph's tree sends common priors only)
    for v in fa.ls():
        np.testing.assert_array_equal(fa.getVal(v), post[v].pts, err_msg=f"{name}:{v}")
        np.testing.assert_allclose(post[v].bw, fa.getVariable(v).bw, rtol=0)


def test_clique_entry_rejects_bad_input(hip_backend):
    from iif_amd.native_host import Belief, clique_solve
    fg = iif.generateChainEuclid(4, vardims=2, priorEvery=2, N=64)
    be = hip_backend(64, 8)
    try:
        bel = {v: Belief(abi.EUCLID2, np.zeros((64, 2)), np.ones(2)) for v in ("x0", "x1")}
        f = fg.getFactor(fg.ls("x0")[0])
        with pytest.raises(ValueError):  # unknown manifold code
            clique_solve(be, fg.solverParams, 1, ["x0", "x1"], 1, 1, [abi.EUCLID2, 9], [f], bel, 1, lists={"itervar": ["x0"]})
        with pytest.raises(ValueError):  # more frontals + separators than variables
            clique_solve(be, fg.solverParams, 1, ["x0", "x1"], 2, 1, [abi.EUCLID2] * 2, [f], bel, 1, lists={"itervar": ["x0"]})
        with pytest.raises(ValueError, match="created for N"):  # solver parameters for another particle count than the context's
            other = iif.SolverParams(N=32)
            clique_solve(be, other, 1, ["x0", "x1"], 1, 1, [abi.EUCLID2] * 2, [f], bel, 1, lists={"itervar": ["x0"]})
        with pytest.raises(ValueError):  # a context with too few slots for the clique
            small = hip_backend(64, 2)
            try:
                clique_solve(small, fg.solverParams, 1, ["x0", "x1"], 1, 1, [abi.EUCLID2] * 2, [f], bel, 1, lists={"itervar": ["x0"]})
            finally:
                small.close()
    finally:
        be.close()


def test_clique_batch_rejects_bad_input_and_takes_an_empty_batch(hip_backend):
    """nbp_clique_solve_batch: a bad request fails the call (no partial results, nothing crashes), a context that cannot hold
    the sum of the requests is refused, and a batch of zero requests is a no-op"""
    import ctypes as C
    from iif_amd import native_host
    from iif_amd.native_host import Belief, clique_solve_batch
    fg = iif.generateChainEuclid(4, vardims=2, priorEvery=2, N=64)
    be = hip_backend(64, 24)
    try:
        f = fg.getFactor(fg.ls("x0")[0])
        mk = lambda: {v: Belief(abi.EUCLID2, np.random.default_rng(1).normal(size=(64, 2)), np.ones(2)) for v in ("x0", "x1")}
        good = ((fg.solverParams, 1, ["x0", "x1"], 1, 1, [abi.EUCLID2] * 2, [f], mk(), 1), dict(lists={"itervar": ["x0"]}))
        bad = ((fg.solverParams, 2, ["x0", "x1"], 2, 1, [abi.EUCLID2] * 2, [f], mk(), 1), dict(lists={"itervar": ["x0"]}))
        assert clique_solve_batch(be, []) == []
        assert clique_solve_batch(be, [good]) == [3]  # NBP_CLIQ_UPSOLVED
        with pytest.raises(ValueError):
            clique_solve_batch(be, [good, bad])
        with pytest.raises(ValueError):  # 24 slots hold one of these cliques, not eight
            clique_solve_batch(be, [((fg.solverParams, k, ["x0", "x1"], 1, 1, [abi.EUCLID2] * 2, [f], mk(), 1),
                                     dict(lists={"itervar": ["x0"]})) for k in range(1, 9)])
        lib = native_host._lib()
        req = (native_host.CliqueRequestC * 1)()  # null params / clique
        assert lib.nbp_clique_solve_batch(be._ctx, req, 1) < 0
        assert lib.nbp_clique_solve_batch(None, req, 1) < 0
        assert lib.nbp_clique_solve_batch(be._ctx, None, 0) == 0
    finally:
        be.close()


def test_concurrent_single_clique_calls_on_one_context_are_merged_and_keep_their_own_errors(hip_backend):
    """nbp_clique_upsolve from several host threads on ONE context (include/nbp_host.h, "CONCURRENT CALLERS"): the library merges the
    calls into batches.  Eight Python threads (ctypes releases the GIL), five of them with good requests -- each must come out
    with the bytes of the lone call -- and three with solver parameters for another particle count, whose batch mates must not
    fail with them: every bad call raises ITS message, every good call succeeds."""
    import threading
    from iif_amd.native_host import Belief, clique_solve
    N = 64
    fg = iif.generateChainEuclid(4, vardims=2, priorEvery=2, N=N)
    f = fg.getFactor(fg.ls("x0")[0])
    rng = np.random.default_rng(5)
    start = {v: (rng.normal(size=(N, 2)) + i, np.full(2, 0.5)) for i, v in enumerate(("x0", "x1"))}

    def fresh():
        return {v: Belief(abi.EUCLID2, p.copy(), b.copy()) for v, (p, b) in start.items()}

    def call(be, sp, seed):
        bel = fresh()
        clique_solve(be, sp, 1, ["x0", "x1"], 1, 1, [abi.EUCLID2] * 2, [f], bel, seed, lists={"itervar": ["x0"]})
        return bel["x0"].pts.copy(), np.asarray(bel["x0"].bw).copy()

    be = hip_backend(N, 256)
    try:
        lone = {seed: call(be, fg.solverParams, seed) for seed in range(1, 6)}
        other = iif.SolverParams(N=32)
        problems, rounds = [], 25

        def good(seed):
            for _ in range(rounds):
                p, b = call(be, fg.solverParams, seed)
                if not (np.array_equal(p, lone[seed][0]) and np.array_equal(b, lone[seed][1])):
                    problems.append(f"seed {seed}: a merged call's result is not the lone call's")

        def bad():
            for _ in range(rounds):
                try:
                    call(be, other, 9)
                    problems.append("a call with parameters for N = 32 on a context for N = 64 was accepted")
                except ValueError as e:
                    if "created for N" not in str(e):
                        problems.append(f"a bad call raised somebody else's message: {e}")

        threads = [threading.Thread(target=good, args=(s,)) for s in range(1, 6)] + [threading.Thread(target=bad) for _ in range(3)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not problems, problems[:5]
    finally:
        be.close()
