"""CPU: the native (C++) host side (include/nbp_host.h) against this repo's Python implementation
(bayestree.py / solver.TreeProgram, itself pinned to the reference's known answers in
tests/test_tree_known_answers.py): identical elimination orders, cliques, potentials, Gibbs schedules,
slot plans and -- byte for byte -- identical stage descriptors.  No device call is made: compiling to a
resident program needs a context and is covered by tests/test_gpu_native_host.py."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import iif_amd_loader

iif = iif_amd_loader.load()
from iif_amd import native_host  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def graphs():
    E2 = iif.ContinuousEuclid(2)
    yield "chain", iif.generateChainEuclid(60, vardims=2, priorEvery=10, N=100)
    yield "kaess", iif.generateGraph_Kaess(iif.SolverParams(N=100))
    yield "linestep", iif.generateGraph_LineStep(8, landmarkPriorsAt=(0, 4), solverParams=iif.SolverParams(N=100))
    yield "lattice", iif.generateSE2Lattice(rows=4, cols=7, N=100, closeEvery=2)
    yield "doors", iif.generateCircularDoors(nposes=120, N=100, sightEvery=5)   # dense landmark nodes + multihypo
    yield "mixture", iif.generateMixtureChain(nvars=30, N=100, priorEvery=7)
    fg = iif.initfg(iif.SolverParams(N=100, gibbsIters=4, limitfixeddown=True))
    for v in ("x1", "x2", "x3"):
        iif.addVariable(fg, v, E2)
    iif.addFactor(fg, ["x1"], iif.Prior(iif.MvNormal(np.zeros(2), np.diag([0.01, 0.01]))))
    iif.addFactor(fg, ["x1"], iif.PartialPrior(E2, iif.Normal(2.0, 1.0), (1,)))
    iif.addFactor(fg, ["x1", "x2"], iif.PartialLinearRelative(E2, iif.Normal(10.0, 1.0), (2,)), nullhypo=0.1)
    iif.addFactor(fg, ["x2"], iif.PartialPrior(E2, iif.Normal(-20.0, 1.0), (1,)))
    iif.addFactor(fg, ["x2", "x3"], iif.LinearRelative(iif.MvNormal([1.0, 1.0], [0.1, 0.1])), inflation=3.0)
    fg.getVariable("x3").ismargin = True
    yield "partial", fg
    fg = iif.initfg(iif.SolverParams(N=100))
    for a in ("x0 x1 x2", "y0 y1 y2"):
        vs = a.split()
        for v in vs:
            iif.addVariable(fg, v, iif.ContinuousScalar)
        iif.addFactor(fg, [vs[0]], iif.Prior(iif.Normal(0, 1)))
        iif.addFactor(fg, [vs[0], vs[1]], iif.LinearRelative(iif.Normal(1, 1)))
        iif.addFactor(fg, [vs[1], vs[2]], iif.LinearRelative(iif.Normal(1, 1)))
    yield "forest", fg
    # pass-through priors (PartialPriorPassThrough): alone on x0, next to a prior on x2, a partial one on a Euclid(2) variable
    SE2 = iif.SpecialEuclidean2
    fg = iif.initfg(iif.SolverParams(N=100))
    rng = np.random.default_rng(0)
    for v in ("x0", "x1", "x2"):
        iif.addVariable(fg, v, SE2)
    iif.addVariable(fg, "l1", E2)
    iif.addFactor(fg, ["x0"], iif.PartialPriorPassThrough(SE2, rng.normal(size=(60, 2)), [0.3, 0.3], (1, 2)), nullhypo=0.2)
    z = iif.MvNormal([1.0, 0.0, 0.1], np.diag([0.01, 0.01, 0.01]))
    iif.addFactor(fg, ["x0", "x1"], iif.ManifoldFactor(z))
    iif.addFactor(fg, ["x1", "x2"], iif.ManifoldFactor(z))
    iif.addFactor(fg, ["x2"], iif.ManifoldPrior(np.zeros(3), iif.MvNormal(np.zeros(3), np.diag([0.01, 0.01, 0.01]))))
    iif.addFactor(fg, ["x2"], iif.PartialPriorPassThrough(SE2, rng.normal(size=(100, 3)), [0.3, 0.3, 0.1]), inflation=2.0)
    iif.addFactor(fg, ["l1"], iif.PartialPriorPassThrough(E2, rng.normal(size=(40, 1)), [0.2], (2,)))
    yield "passthrough", fg


GRAPHS = dict(graphs())


def mark_initialised(fg):
    for v in fg.ls():
        fg.getVariable(v).initialized = True


def test_clique_seam_clock_reads_and_resets():
    native_host.clique_seam_times(1)
    t = native_host.clique_seam_times(0)
    assert set(t) == {"planning_s", "beliefs_in_s", "assembly_s", "launches_s", "beliefs_out_s", "calls"}
    assert all(v == 0.0 for v in t.values())
    native_host.clique_seam_times(1)  # (mode 2 would make later clique calls of this process wait for the device)


def test_header_symbols_are_exported():
    hdr = open(os.path.join(ROOT, "include", "nbp_host.h")).read()
    declared = set(re.findall(r"\b(nbp_(?:graph|tree|clique|resident)_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(native_host.HOST_EXPORTS), declared ^ set(native_host.HOST_EXPORTS)
    lib = iif.abi.load_library()
    for n in declared:
        assert hasattr(lib, n), n


@pytest.mark.parametrize("name", list(GRAPHS))
def test_nested_dissection_order_is_identical(name):
    fg = GRAPHS[name]
    g = native_host.NativeGraph.from_fg(fg)
    assert g.order_nested_dissection() == iif.nestedDissectionOrder(fg)
    g.close()


@pytest.mark.parametrize("name", list(GRAPHS))
@pytest.mark.parametrize("ordering", ["nd", "qr", "natural"])
def test_tree_potentials_and_schedules_are_identical(name, ordering):
    fg = GRAPHS[name]
    order = {"nd": iif.nestedDissectionOrder, "qr": iif.getEliminationOrder, "natural": lambda f: f.ls()}[ordering](fg)
    tree = iif.buildTreeReset(fg, order)
    g = native_host.NativeGraph.from_fg(fg)
    nt = g.build_tree(order)
    assert nt.n_cliques == len(tree.cliques)
    from iif_amd import bayestree
    for k, c in tree.cliques.items():
        n = nt.clique(k)
        assert n["frontals"] == c.frontalIDs and n["separators"] == c.separatorIDs, k
        assert n["parent"] == max(c.parent, 0) and n["children"] == c.children, k
        assert n["potentials"] == c.potentials, k
        assert n["up"] == bayestree.upGibbsSchedule(c, fg.solverParams.gibbsIters), k
        down = bayestree.downSchedule(fg, c, fg.solverParams.gibbsIters) if c.parent >= 0 else []
        assert n["down"] == down, k
    nt.close()
    g.close()


@pytest.mark.parametrize("name", list(GRAPHS))
@pytest.mark.parametrize("snapshot", [False, True])
def test_compiled_stages_are_byte_identical(name, snapshot):
    fg = GRAPHS[name]
    mark_initialised(fg)
    order = iif.nestedDissectionOrder(fg)
    tree = iif.buildTreeReset(fg, order)
    tp = iif.TreeProgram(fg, tree, seed=12345, snapshot=snapshot)
    g = native_host.NativeGraph.from_fg(fg)
    nt = g.build_tree(order)
    assert nt.plan_slots(snapshot) == tp.n_slots
    assert nt.main == tp.main and nt.snap == tp.snap
    nt.schedule(12345)  # host half only: no device needed
    got = nt.stages()
    ctype = {iif.abi.STAGE_PROPOSALS: iif.abi.ProposalDesc, iif.abi.STAGE_PRODUCTS: iif.abi.ProductDesc,
             iif.abi.STAGE_COPIES: iif.abi.CopyDesc, iif.abi.STAGE_COPY_POINTS: iif.abi.CopyDesc}
    assert len(got) == len(tp.stages)
    for s, ((kind, raw), (pk, descs)) in enumerate(zip(got, tp.stages)):
        assert kind == pk, s
        want = bytes((ctype[pk] * len(descs))(*descs)) if descs else b""
        assert raw == want, (s, kind, len(descs))
    st, ps = nt.stats(), tp.stats()
    for k in ("stages", "proposals", "products", "updates_up", "updates_down", "messages", "slots", "alg_bytes"):
        assert st[k] == ps[k], k
    alg = tp.alg_bytes_by_kernel()
    assert st["alg_bytes_proposal"] == alg["nbp_proposal_kernel"] and st["alg_bytes_product"] == alg["nbp_product_kernel"]
    nt.close()
    g.close()


@pytest.mark.parametrize("name", list(GRAPHS))
def test_graph_init_plan_and_stages_are_byte_identical(name):
    """initAll!: same initialisation order, same batching into stages, same descriptors (solver.initStages)"""
    fg = GRAPHS[name]
    for v in fg.ls():
        fg.getVariable(v).initialized = False
    plan, slot, n_slots, stages = iif.solver.initStages(fg, seed=4242)
    g = native_host.NativeGraph.from_fg(fg)
    need, planned = g.init_plan(4242)
    assert planned == [p[0] for p in plan]
    assert need == n_slots
    got = g.init_stages()
    ctype = {iif.abi.STAGE_PROPOSALS: iif.abi.ProposalDesc, iif.abi.STAGE_PRODUCTS: iif.abi.ProductDesc}
    assert len(got) == len(stages)
    for s, ((kind, raw), (pk, descs)) in enumerate(zip(got, stages)):
        assert kind == pk and raw == bytes((ctype[pk] * len(descs))(*descs)), s
    g.close()
    mark_initialised(fg)


def random_graph(seed):
    """random sparse graph: a spanning tree of relative factors plus loop closures, priors, multihypo
    triples, mixtures and nullhypo, on a random manifold"""
    r = np.random.default_rng(seed)
    kind = r.integers(0, 3)
    vt, rel, pri = [(iif.ContinuousScalar, lambda: iif.LinearRelative(iif.Normal(1.0, 0.1)), lambda: iif.Prior(iif.Normal(0.0, 1.0))),
                    (iif.ContinuousEuclid(2), lambda: iif.LinearRelative(iif.MvNormal([1.0, 0.0], [0.1, 0.1])),
                     lambda: iif.Prior(iif.MvNormal(np.zeros(2), [1.0, 1.0]))),
                    (iif.Circular, lambda: iif.CircularCircular(iif.Normal(0.3, 0.1)), lambda: iif.PriorCircular(iif.Normal(0.0, 0.2)))][kind]
    n = int(r.integers(4, 40))
    fg = iif.initfg(iif.SolverParams(N=64, gibbsIters=int(r.integers(1, 5))))
    for i in range(n):
        iif.addVariable(fg, f"v{i}", vt)
    iif.addFactor(fg, ["v0"], pri())
    for i in range(1, n):
        j = int(r.integers(max(0, i - 6), i))
        nh = 0.1 if r.random() < 0.15 else 0.0
        if kind == 0 and r.random() < 0.15:
            iif.addFactor(fg, [f"v{j}", f"v{i}"], iif.Mixture(iif.LinearRelative, (iif.Normal(1.0, 0.1), iif.Normal(2.0, 0.5)), [0.7, 0.3]))
        else:
            iif.addFactor(fg, [f"v{j}", f"v{i}"], rel(), nullhypo=nh)
    for _ in range(int(r.integers(0, n // 3 + 1))):  # loop closures / extra priors / multihypo sightings
        a, b, c = (int(x) for x in r.choice(n, size=3, replace=False))
        u = r.random()
        if u < 0.4:
            iif.addFactor(fg, [f"v{a}", f"v{b}"], rel())
        elif u < 0.6:
            iif.addFactor(fg, [f"v{a}"], pri())
        else:
            iif.addFactor(fg, [f"v{a}", f"v{b}", f"v{c}"], rel(), multihypo=[1.0, 0.5, 0.5])
    if r.random() < 0.3:
        fg.getVariable(f"v{int(r.integers(0, n))}").ismargin = True
    return fg


@pytest.mark.parametrize("seed", range(60))
def test_random_graphs_native_equals_python(seed):
    fg = random_graph(seed)
    # every third graph also exercises the non-default message / measurement modes of the compiler
    fg.solverParams.useMsgLikelihoods = seed % 3 == 1
    fg.solverParams.alwaysFreshMeasurements = seed % 3 != 2
    g = native_host.NativeGraph.from_fg(fg)
    # graph initialisation plan from the uninitialised graph
    plan, slot, n_slots, stages = iif.solver.initStages(fg, seed=seed)
    need, planned = g.init_plan(seed)
    assert planned == [p[0] for p in plan] and need == n_slots
    ctype = {iif.abi.STAGE_PROPOSALS: iif.abi.ProposalDesc, iif.abi.STAGE_PRODUCTS: iif.abi.ProductDesc,
             iif.abi.STAGE_COPIES: iif.abi.CopyDesc, iif.abi.STAGE_DECONV: iif.abi.ProposalDesc,
             iif.abi.STAGE_COPY_POINTS: iif.abi.CopyDesc}
    for (kind, raw), (pk, descs) in zip(g.init_stages(), stages):
        assert kind == pk and raw == bytes((ctype[pk] * len(descs))(*descs))
    g.close()
    # whole-tree program from the initialised graph, two orderings
    mark_initialised(fg)
    g = native_host.NativeGraph.from_fg(fg)
    nd = iif.nestedDissectionOrder(fg)
    assert g.order_nested_dissection() == nd
    for order in (nd, iif.getEliminationOrder(fg)):
        tree = iif.buildTreeReset(fg, order)
        try:
            tp = iif.TreeProgram(fg, tree, seed=seed, snapshot=bool(seed % 2))
        except ValueError:
            continue  # a product wider than NBP_MAXF: both sides refuse
        nt = g.build_tree(order)
        assert nt.plan_slots(bool(seed % 2)) == tp.n_slots
        nt.schedule(seed)
        got = nt.stages()
        assert len(got) == len(tp.stages)
        for s, ((kind, raw), (pk, descs)) in enumerate(zip(got, tp.stages)):
            assert kind == pk, s
            assert raw == (bytes((ctype[pk] * len(descs))(*descs)) if descs else b""), (s, kind)
        nt.close()
    g.close()


@pytest.mark.parametrize("name", list(GRAPHS))
@pytest.mark.parametrize("world", [2, 3, 8])
def test_partition_and_sharded_compile_are_identical(name, world):
    """the multi-rank compile of the native host (nbp_tree_partition / nbp_tree_set_owner) against the Python mirror
    (dist_solver.partition_cliques, solver.TreeProgram(owner=, rank=)): owners, slots, stage bytes and exchange segments
    of EVERY rank"""
    from iif_amd import bayestree
    from iif_amd.dist_solver import partition_cliques
    fg = GRAPHS[name]
    mark_initialised(fg)
    order = iif.nestedDissectionOrder(fg)
    tree = iif.buildTreeReset(fg, order)
    gi = fg.solverParams.gibbsIters
    owner = partition_cliques(tree, world, weight=lambda c: 1 + len(bayestree.upGibbsSchedule(tree.cliques[c], gi)))
    g = native_host.NativeGraph.from_fg(fg)
    nt = g.build_tree(order)
    assert nt.partition(world) == owner
    ctype = {iif.abi.STAGE_PROPOSALS: iif.abi.ProposalDesc, iif.abi.STAGE_PRODUCTS: iif.abi.ProductDesc,
             iif.abi.STAGE_COPIES: iif.abi.CopyDesc, iif.abi.STAGE_COPY_POINTS: iif.abi.CopyDesc, iif.abi.STAGE_DECONV: iif.abi.ProposalDesc}
    for rank in range(world):
        tp = iif.TreeProgram(fg, tree, seed=777, snapshot=True, owner=owner, rank=rank)
        nt.set_owner(owner, rank)
        assert nt.plan_slots(True) == tp.n_slots, rank
        nt.schedule(777)
        got = nt.stages()
        assert len(got) == len(tp.stages), (rank, len(got), len(tp.stages))
        for s, ((kind, raw), (pk, descs)) in enumerate(zip(got, tp.stages)):
            assert kind == pk, (rank, s)
            want = bytes((ctype[pk] * len(descs))(*descs)) if descs else b""
            assert raw == want, (rank, s, kind, len(descs))
        assert nt.segments() == [tuple(x) if x[0] == "run" else (x[0], list(x[1]), list(x[2])) for x in tp.segments], rank
    nt.set_owner(None, 0)
    nt.close()
    g.close()
