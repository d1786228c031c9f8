"""useMsgLikelihoods: the symbolic plan of the joint upward messages (jointmsg.py) against the structural
assertions of the reference's own tests (known answers), and the device-side pieces on the oracle."""
import numpy as np

from parity_utils import iif
from oracle.oracle_backend import OracleBackend
from iif_amd import abi, jointmsg


def _square(second_type):
    # test/testJointEnforcement.jl:13-35 and :130-147
    fg = iif.initfg(iif.SolverParams(N=100))
    E2 = iif.ContinuousEuclid(2)
    r = np.random.default_rng(0)
    for i in range(3):
        iif.addVariable(fg, f"x{i}", E2)
        iif.initVariable(fg, f"x{i}", r.normal(size=(100, 2)) + 10 * i, backend=OracleBackend)
    Z = iif.MvNormal([10.0, 10.0], np.eye(2))
    iif.addFactor(fg, ["x0", "x1"], iif.LinearRelative(Z))
    iif.addFactor(fg, ["x1", "x2"], iif.LinearRelative(Z))
    iif.addVariable(fg, "x3", E2)
    if second_type == "EuclidDistance":
        iif.addFactor(fg, ["x2", "x3"], iif.EuclidDistance(iif.Normal(10, 1)))
        iif.addFactor(fg, ["x0", "x3"], iif.EuclidDistance(iif.Normal(30, 1)))
    else:
        iif.addFactor(fg, ["x2", "x3"], iif.LinearRelative(Z))
        iif.addFactor(fg, ["x0", "x3"], iif.LinearRelative(Z))
    iif.initAll(fg, backend=OracleBackend)
    tree = iif.buildTreeReset(fg, ["x3", "x1", "x2", "x0"])
    cid = [c for c, cl in tree.cliques.items() if "x3" in cl.frontalIDs][0]
    return fg, tree, cid


def test_disjoint_clique_joint_sends_priors_only():
    # test/testJointEnforcement.jl:84-92, 116-118: the path x0 - x3 - x2 is homogeneous but of the wrong type
    fg, tree, cid = _square("EuclidDistance")
    cl = tree.cliques[cid]
    assert sorted(cl.separatorIDs) == ["x0", "x2"]
    J = jointmsg.plan_joint_messages(fg, tree)[cid]
    hom, types = jointmsg.isPathFactorsHomogeneous(cl.allIDs, J.factors, "x0", "x2")
    assert hom and types == ["EuclidDistance"]
    assert J.relatives == [] and sorted(J.priors) == ["x0", "x2"]


def test_homogeneous_clique_joint_sends_one_relative():
    # test/testJointEnforcement.jl:176-213: one LinearRelative differential between x0 and x2, one class, no
    # prior (the clique holds no prior potential)
    fg, tree, cid = _square("LinearRelative")
    J = jointmsg.plan_joint_messages(fg, tree)[cid]
    assert len(J.relatives) == 1 and sorted(J.relatives[0][:2]) == ["x0", "x2"] and J.relatives[0][3] == abi.F_LINREL
    assert J.priors == [] and not J.hasPriors
    classes = jointmsg._find_subgraph_classes(fg, tree.cliques[cid], J.factors, J.relatives)
    assert len(classes) == 1 and sorted(list(classes.values())[0]) == ["x0", "x2"]


def _caesar_ring():
    fg = iif.initfg(iif.SolverParams(N=100, useMsgLikelihoods=True))
    for v in ["x0", "x1", "x2", "x3", "x4", "x5", "x6"]:
        iif.addVariable(fg, v, iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal()))
    for a, b in [("x0", "x1"), ("x1", "x2"), ("x2", "x3"), ("x3", "x4"), ("x4", "x5"), ("x5", "x6")]:
        iif.addFactor(fg, [a, b], iif.LinearRelative(iif.Normal()))
    iif.addVariable(fg, "l1", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0", "l1"], iif.LinearRelative(iif.Normal()))
    iif.addFactor(fg, ["x6", "l1"], iif.LinearRelative(iif.Normal()))
    return fg


def test_caesar_ring_parent_receives_two_differentials():
    # test/testUseMsgLikelihoods.jl:44-75: clique 2 = {x6 | x4, x0} has three variables and no factor of its own;
    # the messages of cliques 4 and 5 add exactly two factors, one of them a LinearRelative(::MKD) on (x0, x6)
    fg = _caesar_ring()
    iif.initAll(fg, backend=OracleBackend)
    tree = iif.buildTreeReset(fg, ["x3", "x5", "l1", "x1", "x6", "x4", "x2", "x0"])
    cl = tree.cliques[2]
    assert sorted(cl.allIDs) == ["x0", "x4", "x6"] and cl.potentials == []
    J = jointmsg.plan_joint_messages(fg, tree)
    assert sorted(tree.cliques[2].children) == [4, 5]
    assert [f.tag for f in J[2].factors] == ["d", "d"]
    assert any(sorted(f.variables) == ["x0", "x6"] and f.typename == "LinearRelative" for f in J[2].factors)
    # the root holds the only prior; no clique below it has one, so no common prior travels (hasPriors)
    assert J[1].hasPriors and not any(J[c].hasPriors for c in J if c != 1)
    # the compiled program: one deconvolution per differential, proposals that name its KDE
    tp = iif.TreeProgram(fg, tree, seed=3)
    ndec = sum(len(d) for k, d in tp.stages if k == abi.STAGE_DECONV)
    assert ndec == sum(len(J[c].relatives) for c in J) == 5
    kde = {d.meas_kde - 1 for k, ds in tp.stages if k == abi.STAGE_PROPOSALS for d in ds if d.meas_kde}
    assert kde == set(tp.D.values())


def test_differential_factor_roundtrip_on_the_oracle():
    # the numeric half: deconvolution of two beliefs that differ by a known offset, then a proposal through the
    # factor whose measurement is that KDE reproduces the second belief (LinearRelative(::MKD))
    N = 200
    r = np.random.default_rng(5)
    man = abi.EUCLID2
    be = OracleBackend(N, 5)
    a = r.normal(size=(N, 2)) * 0.3
    b = a + np.array([4.0, -2.0]) + r.normal(size=(N, 2)) * 0.05
    be.slot_write(0, man, a, np.ones(2))
    be.slot_write(1, man, b, np.ones(2))
    be.run_bandwidth([0, 1], [man, man])
    fg = iif.initfg(iif.SolverParams(N=N))
    E2 = iif.ContinuousEuclid(2)
    iif.addVariable(fg, "a", E2)
    iif.addVariable(fg, "b", E2)
    from iif_amd.solver import _default_relative
    from iif_amd.factorgraph import DFGFactor, DifferentialRelative
    dummy = DFGFactor("dummy", ["a", "b"], _default_relative(abi.F_LINREL, E2), None, 0.0, 5.0)
    slot = {"a": 0, "b": 1}
    dec = iif.proposal_desc(fg, dummy, "b", lambda v: slot[v], 2, 77)
    diff = DFGFactor("diff", ["a", "b"], DifferentialRelative(abi.F_LINREL, 2), None, 0.0, 5.0)
    prop = iif.proposal_desc(fg, diff, "b", lambda v: slot[v], 3, 78)
    assert prop.meas_kde == 3
    prog = be.program([(abi.STAGE_DECONV, [dec]), (abi.STAGE_PROPOSALS, [prop])])
    prog.run()
    z, zbw = be.slot_read(2, man)
    assert np.allclose(z, b - a, atol=2e-3)  # the inverse of r = z - (b - a), to the Nelder-Mead tolerance
    assert (zbw > 0).all()
    out, _ = be.slot_read(3, man)
    assert np.abs(out.mean(axis=0) - b.mean(axis=0)).max() < 0.15
    be.close()


def test_batching_by_dependency_or_by_level_gives_the_same_posteriors():
    # the random streams are keyed by (pass, clique, step): how cliques are batched into stages changes nothing
    from test_native_host import mark_initialised, random_graph
    for seed in (1, 4, 8, 11):  # plain, joint-message and stored-measurement graphs
        fg = random_graph(seed)
        fg.solverParams.useMsgLikelihoods = seed % 3 == 1
        fg.solverParams.alwaysFreshMeasurements = seed % 3 != 2
        mark_initialised(fg)
        tree = iif.buildTreeReset(fg, iif.nestedDissectionOrder(fg))
        res = []
        for asap in (True, False):
            tp = iif.TreeProgram(fg, tree, seed=seed, asap=asap)
            be = OracleBackend(fg.solverParams.N, tp.n_slots, 0, threads=4)
            for v in fg.ls():
                var = fg.getVariable(v)
                be.slot_write(tp.main[v], var.varType.manifold, var.val, var.bw)
            be.program(tp.stages).run()
            res.append({v: be.slot_read(tp.main[v], fg.getVariable(v).varType.manifold) for v in fg.ls()})
            be.close()
        n_dep = sum(1 for k, _ in iif.TreeProgram(fg, tree, seed=seed, asap=True).stages if k == abi.STAGE_PRODUCTS)
        n_lev = sum(1 for k, _ in iif.TreeProgram(fg, tree, seed=seed, asap=False).stages if k == abi.STAGE_PRODUCTS)
        assert n_dep <= n_lev
        for v in fg.ls():
            np.testing.assert_array_equal(res[0][v][0], res[1][v][0])
            np.testing.assert_array_equal(res[0][v][1], res[1][v][1])
