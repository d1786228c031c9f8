"""CPU: exact (integer / symbolic) known answers of the reference's own tests for the Bayes tree and the
clique-factor assignment: given the same elimination order, clique frontals,
separators, parent/child structure and the factors each clique owns must be identical."""
import iif_amd_loader

iif = iif_amd_loader.load()
S = iif.ContinuousScalar
LR = lambda: iif.LinearRelative(iif.Normal(0.0, 1.0))  # noqa: E731
PR = lambda: iif.Prior(iif.Normal(0.0, 1.0))  # noqa: E731


def clique_of(tree, v):
    return next(c for c in tree.cliques.values() if v in c.frontalIDs)


def children(tree, c):
    return [tree.cliques[k] for k in c.children]


def caesar_ring_1d():
    # CanonicalGraphExamples.jl:123-147
    fg = iif.initfg()
    for i in range(7):
        iif.addVariable(fg, f"x{i}", S)
    iif.addFactor(fg, ["x0"], PR())
    for i in range(6):
        iif.addFactor(fg, [f"x{i}", f"x{i+1}"], LR())
    iif.addVariable(fg, "l1", S)
    iif.addFactor(fg, ["x0", "l1"], LR())
    iif.addFactor(fg, ["x6", "l1"], LR())
    return fg


def test_elimination_order_is_kept():
    # test/testJunctionTreeConstruction.jl:8-17
    fg = iif.generateGraph_Kaess()
    vo = ["l1", "l2", "x1", "x2", "x3"]
    tree = iif.buildTreeReset(fg, vo)
    assert tree.eliminationOrder == vo


def test_caesar_ring_1d_symbolic_tree():
    # test/testJunctionTreeConstruction.jl:19-66
    fg = caesar_ring_1d()
    tree = iif.buildTreeReset(fg, ["x0", "x2", "x4", "x6", "x1", "l1", "x5", "x3"])
    assert len(tree.cliques) == 6
    C0 = clique_of(tree, "x3")
    assert set(C0.frontalIDs) == {"x3", "x5", "l1"} and C0.separatorIDs == [] and C0.parent < 0
    cC0 = children(tree, C0)
    assert len(cC0) == 3
    C1 = clique_of(tree, "x1")
    assert C1 in cC0 and C1.frontalIDs == ["x1"] and set(C1.separatorIDs) == {"x3", "l1"}
    cC1 = children(tree, C1)
    assert len(cC1) == 2
    C4 = clique_of(tree, "x2")
    assert C4 in cC1 and C4.frontalIDs == ["x2"] and set(C4.separatorIDs) == {"x3", "x1"}
    C5 = clique_of(tree, "x0")
    assert C5 in cC1 and C5.frontalIDs == ["x0"] and set(C5.separatorIDs) == {"l1", "x1"}
    C2 = clique_of(tree, "x6")
    assert C2 in cC0 and C2.frontalIDs == ["x6"] and set(C2.separatorIDs) == {"l1", "x5"}
    C3 = clique_of(tree, "x4")
    assert C3 in cC0 and C3.frontalIDs == ["x4"] and set(C3.separatorIDs) == {"x3", "x5"}


def test_kaess_tree_listing():
    # test/testTreeFunctions.jl:52-80
    fg = iif.generateGraph_Kaess()
    tree = iif.buildTreeReset(fg, ["l2", "l1", "x1", "x2", "x3"])
    assert len(tree.cliques) == 3
    root = clique_of(tree, "x3")
    assert root.parent < 0 and set(root.frontalIDs) == {"x3", "x2"} and len(root.children) == 2
    assert set(clique_of(tree, "x1").frontalIDs) == {"x1", "l1"}
    assert clique_of(tree, "l2").frontalIDs == ["l2"]


def test_clique_factors_458_example_1():
    # test/testCliqueFactors.jl:9-98
    fg = iif.initfg()
    for v in ["x0", "x1", "x2", "x3", "x4", "l0", "l1"]:
        iif.addVariable(fg, v, S)
    for a, b in [("x0", "x1"), ("x1", "x2"), ("x2", "x3"), ("x3", "x4"), ("x0", "l0"), ("x2", "l0"), ("x0", "l1"), ("x2", "l1")]:
        iif.addFactor(fg, [a, b], LR())
    iif.addFactor(fg, ["x0"], PR())
    iif.addFactor(fg, ["l0"], PR())
    tree = iif.buildTreeReset(fg, ["x2", "x0", "l0", "x3", "x1", "l1", "x4"])
    fr = [set(clique_of(tree, v).frontalIDs) for v in ("x0", "l0", "x4")]
    assert not (fr[0] & fr[1]) and not (fr[1] & fr[2]) and not (fr[0] & fr[2])
    assert fr[0] | fr[1] | fr[2] == set(fg.ls())
    C3, C2, C1 = clique_of(tree, "x0"), clique_of(tree, "l0"), clique_of(tree, "x4")
    assert set(C3.allIDs) == {"x0", "x1", "l0", "l1"}
    assert set(C3.potentials) == {"x0l0f1", "x0l1f1", "x0x1f1", "x0f1"}
    assert set(C2.allIDs) == {"x3", "x2", "x1", "l0", "l1"}
    assert set(C2.potentials) == {"x1x2f1", "x2x3f1", "x2l0f1", "x2l1f1", "l0f1"}
    assert set(C1.allIDs) == {"x3", "x4", "x1", "l1"}
    assert set(C1.potentials) == {"x3x4f1"}
    assert set(C1.potentials) | set(C2.potentials) | set(C3.potentials) == set(fg.lsf())


def test_clique_factors_458_example_2():
    # test/testCliqueFactors.jl:102-163
    fg = iif.initfg()
    for v in ["x0", "x1", "x2", "x3", "lm0", "lm3"]:
        iif.addVariable(fg, v, S)
    for a, b in [("x0", "x1"), ("x1", "x2"), ("x2", "x3"), ("x0", "lm0"), ("x1", "lm0"), ("x2", "lm3"), ("x3", "lm3")]:
        iif.addFactor(fg, [a, b], LR())
    tree = iif.buildTreeReset(fg, ["x0", "x2", "x1", "lm3", "lm0", "x3"])
    C1, C2, C3 = (set(clique_of(tree, v).potentials) for v in ("x3", "x2", "x0"))
    assert C1 == {"x1lm0f1", "x3lm3f1"}
    assert C2 == {"x1x2f1", "x2x3f1", "x2lm3f1"}
    assert C3 == {"x0x1f1", "x0lm0f1"}
    assert C1 | C2 | C3 == set(fg.lsf())


def test_linestep_clique_frontals_separators_potentials():
    # test/testCliqueFactors.jl:167-204: generateGraph_LineStep(4, landmarkPriorsAt=[0,4]) with the DEFAULT
    # (QR of the biadjacency matrix) ordering: two cliques with exactly these members
    fg = iif.generateGraph_LineStep(4, landmarkPriorsAt=(0, 4))
    tree = iif.buildTreeReset(fg, iif.getEliminationOrder(fg))
    assert len(tree.cliques) == 2
    c1, c2 = tree.cliques[1], tree.cliques[2]
    assert set(c1.frontalIDs) == {"x0", "lm0", "x2"} and c1.separatorIDs == []
    assert set(c1.potentials) == {"lm0f1", "x0x2f1", "x0lm0f1", "x0f1", "x2lm0f1"}
    assert set(c2.frontalIDs) == {"x4", "lm4"} and c2.separatorIDs == ["x2"]
    assert set(c2.potentials) == {"x2lm4f1", "x2x4f1", "x4lm4f1", "lm4f1"}
    assert clique_of(tree, "x2") is c1 and c2.parent == 1


def test_isam2_example_has_three_cliques():
    # test/testBayesTreeiSAM2Example.jl:5-52: the iSAM2 paper example, order [l1, l2, x1, x2, x3] -> 3 cliques
    fg = iif.initfg(iif.SolverParams(N=100))
    iif.addVariable(fg, "x1", iif.ContinuousScalar)
    iif.addFactor(fg, ["x1"], iif.Prior(iif.Normal()))
    iif.addVariable(fg, "x2", iif.ContinuousScalar)
    iif.addFactor(fg, ["x1", "x2"], iif.LinearRelative(iif.Normal()))
    iif.addVariable(fg, "x3", iif.ContinuousScalar)
    iif.addFactor(fg, ["x2", "x3"], iif.LinearRelative(iif.Normal()))
    iif.addVariable(fg, "l1", iif.ContinuousScalar)
    iif.addFactor(fg, ["x1", "l1"], iif.LinearRelative(iif.Normal()))
    iif.addFactor(fg, ["x2", "l1"], iif.LinearRelative(iif.Normal()))
    iif.addVariable(fg, "l2", iif.ContinuousScalar)
    iif.addFactor(fg, ["x3", "l2"], iif.LinearRelative(iif.Normal()))
    tree = iif.buildTreeReset(fg, ["l1", "l2", "x1", "x2", "x3"])
    assert len(tree.cliques) == 3
    # every factor is the potential of exactly one clique
    pots = [f for cl in tree.cliques.values() for f in cl.potentials]
    assert sorted(pots) == sorted(fg.lsf())


def test_tree_message_utils_structure():
    # test/testTreeMessageUtils.jl:6-43: LineStep(8) with only the two end sightings of lm0, fixed elimination
    # order: cliques 2..8 send up messages, clique 2's message carries {x0, x4}, the messages stacked by
    # variable cover {lm0, x0, x2, x4, x6, x7}, x0 is in three of them, x4 in those of cliques (depth)
    # 4 (2), 6 (3), 2 (1), 8 (1); clique 7 is a child of clique 3
    N = 8
    fg = iif.generateGraph_LineStep(N, poseEvery=1, landmarkEvery=N + 1, posePriorsAt=(0,), landmarkPriorsAt=(), sightDistance=N + 1)
    for i in range(1, N):
        iif.deleteFactor(fg, f"x{i}lm0f1")
    tree = iif.buildTreeReset(fg, ["x3", "x8", "x5", "x1", "x6", "lm0", "x7", "x4", "x2", "x0"])
    senders = sorted(c for c, cl in tree.cliques.items() if cl.parent >= 0)
    assert senders == list(range(2, 9))
    assert sorted(tree.cliques[2].separatorIDs) == ["x0", "x4"]
    stacked = {}
    depths = tree.depths()
    for c in senders:
        for v in tree.cliques[c].separatorIDs:
            stacked.setdefault(v, []).append((c, depths[c]))
    assert sorted(stacked) == ["lm0", "x0", "x2", "x4", "x6", "x7"]
    assert len(stacked["x0"]) == 3
    assert sorted(stacked["x4"]) == sorted([(4, 2), (6, 3), (2, 1), (8, 1)])
    assert tree.cliques[7].parent == 3
    # the native host builds the same tree
    from iif_amd import native_host
    g = native_host.NativeGraph.from_fg(fg)
    nt = g.build_tree(tree.eliminationOrder)
    assert nt.n_cliques == 8
    for c in range(1, 9):
        info = nt.clique(c)
        assert sorted(info["frontals"]) == sorted(tree.cliques[c].frontalIDs)
        assert sorted(info["separators"]) == sorted(tree.cliques[c].separatorIDs)
        assert info["parent"] == max(tree.cliques[c].parent, 0)


def test_is_partial():
    # test/testPartialFactors.jl:6-25
    fg = iif.initfg()
    iif.addVariable(fg, "x0", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal()))
    assert not iif.isPartial(fg.getFactor("x0f1"))
    fg = iif.initfg()
    E2 = iif.ContinuousEuclid(2)
    iif.addVariable(fg, "x1", E2)
    iif.addFactor(fg, ["x1"], iif.PartialPrior(E2, iif.Normal(), (1,)))
    assert iif.isPartial(fg.getFactor("x1f1"))
