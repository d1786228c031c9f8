"""A minimal stand-in for the reference's CliqueStateMachine in the tests: it walks the Bayes tree on the host and
makes, per clique, exactly the two calls the CSM makes -- upGibbsCliqueDensity (SolveTree.jl:164-239, called at
CliqStateMachineUtils.jl:375-385) and solveCliqDownFrontalProducts! (CliqStateMachineUtils.jl:479-571) -- through the
per-clique C entry points nbp_clique_upsolve / nbp_clique_downsolve.  Message assembly (separator beliefs up, parent
values down, frontals back to the graph) is done here in numpy, as the Julia CSM does it in Julia."""
from parity_utils import iif

from iif_amd.native_host import Belief, clique_solve, clique_solve_batch


def solve_tree_by_clique_calls(fg, tree, backend, seed):
    """-> {label: Belief} posteriors after one up + one down pass; `fg` must be initialised; `backend`: a HipBackend"""
    sp = fg.solverParams
    man = {v: fg.getVariable(v).varType.manifold for v in fg.ls()}
    marg = {v: fg.getVariable(v).ismargin for v in fg.ls()}
    main = {v: Belief(man[v], fg.getVariable(v).val, fg.getVariable(v).bw) for v in fg.ls()}
    sub = {}      # clique -> {label: Belief}: the clique sub graph (deep copy, SubGraphFunctions.jl:48)
    status = {}
    # ---- up pass: children before parents ---------------------------------------------------------------------
    for cid in tree.postorder():
        cl = tree.cliques[cid]
        labels = list(cl.frontalIDs) + list(cl.separatorIDs)
        bel = {v: main[v].copy() for v in labels}
        msgs = [(v, sub[ch][v]) for ch in cl.children for v in tree.cliques[ch].separatorIDs]  # prepCliqueMsgUp of the children
        lists = {"directFrtlMsg": cl.directFrtlMsgIDs, "msgskip": cl.msgskipIDs, "itervar": cl.itervarIDs,
                 "directPriorMsg": cl.directPriorMsgIDs}
        status[cid] = clique_solve(backend, sp, cid, labels, len(cl.frontalIDs), len(cl.separatorIDs), [man[v] for v in labels],
                                   [fg.getFactor(f) for f in cl.potentials], bel, seed, down=False,
                                   ismargin=[marg[v] for v in labels], lists=lists, msgs=msgs)
        sub[cid] = bel
    # ---- roots: the up-solved frontals are the posterior (CliqueStateMachine.jl, preDownSolve root branch) ----------
    post = {}
    for r in tree.roots:
        for v in tree.cliques[r].frontalIDs:
            main[v] = sub[r][v].copy()
            post[v] = main[v]
    # ---- down pass: parents before children ---------------------------------------------------------------------
    depths = tree.depths()
    for cid in sorted(tree.cliques, key=lambda c: (depths[c], c)):
        cl = tree.cliques[cid]
        if cl.parent < 0:
            continue
        for s in cl.separatorIDs:  # updateSubFgFromDownMsgs! (TreeMessageUtils.jl:66-84): the parent's values
            sub[cid][s].pts[:] = sub[cl.parent][s].pts
        factors = []
        for v in cl.frontalIDs:  # every factor of the frontals (addDownVariableFactors!, CliqueStateMachine.jl:823-835)
            for f in fg.ls(v):
                if f not in factors:
                    factors.append(f)
        inclq = list(cl.frontalIDs) + list(cl.separatorIDs)
        others = []
        for f in factors:
            for u in fg.getFactor(f).variables:
                if u not in inclq and u not in others:
                    others.append(u)
        labels = inclq + others
        bel = {v: (sub[cid][v] if v in sub[cid] else main[v].copy()) for v in labels}
        # the factor order of a variable's product is the graph's (fg.ls(v)), as in the whole-tree compile
        order = [f for f in fg.lsf() if f in factors]
        status[cid] = clique_solve(backend, sp, cid, labels, len(cl.frontalIDs), len(cl.separatorIDs), [man[v] for v in labels],
                                   [fg.getFactor(f) for f in order], bel, seed, down=True, ismargin=[marg[v] for v in labels])
        for v in cl.frontalIDs:
            post[v] = bel[v]
    return post, status


def solve_tree_by_clique_calls_joint(fg, tree, backend, seed, batched=False):
    """the same walk with joint upward messages (SolverParams.useMsgLikelihoods): the symbolic half -- which differential
    factors and common priors a clique sends up, which of them its parent keeps (addLikelihoodsDifferentialCHILD!,
    addMsgFactors!; iif_amd.jointmsg) -- is done here, as the Julia CSM does it; the numeric half goes through
    nbp_clique_upsolve_joint (approxDeconv + manikde! of every differential pair on the way up) and the measurement KDEs
    of nbp_clique_desc.factor_meas_kde (on the way in).  batched: the cliques of a tree level in one
    nbp_clique_solve_batch call (diff_out of every request).  -> ({label: Belief}, {clique: status})"""
    from iif_amd import jointmsg
    from iif_amd.factorgraph import DFGFactor, DifferentialRelative
    sp = fg.solverParams
    plan = jointmsg.plan_joint_messages(fg, tree)
    man = {v: fg.getVariable(v).varType.manifold for v in fg.ls()}
    marg = {v: fg.getVariable(v).ismargin for v in fg.ls()}
    main = {v: Belief(man[v], fg.getVariable(v).val, fg.getVariable(v).bw) for v in fg.ls()}
    sub, status, dkde = {}, {}, {}   # dkde[(clique, i)]: the KDE of differential factor i of that clique's message

    def subgraph(cid):
        """(factors, measurement KDEs, message priors) of the clique sub graph, in the order of the plan"""
        facs, kdes, msgs = [], [], []
        for f in plan[cid].factors:
            if f.tag == "f":
                facs.append(fg.getFactor(f.ref))
                kdes.append(None)
            elif f.tag == "d":
                a, b, _, kind = plan[f.ref[0]].relatives[f.ref[1]]
                facs.append(DFGFactor(f"diff{f.ref[0]}_{f.ref[1]}", [a, b], DifferentialRelative(kind, -1), None, 0.0, sp.inflation))
                kdes.append(dkde[f.ref])
            else:
                msgs.append((f.ref[1], sub[f.ref[0]][f.ref[1]]))
        return facs, kdes, msgs

    def run(calls):
        return clique_solve_batch(backend, calls) if batched else [clique_solve(backend, *a, **kw) for a, kw in calls]

    depths = tree.depths()
    levels = sorted(set(depths.values()))
    for d in reversed(levels):
        ids = sorted(c for c in tree.cliques if depths[c] == d)
        calls = []
        for cid in ids:
            cl = tree.cliques[cid]
            labels = list(cl.frontalIDs) + list(cl.separatorIDs)
            sub[cid] = {v: main[v].copy() for v in labels}
            facs, kdes, msgs = subgraph(cid)
            lists = {"directFrtlMsg": cl.directFrtlMsgIDs, "msgskip": cl.msgskipIDs, "itervar": cl.itervarIDs,
                     "directPriorMsg": cl.directPriorMsgIDs}
            diffs = [(a, b, kind) for (a, b, _, kind) in plan[cid].relatives] if cl.parent >= 0 else []
            calls.append(((sp, cid, labels, len(cl.frontalIDs), len(cl.separatorIDs), [man[v] for v in labels], facs, sub[cid], seed),
                          dict(down=False, ismargin=[marg[v] for v in labels], lists=lists, msgs=msgs, meas_kdes=kdes, diffs=diffs)))
        for cid, r in zip(ids, run(calls)):
            if isinstance(r, tuple):
                status[cid], out = r
                for i, b in enumerate(out):
                    dkde[(cid, i)] = b
            else:
                status[cid] = r
    post = {}
    for r in tree.roots:
        for v in tree.cliques[r].frontalIDs:
            main[v] = sub[r][v].copy()
            post[v] = main[v]
    for d in levels:
        ids = sorted(c for c in tree.cliques if depths[c] == d and tree.cliques[c].parent >= 0)
        calls = []
        for cid in ids:
            cl = tree.cliques[cid]
            for s in cl.separatorIDs:
                sub[cid][s].pts[:] = sub[cl.parent][s].pts
            # no addDownVariableFactors! in this mode: the down solve works on the clique sub graph as the up solve left it,
            # minus the common priors (CliqueStateMachine.jl:558)
            facs, kdes, _ = subgraph(cid)
            labels = list(cl.frontalIDs) + list(cl.separatorIDs)
            calls.append(((sp, cid, labels, len(cl.frontalIDs), len(cl.separatorIDs), [man[v] for v in labels], facs, sub[cid], seed),
                          dict(down=True, ismargin=[marg[v] for v in labels], meas_kdes=kdes)))
        if not calls:
            continue
        for cid, st in zip(ids, run(calls)):
            status[cid] = st
            for v in tree.cliques[cid].frontalIDs:
                post[v] = sub[cid][v]
    return post, status


def solve_tree_by_level_batches(fg, tree, backend, seed):
    """the walk of solve_tree_by_clique_calls with the cliques of one tree level in ONE nbp_clique_solve_batch call (they do
    not depend on each other: the reference runs them as concurrent tasks).  Message assembly stays here, on the host.
    -> ({label: Belief}, {clique: status})"""
    sp = fg.solverParams
    man = {v: fg.getVariable(v).varType.manifold for v in fg.ls()}
    marg = {v: fg.getVariable(v).ismargin for v in fg.ls()}
    main = {v: Belief(man[v], fg.getVariable(v).val, fg.getVariable(v).bw) for v in fg.ls()}
    depths = tree.depths()
    levels = sorted(set(depths.values()))
    sub, status, post = {}, {}, {}
    for d in reversed(levels):  # up pass: the deepest level first
        ids = sorted(c for c in tree.cliques if depths[c] == d)
        calls = []
        for cid in ids:
            cl = tree.cliques[cid]
            labels = list(cl.frontalIDs) + list(cl.separatorIDs)
            sub[cid] = {v: main[v].copy() for v in labels}
            msgs = [(v, sub[ch][v]) for ch in cl.children for v in tree.cliques[ch].separatorIDs]
            lists = {"directFrtlMsg": cl.directFrtlMsgIDs, "msgskip": cl.msgskipIDs, "itervar": cl.itervarIDs,
                     "directPriorMsg": cl.directPriorMsgIDs}
            calls.append(((sp, cid, labels, len(cl.frontalIDs), len(cl.separatorIDs), [man[v] for v in labels],
                           [fg.getFactor(f) for f in cl.potentials], sub[cid], seed),
                          dict(down=False, ismargin=[marg[v] for v in labels], lists=lists, msgs=msgs)))
        for cid, st in zip(ids, clique_solve_batch(backend, calls)):
            status[cid] = st
    for r in tree.roots:
        for v in tree.cliques[r].frontalIDs:
            main[v] = sub[r][v].copy()
            post[v] = main[v]
    for d in levels:  # down pass: parents before children
        ids = sorted(c for c in tree.cliques if depths[c] == d and tree.cliques[c].parent >= 0)
        calls = []
        for cid in ids:
            cl = tree.cliques[cid]
            for s in cl.separatorIDs:
                sub[cid][s].pts[:] = sub[cl.parent][s].pts
            factors = []
            for v in cl.frontalIDs:
                for f in fg.ls(v):
                    if f not in factors:
                        factors.append(f)
            inclq = list(cl.frontalIDs) + list(cl.separatorIDs)
            others = []
            for f in factors:
                for u in fg.getFactor(f).variables:
                    if u not in inclq and u not in others:
                        others.append(u)
            labels = inclq + others
            bel = {v: (sub[cid][v] if v in sub[cid] else main[v].copy()) for v in labels}
            sub[cid] = bel
            order = [f for f in fg.lsf() if f in factors]
            calls.append(((sp, cid, labels, len(cl.frontalIDs), len(cl.separatorIDs), [man[v] for v in labels],
                           [fg.getFactor(f) for f in order], bel, seed), dict(down=True, ismargin=[marg[v] for v in labels])))
        if not calls:
            continue
        for cid, st in zip(ids, clique_solve_batch(backend, calls)):
            status[cid] = st
            for v in tree.cliques[cid].frontalIDs:
                post[v] = sub[cid][v]
    return post, status
