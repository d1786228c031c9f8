"""Upward messages as joint likelihoods: `SolverParams.useMsgLikelihoods = true`.

The symbolic half (which differential factors and which message priors a clique sends up and which of
them its parent keeps) of
  addLikelihoodsDifferentialCHILD!   src/services/TreeMessageUtils.jl:279-335
  _findSubgraphsFactorType           :126-193
  _calcCandidatePriorBest            :339-376
  _generateSubgraphMsgPriors         :394-420
  _generateMsgJointRelativesPriors   :430-456
  addLikelihoodPriorCommon!          :463-474   (parent side)
  addMsgFactors!                     :538-578   (parent side)
  prepCliqueMsgUp                    :667-703   (msg.hasPriors)
The numeric half (approxDeconv between two separator beliefs -> manikde! -> relative factor whose
measurement is that KDE) is a NBP_STAGE_DECONV op plus proposals with `meas_kde` (include/nbp.h).

`isPathFactorsHomogeneous` lives in DistributedFactorGraphs (not vendored): the factors on ONE shortest
path between the two variables all have the same type name (pinned by test/testJointEnforcement.jl:57-62).
Where the reference iterates a Dict (arbitrary order) this module iterates in separator order.
"""
from collections import deque

from . import abi


def selectFactorType(vt1, vt2):
    """services/DefaultNodeTypes.jl:12-31: Position{N} pairs -> LinearRelative{N}; otherwise the type named
    T1T2 (CircularCircular in IIF; SE(2) pairs resolve to the SE(2) ManifoldFactor, RoME's Pose2Pose2).
    -> (type name, nbp factor kind) or None when the reference has no such type."""
    m1, m2 = vt1.manifold, vt2.manifold
    if m1 != m2:
        return None
    if m1 in (abi.EUCLID1, abi.EUCLID2, abi.EUCLID3):
        return ("LinearRelative", abi.F_LINREL)
    if m1 == abi.CIRCULAR:
        return ("CircularCircular", abi.F_CIRCULAR)
    if m1 == abi.SE2:
        return ("ManifoldFactor", abi.F_SE2)
    return None


class SubFactor:
    """one factor of a clique sub graph: a clique potential ("f"), a differential factor received from a
    child ("d"), or a message prior received from a child ("p", tag __UPWARD_COMMON__)"""
    __slots__ = ("tag", "ref", "variables", "typename", "is_prior", "kind")

    def __init__(self, tag, ref, variables, typename, is_prior, kind=0):
        self.tag, self.ref, self.variables = tag, ref, list(variables)
        self.typename, self.is_prior, self.kind = typename, is_prior, kind

    def __repr__(self):
        return f"SubFactor({self.tag}, {self.ref}, {self.variables}, {self.typename})"


def shortest_path_factors(variables, factors, frm, to, typename=None):
    """findShortestPathDijkstra on the bipartite graph (unit weights): the factors along one shortest path,
    or None when `to` cannot be reached.  typename: only factors of that type may be crossed."""
    if frm == to:
        return []
    by_var = {v: [] for v in variables}
    for i, f in enumerate(factors):
        if typename is not None and f.typename != typename:
            continue
        for v in f.variables:
            if v in by_var:
                by_var[v].append(i)
    prev = {frm: None}
    dq = deque([frm])
    while dq:
        v = dq.popleft()
        for i in by_var[v]:
            for u in factors[i].variables:
                if u in by_var and u not in prev:
                    prev[u] = (v, i)
                    if u == to:
                        path = []
                        while prev[u] is not None:
                            u, fi = prev[u]
                            path.append(factors[fi])
                        return path[::-1]
                    dq.append(u)
    return None


def isPathFactorsHomogeneous(variables, factors, frm, to):
    pth = shortest_path_factors(variables, factors, frm, to)
    types = []
    for f in (pth or []):
        if f.typename not in types:
            types.append(f.typename)
    return len(types) == 1, types


class CliqueJoint:
    """what one clique's sub graph holds during its up solve, and the joint message it sends up"""

    def __init__(self):
        self.factors = []       # SubFactor list of the clique sub graph (potentials + child messages)
        self.relatives = []     # [(sym1, sym2, typename, kind)] differential factors sent up
        self.priors = []        # separator variables that get a MsgPrior in the message
        self.hasPriors = False  # msg.hasPriors


def _find_subgraph_classes(fg, cl, factors, relatives):
    """_findSubgraphsFactorType: separators grouped by connectivity through factors of the default type"""
    seps = list(cl.separatorIDs)
    count = {s: 0 for s in seps}
    for (a, b, _, _) in relatives:
        count[a] += 1
        count[b] += 1
    cls, new = {}, 0
    for s in seps:
        if count[s] == 0:
            new += 1
            cls[s] = new
    outer = [s for s in seps if s not in cls]
    for k1 in outer:
        if k1 not in cls:
            new += 1
            cls[k1] = new
        for k2 in [s for s in seps if s not in cls]:
            sel = selectFactorType(fg.getVariable(k1).varType, fg.getVariable(k2).varType)
            pth = shortest_path_factors(cl.allIDs, factors, k1, k2, sel[0]) if sel is not None else None
            if not pth:
                new += 1
                cls[k2] = new
            else:
                cls[k2] = cls[k1]
    allc = {}
    for s in seps:
        allc.setdefault(cls[s], []).append(s)  # (isInitialized: separators are solved by now)
    return allc


def _candidate_prior_best(fg, factors, syms):
    """_calcCandidatePriorBest: highest dimension, then most factors in the clique sub graph"""
    dims = [fg.getVariable(s).varType.dim for s in syms]
    md = max(dims)
    cand = [s for s, d in zip(syms, dims) if d == md]
    adj = [sum(1 for f in factors if s in f.variables) for s in cand]
    best = max(adj)
    return cand[adj.index(best)]  # stable descending sort: first of the ties


def plan_joint_messages(fg, tree):
    """-> {clique id: CliqueJoint}, children before parents"""
    heights = tree.heights()
    out = {}
    for cid in sorted(tree.cliques, key=lambda c: (heights[c], c)):
        cl = tree.cliques[cid]
        J = CliqueJoint()
        for f in cl.potentials:
            fc = fg.getFactor(f)
            J.factors.append(SubFactor("f", f, fc.variables, type(fc.fnc).__name__, fc.fnc.is_prior, fc.fnc.kind))
        # addMsgFactors!(subfg, msg, UpwardPass): differentials, then the common priors "only if necessary"
        for ch in cl.children:
            M = out[ch]
            for idx, (a, b, tn, kind) in enumerate(M.relatives):
                J.factors.append(SubFactor("d", (ch, idx), [a, b], tn, False, kind))
            for v in M.priors:
                if M.hasPriors or not any(v in f.variables for f in J.factors):
                    J.factors.append(SubFactor("p", (ch, v), [v], "MsgPrior", True, abi.F_MSGPRIOR))
        # prepCliqueMsgUp: hasPriors = any prior in the sub graph, message priors included (:690)
        J.hasPriors = any(f.is_prior for f in J.factors)
        own_priors = any(f.is_prior for f in J.factors if f.tag == "f")  # :442
        if cl.parent >= 0:
            seps = list(cl.separatorIDs)
            dims = [fg.getVariable(s).varType.dim for s in seps]
            dec = [seps[i] for i in sorted(range(len(seps)), key=lambda i: -dims[i])]  # sortperm(rev=true), stable
            acc = dec[::-1]
            already = []
            for s1 in dec:
                already.append(s1)
                for s2 in [s for s in acc if s not in already]:
                    hom, types = isPathFactorsHomogeneous(cl.allIDs, J.factors, s1, s2)
                    if not hom:
                        continue
                    sel = selectFactorType(fg.getVariable(s1).varType, fg.getVariable(s2).varType)
                    if sel is not None and sel[0] == types[0]:
                        J.relatives.append((s1, s2, sel[0], sel[1]))
            classes = _find_subgraph_classes(fg, cl, J.factors, J.relatives)
            for _, syms in sorted(classes.items()):
                if len(syms) == 1 or own_priors:
                    J.priors.append(_candidate_prior_best(fg, J.factors, syms))
        out[cid] = J
    return out
