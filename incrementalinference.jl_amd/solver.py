"""Host-side mirror of the reference's hot-path entry points, compiled to libnbp descriptor
batches.  Julia `foo!` -> Python `foo`:

  approxConvBelief / approxConv     src/services/ApproxConv.jl:4-47
  proposalbeliefs (descriptor form) src/services/ApproxConv.jl:238-304
  propagateBelief                   src/services/GraphProductOperations.jl:16-64
  localProduct / localProductAndUpdate   :93-155
  initAll / doautoinit              src/services/GraphInit.jl:132-199, 495-555
  solveTree                         src/services/SolverAPI.jl:326-493, with the clique work of
      upGibbsCliqueDensity (src/services/SolveTree.jl:164-239) and solveCliqDownFrontalProducts!
      (src/CliqueStateMachine/services/CliqStateMachineUtils.jl:479-571) batched per tree level
      into one device-resident program (the CliqueStateMachine's rendezvous order,
      CliqueStateMachine.jl:221-234 / 617-629, becomes the stage order).

Every function takes a `backend` class/instance implementing backend.HipBackend's interface; the
default is the HIP library and there is no CPU fallback in this package.
"""
import time

import numpy as np

from . import abi, bayestree, jointmsg
from .backend import HipBackend
from .factorgraph import DFGFactor, DifferentialRelative, MsgPrior, PartialPriorPassThrough
from .seeds import op_seed

PASS_INIT, PASS_UP, PASS_DOWN, PASS_UNIT = 0, 1, 2, 3
PRODUCT_ID = 0xFFFF


# ------------------------------------------------------------------------------------------------
# descriptor builders
# ------------------------------------------------------------------------------------------------
def proposal_desc(fg, fct, target, slot_of, out_slot, seed, nullSurplus=0.0, mhidx_in=-1, mhidx_out=-1,
                  skip_bandwidth=False, inflateCycles=None, isinit=None, meas_seed=0, alone=False):
    """One approxConvBelief(dfg, fct, target) as a libnbp descriptor (ApproxConv.jl:4-45 +
    evalFactor kwargs, EvalFactor.jl:571-603)."""
    sp = fg.solverParams
    d = abi.ProposalDesc()
    fnc = fct.fnc
    vt = fg.getVariable(target).varType
    d.factor_kind = fnc.kind
    d.manifold = vt.manifold
    d.out_slot = out_slot
    d.inflate_cycles = sp.inflateCycles if inflateCycles is None else inflateCycles
    d.mhidx_in, d.mhidx_out = mhidx_in, mhidx_out
    d.skip_bandwidth = int(skip_bandwidth)
    d.partial_mask = getattr(fnc, "partial_mask", 0)
    d.inflation = fct.inflation
    d.spread_nh = sp.spreadNH
    d.nullhypo = max(fct.nullhypo, nullSurplus)  # EvalFactor.jl:352
    d.seed = seed
    d.meas_seed = meas_seed  # needFreshMeasurements = false: reuse the samples of the op with that seed
    if isinstance(fnc, PartialPriorPassThrough):
        # the density is the proposal (ApproxConv.jl:196-227): no hypotheses, no fit; alone in its update it keeps its
        # own particle count (alone = 1; 2 = graph initialisation tops it up to N), in a product it is resampled to N
        d.nvars, d.sfidx, d.ncomp = 1, 0, 1
        d.var_slot[0], d.var_slot[1] = slot_of(target), fnc.slot
        d.comp[0][0] = 1.0
        d.nullhypo, d.skip_bandwidth, d.keep_count = 0.0, 1, int(alone)
        return d
    if isinstance(fnc, MsgPrior):
        d.nvars, d.sfidx = 1, 0
        d.var_slot[0] = slot_of(target)
        d.var_slot[1] = fnc.slot
        d.ncomp = 1
        d.comp[0][0] = 1.0
        return d
    d.nvars = len(fct.variables)
    d.sfidx = fct.variables.index(target)
    for i, v in enumerate(fct.variables):
        d.var_slot[i] = slot_of(v)
    flat = getattr(fnc, "_comp_flat", None)  # measurement model, flattened once per factor
    if flat is None:
        comps = fnc.components()
        flat = []
        for comp in comps:
            w, mu, L = comp[:3]
            row = [0.0] * abi.COMP_STRIDE
            row[0] = w
            for i in range(min(3, len(mu))):
                row[1 + i] = float(mu[i])
            for i in range(min(3, L.shape[0])):
                for j in range(i + 1):
                    row[4 + 3 * i + j] = float(L[i, j])
            if len(comp) > 3 and comp[3] != abi.DIST_GAUSSIAN:  # scalar Uniform / Rayleigh / table: the family rides in the last slot
                if len(mu) != 1:
                    raise ValueError("Uniform / Rayleigh / AliasingScalarSampler measurements are scalar")
                row[12] = float(comp[3])
            flat.append(row)
        try:
            fnc._comp_flat = flat
        except AttributeError:
            pass
    d.ncomp = len(flat)
    for c, row in enumerate(flat):
        d.comp[c][:] = row
    if getattr(fnc, "table", None) is not None:  # AliasingScalarSampler: its table lives in the factor's own slot
        if len(fct.variables) >= abi.MAXV:
            raise ValueError("a factor with an AliasingScalarSampler takes at most MAXV - 1 variables")
        d.var_slot[abi.MAXV - 1] = fnc.slot
    if getattr(fnc, "meas_slot", None) is not None:  # measurement = the KDE in that slot
        d.meas_kde = fnc.meas_slot + 1
    if fct.multihypo is not None:
        # isinit flags: uninitialised hypotheses are suppressed (ExplicitDiscreteMarginalizations.jl:161-172)
        flags = 1 | 0x80
        for i, v in enumerate(fct.variables):
            if (isinit[v] if isinit is not None else fg.getVariable(v).initialized):
                flags |= 1 << (8 + i)
        d.has_multihypo = flags
        for i, p in enumerate(fct.multihypo):
            d.multihypo[i] = p
    return d


def product_desc(manifold, in_slots, out_slot, seed, niter=1, labels_out=-1, partials=None, old_slot=-1):
    """AMP.manifoldProduct(dens, M; Niter, oldPoints) descriptor.  `partials[i]` = coordinate bit mask of
    a partial input density (0 = full); coordinates no density informs come from `old_slot`."""
    d = abi.ProductDesc()
    d.manifold, d.nfactors, d.niter, d.out_slot = manifold, len(in_slots), niter, out_slot
    for i, s in enumerate(in_slots):
        d.in_slot[i] = s
    d.old_slot = -1
    if partials is not None and any(partials):
        for i, m in enumerate(partials):
            d.in_partial[i] = m
        d.old_slot = old_slot
    d.labels_out, d.seed = labels_out, seed
    return d


def has_density(fnc):
    """a factor that keeps something in a device slot of its own: the density of a PartialPriorPassThrough, or the table of
    an AliasingScalarSampler measurement"""
    return isinstance(fnc, PartialPriorPassThrough) or getattr(fnc, "table", None) is not None


def passthrough_factors(fg, labels=None):
    """labels of the factors with a density slot (of the whole graph, or among `labels`)"""
    return [f for f in (fg.lsf() if labels is None else labels) if has_density(fg.getFactor(f).fnc)]


def _plan_densities(fg, labels, first_slot):
    """give every pass-through factor among `labels` a slot from `first_slot` on; returns the number of slots taken"""
    pts = passthrough_factors(fg, labels)
    for i, f in enumerate(pts):
        fg.getFactor(f).fnc.slot = first_slot + i
    return len(pts)


def write_densities(fg, be, labels=None):
    for f in passthrough_factors(fg, labels):
        fnc = fg.getFactor(f).fnc
        pts, bw = fnc.density_belief(be.N)
        be.belief_write(fnc.slot, fnc.varType.manifold if isinstance(fnc, PartialPriorPassThrough) else fnc.density_manifold, pts, bw)


def _partials(fcts):
    return [getattr(f.fnc, "partial_mask", 0) for f in fcts]


def _null_surplus(fg, factors):
    """proposalbeliefs!, ApproxConv.jl:255-265: relative non-multihypo siblings of a multihypo
    factor get nullSurplusAdd."""
    ns = [0.0] * len(factors)
    if any(f.isMultihypo for f in factors):
        for i, f in enumerate(factors):
            if not f.fnc.is_prior and not f.isMultihypo:
                ns[i] = fg.solverParams.nullSurplusAdd
    return ns


def _make_backend(backend, N, n_slots, side_ints=0):
    if backend is None:
        backend = HipBackend
    if isinstance(backend, type) or callable(backend) and not hasattr(backend, "slot_write"):
        return backend(N, n_slots, side_ints=side_ints), True
    return backend, False


# ------------------------------------------------------------------------------------------------
# unit level: factor seam and variable seam
# ------------------------------------------------------------------------------------------------
def approxConvBelief(fg, fctlabel, target, backend=None, seed=0, nullSurplus=0.0, mhidx=None, return_mhidx=False):
    """approxConvBelief(dfg, fct, target) -> (points N x P, bandwidth D)   ApproxConv.jl:4-45.
    The stored belief of `target` is never modified (CalcFactor.jl:543-548)."""
    fct = fg.getFactor(fctlabel)
    N = fg.solverParams.N
    labels = list(fct.variables)
    slot = {v: i for i, v in enumerate(labels)}
    out = len(labels)
    nd = _plan_densities(fg, [fctlabel], out + 1)
    be, own = _make_backend(backend, N, out + 1 + nd, side_ints=2 * N)
    try:
        for v in labels:
            var = fg.getVariable(v)
            be.belief_write(slot[v], var.varType.manifold, var.val, var.bw)
        write_densities(fg, be, [fctlabel])
        mh_in = -1
        if mhidx is not None:
            be.side_write(0, np.asarray(mhidx, dtype=np.int32))
            mh_in = 0
        d = proposal_desc(fg, fct, target, slot.__getitem__, out, op_seed(seed, PASS_UNIT, 0, 0, 0),
                          nullSurplus=nullSurplus, mhidx_in=mh_in, mhidx_out=N, alone=True)
        be.run_proposals([d])
        pts, bw, _ = be.belief_read(out, fg.getVariable(target).varType.manifold)  # a pass-through density keeps its count
        used = be.side_read(N, N)
    finally:
        if own:
            be.close()
    return (pts, bw, used) if return_mhidx else (pts, bw)


def findShortestPath(fg, frm, to):
    """Shortest path over the bipartite variable/factor graph (unit edge weights), the role
    `findShortestPathDijkstra` plays at ApproxConv.jl:100.  Returns [frm, ..., to]."""
    import collections
    isfct = lambda l: l in fg.factors
    prev = {frm: None}
    q = collections.deque([frm])
    while q:
        u = q.popleft()
        if u == to:
            break
        nbrs = fg.getFactor(u).variables if isfct(u) else fg.ls(u)
        for w in nbrs:
            if w not in prev:
                prev[w] = u
                q.append(w)
    if to not in prev:
        raise ValueError(f"no path from {frm} to {to}")
    path = [to]
    while prev[path[-1]] is not None:
        path.append(prev[path[-1]])
    return path[::-1]


def approxConvBeliefPath(fg, frm, target, backend=None, seed=0, path=None):
    """approxConvBelief(dfg, from, target): the sequential chain of convolutions along the shortest
    factor path, starting from the belief stored in `from` (a variable) or from a fresh prior
    proposal (`from` a unary factor).  ApproxConv.jl:75-159.  Nothing in `fg` is modified: the chain
    runs on device slots (the reference's temporary graph `tfg`); variables adjacent to the path
    keep their `fg` beliefs (:114-115)."""
    path = list(path) if path else findShortestPath(fg, frm, target)
    if path[0] != frm:
        raise ValueError("sanity check failing for shortest path function")
    isfct = [l in fg.factors for l in path]
    fcts = [l for l, f in zip(path, isfct) if f]
    labels = []
    for f in fcts:
        for v in fg.getFactor(f).variables:
            if v not in labels:
                labels.append(v)
    N = fg.solverParams.N
    slot = {v: i for i, v in enumerate(labels)}
    scratch = len(labels)
    be, own = _make_backend(backend, N, scratch + 1)
    try:
        for v in labels:
            var = fg.getVariable(v)
            be.belief_write(slot[v], var.varType.manifold, var.val, var.bw)
        for k, (l, f) in enumerate(zip(path, isfct)):
            if not f:
                continue
            nxt = path[k + 1]
            d = proposal_desc(fg, fg.getFactor(l), nxt, slot.__getitem__, scratch, op_seed(seed, PASS_UNIT, 0, k, 0))
            be.run_proposals([d])
            be.run_copies([abi.CopyDesc(scratch, slot[nxt])])
        out = be.slot_read(slot[target], fg.getVariable(target).varType.manifold)
    finally:
        if own:
            be.close()
    return out


def approxConv(fg, frm, target, **kw):
    """approxConv(dfg, from, target) = getPoints(approxConvBelief(...))   ApproxConv.jl:47.  `from` is
    a factor adjacent to `target` (direct request, :91-95) or any variable / prior factor (chain)."""
    if frm in fg.factors and target in fg.getFactor(frm).variables:
        return approxConvBelief(fg, frm, target, **kw)[0]
    return approxConvBeliefPath(fg, frm, target, **kw)[0]


def approxDeconv(fg, fctlabel, backend=None, seed=0):
    """approxDeconv(dfg, fctsym) -> (predicted, measured)   services/DeconvUtils.jl:162-189, :32-160.
    Inverse solve: per particle the measurement that makes the factor residual zero for the stored
    points of its variables ("predicted", N x zDim tangent coordinates), next to N freshly sampled
    measurements ("measured", also the starting points of the search).  Multihypo factors are not
    supported (the reference's own limitation, issues #467/#927)."""
    fct = fg.getFactor(fctlabel)
    fnc = fct.fnc
    N = fg.solverParams.N
    if fnc.is_prior:
        # r = z - x: the predicted measurement of particle n is the point itself (DefaultPrior.jl:17)
        comps = fnc.components()
        rng_pts = _sample_components(comps, N, seed)
        var = fg.getVariable(fct.variables[0])
        pred = var.val if var.varType.P == var.varType.dim else _coords(var.varType, var.val)
        return np.array(pred, dtype=float), rng_pts
    if fct.multihypo is not None:
        raise NotImplementedError("approxDeconv on multihypo factors (reference issues #467, #927)")
    labels = list(fct.variables)
    be, own = _make_backend(backend, N, len(labels) + 2)
    try:
        for i, v in enumerate(labels):
            var = fg.getVariable(v)
            be.belief_write(i, var.varType.manifold, var.val, var.bw)
        out, ms = len(labels), len(labels) + 1
        d = proposal_desc(fg, fct, labels[-1], lambda v: labels.index(v), out, op_seed(seed, PASS_UNIT, 0, 0, 0))
        be.run_deconv([d], [ms])
        zdim = fnc.zdim or fg.getVariable(labels[-1]).varType.dim
        zman = {1: abi.EUCLID1, 2: abi.EUCLID2, 3: abi.EUCLID3}[zdim]
        pred, _ = be.slot_read(out, zman)
        meas, _ = be.slot_read(ms, zman)
    finally:
        if own:
            be.close()
    return pred, meas


def _coords(varType, pts):
    if varType.manifold == abi.SE2:
        return np.stack([pts[:, 0], pts[:, 1], np.arctan2(pts[:, 3], pts[:, 2])], axis=1)
    return pts


def _sample_components(comps, N, seed):
    """host-side sampleFactor for the trivial prior case of approxDeconv"""
    rng = np.random.default_rng(seed)
    w = np.array([c[0] for c in comps])
    lbl = rng.choice(len(comps), size=N, p=w / w.sum())
    out = []
    for n in range(N):
        comp = comps[lbl[n]]
        mu, L, fam = comp[1], comp[2], (comp[3] if len(comp) > 3 else abi.DIST_GAUSSIAN)
        if fam == abi.DIST_TABLE:
            raise NotImplementedError("approxDeconv of a prior with an AliasingScalarSampler")
        if fam == abi.DIST_UNIFORM:
            out.append(np.asarray(mu) + np.asarray(L)[0] * rng.uniform())
        elif fam == abi.DIST_RAYLEIGH:
            out.append(np.asarray(L)[0] * np.sqrt(-2.0 * np.log(1.0 - rng.uniform())))
        else:
            out.append(np.asarray(mu) + np.asarray(L) @ rng.normal(size=len(mu)))
    return np.array(out)


def propagateBelief(fg, destlbl, factors=None, backend=None, seed=0, return_proposals=False):
    """propagateBelief(dfg, destvar, factors) -> ((pts, bw), ipc)   GraphProductOperations.jl:16-64:
    one proposal per factor (proposalbeliefs!) then AMP.manifoldProduct(dens; Niter=1, N)."""
    sp = fg.solverParams
    N = sp.N
    flabels = list(fg.ls(destlbl)) if factors is None else list(factors)
    fcts = [fg.getFactor(f) for f in flabels]
    if not fcts:
        raise ValueError(f"propagateBelief: no factors for {destlbl}")
    labels = []
    for f in fcts:
        for v in f.variables:
            if v not in labels:
                labels.append(v)
    if destlbl not in labels:
        labels.append(destlbl)
    slot = {v: i for i, v in enumerate(labels)}
    nv = len(labels)
    nd = _plan_densities(fg, flabels, nv + len(fcts) + 1)
    be, own = _make_backend(backend, N, nv + len(fcts) + 1 + nd)
    man = fg.getVariable(destlbl).varType.manifold
    try:
        for v in labels:
            var = fg.getVariable(v)
            be.belief_write(slot[v], var.varType.manifold, var.val, var.bw)
        write_densities(fg, be, flabels)
        ns = _null_surplus(fg, fcts)
        descs = [proposal_desc(fg, f, destlbl, slot.__getitem__, nv + i, op_seed(seed, PASS_UNIT, 0, 0, i + 1),
                               nullSurplus=ns[i], alone=len(fcts) == 1) for i, f in enumerate(fcts)]
        be.run_proposals(descs)
        out = nv + len(fcts)
        be.run_products([product_desc(man, [nv + i for i in range(len(fcts))], out,
                                      op_seed(seed, PASS_UNIT, 0, 0, PRODUCT_ID), sp.productNiter,
                                      partials=_partials(fcts), old_slot=slot[destlbl])])
        pts, bw, ipc = be.belief_read(out, man)  # the belief may hold a pass-through density's own point count
        props = [be.belief_read(nv + i, man)[:2] for i in range(len(fcts))] if return_proposals else None
    finally:
        if own:
            be.close()
    # ipc = the sum over the factors of ones(D) (`fct_ipc = ones(vardim)`, ApproxConv.jl:277,298-303), produced on the device
    if return_proposals:
        return (pts, bw), ipc, props
    return (pts, bw), ipc


def localProduct(fg, sym, **kw):
    """localProduct(dfg, sym) -> (product, proposals, factor labels, ipc)   GraphProductOperations.jl:93-120"""
    lb = list(fg.ls(sym))
    (pts, bw), ipc, props = propagateBelief(fg, sym, lb, return_proposals=True, **kw)
    return (pts, bw), props, lb, ipc


def setValKDE(fg, sym, pts, bw, setinit=True):
    """setValKDE!(vari, mkd, setinit, ipc)   FactorGraph.jl:250-263"""
    v = fg.getVariable(sym)
    v.val, v.bw = np.array(pts, dtype=float), np.array(bw, dtype=float)
    if setinit:
        v.initialized = True


def manikde(varType, pts, backend=None):
    """manikde!(M, pts) -> (pts, bw): the automatic per-coordinate bandwidth of a point set
    (call sites ApproxConv.jl:38,41, FGOSUtils.jl:118-128) through `nbp_run_bandwidth`."""
    pts = np.asarray(pts, dtype=np.float64)
    N = pts.shape[0]
    be, own = _make_backend(backend, N, 1)
    try:
        be.slot_write(0, varType.manifold, pts, np.ones(varType.dim))
        be.run_bandwidth([0], [varType.manifold])
        out = be.slot_read(0, varType.manifold)
    finally:
        if own:
            be.close()
    return out


def initVariable(fg, sym, pts, backend=None):
    """initVariable!(dfg, sym, pts): fit the KDE bandwidth of `pts` and store the belief
    (services/GraphInit.jl, setValKDE! at FactorGraph.jl:250-297).  N follows the point count."""
    var = fg.getVariable(sym)
    p, bw = manikde(var.varType, pts, backend=backend)
    setValKDE(fg, sym, p, bw)


def localProductAndUpdate(fg, sym, setkde=True, **kw):
    """localProductAndUpdate!(dfg, sym, setkde)   GraphProductOperations.jl:136-155"""
    (pts, bw), _, lbl, ipc = localProduct(fg, sym, **kw)
    if setkde and len(pts):
        setValKDE(fg, sym, pts, bw, False)
    return (pts, bw), ipc, lbl


# ------------------------------------------------------------------------------------------------
# graph initialisation (GraphInit.jl:132-199, 495-555)
# ------------------------------------------------------------------------------------------------
def _init_plan(fg):
    """Simulate initAll!'s sweep on the host: ordered list of (variable, usable factor labels)."""
    init = {v: fg.getVariable(v).initialized for v in fg.ls()}
    plan = []
    for _ in range(10):
        repeat = False
        for sym in fg.ls():
            if init[sym]:
                continue
            use = []
            for fl in fg.ls(sym):
                f = fg.getFactor(fl)
                others = [v for v in f.variables if v != sym]
                ok = all(init[v] for v in others)  # priors and general n-ary cases, GraphInit.jl:90
                if not ok and f.isMultihypo:
                    # at least one hypothesis available (GraphInit.jl:94-105, FactorGraph.jl:772-784)
                    cer = [v for v, p in zip(f.variables, f.multihypo) if p == 0.0]
                    unc = [v for v, p in zip(f.variables, f.multihypo) if p > 0.0]
                    ok = (sym in cer and any(init[v] for v in unc)) or (sym in unc and all(init[v] for v in cer))
                if ok:
                    use.append(fl)
            if use:
                plan.append((sym, use, dict(init)))  # + the initialisation state the proposals see
                init[sym] = True
            else:
                repeat = True
        if not repeat:
            break
    return plan, init


def initStages(fg, seed=0):
    """The stage list of initAll!: (plan, slot map, n_slots, stages).  Independent inits are batched into
    one PROPOSALS + PRODUCTS stage pair; slots 0..V-1 are the variables (add order), the rest scratch."""
    sp = fg.solverParams
    plan, _ = _init_plan(fg)
    labels = fg.ls()
    slot = {v: i for i, v in enumerate(labels)}
    V = len(labels)
    if not plan:
        return plan, slot, V, []
    maxF = max(len(u) for _, u, _ in plan)
    # group consecutive independent inits into stages
    groups, cur, produced = [], [], set()
    for sym, use, st in plan:
        deps = set()
        for fl in use:
            deps.update(v for v in fg.getFactor(fl).variables if v != sym)
        if deps & produced:
            groups.append(cur)
            cur, produced = [], set()
        cur.append((sym, use, st))
        produced.add(sym)
    if cur:
        groups.append(cur)
    width = max(len(g) for g in groups)
    nd = _plan_densities(fg, None, V + width * maxF)  # V beliefs | proposal scratch | the pass-through densities
    stages = []
    for gi, g in enumerate(groups):
        props, prods = [], []
        for ci, (sym, use, st) in enumerate(g):
            fcts = [fg.getFactor(f) for f in use]
            ns = _null_surplus(fg, fcts)
            base = V + ci * maxF
            for i, f in enumerate(fcts):
                props.append(proposal_desc(fg, f, sym, slot.__getitem__, base + i,
                                           op_seed(seed, PASS_INIT, slot[sym], 0, i + 1), nullSurplus=ns[i], isinit=st,
                                           alone=2 if len(fcts) == 1 else 0))  # 2: resample(bel, N), GraphInit.jl:174-177
            prods.append(product_desc(fg.getVariable(sym).varType.manifold, [base + i for i in range(len(fcts))],
                                      slot[sym], op_seed(seed, PASS_INIT, slot[sym], 0, PRODUCT_ID), sp.productNiter,
                                      partials=_partials(fcts), old_slot=slot[sym]))
        stages.append((abi.STAGE_PROPOSALS, props))
        stages.append((abi.STAGE_PRODUCTS, prods))
    return plan, slot, V + width * maxF + nd, stages


def initAll(fg, backend=None, seed=0):
    """initAll!(dfg): initialise every variable from already-initialised neighbours in add-history
    order; each init is a propagateBelief over the usable factors.  Independent inits are batched
    into one launch."""
    N = fg.solverParams.N
    plan, slot, n_slots, stages = initStages(fg, seed)
    if not plan:
        return 0
    be, own = _make_backend(backend, N, n_slots)
    try:
        for v in fg.ls():
            var = fg.getVariable(v)
            be.belief_write(slot[v], var.varType.manifold, var.val, var.bw)
        write_densities(fg, be)
        prog = None
        try:
            prog = be.program(stages)
            prog.run()
            be.synchronize()
            for sym, _, _ in plan:  # a variable initialised from one pass-through density carries that density's point count
                pts, bw, _ = be.belief_read(slot[sym], fg.getVariable(sym).varType.manifold)
                setValKDE(fg, sym, pts, bw, True)
        finally:  # the program goes before its context, on the error path too
            if prog is not None:
                prog.close()
    finally:
        if own:
            be.close()
    return len(plan)


# ------------------------------------------------------------------------------------------------
# tree solve
# ------------------------------------------------------------------------------------------------
def _default_relative(kind, varType):
    """`_sft()` of addLikelihoodsDifferentialCHILD!: the default-constructed relative factor whose samples
    start the deconvolution search (LinearRelative{N}() = MvNormal(0, I), Factors/LinearRelative.jl:17-22)"""
    from .factorgraph import CircularCircular, LinearRelative, ManifoldFactor, MvNormal, Normal
    if kind == abi.F_LINREL:
        return LinearRelative(MvNormal(np.zeros(varType.dim), np.eye(varType.dim)))
    if kind == abi.F_CIRCULAR:
        return CircularCircular(Normal(0.0, 1.0))
    return ManifoldFactor(MvNormal(np.zeros(3), np.eye(3)))


class TreeProgram:
    """The whole up+down solve of a Bayes tree compiled into libnbp stages.

    Slot plan (DESIGN.md "HBM layout"):  main[v] | snap[v] | clique-local beliefs B[c,v] | ghost
    slots for messages arriving from other ranks | per-clique proposal scratch.  The clique-local
    copies are the reference's deep-copied `cliqSubFg` (SubGraphFunctions.jl:48); an up message is
    the child's separator slot read in place by the parent's MsgPrior proposal; a down message is a
    slot copy parent -> child (updateSubFgFromDownMsgs!, TreeMessageUtils.jl:66-84).

    Multi-GPU: `owner` maps every clique to a rank; a rank compiles only its own cliques.  Tree
    edges whose two cliques live on different ranks become exchange items between the stage
    segments (`self.segments`): the separator beliefs are sent slot-by-slot, point to point."""

    def __init__(self, fg, tree, seed=0, snapshot=False, owner=None, rank=0, asap=None):
        self.fg, self.tree, self.seed = fg, tree, seed
        self.asap = asap  # None / True: cliques batched by dependency; False: by tree level (kept for the equivalence test)
        self.rank = rank
        self.owner = owner or {c: 0 for c in tree.cliques}
        sp = fg.solverParams
        labels = fg.ls()
        self.main = {v: i for i, v in enumerate(labels)}
        nxt = len(labels)
        # optional snapshot of the initial beliefs so that the program can be replayed (bench):
        # stage 0 restores main[v] from snap[v]
        self.snap = None
        if snapshot:
            self.snap = {v: nxt + i for i, v in enumerate(labels)}
            nxt += len(labels)
        self.dens0 = nxt  # one slot per pass-through density, written by the caller like the initial beliefs
        nxt += _plan_densities(fg, None, nxt)
        self.B = {}
        self.ghost = {}
        self.scratch = {}
        # useMsgLikelihoods: the symbolic plan of the joint messages (jointmsg.py); D[(c, i)] = slot of the
        # KDE of the i-th differential factor clique c sends up
        self.joint, self.D = None, {}
        if getattr(sp, "useMsgLikelihoods", False):
            self.joint = jointmsg.plan_joint_messages(fg, tree)  # the same plan on every rank
        self.upsched, self.dnsched, self.upfacs, self.dnfacs = {}, {}, {}, {}
        self.uprounds, self.dnrounds = {}, {}  # per clique: the schedule's steps grouped into rounds of commuting steps
        self.upfresh = {}     # per clique: does step k of the up schedule draw fresh measurements?
        self._meas_seed = {}  # (clique, factor entry) -> seed of the op that last drew its measurements
        self.heights, self.depths = tree.heights(), tree.depths()
        mine = [c for c in tree.cliques if self.owner[c] == rank]
        self.cliques = mine
        for cid in tree.cliques:
            self._plan_symbolic(cid)
        for cid in mine:
            cl = tree.cliques[cid]
            for v in cl.allIDs:
                self.B[(cid, v)] = nxt
                nxt += 1
            for ch in cl.children:  # messages from children that live on another rank land in ghost slots
                if self.owner[ch] != rank:
                    for v in tree.cliques[ch].separatorIDs:
                        self.ghost[(ch, v)] = nxt
                        nxt += 1
                    if self.joint is not None:  # ... and so do the KDEs of its differential factors
                        for i in range(len(self.joint[ch].relatives)):
                            self.D[(ch, i)] = nxt
                            nxt += 1
            if self.joint is not None:
                for i in range(len(self.joint[cid].relatives)):
                    self.D[(cid, i)] = nxt
                    nxt += 1
            upf, sched, dnf = self.upfacs[cid], self.upsched[cid], self.dnfacs[cid]
            maxf = max([len(upf[v]) for v in sched] + [len(dnf[v]) for v in self.dnsched[cid]] + [1])
            if maxf > abi.MAXF:
                raise ValueError(f"clique {cid}: {maxf} densities in one product exceeds NBP_MAXF")
            conc = max([len(g) for g in self.uprounds[cid] + self.dnrounds[cid]] + [1])  # steps of one round run side by side
            self.scratch[cid] = (nxt, maxf)
            nxt += maxf * conc
        self.n_slots = nxt
        self.stages = []
        self.stage_pass = []
        self.segments = []  # ("run", first_stage, last_stage) | ("xchg", sends, recvs)
        self._seg_start = 0
        self.n_updates_up = 0
        self.n_updates_down = 0
        self.alg_bytes = 0
        self.alg = {"nbp_proposal_kernel": 0, "nbp_prep_kernel": 0, "nbp_product_kernel": 0, "nbp_bandwidth_kernel": 0}
        self._compile()

    def _plan_symbolic(self, cid):
        """the symbolic half of one clique's plan (no slots): factor lists and schedules of its up and down solve --
        computed for EVERY clique on every rank, because the stage times of a rank's cliques depend on the
        schedule lengths of cliques elsewhere in the tree"""
        fg, tree, sp = self.fg, self.tree, self.fg.solverParams
        cl = tree.cliques[cid]
        # up-solve factor lists per variable: clique potentials touching v + child messages on v
        upf = {}
        for v in cl.allIDs:
            if self.joint is not None:  # potentials + the children's differentials + their common priors
                # (the common message priors last: the reference iterates a Dict here, and with this order a clique call
                #  -- factors first, then messages, nbp_clique_upsolve -- numbers the densities of a variable the same way)
                lst = [(f.tag, f.ref) for f in self.joint[cid].factors if v in f.variables and f.tag != "p"] + \
                      [("m", f.ref[0]) for f in self.joint[cid].factors if v in f.variables and f.tag == "p"]
            else:
                lst = [("f", f) for f in cl.potentials if v in fg.getFactor(f).variables]
                for ch in cl.children:
                    if v in tree.cliques[ch].separatorIDs:
                        lst.append(("m", ch))
            upf[v] = lst
        # doFMCIteration skips marginalized variables (SolveTree.jl:61)
        full = [(v, it) for v, it in bayestree.upGibbsSchedule(cl, sp.gibbsIters, with_iteration=True)
                if upf[v] and not fg.getVariable(v).ismargin]
        sched = [v for v, _ in full]
        # needFreshMeasurements = iter == 1 || alwaysFreshMeasurements (SolveTree.jl:119)
        self.upfresh[cid] = [it == 1 or sp.alwaysFreshMeasurements for _, it in full]
        self.upsched[cid], self.upfacs[cid] = sched, upf
        if self.joint is None:
            dnf = {v: [("f", f) for f in fg.ls(v)] for v in cl.frontalIDs}
            dsch = bayestree.downSchedule(fg, cl) if cl.parent >= 0 else []  # MCIters = 3: its own default (:485), not gibbsIters
        else:
            # no addDownVariableFactors! (CliqueStateMachine.jl:823): the down solve works on the clique sub
            # graph as the up solve left it -- potentials + the children's differentials (:558 removes
            # only the __UPWARD_COMMON__ priors)
            kept = [f for f in self.joint[cid].factors if f.tag != "p"]
            dnf = {v: [(f.tag, f.ref) for f in kept if v in f.variables] for v in cl.frontalIDs}
            frs = set(cl.frontalIDs)
            itv = {v for f in kept if len([u for u in f.variables if u in frs]) > 1 for v in f.variables if v in frs}
            skip = {v for v in cl.frontalIDs if sp.limitfixeddown and fg.getVariable(v).ismargin}
            dsch = ([v for v in cl.frontalIDs if v not in itv and v not in skip and dnf[v]]
                    + [v for v in cl.frontalIDs if v in itv and v not in skip] * 3) if cl.parent >= 0 else []  # MCIters = 3
        self.dnfacs[cid] = dnf
        self.dnsched[cid] = dsch
        self.uprounds[cid] = self._rounds(sched, upf)
        self.dnrounds[cid] = self._rounds(dsch, dnf)

    def _entry_variables(self, entry):
        kind, ref = entry
        if kind == "f":
            return self.fg.getFactor(ref).variables
        if kind == "d":
            return list(self.joint[ref[0]].relatives[ref[1]][:2])
        return ()  # a message prior touches its own variable only

    def _rounds(self, sched, facs):
        """Steps of one clique's Gibbs schedule that can share a stage.  The reference runs the steps one after the other
        (fmcmc!, SolveTree.jl:97-160); step j only sees step i < j through the belief of i's variable, so two steps whose
        variables are different and share no factor commute -- they read and write disjoint beliefs, and the random
        streams are keyed by the step index, not by the execution order.  round[j] = 1 + the latest round among the
        earlier steps j conflicts with; a round is one PROPOSALS + PRODUCTS stage pair.  A chain clique {f | a, b} with
        factors (a,f), (f,b) runs f | a b | f | a b | f | a b: six rounds instead of nine."""
        reads = {}
        for v in set(sched):
            r = set()
            for e in facs[v]:
                r.update(self._entry_variables(e))
            r.discard(v)
            reads[v] = r
        rnd = []
        for j, v in enumerate(sched):
            r = 0
            for i in range(j):
                u = sched[i]
                if u == v or u in reads[v] or v in reads[u]:
                    r = max(r, rnd[i] + 1)
            rnd.append(r)
        groups = [[] for _ in range(max(rnd) + 1)] if rnd else []
        for j, r in enumerate(rnd):
            groups[r].append(j)
        return groups

    # -- helpers ------------------------------------------------------------------------------
    def _account(self, man, F_in):
        N = self.fg.solverParams.N
        P, D = abi.MANIFOLD_P[man], abi.MANIFOLD_DIM[man]
        self.alg_bytes += (F_in + 2) * N * P * 8 + (F_in + 1) * D * 8  # B_upd, SURVEY 8(d)
        # split of B_upd over the kernels of one update (DESIGN.md "algorithmic bytes"): the proposal
        # kernel reads the F_in operand beliefs + the variable's own old belief; the product kernel
        # writes the new points; the bandwidth kernels write the (F_in + 1) bandwidth vectors.
        self.alg["nbp_proposal_kernel"] += (F_in + 1) * N * P * 8
        self.alg["nbp_prep_kernel"] += (F_in + 1) * D * 8
        self.alg["nbp_product_kernel"] += N * P * 8

    def _msg_slot(self, child, v):
        return self.B[(child, v)] if self.owner[child] == self.rank else self.ghost[(child, v)]

    def _update_ops(self, cid, v, entries, slot_of, out_slot, passid, step, fresh=True, lane=0):
        fg, sp = self.fg, self.fg.solverParams
        base, maxf = self.scratch[cid]
        base += lane * maxf  # `lane`: position of this step within its round
        fcts = []
        for kind, ref in entries:
            if kind == "f":
                fcts.append(fg.getFactor(ref))
            elif kind == "d":  # differential factor of a child's joint message (addLikelihoodsDifferential!)
                a, b, _, fk = self.joint[ref[0]].relatives[ref[1]]
                fcts.append(DFGFactor(f"diff{ref[0]}_{ref[1]}", [a, b], DifferentialRelative(fk, self.D[ref]), None, 0.0, sp.inflation))
            else:  # child message -> MsgPrior (generateMsgPrior, TreeMessageUtils.jl:86-89)
                fcts.append(DFGFactor(f"msg{ref}_{v}", [v], MsgPrior(self._msg_slot(ref, v)), None, 0.0, sp.inflation))
        ns = _null_surplus(fg, fcts)
        props = []
        for i, f in enumerate(fcts):
            # one stored measurement per factor object: a user / differential factor is shared by its variables,
            # a message is one MsgPrior factor per separator variable
            sd, key = op_seed(self.seed, passid, cid, step, i + 1), (cid, entries[i], v if entries[i][0] == "m" else None)
            if fresh:  # the factor's ccw.measurement is overwritten (CalcFactor.jl:492-510)
                self._meas_seed[key] = sd
            props.append(proposal_desc(fg, f, v, slot_of, base + i, sd, nullSurplus=ns[i],
                                       meas_seed=0 if fresh else self._meas_seed.get(key, 0), alone=len(fcts) == 1))
        man = fg.getVariable(v).varType.manifold
        prod = product_desc(man, [base + i for i in range(len(fcts))], out_slot,
                            op_seed(self.seed, passid, cid, step, PRODUCT_ID), sp.productNiter,
                            partials=_partials(fcts), old_slot=slot_of(v))
        self._account(man, sum(0 if f.fnc.is_prior and not isinstance(f.fnc, MsgPrior) else 1 for f in fcts))
        return props, prod

    def _add(self, kind, descs, tag):
        self.stages.append((kind, descs))
        self.stage_pass.append(tag)

    def _exchange(self, edges):
        """edges: [(src_rank, src_slot_key, dst_rank, dst_slot_key)] in a globally agreed order"""
        sends = [(dr, sk()) for (sr, sk, dr, dk) in edges if sr == self.rank and dr != self.rank]
        recvs = [(sr, dk()) for (sr, sk, dr, dk) in edges if dr == self.rank and sr != self.rank]
        if sends or recvs:
            # an empty copy stage forces libnbp to flush the deferred bandwidth fits before slots travel
            self._add(abi.STAGE_COPIES, [], "flush")
            self.segments.append(("run", self._seg_start, len(self.stages)))
            self.segments.append(("xchg", sends, recvs))
            self._seg_start = len(self.stages)

    def _deconv_descs(self, cliques):
        """prepCliqueMsgUp -> addLikelihoodsDifferentialCHILD!: approxDeconv between the solved separator beliefs
        of every differential pair, manikde! of the predicted measurements"""
        fg, sp, dec = self.fg, self.fg.solverParams, []
        for c in cliques:
            for i, (a, b, _, fk) in enumerate(self.joint[c].relatives):
                dflt = DFGFactor(f"dummy{c}_{i}", [a, b], _default_relative(fk, fg.getVariable(a).varType), None, 0.0, sp.inflation)
                dec.append(proposal_desc(fg, dflt, b, lambda u, c=c: self.B[(c, u)], self.D[(c, i)],
                                         op_seed(self.seed, PASS_UP, c, 0x4000 + i, 0)))
        return dec

    def _compile_up_asap(self):
        """Up pass: a clique starts its schedule in the stage after its last child finished (the rendezvous of
        the CliqueStateMachine, CliqueStateMachine.jl:221-234: a parent waits for its own children, not for a
        whole tree level).  Stage t batches step t - start[c] of every clique of this rank that is running; the
        critical path is the longest root-to-leaf sum of schedule lengths instead of the sum of the per-level
        maxima.  The stage times are those of the WHOLE tree (the same on every rank), so a message that crosses
        ranks is exchanged at a point both sides agree on: before stage t for the cliques that finished at t.
        The random streams are keyed by (clique, step), so the results do not depend on the batching."""
        tree, owner, rank = self.tree, self.owner, self.rank
        start, finish = {}, {}
        for c in sorted(tree.cliques, key=lambda c: (self.heights[c], c)):
            start[c] = max([finish[x] for x in tree.cliques[c].children] + [0])
            finish[c] = start[c] + len(self.uprounds[c])
        for t in range(max(finish.values()) if finish else 0):
            done = [c for c in sorted(tree.cliques) if finish[c] == t and tree.cliques[c].parent >= 0]
            if self.joint is not None:  # the joint messages of the cliques that have just finished
                dec = self._deconv_descs([c for c in self.cliques if c in done])
                if dec:
                    self._add(abi.STAGE_DECONV, dec, "up")
            self._exchange(self._up_edges([c for c in done if owner[c] != owner[tree.cliques[c].parent]]))
            props, prods = [], []
            for c in self.cliques:
                if not (start[c] <= t < finish[c]):
                    continue
                for lane, k in enumerate(self.uprounds[c][t - start[c]]):
                    v = self.upsched[c][k]
                    p, q = self._update_ops(c, v, self.upfacs[c][v], lambda u, c=c: self.B[(c, u)], self.B[(c, v)], PASS_UP, k,
                                            fresh=self.upfresh[c][k], lane=lane)
                    props += p
                    prods.append(q)
                    self.n_updates_up += 1
            if prods:
                self._add(abi.STAGE_PROPOSALS, props, "up")
                self._add(abi.STAGE_PRODUCTS, prods, "up")

    def _up_edges(self, cliques):
        """up messages that cross a rank boundary: the child's separator beliefs (and, in joint-message mode, the
        KDEs of its differential factors) -> the parent rank's landing slots"""
        tree, owner, edges = self.tree, self.owner, []
        for c in cliques:
            cl = tree.cliques[c]
            for v in cl.separatorIDs:
                edges.append((owner[c], (lambda c=c, v=v: self.B[(c, v)]), owner[cl.parent], (lambda c=c, v=v: self.ghost[(c, v)])))
            for i in range(len(self.joint[c].relatives) if self.joint is not None else 0):
                # D[(c, i)] names the sender's own slot on its rank and the landing slot on the parent's
                edges.append((owner[c], (lambda c=c, i=i: self.D[(c, i)]), owner[cl.parent], (lambda c=c, i=i: self.D[(c, i)])))
        return edges

    def _compile_down_asap(self):
        """Down pass, batched by dependency like the up pass: a clique receives its parent's separator values
        (a points-only copy, or an exchange when the parent lives on another rank) and starts in the stage after
        the parent's last update."""
        tree, owner, rank = self.tree, self.owner, self.rank
        start, finish = {}, {}
        for c in sorted(tree.cliques, key=lambda c: (self.depths[c], c)):
            par = tree.cliques[c].parent
            start[c] = finish[par] if par >= 0 else 0
            finish[c] = start[c] + len(self.dnrounds[c])
        for t in range(max(finish.values()) + 1 if finish else 0):
            # down messages of the cliques that start now: one round, or one more per link of a chain of cliques
            # without updates of their own, which hand the values on within the same time step
            starting = [c for c in sorted(tree.cliques) if start[c] == t and tree.cliques[c].parent >= 0]
            rnd = {}
            for c in sorted(starting, key=lambda c: (self.depths[c], c)):
                rnd[c] = rnd[tree.cliques[c].parent] + 1 if tree.cliques[c].parent in rnd else 0
            for r in sorted(set(rnd.values())):
                self._exchange([(owner[tree.cliques[c].parent], (lambda p=tree.cliques[c].parent, v=v: self.B[(p, v)]), owner[c],
                                 (lambda c=c, v=v: self.B[(c, v)]))
                                for c in starting if rnd[c] == r and owner[c] != owner[tree.cliques[c].parent]
                                for v in tree.cliques[c].separatorIDs])
                msg = [abi.CopyDesc(self.B[(tree.cliques[c].parent, s)], self.B[(c, s)])
                       for c in self.cliques if c in rnd and rnd[c] == r and owner[tree.cliques[c].parent] == rank
                       for s in tree.cliques[c].separatorIDs]
                if msg:
                    self._add(abi.STAGE_COPY_POINTS, msg, "down")
            props, prods = [], []
            for c in self.cliques:
                if not (start[c] <= t < finish[c]):
                    continue
                inclq = set(tree.cliques[c].allIDs)

                def slot_of(u, c=c, inclq=inclq):
                    return self.B[(c, u)] if u in inclq else self.main[u]

                for lane, k in enumerate(self.dnrounds[c][t - start[c]]):
                    v = self.dnsched[c][k]
                    p, q = self._update_ops(c, v, self.dnfacs[c][v], slot_of, self.B[(c, v)], PASS_DOWN, k, lane=lane)
                    props += p
                    prods.append(q)
                    self.n_updates_down += 1
            if prods:
                self._add(abi.STAGE_PROPOSALS, props, "down")
                self._add(abi.STAGE_PRODUCTS, prods, "down")

    def _compile(self):
        tree, fg, rank, owner = self.tree, self.fg, self.rank, self.owner
        if self.snap is not None:
            self._add(abi.STAGE_COPIES, [abi.CopyDesc(self.snap[v], self.main[v]) for v in fg.ls()], "copy")
        # deep copy of the clique sub graphs (SubGraphFunctions.jl:48)
        self._add(abi.STAGE_COPIES, [abi.CopyDesc(self.main[v], self.B[(c, v)]) for c in self.cliques
                                     for v in tree.cliques[c].allIDs], "copy")
        allc = sorted(tree.cliques)
        sp = fg.solverParams
        if not sp.upsolve and not sp.downsolve:
            raise ValueError("must attempt either up or down solve")  # CliqueStateMachine.jl:60
        # ---- up pass: leaves first ------------------------------------------------------------
        # upsolve = false: the cliques are "up-recycled" (tryDownSolveOnly_StateMachine, :485-529): no
        # update runs and the down solve works from the stored beliefs
        maxh = max(self.heights.values())
        single = True if self.asap is None else bool(self.asap)  # "single": batch by dependency (every program by default)
        if sp.upsolve and single:
            self._compile_up_asap()
        for h in (range(maxh + 1) if sp.upsolve and not single else ()):
            level = [c for c in self.cliques if self.heights[c] == h]
            nsteps = max([len(self.upsched[c]) for c in level] + [0])
            for k in range(nsteps):
                props, prods = [], []
                for c in level:
                    sched = self.upsched[c]
                    if k >= len(sched):
                        continue
                    v = sched[k]
                    p, q = self._update_ops(c, v, self.upfacs[c][v], lambda u, c=c: self.B[(c, u)], self.B[(c, v)], PASS_UP, k,
                                            fresh=self.upfresh[c][k])
                    props += p
                    prods.append(q)
                    self.n_updates_up += 1
                self._add(abi.STAGE_PROPOSALS, props, "up")
                self._add(abi.STAGE_PRODUCTS, prods, "up")
            if self.joint is not None:
                dec = self._deconv_descs(level)
                if dec:
                    self._add(abi.STAGE_DECONV, dec, "up")
            # up messages that cross a rank boundary: child's separator beliefs -> parent's ghost slots
            edges = []
            for c in allc:
                cl = tree.cliques[c]
                if self.heights[c] == h and cl.parent >= 0 and owner[c] != owner[cl.parent]:
                    for v in cl.separatorIDs:
                        edges.append((owner[c], (lambda c=c, v=v: self.B[(c, v)]), owner[cl.parent],
                                      (lambda c=c, v=v: self.ghost[(c, v)])))
                    for i in range(len(self.joint[c].relatives) if self.joint is not None else 0):
                        # D[(c, i)] names the sender's own slot on its rank and the landing slot on the parent's
                        edges.append((owner[c], (lambda c=c, i=i: self.D[(c, i)]), owner[cl.parent],
                                      (lambda c=c, i=i: self.D[(c, i)])))
            self._exchange(edges)
        if not sp.downsolve:
            # postUpSolve -> updateFromSubgraph (:595-599): every clique hands its up-solved frontals back
            self._add(abi.STAGE_COPIES, [abi.CopyDesc(self.B[(c, v)], self.main[v]) for c in self.cliques
                                         for v in tree.cliques[c].frontalIDs], "up")
            self.segments.append(("run", self._seg_start, len(self.stages)))
            return
        # roots: posterior = up-solve result (CliqueStateMachine.jl preDownSolve root branch)
        self._add(abi.STAGE_COPIES, [abi.CopyDesc(self.B[(r, v)], self.main[v]) for r in tree.roots if owner[r] == rank
                                     for v in tree.cliques[r].frontalIDs], "down")
        # ---- down pass: root first --------------------------------------------------------------
        maxd = max(self.depths.values())
        done = []
        if single:
            done = [c for dpt in range(1, maxd + 1) for c in self.cliques if self.depths[c] == dpt]
            self._compile_down_asap()
        for dpt in (range(1, maxd + 1) if not single else ()):
            # down messages that cross a rank boundary: parent's values of the child's separators
            edges = []
            for c in allc:
                cl = tree.cliques[c]
                if self.depths[c] == dpt and owner[c] != owner[cl.parent]:
                    for v in cl.separatorIDs:
                        edges.append((owner[cl.parent], (lambda p=cl.parent, v=v: self.B[(p, v)]), owner[c],
                                      (lambda c=c, v=v: self.B[(c, v)])))
            self._exchange(edges)
            level = [c for c in self.cliques if self.depths[c] == dpt]
            # down message: separators := parent's values (updateSubFgFromDownMsgs!).  The child reads them as
            # points (relative proposals), never as a density: a points-only copy, which does not wait for the
            # parent's pending bandwidth fits -- those ride with the next prep launch instead
            msg = [abi.CopyDesc(self.B[(tree.cliques[c].parent, s)], self.B[(c, s)]) for c in level
                   if owner[tree.cliques[c].parent] == rank for s in tree.cliques[c].separatorIDs]
            self._add(abi.STAGE_COPY_POINTS, msg, "down")
            nsteps = max([len(self.dnsched[c]) for c in level] + [0])
            for k in range(nsteps):
                props, prods = [], []
                for c in level:
                    sched = self.dnsched[c]
                    if k >= len(sched):
                        continue
                    v = sched[k]
                    inclq = set(tree.cliques[c].allIDs)

                    def slot_of(u, c=c, inclq=inclq):
                        return self.B[(c, u)] if u in inclq else self.main[u]

                    p, q = self._update_ops(c, v, self.dnfacs[c][v], slot_of, self.B[(c, v)], PASS_DOWN, k)
                    props += p
                    prods.append(q)
                    self.n_updates_down += 1
                self._add(abi.STAGE_PROPOSALS, props, "down")
                self._add(abi.STAGE_PRODUCTS, prods, "down")
            done += level
        # transferUpdateSubGraph!: frontals -> main graph (CliqueStateMachine.jl:928-966), once for the whole pass:
        # during the pass main[u] is only read for variables of cliques further down, which are written later
        # anyway, so the deferral changes no value -- and the bandwidth fits of all those beliefs run in one
        # chip-filling launch instead of one small launch per tree level
        self._add(abi.STAGE_COPIES, [abi.CopyDesc(self.B[(c, v)], self.main[v]) for c in done
                                     for v in tree.cliques[c].frontalIDs], "down")
        self.segments.append(("run", self._seg_start, len(self.stages)))

    def alg_bytes_by_kernel(self):
        return dict(self.alg)

    @property
    def n_messages(self):
        """one LikelihoodMessage per tree edge and direction (CliqueStateMachine.jl:590-593, 900-903);
        counted over the WHOLE tree (all ranks)"""
        sp = self.fg.solverParams
        return (int(sp.upsolve) + int(sp.downsolve)) * sum(1 for c in self.tree.cliques.values() if c.parent >= 0)

    def stats(self):
        np_, nq = 0, 0
        for kind, d in self.stages:
            if kind == abi.STAGE_PROPOSALS:
                np_ += len(d)
            elif kind == abi.STAGE_PRODUCTS:
                nq += len(d)
        return {"stages": len(self.stages), "proposals": np_, "products": nq, "updates_up": self.n_updates_up,
                "updates_down": self.n_updates_down, "messages": self.n_messages, "slots": self.n_slots,
                "alg_bytes": self.alg_bytes, "cliques": len(self.cliques)}


def _initialised_subgraph(fg):
    """The part of the graph a tree solve can work on.  Variables graph initialisation could not reach (no
    path to a prior) stay uninitialised and unsolved in the reference -- their cliques give up in tree
    initialisation (CliqueStateMachine.jl:562-575, test/testBasicTreeInit.jl:81-106) -- so they and the factors
    touching them are left out here.  Returns `fg` itself when everything is initialised."""
    out = [v for v in fg.ls() if not fg.getVariable(v).initialized]
    if not out:
        return fg
    from .factorgraph import FactorGraph
    sub = FactorGraph(fg.solverParams)
    drop = set(out)
    for v in fg.ls():
        if v not in drop:
            sub.variables[v] = fg.getVariable(v)  # shared objects: results land in the caller's graph
            sub._adj[v] = []
    for f in fg.lsf():
        fc = fg.getFactor(f)
        if not drop.intersection(fc.variables):
            sub.factors[f] = fc
            for v in fc.variables:
                sub._adj[v].append(f)
    return sub


def solveTree(fg, tree=None, eliminationOrder=None, backend=None, seed=0, ordering="qr", return_timing=False, native=None):
    """solveTree!(dfg; eliminationOrder) -> tree   (SolverAPI.jl:326-493).
    graphinit -> buildTreeReset! -> up pass -> down pass -> posteriors written back to `fg`."""
    sp = fg.solverParams
    t0 = time.perf_counter()
    if sp.graphinit:
        # a backend INSTANCE is sized for the tree solve: graph initialisation makes its own from the same class
        initAll(fg, backend=backend if (backend is None or isinstance(backend, type) or not hasattr(backend, 'slot_write'))
                else type(backend), seed=seed)
    t1 = time.perf_counter()
    whole = fg
    fg = _initialised_subgraph(fg)
    if fg is not whole:
        if tree is not None:
            raise ValueError("a prebuilt tree needs every variable initialised: "
                             + ", ".join(v for v in whole.ls() if not whole.getVariable(v).initialized))
        if eliminationOrder is not None:
            eliminationOrder = [v for v in eliminationOrder if v in fg.variables]
    if tree is None:
        tree = bayestree.buildTreeReset(fg, eliminationOrder, ordering)
    t2 = time.perf_counter()
    # the schedule is compiled by the native host (nbp_host.h) when the solve runs on libnbp, by the
    # Python mirror otherwise (oracle backend in the tests); both produce the same descriptors
    use_native = native if native is not None else (backend is None or backend is HipBackend or isinstance(backend, HipBackend)
                                                    or getattr(backend, "is_hip", False))
    if use_native:
        from . import native_host
        ng = native_host.NativeGraph.from_fg(fg)
        tp = ng.build_tree(tree.eliminationOrder)
        tp.plan_slots(False)
        ng.place_densities(fg, tp.density_slot0())
    else:
        tp = TreeProgram(fg, tree, seed=seed)
    be, own = _make_backend(backend, sp.N, tp.n_slots)
    try:
        for v in fg.ls():
            var = fg.getVariable(v)
            be.belief_write(tp.main[v], var.varType.manifold, var.val, var.bw)
        write_densities(fg, be)
        prog = None
        try:
            prog = tp.compile(be, seed) if use_native else be.program(tp.stages, lazy_bandwidth=True)
            t3 = time.perf_counter()
            prog.run()
            be.synchronize()
            t4 = time.perf_counter()
            for v in fg.ls():
                var = fg.getVariable(v)
                pts, bw, _ = be.belief_read(tp.main[v], var.varType.manifold)
                setValKDE(fg, v, pts, bw, True)
                var.solvedCount += 1
        finally:  # the program goes before its context, on the error path too
            if prog is not None:
                prog.close()
    finally:
        if own:
            be.close()
    if return_timing:
        st = tp.stats()
        st.setdefault("cliques", len(tree.cliques))
        return tree, {"init_s": t1 - t0, "tree_s": t2 - t1, "compile_s": t3 - t2, "solve_s": t4 - t3, **st}
    return tree
