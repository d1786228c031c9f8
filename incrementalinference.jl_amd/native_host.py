"""ctypes binding of the native (C++) host side, include/nbp_host.h: graph container, nested-dissection
ordering, Bayes tree + clique potentials + Gibbs schedules, and the whole-tree schedule compiler.
`NativeGraph.from_fg(fg)` mirrors a Python FactorGraph; everything after that runs in libnbp."""
import ctypes as C

import numpy as np

from . import abi

i32, i64, f64 = C.c_int32, C.c_int64, C.c_double


SOLVER_STORED_MEASUREMENTS, SOLVER_MSG_LIKELIHOODS = 1, 2  # enum nbp_solver_flag


class SolverParamsC(C.Structure):
    _fields_ = [("N", i32), ("gibbs_iters", i32), ("inflate_cycles", i32), ("product_niter", i32), ("upsolve", i32),
                ("downsolve", i32), ("limitfixeddown", i32), ("flags", i32), ("spread_nh", f64), ("inflation", f64),
                ("null_surplus_add", f64)]


class FactorSpec(C.Structure):
    _fields_ = [("factor_kind", i32), ("nvars", i32), ("vars", i32 * abi.MAXV), ("ncomp", i32), ("has_multihypo", i32),
                ("partial_mask", i32), ("multihypo", f64 * abi.MAXV), ("nullhypo", f64), ("inflation", f64),
                ("comp", (f64 * abi.COMP_STRIDE) * abi.MAXC)]


class CliqueInfo(C.Structure):
    _fields_ = [(k, i32) for k in ("parent", "nfrontals", "nseparators", "nchildren", "npotentials", "nup", "ndown")]


class TreeStats(C.Structure):
    _fields_ = [(k, i64) for k in ("stages", "proposals", "products", "updates_up", "updates_down", "messages", "slots",
                                   "alg_bytes", "alg_bytes_proposal", "alg_bytes_prep", "alg_bytes_product")]


Xfer = abi.Xfer
XCHG_FN = C.CFUNCTYPE(i32, C.c_void_p, C.POINTER(Xfer), i32, C.POINTER(Xfer), i32)  # nbp_exchange_fn


class TreeBeliefC(C.Structure):
    """nbp_tree_belief: TreeBelief (val, bw, infoPerCoord) in host buffers"""
    _fields_ = [("pts", C.POINTER(f64)), ("bw", C.POINTER(f64)), ("ipc", C.POINTER(f64)), ("n_pts", i32), ("handle", i32)]


class CliqueDescC(C.Structure):
    _fields_ = [("clique_id", i32), ("nvars", i32), ("nfrontals", i32), ("nseparators", i32),
                ("manifold", C.POINTER(i32)), ("ismargin", C.POINTER(i32)), ("nfactors", i32), ("factors", C.POINTER(FactorSpec)),
                ("n_direct_frtl_msg", i32), ("n_msgskip", i32), ("n_itervar", i32), ("n_direct_prior_msg", i32),
                ("direct_frtl_msg", C.POINTER(i32)), ("msgskip", C.POINTER(i32)), ("itervar", C.POINTER(i32)),
                ("direct_prior_msg", C.POINTER(i32)), ("nmsgs", i32), ("msg_var", C.POINTER(i32)),
                ("msg_belief", C.POINTER(TreeBeliefC)), ("factor_density", C.POINTER(TreeBeliefC)),
                ("factor_meas_kde", C.POINTER(TreeBeliefC)), ("n_diff", i32), ("reserved_", i32),
                ("diff_a", C.POINTER(i32)), ("diff_b", C.POINTER(i32)), ("diff_kind", C.POINTER(i32))]


CLIQ_UPSOLVED, CLIQ_DOWNSOLVED = 3, 5  # enum nbp_cliq_status

HOST_EXPORTS = ["nbp_graph_create", "nbp_graph_destroy", "nbp_graph_add_variable", "nbp_graph_add_factor",
                "nbp_graph_set_variable_flags", "nbp_graph_num_variables", "nbp_graph_num_factors",
                "nbp_graph_order_nested_dissection", "nbp_graph_init_plan", "nbp_graph_init_num_variables", "nbp_graph_init_variables",
                "nbp_graph_init_num_stages", "nbp_graph_init_stage", "nbp_graph_init_compile", "nbp_tree_build", "nbp_tree_destroy", "nbp_tree_num_cliques",
                "nbp_tree_clique", "nbp_tree_clique_idlists", "nbp_tree_max_schedule", "nbp_tree_plan_slots", "nbp_tree_main_slots", "nbp_tree_compile", "nbp_tree_schedule",
                "nbp_tree_get_stats", "nbp_tree_num_stages", "nbp_tree_stage", "nbp_clique_slots", "nbp_clique_upsolve", "nbp_clique_upsolve_joint", "nbp_clique_downsolve", "nbp_clique_solve_batch", "nbp_clique_seam_times",
                "nbp_clique_submit_batch", "nbp_clique_wait", "nbp_resident_write", "nbp_resident_read", "nbp_resident_copy",
                "nbp_tree_partition", "nbp_tree_set_owner", "nbp_tree_num_segments", "nbp_tree_segment",
                "nbp_tree_run_sharded", "nbp_tree_run_sharded_cb",
                "nbp_graph_num_densities", "nbp_graph_density_factors", "nbp_graph_init_density_slot0", "nbp_tree_density_slot0"]

_declared = False


def _lib():
    global _declared
    lib = abi.load_library()
    if not _declared:
        vp, ip = C.c_void_p, C.POINTER(i32)
        lib.nbp_graph_create.argtypes = [C.POINTER(SolverParamsC), C.POINTER(vp)]
        lib.nbp_graph_destroy.argtypes = [vp]
        lib.nbp_graph_add_variable.argtypes = [vp, i32]
        lib.nbp_graph_add_factor.argtypes = [vp, C.POINTER(FactorSpec)]
        lib.nbp_graph_set_variable_flags.argtypes = [vp, i32, i32, i32]
        lib.nbp_graph_num_variables.argtypes = [vp]
        lib.nbp_graph_num_factors.argtypes = [vp]
        lib.nbp_graph_order_nested_dissection.argtypes = [vp, ip]
        lib.nbp_graph_init_plan.argtypes = [vp, C.c_uint64]
        lib.nbp_graph_init_num_variables.argtypes = [vp]
        lib.nbp_graph_num_densities.argtypes = [vp]
        lib.nbp_graph_density_factors.argtypes = [vp, ip]
        lib.nbp_graph_init_density_slot0.argtypes = [vp]
        lib.nbp_tree_density_slot0.argtypes = [vp]
        lib.nbp_graph_init_variables.argtypes = [vp, ip]
        lib.nbp_graph_init_num_stages.argtypes = [vp]
        lib.nbp_graph_init_stage.argtypes = [vp, i32, ip, ip, vp, i64]
        lib.nbp_graph_init_compile.argtypes = [vp, vp, C.POINTER(vp)]
        lib.nbp_tree_build.argtypes = [vp, ip, i32, C.POINTER(vp)]
        lib.nbp_tree_destroy.argtypes = [vp]
        lib.nbp_tree_num_cliques.argtypes = [vp]
        lib.nbp_tree_clique.argtypes = [vp, i32, C.POINTER(CliqueInfo), ip, ip, ip, ip, ip, ip]
        lib.nbp_tree_max_schedule.argtypes = [vp]
        lib.nbp_tree_clique_idlists.argtypes = [vp, i32, ip, ip, ip, ip, ip]
        lib.nbp_tree_plan_slots.argtypes = [vp, i32]
        lib.nbp_tree_main_slots.argtypes = [vp, ip, ip]
        lib.nbp_tree_compile.argtypes = [vp, vp, C.c_uint64, C.POINTER(vp)]
        lib.nbp_tree_schedule.argtypes = [vp, C.c_uint64]
        lib.nbp_tree_get_stats.argtypes = [vp, C.POINTER(TreeStats)]
        lib.nbp_tree_num_stages.argtypes = [vp]
        lib.nbp_tree_stage.argtypes = [vp, i32, ip, ip, vp, i64]
        lib.nbp_tree_partition.argtypes = [vp, i32, ip]
        lib.nbp_tree_set_owner.argtypes = [vp, ip, i32]
        lib.nbp_tree_num_segments.argtypes = [vp]
        lib.nbp_tree_segment.argtypes = [vp, i32, ip, ip, ip, ip, ip, C.POINTER(Xfer), C.POINTER(Xfer), i32]
        lib.nbp_tree_run_sharded.argtypes = [vp, vp, vp, vp]
        lib.nbp_tree_run_sharded_cb.argtypes = [vp, vp, XCHG_FN, vp]
        lib.nbp_clique_slots.argtypes = [C.POINTER(CliqueDescC)]
        for fn in (lib.nbp_clique_upsolve, lib.nbp_clique_downsolve):
            fn.argtypes = [vp, C.POINTER(SolverParamsC), C.POINTER(CliqueDescC), C.c_uint64, C.POINTER(TreeBeliefC), ip]
        lib.nbp_clique_upsolve_joint.argtypes = [vp, C.POINTER(SolverParamsC), C.POINTER(CliqueDescC), C.c_uint64, C.POINTER(TreeBeliefC),
                                                 C.POINTER(TreeBeliefC), ip]
        lib.nbp_clique_solve_batch.argtypes = [vp, C.POINTER(CliqueRequestC), i32]
        lib.nbp_clique_seam_times.argtypes = [C.POINTER(C.c_double), i32]
        for n in HOST_EXPORTS:
            getattr(lib, n).restype = i32
        _declared = True
    return lib


def _check(rc):
    if rc < 0:
        msg = f"libnbp host status {rc}: {abi.load_library().nbp_last_error().decode()}"
        raise (ValueError if rc in (-1, -4) else RuntimeError)(msg)  # NBP_ERR_ARG / NBP_ERR_RANGE: bad input
    return rc


def _stages_of(getter, count):
    esz = {abi.STAGE_PROPOSALS: C.sizeof(abi.ProposalDesc), abi.STAGE_PRODUCTS: C.sizeof(abi.ProductDesc),
           abi.STAGE_COPIES: C.sizeof(abi.CopyDesc), abi.STAGE_DECONV: C.sizeof(abi.ProposalDesc),
           abi.STAGE_COPY_POINTS: C.sizeof(abi.CopyDesc)}
    out = []
    for s in range(count):
        kind, n = i32(), i32()
        _check(getter(s, C.byref(kind), C.byref(n), None, 0))
        nb = n.value * esz[kind.value]
        buf = (C.c_char * max(1, nb))()
        _check(getter(s, C.byref(kind), C.byref(n), buf, nb))
        out.append((kind.value, bytes(buf[:nb])))
    return out


def solver_params_c(sp):
    flags = (0 if sp.alwaysFreshMeasurements else SOLVER_STORED_MEASUREMENTS) | \
            (SOLVER_MSG_LIKELIHOODS if getattr(sp, "useMsgLikelihoods", False) else 0)
    return SolverParamsC(sp.N, sp.gibbsIters, sp.inflateCycles, sp.productNiter, int(sp.upsolve), int(sp.downsolve),
                         int(getattr(sp, "limitfixeddown", False)), flags, sp.spreadNH, sp.inflation, sp.nullSurplusAdd)


def factor_spec(f, index):
    """DFGFactor -> nbp_factor_spec; `index`: label -> variable id"""
    s = FactorSpec()
    s.factor_kind, s.nvars = f.fnc.kind, len(f.variables)
    for i, v in enumerate(f.variables):
        s.vars[i] = index[v]
    comps = f.fnc.components()
    s.ncomp = len(comps)
    for c, comp in enumerate(comps):
        w, mu, L = comp[:3]
        s.comp[c][0] = w
        for i in range(min(3, len(mu))):
            s.comp[c][1 + i] = float(mu[i])
        for i in range(min(3, L.shape[0])):
            for j in range(i + 1):
                s.comp[c][4 + 3 * i + j] = float(L[i, j])
        if len(comp) > 3 and comp[3] != abi.DIST_GAUSSIAN:  # scalar Uniform / Rayleigh (enum nbp_dist)
            s.comp[c][12] = float(comp[3])
    if f.multihypo is not None:
        s.has_multihypo = 1
        for i, p in enumerate(f.multihypo):
            s.multihypo[i] = p
    s.nullhypo, s.inflation = f.nullhypo, f.inflation
    s.partial_mask = getattr(f.fnc, "partial_mask", 0)
    return s


class Belief:
    """host-side TreeBelief: pts (N x P), bw (D), ipc (D)"""

    def __init__(self, manifold, pts, bw=None, ipc=None):
        D = abi.MANIFOLD_DIM[manifold]
        self.manifold = manifold
        self.pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, abi.MANIFOLD_P[manifold]).copy()
        self.bw = np.zeros(D) if bw is None else np.ascontiguousarray(bw, dtype=np.float64).copy()
        self.ipc = np.zeros(D) if ipc is None else np.ascontiguousarray(ipc, dtype=np.float64).copy()

    def copy(self):
        return Belief(self.manifold, self.pts, self.bw, self.ipc)

    def c(self, capacity=None):
        """the C view; `capacity`: room for that many points (a clique call hands N points back whatever came in)"""
        dp = C.POINTER(f64)
        n = self.pts.shape[0]
        if capacity is not None and capacity > n:
            buf = np.zeros((capacity, self.pts.shape[1]))
            buf[:n] = self.pts
            self._buf, self._n_in = buf, n
        else:
            self._buf, self._n_in = self.pts, n
        return TreeBeliefC(self._buf.ctypes.data_as(dp), self.bw.ctypes.data_as(dp), self.ipc.ctypes.data_as(dp), n, 0)

    def take(self, cview):
        """after a call that wrote this belief: adopt the points that came back"""
        self.pts = self._buf[:cview.n_pts]


class _CliqueCall:
    """one prepared clique call: the C structures (kept alive here) and where the results go"""


def _clique_prepare(backend, sp, clique_id, variables, nfrontals, nseparators, manifolds, factors, beliefs, seed, down=False,
                    ismargin=None, lists=None, msgs=(), meas_kdes=None, diffs=()):
    """nbp_clique_upsolve / nbp_clique_downsolve (include/nbp_host.h) -- the per-clique seam of the CliqueStateMachine:
    upGibbsCliqueDensity (SolveTree.jl:164-239) / solveCliqDownFrontalProducts! (CliqStateMachineUtils.jl:479-571).

    variables: labels, frontals first, then separators, then (down) the others; factors: DFGFactor list; beliefs:
    {label: Belief} (updated in place); lists: {"directFrtlMsg", "msgskip", "itervar", "directPriorMsg"} of labels;
    msgs: [(label, Belief)] the children's up messages.  Joint messages (useMsgLikelihoods): meas_kdes = per factor None or
    the Belief (Euclid(zdim), measurement coordinates) of a differential factor received from a child; diffs =
    [(label_a, label_b, kind)] the differential factors to send up -- then the call returns (status, [Belief]).
    Returns the CliqStatus code otherwise."""
    lib = _lib()
    idx = {v: i for i, v in enumerate(variables)}
    q = CliqueDescC()
    q.clique_id, q.nvars, q.nfrontals, q.nseparators = clique_id, len(variables), nfrontals, nseparators
    man = (i32 * len(variables))(*manifolds)
    q.manifold = man
    mg = (i32 * len(variables))(*[int(bool(m)) for m in (ismargin or [0] * len(variables))])
    q.ismargin = mg
    specs = (FactorSpec * max(1, len(factors)))(*[factor_spec(f, idx) for f in factors])
    q.nfactors, q.factors = len(factors), specs
    keep = [man, mg, specs]
    for name, field in (("directFrtlMsg", "direct_frtl_msg"), ("msgskip", "msgskip"), ("itervar", "itervar"), ("directPriorMsg", "direct_prior_msg")):
        l = [idx[v] for v in (lists or {}).get(name, [])]
        arr = (i32 * max(1, len(l)))(*l)
        keep.append(arr)
        setattr(q, "n_" + field, len(l))
        setattr(q, field, arr)
    mv = (i32 * max(1, len(msgs)))(*[idx[v] for v, _ in msgs])
    mb = (TreeBeliefC * max(1, len(msgs)))(*[b.c() for _, b in msgs])
    q.nmsgs, q.msg_var, q.msg_belief = len(msgs), mv, mb
    dens = []  # the densities of the pass-through priors ride along with their factors
    for f in factors:
        if f.fnc.kind == abi.F_PASSTHROUGH:
            pts, bw = f.fnc.density_belief()
            dens.append(Belief(f.fnc.varType.manifold, pts, bw))
        elif getattr(f.fnc, "table", None) is not None:  # the table of an AliasingScalarSampler measurement
            pts, bw = f.fnc.density_belief()
            dens.append(Belief(f.fnc.density_manifold, pts, bw))
        else:
            dens.append(None)
    if any(d is not None for d in dens):
        fd = (TreeBeliefC * len(factors))(*[d.c() if d is not None else TreeBeliefC() for d in dens])
        q.factor_density = fd
        keep += [fd, dens]
    if meas_kdes is not None and any(m is not None for m in meas_kdes):
        mk = (TreeBeliefC * len(factors))(*[m.c() if m is not None else TreeBeliefC() for m in meas_kdes])
        q.factor_meas_kde = mk
        keep += [mk, meas_kdes]
    diff_out = []
    if diffs:
        da = (i32 * len(diffs))(*[idx[a] for a, _, _ in diffs])
        db = (i32 * len(diffs))(*[idx[b] for _, b, _ in diffs])
        dk = (i32 * len(diffs))(*[k for _, _, k in diffs])
        q.n_diff, q.diff_a, q.diff_b, q.diff_kind = len(diffs), da, db, dk
        keep += [da, db, dk]
        for a, _, k in diffs:
            zd = abi.MANIFOLD_DIM[manifolds[idx[a]]] if k == abi.F_LINREL else (3 if k == abi.F_SE2 else 1)
            diff_out.append(Belief(zd, np.zeros((sp.N, zd)), np.zeros(zd)))  # Euclid(zd): the manifold code is the dimension
    k = _CliqueCall()
    k.need = _check(lib.nbp_clique_slots(C.byref(q)))
    k.q, k.keep, k.variables, k.beliefs, k.seed, k.down = q, keep, variables, beliefs, seed, bool(down)
    k.bel = (TreeBeliefC * len(variables))(*[beliefs[v].c(capacity=sp.N) for v in variables])
    k.p = solver_params_c(sp)
    k.joint = bool(diffs) and not down
    k.diff_out = diff_out
    k.dout = (TreeBeliefC * len(diffs))(*[b.c() for b in diff_out]) if k.joint else None
    return k


def _clique_finish(k, status):
    if k.joint:
        for i, b in enumerate(k.diff_out):
            b.take(k.dout[i])
    for i, v in enumerate(k.variables):
        k.beliefs[v].take(k.bel[i])
    return (status, k.diff_out) if k.joint else status


def clique_solve(backend, *args, **kwargs):
    """one clique call; arguments and return value: see _clique_prepare"""
    lib = _lib()
    k = _clique_prepare(backend, *args, **kwargs)
    if k.need > backend.n_slots:
        raise ValueError(f"the context has {backend.n_slots} slots, this clique needs {k.need}")
    status = i32(0)
    if k.joint:
        _check(lib.nbp_clique_upsolve_joint(backend._ctx, C.byref(k.p), C.byref(k.q), C.c_uint64(k.seed), k.bel, k.dout, C.byref(status)))
    else:
        fn = lib.nbp_clique_downsolve if k.down else lib.nbp_clique_upsolve
        _check(fn(backend._ctx, C.byref(k.p), C.byref(k.q), C.c_uint64(k.seed), k.bel, C.byref(status)))
    return _clique_finish(k, status.value)


class CliqueRequestC(C.Structure):
    """nbp_clique_request (include/nbp_host.h)"""
    _fields_ = [("params", C.POINTER(SolverParamsC)), ("clique", C.POINTER(CliqueDescC)), ("seed", C.c_uint64),
                ("beliefs", C.POINTER(TreeBeliefC)), ("diff_out", C.POINTER(TreeBeliefC)), ("down", i32), ("status", i32)]


def clique_solve_batch(backend, calls):
    """nbp_clique_solve_batch: cliques that do not depend on each other (a tree level) in ONE call.  calls = [(args, kwargs)]
    of clique_solve without the backend; -> the list of what clique_solve returns for each."""
    lib = _lib()
    ks = [_clique_prepare(backend, *a, **kw) for a, kw in calls]
    need = sum(k.need for k in ks)
    if need > backend.n_slots:
        raise ValueError(f"the context has {backend.n_slots} slots, these cliques need {need}")
    req = (CliqueRequestC * max(1, len(ks)))()
    for r, k in zip(req, ks):
        r.params, r.clique, r.seed, r.beliefs, r.down = C.pointer(k.p), C.pointer(k.q), k.seed, k.bel, int(k.down)
        if k.joint:
            r.diff_out = k.dout
    _check(lib.nbp_clique_solve_batch(backend._ctx, req, len(ks)))
    return [_clique_finish(k, r.status) for r, k in zip(req, ks)]




class NativeGraph:
    def __init__(self, sp):
        self.lib = _lib()
        p = solver_params_c(sp)
        self._g = C.c_void_p()
        _check(self.lib.nbp_graph_create(C.byref(p), C.byref(self._g)))
        self.labels, self.flabels = [], []

    @classmethod
    def from_fg(cls, fg):
        g = cls(fg.solverParams)
        idx = {}
        for v in fg.ls():
            var = fg.getVariable(v)
            idx[v] = _check(g.lib.nbp_graph_add_variable(g._g, var.varType.manifold))
            g.lib.nbp_graph_set_variable_flags(g._g, idx[v], int(var.initialized), int(var.ismargin))
            g.labels.append(v)
        for fl in fg.lsf():
            s = factor_spec(fg.getFactor(fl), idx)
            _check(g.lib.nbp_graph_add_factor(g._g, C.byref(s)))
            g.flabels.append(fl)
        g.index = idx
        return g

    def init_plan(self, seed):
        """-> (slots needed, [labels initialised by the plan])"""
        n = _check(self.lib.nbp_graph_init_plan(self._g, C.c_uint64(seed)))
        k = self.lib.nbp_graph_init_num_variables(self._g)
        out = (i32 * max(1, k))()
        _check(self.lib.nbp_graph_init_variables(self._g, out))
        return n, [self.labels[out[i]] for i in range(k)]

    def density_factors(self):
        """labels of the pass-through priors, in the order of their density slots"""
        k = self.lib.nbp_graph_num_densities(self._g)
        out = (i32 * max(1, k))()
        _check(self.lib.nbp_graph_density_factors(self._g, out))
        return [self.flabels[out[i]] for i in range(k)]

    def init_density_slot0(self):
        return _check(self.lib.nbp_graph_init_density_slot0(self._g))

    def place_densities(self, fg, slot0):
        """tell the factor objects which slot their density goes to (solver.write_densities writes them)"""
        for i, fl in enumerate(self.density_factors()):
            fg.getFactor(fl).fnc.slot = slot0 + i

    def init_stages(self):
        return _stages_of(lambda s, kind, n, buf, cap: self.lib.nbp_graph_init_stage(self._g, s, kind, n, buf, cap),
                          self.lib.nbp_graph_init_num_stages(self._g))

    def init_compile(self, backend):
        from .backend import HipProgram
        p = C.c_void_p()
        _check(self.lib.nbp_graph_init_compile(self._g, backend._ctx, C.byref(p)))
        prog = HipProgram.__new__(HipProgram)
        prog.backend, prog._p, prog.n_stages = backend, p, self.lib.nbp_graph_init_num_stages(self._g)
        backend._programs.add(prog)
        return prog

    def order_nested_dissection(self):
        out = (i32 * len(self.labels))()
        _check(self.lib.nbp_graph_order_nested_dissection(self._g, out))
        return [self.labels[i] for i in out]

    def build_tree(self, order):
        arr = (i32 * len(order))(*[self.index[v] for v in order])
        t = C.c_void_p()
        _check(self.lib.nbp_tree_build(self._g, arr, len(order), C.byref(t)))
        return NativeTree(self, t)

    def close(self):
        if self._g:
            self.lib.nbp_graph_destroy(self._g)
            self._g = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class NativeTree:
    def __init__(self, graph, handle):
        self.g, self.lib, self._t = graph, graph.lib, handle

    @property
    def n_cliques(self):
        return self.lib.nbp_tree_num_cliques(self._t)

    def clique(self, k):
        nv, nf, ms = len(self.g.labels), len(self.g.flabels), max(1, self.lib.nbp_tree_max_schedule(self._t))
        info = CliqueInfo()
        fr, sp, ch = (i32 * nv)(), (i32 * nv)(), (i32 * max(1, self.n_cliques))()
        po, up, dn = (i32 * max(1, nf))(), (i32 * ms)(), (i32 * ms)()
        _check(self.lib.nbp_tree_clique(self._t, k, C.byref(info), fr, sp, ch, po, up, dn))
        L, F = self.g.labels, self.g.flabels
        return {"parent": info.parent, "frontals": [L[fr[i]] for i in range(info.nfrontals)],
                "separators": [L[sp[i]] for i in range(info.nseparators)], "children": [ch[i] for i in range(info.nchildren)],
                "potentials": [F[po[i]] for i in range(info.npotentials)], "up": [L[up[i]] for i in range(info.nup)],
                "down": [L[dn[i]] for i in range(info.ndown)]}

    def partition(self, world):
        """owner rank of every clique: {clique id: rank} (dist_solver.partition_cliques in C++)"""
        out = (i32 * max(1, self.n_cliques))()
        _check(self.lib.nbp_tree_partition(self._t, world, out))
        return {k + 1: out[k] for k in range(self.n_cliques)}

    def set_owner(self, owner, rank):
        """compile this rank's share only (owner: {clique id: rank} or None for a single rank)"""
        if owner is None:
            _check(self.lib.nbp_tree_set_owner(self._t, None, 0))
            return
        arr = (i32 * self.n_cliques)(*[owner[k + 1] for k in range(self.n_cliques)])
        _check(self.lib.nbp_tree_set_owner(self._t, arr, rank))

    def segments(self):
        """[("run", first, last) | ("xchg", [(peer, slot)], [(peer, slot)])] of the last compile (solver.TreeProgram.segments)"""
        out = []
        for i in range(self.lib.nbp_tree_num_segments(self._t)):
            kind, a, b, ns, nr = i32(), i32(), i32(), i32(), i32()
            _check(self.lib.nbp_tree_segment(self._t, i, C.byref(kind), C.byref(a), C.byref(b), C.byref(ns), C.byref(nr), None, None, 0))
            if kind.value == 0:
                out.append(("run", a.value, b.value))
                continue
            cap = max(1, ns.value, nr.value)
            sx, rx = (Xfer * cap)(), (Xfer * cap)()
            _check(self.lib.nbp_tree_segment(self._t, i, None, None, None, None, None, sx, rx, cap))
            out.append(("xchg", [(sx[k].peer, sx[k].slot) for k in range(ns.value)], [(rx[k].peer, rx[k].slot) for k in range(nr.value)]))
        return out

    def run_sharded(self, prog, backend=None, exchange=None):
        """one solve of this rank's share from C (nbp_tree_run_sharded): the stage segments of the last compile and the
        separator exchanges between them.  `backend` with a communicator (HipBackend.comm_create): grouped RCCL send /
        recv on the library stream; `exchange(sends, recvs)`: the caller's transport, called back from the C loop."""
        if exchange is None:
            comm = getattr(backend, "_comm", None) if backend is not None else None
            _check(self.lib.nbp_tree_run_sharded(self._t, prog._p, backend._ctx if comm else None, comm))
            return
        err = []

        def cb(_user, sx, ns, rx, nr):
            try:
                exchange([(sx[k].peer, sx[k].slot) for k in range(ns)], [(rx[k].peer, rx[k].slot) for k in range(nr)])
                return 0
            except Exception as e:  # noqa: BLE001 -- reported after the C loop returns
                err.append(e)
                return -2
        fn = XCHG_FN(cb)
        rc = self.lib.nbp_tree_run_sharded_cb(self._t, prog._p, fn, None)
        if err:
            raise err[0]
        _check(rc)

    def density_slot0(self):
        return _check(self.lib.nbp_tree_density_slot0(self._t))

    def plan_slots(self, snapshot=False):
        self.n_slots = _check(self.lib.nbp_tree_plan_slots(self._t, int(snapshot)))
        nv = len(self.g.labels)
        m, s = (i32 * nv)(), (i32 * nv)()
        _check(self.lib.nbp_tree_main_slots(self._t, m, s if snapshot else None))
        self.main = {v: m[i] for i, v in enumerate(self.g.labels)}
        self.snap = {v: s[i] for i, v in enumerate(self.g.labels)} if snapshot else None
        return self.n_slots

    def compile(self, backend, seed):
        """-> a program object with run/reseed/close (backend.HipProgram interface) living on `backend`"""
        from .backend import HipProgram
        p = C.c_void_p()
        _check(self.lib.nbp_tree_compile(self._t, backend._ctx, C.c_uint64(seed), C.byref(p)))
        prog = HipProgram.__new__(HipProgram)
        prog.backend, prog._p, prog.n_stages = backend, p, self.lib.nbp_tree_num_stages(self._t)
        backend._programs.add(prog)
        return prog

    def schedule(self, seed):
        """build the stage descriptors without touching a device"""
        _check(self.lib.nbp_tree_schedule(self._t, C.c_uint64(seed)))

    def stats(self):
        st = TreeStats()
        _check(self.lib.nbp_tree_get_stats(self._t, C.byref(st)))
        return {k: getattr(st, k) for k, _ in TreeStats._fields_}

    def stages(self):
        """[(kind, bytes)] of the last compile (tests compare them with solver.TreeProgram's)"""
        return _stages_of(lambda s, kind, n, buf, cap: self.lib.nbp_tree_stage(self._t, s, kind, n, buf, cap),
                          self.lib.nbp_tree_num_stages(self._t))

    def close(self):
        if self._t:
            self.lib.nbp_tree_destroy(self._t)
            self._t = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def clique_seam_times(mode=0):
    """nbp_clique_seam_times: host wall clock the clique calls of this process have spent, by phase (seconds) --
    planning, beliefs in, program assembly, launches (+ device when mode 2 was set), beliefs out -- and the number of calls.
    mode 1 resets, mode 2 resets and makes every call wait for the device after its launches."""
    out = (C.c_double * 6)()
    _check(_lib().nbp_clique_seam_times(out, mode))
    return dict(zip(("planning_s", "beliefs_in_s", "assembly_s", "launches_s", "beliefs_out_s", "calls"), list(out)))
