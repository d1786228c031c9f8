"""Host-side mirror of the reference's graph API for the hot path.

Names follow IncrementalInference.jl / DistributedFactorGraphs.jl (Julia `foo!` -> Python `foo`):
variables  ContinuousScalar, ContinuousEuclid(N), Circular, SpecialEuclidean2
           (src/Variables/DefaultVariables.jl:9-19,37-38,52; test/testSpecialEuclidean2Mani.jl:14)
factors    Prior, PriorCircular, ManifoldPrior, LinearRelative, CircularCircular, ManifoldFactor,
           EuclidDistance, Mixture, MsgPrior   (src/Factors/*.jl)
graph      initfg, addVariable, addFactor(multihypo=, nullhypo=, inflation=), getVariable, ls, lsf
           (src/services/FactorGraph.jl:587-632, 824-875)
params     SolverParams with the reference defaults (src/entities/SolverParams.jl:12-75)

Only bookkeeping lives here; all particle arithmetic happens in libnbp behind include/nbp.h.
"""
from dataclasses import dataclass, field

import numpy as np

from . import abi


# ------------------------------------------------------------------------------------------------
# distributions (only what the factor set samples from)
# ------------------------------------------------------------------------------------------------
class Normal:
    def __init__(self, mu=0.0, sigma=1.0):
        self.mu, self.sigma = float(mu), float(sigma)

    def mean_sqrtcov(self):
        return np.array([self.mu]), np.array([[self.sigma]])


class Uniform:
    """Uniform(a, b): a scalar measurement drawn uniformly on [a, b] (Distributions.jl; test/testMixturePrior.jl:30,
    test/testPackingMixtures.jl:20).  In a measurement component: mean slot = a, L[0][0] = b - a, family code 1."""
    family = abi.DIST_UNIFORM

    def __init__(self, a=0.0, b=1.0):
        self.a, self.b = float(a), float(b)
        if not self.b > self.a:
            raise ValueError("Uniform(a, b) needs a < b")

    def mean_sqrtcov(self):
        return np.array([self.a]), np.array([[self.b - self.a]])


class Rayleigh:
    """Rayleigh(sigma): z = sigma * sqrt(-2 log u) (test/testCompareVariablesFactors.jl:106 `LinearRelative(Rayleigh())`).
    In a measurement component: L[0][0] = sigma, family code 2."""
    family = abi.DIST_RAYLEIGH

    def __init__(self, sigma=1.0):
        self.sigma = float(sigma)
        if not self.sigma > 0:
            raise ValueError("Rayleigh(sigma) needs sigma > 0")

    def mean_sqrtcov(self):
        return np.array([0.0]), np.array([[self.sigma]])


class AliasingScalarSampler:
    """AliasingScalarSampler(domain, weights; SNRfloor = 0) (entities/AliasScalarSampling.jl:13-55): a scalar measurement
    that takes the value domain[i] with probability weights[i].  The constructor's conditioning of the weights is the
    reference's: negative weights to zero, normalise, subtract the SNRfloor quantile, clip, (keep the unfloored pmf when
    nothing is left), normalise.  On the device: a table in a slot (enum nbp_dist NBP_DIST_TABLE), family code 3."""
    family = abi.DIST_TABLE

    def __init__(self, domain, weights, SNRfloor=0.0):
        x, p = np.asarray(domain, dtype=float).ravel(), np.asarray(weights, dtype=float).ravel().copy()
        if x.size != p.size or x.size == 0:
            raise ValueError("AliasingScalarSampler: domain and weights must have the same, non-zero length")
        p[p < 0.0] = 0.0
        p /= p.sum()
        p2 = p - np.quantile(p, SNRfloor)
        p2[p2 < 0.0] = 0.0
        if p2.sum() > 1e-10:
            p = p2
        p = p / p.sum()
        if np.isnan(p).any():
            raise ValueError("AliasingScalarSampler got NaN because of particular values in p_x")
        self.domain, self.weights = x, p

    def mean_sqrtcov(self):
        return np.zeros(1), np.zeros((1, 1))

    def table_belief(self, N=None):
        """the table as the device holds it: a belief on Euclid(2), row 0 the domain, row 1 the cumulative weights.
        `N`: the particle count of the context that will hold it -- a slot takes at most N rows, and a longer table would be
        cut to its first N entries (the tail's mass would land on entry N - 1): refused, here and by nbp_clique_*"""
        if N is not None and self.domain.size > N:
            raise ValueError(f"AliasingScalarSampler: a table of {self.domain.size} entries does not fit a device slot of N = {N} rows "
                             "(libnbp holds the table in a belief slot; solve with N >= the table length)")
        cum = np.cumsum(self.weights)
        cum[-1] = 1.0
        return np.stack([self.domain, cum], axis=1), np.ones(2)


class MvNormal:
    """MvNormal(mu, Sigma).  Like Distributions.jl, a vector second argument is a vector of
    standard deviations (`MvNormal(mu, sigma::Vector)`), a matrix is the covariance."""

    def __init__(self, mu, cov):
        self.mu = np.atleast_1d(np.asarray(mu, dtype=float))
        cov = np.asarray(cov, dtype=float)
        if cov.ndim == 0:
            cov = np.eye(self.mu.size) * float(cov) ** 2
        elif cov.ndim == 1:
            cov = np.diag(cov ** 2)
        self.cov = cov

    def mean_sqrtcov(self):
        if getattr(self, "_chol", None) is None:
            self._chol = np.linalg.cholesky(self.cov)
        return self.mu, self._chol


# ------------------------------------------------------------------------------------------------
# variable types
# ------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class VariableType:
    name: str
    manifold: int

    @property
    def dim(self):
        return abi.MANIFOLD_DIM[self.manifold]

    @property
    def P(self):
        return abi.MANIFOLD_P[self.manifold]

    def identity(self):
        """getPointIdentity(varType) -- ManifoldsExtentions.jl:102-155"""
        if self.manifold == abi.SE2:
            return np.array([0.0, 0.0, 1.0, 0.0, 0.0, 1.0])
        return np.zeros(self.P)


ContinuousScalar = VariableType("ContinuousScalar", abi.EUCLID1)
Circular = VariableType("Circular", abi.CIRCULAR)
SpecialEuclidean2 = VariableType("SpecialEuclidean2", abi.SE2)


def ContinuousEuclid(n):
    return VariableType(f"ContinuousEuclid{{{n}}}", {1: abi.EUCLID1, 2: abi.EUCLID2, 3: abi.EUCLID3}[n])


# ------------------------------------------------------------------------------------------------
# factors
# ------------------------------------------------------------------------------------------------
class _Factor:
    kind = 0
    is_prior = False
    zdim = None  # None = dimension of the variable

    def components(self):
        """list of (weight, mean, sqrtcov, family) measurement components; family: abi.DIST_* (scalar measurements)"""
        mu, L = self.Z.mean_sqrtcov()
        return [(1.0, mu, L, getattr(self.Z, "family", abi.DIST_GAUSSIAN))]

    # a factor whose measurement model (or one component of it) is an AliasingScalarSampler keeps the sampler's table in a
    # device slot of its own, planned and written like the density of a PartialPriorPassThrough: `slot`, `density_belief()`
    # and `density_manifold` are what the slot planners (solver._plan_densities, native_host.place_densities) read
    slot = None
    density_manifold = abi.EUCLID2

    @property
    def table(self):
        zs = getattr(self, "comps", None) or [getattr(self, "Z", None)]
        tb = [z for z in zs if isinstance(z, AliasingScalarSampler)]
        if len(tb) > 1:
            raise ValueError("one AliasingScalarSampler per factor")
        return tb[0] if tb else None

    def density_belief(self, N=None):
        return self.table.table_belief(N)


class Prior(_Factor):
    """Prior(Z): r = z - x   (Factors/DefaultPrior.jl:17)"""
    kind, is_prior = abi.F_PRIOR, True

    def __init__(self, Z):
        self.Z = Z


class PriorCircular(Prior):
    """PriorCircular(Z)   (Factors/Circular.jl:56-74)"""


class ManifoldPrior(_Factor):
    """ManifoldPrior(M, p, Z): point = retract(p, hat(rand(Z)))   (Factors/GenericFunctions.jl:181-214).
    `p` is given in tangent coordinates at the identity ((x, y, theta) for SE(2))."""
    kind, is_prior = abi.F_PRIOR, True

    def __init__(self, p, Z):
        self.p, self.Z = np.atleast_1d(np.asarray(p, dtype=float)), Z

    def components(self):
        mu, L = self.Z.mean_sqrtcov()
        return [(1.0, self.p + mu, L)]


class LinearRelative(_Factor):
    """LinearRelative(Z): r = z - (x2 - x1)   (Factors/LinearRelative.jl:42-49)"""
    kind = abi.F_LINREL

    def __init__(self, Z):
        self.Z = Z


def _partial_mask(partial, dim):
    """`.partial` tuple of 1-based coordinate indices (the reference's convention) -> bit mask"""
    partial = tuple(int(p) for p in partial)
    if not partial or sorted(set(partial)) != list(partial) or partial[0] < 1 or partial[-1] > dim:
        raise ValueError(f"partial must be increasing 1-based coordinate indices within 1..{dim}: {partial}")
    m = 0
    for p in partial:
        m |= 1 << (p - 1)
    return m


class PartialPrior(Prior):
    """PartialPrior(varType, Z, partial): a prior on the coordinates `partial` (1-based) only
    (Factors/PartialPrior.jl; evaluation at EvalFactor.jl:457-538).  dim(Z) == len(partial)."""

    def __init__(self, varType, Z, partial):
        self.Z, self.partial = Z, tuple(partial)
        self.partial_mask = _partial_mask(partial, varType.dim)
        if len(Z.mean_sqrtcov()[0]) != len(self.partial):
            raise ValueError("PartialPrior: dim(Z) must equal len(partial)")


class PartialLinearRelative(LinearRelative):
    """r = z - (x2[k] - x1[k]) on the coordinates k in `partial` (one or two of them): the `DevelopPartialPairwise`
    factor of test/testpartialconstraint.jl:31-45 and its two-coordinate sibling (`.partial` relative factors solve, and
    inflate, only their partial coordinates -- with BFGS whatever their number: EvalFactor.jl:184-198,
    NumericalCalculations.jl:108,424).  Z has len(partial) dimensions."""

    def __init__(self, varType, Z, partial=(2,)):
        self.Z, self.partial = Z, tuple(partial)
        if not 1 <= len(self.partial) <= 2 or len(self.partial) >= varType.dim:
            raise ValueError("PartialLinearRelative: one or two partial coordinates, fewer than the variable has")
        self.partial_mask = _partial_mask(partial, varType.dim)


class CircularCircular(_Factor):
    """CircularCircular(Z)   (Factors/Circular.jl:24-28)"""
    kind, zdim = abi.F_CIRCULAR, 1

    def __init__(self, Z):
        self.Z = Z


class ManifoldFactor(_Factor):
    """ManifoldFactor(SpecialEuclidean(2), Z), Z on the Lie algebra (dx, dy, dtheta)
    (Factors/GenericFunctions.jl:39-44, 98-100)"""
    kind, zdim = abi.F_SE2, 3

    def __init__(self, Z):
        self.Z = Z


class PartialManifoldFactor(ManifoldFactor):
    """A relative factor on SE(2) that constrains only the residual components in `partial` (1-based: 1, 2 = the
    translation in the frame of the first pose, 3 = the heading): the reference's `.partial` mechanism applied to
    ManifoldFactor's residual -- the residual "must deal with the partial" itself and gets the full points, entropy goes
    on the partial coordinates only, the search is BFGS over the whole point (EvalFactor.jl:184-198,
    NumericalCalculations.jl:424-446).  Z stays on the full Lie algebra (dx, dy, dtheta); components outside `partial`
    are sampled and ignored."""

    def __init__(self, varType, Z, partial):
        self.Z, self.partial = Z, tuple(partial)
        if varType.manifold != abi.SE2 or not 1 <= len(self.partial) <= 2:
            raise ValueError("PartialManifoldFactor: an SE(2) variable and one or two partial components")
        self.partial_mask = _partial_mask(partial, varType.dim)


class EuclidDistance(_Factor):
    """EuclidDistance(Z): r = z - ||x2 - x1||   (Factors/EuclidDistance.jl:20)"""
    kind, zdim = abi.F_EUCLIDDIST, 1

    def __init__(self, Z):
        self.Z = Z


class Mixture(_Factor):
    """Mixture(mechanics, components, diversity): per-particle component label ~ Categorical
    (Factors/Mixture.jl:38-155).  `mechanics` is a factor class or instance."""

    def __init__(self, mechanics, components, diversity):
        self.mechanics = mechanics(components[0]) if isinstance(mechanics, type) else mechanics
        self.comps = list(components)
        self.diversity = np.asarray(diversity, dtype=float)
        if len(self.comps) != self.diversity.size or len(self.comps) > abi.MAXC:
            raise ValueError("Mixture: components/diversity mismatch or too many components")
        self.kind, self.is_prior, self.zdim = self.mechanics.kind, self.mechanics.is_prior, self.mechanics.zdim

    def components(self):
        out = []
        for w, z in zip(self.diversity / self.diversity.sum(), self.comps):
            mu, L = z.mean_sqrtcov()
            if isinstance(self.mechanics, ManifoldPrior):
                mu = self.mechanics.p + mu
            out.append((float(w), mu, L, getattr(z, "family", abi.DIST_GAUSSIAN)))
        return out


class PartialPriorPassThrough(_Factor):
    """PartialPriorPassThrough(Z, partial): a prior whose density is handed to inference as it is -- no sampling, no
    solve, no bandwidth fit (Factors/PartialPriorPassThrough.jl; calcProposalBelief dispatch, ApproxConv.jl:196-227).
    `points` (n x len(partial)) and `bw` (len(partial)) are that density, `Z.heatmap.densityFnc` of the reference's
    HeatmapGridDensity / LevelSetGridNormal; n need not equal the solver's N."""
    kind, is_prior = abi.F_PASSTHROUGH, True

    def __init__(self, varType, points, bw, partial=None):
        self.varType = varType
        self.partial = tuple(partial) if partial is not None else tuple(range(1, varType.dim + 1))
        self.partial_mask = _partial_mask(self.partial, varType.dim) if len(self.partial) < varType.dim else 0
        pts = np.atleast_2d(np.asarray(points, dtype=float))
        if pts.shape[1] != len(self.partial) or len(np.atleast_1d(bw)) != len(self.partial):
            raise ValueError("PartialPriorPassThrough: points / bw must have len(partial) columns")
        self.points, self.bw = pts, np.atleast_1d(np.asarray(bw, dtype=float))
        self.slot = None  # device slot of the density, set by whoever plans the slots

    def components(self):
        return [(1.0, np.zeros(1), np.zeros((1, 1)))]

    def density_belief(self, N=None):
        """(points n x P in the variable's point layout, bw D): the partial coordinates filled, the others zero
        (`N`, the context's particle count, matters to sampler tables only: a density with more points is cut to N)"""
        D, n = self.varType.dim, self.points.shape[0]
        c, b = np.zeros((n, D)), np.zeros(D)
        for i, k in enumerate(self.partial):
            c[:, k - 1], b[k - 1] = self.points[:, i], self.bw[i]
        if self.varType.manifold == abi.SE2:
            th = c[:, 2]
            c = np.stack([c[:, 0], c[:, 1], np.cos(th), np.sin(th), -np.sin(th), np.cos(th)], axis=1)
        return c, b


class MsgPrior(_Factor):
    """MsgPrior(belief): tree message as a prior (Factors/MsgPrior.jl:10-36,
    services/TreeMessageUtils.jl:86-89).  `slot` holds the TreeBelief (val + bw) on the device."""
    kind, is_prior = abi.F_MSGPRIOR, True

    def __init__(self, slot):
        self.slot = slot

    def components(self):
        return [(1.0, np.zeros(1), np.zeros((1, 1)))]


class DifferentialRelative(_Factor):
    """LinearRelative(::MKD) / CircularCircular(::MKD) / their SE(2) twin: a relative factor whose
    measurement is the kernel density estimate held in device slot `meas_slot` -- the differential factors of
    the useMsgLikelihoods upward messages (services/TreeMessageUtils.jl:279-335,
    Factors/LinearRelative.jl:32).  kind = abi.F_LINREL / F_CIRCULAR / F_SE2."""

    def __init__(self, kind, meas_slot):
        self.kind, self.meas_slot = kind, meas_slot
        self.zdim = {abi.F_CIRCULAR: 1, abi.F_SE2: 3}.get(kind)

    def components(self):
        return [(1.0, np.zeros(3), np.eye(3))]  # unused by the device when meas_kde is set


# ------------------------------------------------------------------------------------------------
# SolverParams (entities/SolverParams.jl:12-75) -- hot-path knobs only
# ------------------------------------------------------------------------------------------------
@dataclass
class SolverParams:
    N: int = 100
    spreadNH: float = 3.0
    inflation: float = 5.0
    nullSurplusAdd: float = 0.3
    inflateCycles: int = 3
    gibbsIters: int = 3
    alwaysFreshMeasurements: bool = True
    graphinit: bool = True
    upsolve: bool = True
    downsolve: bool = True
    limitfixeddown: bool = False  # skip marginalized frontals in the down solve (CliqStateMachineUtils.jl:499)
    productNiter: int = 1  # AMP.manifoldProduct(...; Niter=1), GraphProductOperations.jl:56
    useMsgLikelihoods: bool = False  # upward messages as joint likelihoods (SolverParams.jl:25, jointmsg.py)


@dataclass
class DFGVariable:
    label: str
    varType: VariableType
    initialized: bool = False
    val: np.ndarray = None  # N x P points (host copy of the posterior)
    bw: np.ndarray = None
    solvedCount: int = 0
    ismargin: bool = False


@dataclass
class DFGFactor:
    label: str
    variables: list
    fnc: _Factor
    multihypo: np.ndarray = None  # parsed Categorical p (certain -> 0.0) or None
    nullhypo: float = 0.0
    inflation: float = 5.0
    tags: set = field(default_factory=set)

    @property
    def isMultihypo(self):
        return self.multihypo is not None


def parseusermultihypo(multihypo, nullhypo):
    """services/FactorGraph.jl:634-655"""
    if multihypo is None or len(multihypo) == 0:
        return None, float(nullhypo)
    mh = np.asarray(multihypo, dtype=float).copy()
    mh[mh > 1 - 1e-10] = 0.0
    frac = mh.sum() % 1
    if not (abs(frac) < 1e-10 or 1 - 1e-10 < frac):
        raise ValueError("ensure multihypo sums to a (or nearly, 1e-10) integer, see #1086")
    if not np.isclose(mh[mh > 1e-10].sum(), 1.0):
        raise ValueError("fractional multihypo entries must sum to 1")
    mh /= mh.sum()
    return mh, float(nullhypo)


class FactorGraph:
    """In-memory graph (the LocalDFG role)."""

    def __init__(self, solverParams=None):
        self.solverParams = solverParams or SolverParams()
        self.variables = {}
        self.factors = {}
        self._adj = {}  # variable label -> [factor labels] in insertion order

    # -- DFG-style accessors -------------------------------------------------------------------
    def ls(self, var=None):
        return list(self.variables) if var is None else list(self._adj[var])

    def lsf(self):
        return list(self.factors)

    def listNeighbors(self, label):
        return list(self._adj[label]) if label in self.variables else list(self.factors[label].variables)

    def getVariable(self, label):
        return self.variables[label]

    def getFactor(self, label):
        return self.factors[label]

    def getVal(self, label):
        return self.variables[label].val

    def isInitialized(self, label):
        return self.variables[label].initialized


def isPartial(fct):
    """isPartial(fct) (test/testPartialFactors.jl:6-25): does the factor inform only some coordinates?"""
    return getattr(fct.fnc if hasattr(fct, "fnc") else fct, "partial_mask", 0) != 0


def deleteFactor(fg, label):
    """deleteFactor!(dfg, label)"""
    f = fg.factors.pop(label)
    for v in f.variables:
        fg._adj[v].remove(label)
    return f


def initfg(solverParams=None):
    return FactorGraph(solverParams)


def getSolverParams(fg):
    return fg.solverParams


def addVariable(fg, label, varType, N=None):
    """addVariable!(dfg, label, varType)   (services/FactorGraph.jl:587-632)"""
    if label in fg.variables:
        raise KeyError(f"variable {label} already exists")
    N = N or fg.solverParams.N
    v = DFGVariable(label, varType, False, np.tile(varType.identity(), (N, 1)), np.zeros(varType.dim))
    fg.variables[label] = v
    fg._adj[label] = []
    return v


def addFactor(fg, variables, fnc, multihypo=None, nullhypo=0.0, inflation=None, label=None, tags=()):
    """addFactor!(dfg, Xi, usrfnc; multihypo, nullhypo, inflation)   (services/FactorGraph.jl:824-875)"""
    variables = list(variables)
    for v in variables:
        if v not in fg.variables:
            raise KeyError(f"variable {v} not in graph")
    if multihypo is not None and len(multihypo) and len(multihypo) != len(variables):
        raise ValueError("When using multihypo=[...], the number of variables and multihypo probabilities must match.")
    if fnc.is_prior and len(variables) != 1:
        raise ValueError("priors are unary factors")
    if len(variables) > abi.MAXV:
        raise ValueError(f"at most {abi.MAXV} variables per factor")
    mh, nh = parseusermultihypo(multihypo, nullhypo)
    if label is None:
        base = "".join(variables) + "f"
        k = 1
        while f"{base}{k}" in fg.factors:
            k += 1
        label = f"{base}{k}"
    f = DFGFactor(label, variables, fnc, mh, nh,
                  fg.solverParams.inflation if inflation is None else float(inflation), set(tags))
    fg.factors[label] = f
    for v in variables:
        fg._adj[v].append(label)
    return f
