"""MI355X-native drop-in for IncrementalInference.jl's per-clique nonparametric
Chapman-Kolmogorov hot path (approxConv -> per-particle solve -> manifold KDE product ->
clique Gibbs schedule).  Compute lives in csrc/libnbp.so (hand-written HIP, gfx950) behind the
C ABI of include/nbp.h; this package is the host-side mirror of the reference's API for that
path.  There is no CPU fallback: the compute entry points raise when libnbp.so or a GPU is
missing."""
from . import abi, bayestree, canonical, seeds  # noqa: F401
from .backend import HipBackend, NbpError  # noqa: F401
from .bayestree import (buildTreeFromOrdering, buildTreeReset, getEliminationOrder,  # noqa: F401
                        nestedDissectionOrder)
from .canonical import (generateChainEuclid, generateCircularDoors, generateGraph_Kaess,  # noqa: F401
                        generateGraph_LineStep, generateMixtureChain, generateSE2Lattice)
from .factorgraph import (AliasingScalarSampler, Circular, CircularCircular, ContinuousEuclid, ContinuousScalar,  # noqa: F401
                          EuclidDistance, LinearRelative, ManifoldFactor, ManifoldPrior, Mixture,
                          MsgPrior, MvNormal, Normal, PartialLinearRelative, PartialManifoldFactor, PartialPrior, PartialPriorPassThrough, Prior, Rayleigh, Uniform, PriorCircular, SolverParams,
                          SpecialEuclidean2, addFactor, addVariable, deleteFactor, getSolverParams, initfg, isPartial)
from .solver import (TreeProgram, approxConv, approxConvBelief, approxConvBeliefPath, approxDeconv, findShortestPath,  # noqa: F401
                     product_desc, proposal_desc,
                     initAll, initVariable, localProduct, localProductAndUpdate, manikde, propagateBelief,
                     setValKDE, solveTree)
