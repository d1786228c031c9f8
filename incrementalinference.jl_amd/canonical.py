"""Synthetic graph generators: the reference's canonical test graphs plus the benchmark configs
of BASELINE.json / SURVEY.md 8(d).

  generateGraph_LineStep   src/services/CanonicalGraphExamples.jl:154-240
  generateGraph_Kaess      src/services/CanonicalGraphExamples.jl:15-36
  config 3 (Circular doors, multihypo), config 4 (SE(2) lattice), config 5 (Mixture chain):
  SURVEY.md 8(d) definitions, factor forms from test/testMultiHypo3Door.jl:57,
  test/testSpecialEuclidean2Mani.jl:41-57,176-190 and src/Factors/Mixture.jl:114-155.
"""
import numpy as np

from .factorgraph import (Circular, CircularCircular, ContinuousEuclid, ContinuousScalar, LinearRelative,
                          ManifoldFactor, ManifoldPrior, Mixture, MvNormal, Normal, Prior, PriorCircular,
                          SolverParams, SpecialEuclidean2, addFactor, addVariable, initfg)


def generateGraph_LineStep(lineLength, poseEvery=2, landmarkEvery=4, posePriorsAt=(0,), landmarkPriorsAt=(),
                           sightDistance=4, vardims=1, sigma_pose_prior=0.1, sigma_lm_prior=0.1,
                           sigma_pose_pose=0.1, sigma_pose_lm=0.1, solverParams=None):
    fg = initfg(solverParams or SolverParams())
    vtype = ContinuousScalar if vardims == 1 else ContinuousEuclid(vardims)

    def xNoise(i, s):
        return Normal(i, s) if vardims == 1 else MvNormal(np.full(vardims, float(i)), s)

    x, lm = [], []
    for i in range(lineLength + 1):
        if i % poseEvery == 0:
            x.append(i)
            addVariable(fg, f"x{i}", vtype)
            if i in posePriorsAt:
                addFactor(fg, [f"x{i}"], Prior(xNoise(i, sigma_pose_prior)))
            if i > 0:
                addFactor(fg, [f"x{i - poseEvery}", f"x{i}"], LinearRelative(xNoise(poseEvery, sigma_pose_pose)))
        if landmarkEvery != 0 and i % landmarkEvery == 0:
            lm.append(i)
            addVariable(fg, f"lm{i}", vtype)
            if i in landmarkPriorsAt:
                addFactor(fg, [f"lm{i}"], Prior(xNoise(i, sigma_lm_prior)))
    for xi in x:
        for lmi in lm:
            if abs(lmi - xi) < sightDistance:
                addFactor(fg, [f"x{xi}", f"lm{lmi}"], LinearRelative(xNoise(lmi - xi, sigma_pose_lm)))
    return fg


def generateGraph_Kaess(solverParams=None):
    """Kaess et al. Bayes tree example: x1..x3, l1, l2 (CanonicalGraphExamples.jl:15-36)"""
    fg = initfg(solverParams or SolverParams())
    for v in ["x1", "x2", "x3"]:
        addVariable(fg, v, ContinuousScalar)
    addFactor(fg, ["x1"], Prior(Normal(0, 1)))
    addFactor(fg, ["x1", "x2"], LinearRelative(Normal(0, 1)))
    addFactor(fg, ["x2", "x3"], LinearRelative(Normal(0, 1)))
    for v in ["l1", "l2"]:
        addVariable(fg, v, ContinuousScalar)
    addFactor(fg, ["x1", "l1"], LinearRelative(Normal(0, 1)))
    addFactor(fg, ["x2", "l1"], LinearRelative(Normal(0, 1)))
    addFactor(fg, ["x3", "l2"], LinearRelative(Normal(0, 1)))
    return fg


def generateChainEuclid(nvars, vardims=2, priorEvery=100, N=200, sigma=0.1):
    """BASELINE config 2 / 2': generateGraph_LineStep(nvars-1; poseEvery=1, landmarkEvery=0,
    vardims, posePriorsAt=0:priorEvery:...), truth x_i = (i, ..., i)."""
    sp = SolverParams(N=N)
    return generateGraph_LineStep(nvars - 1, poseEvery=1, landmarkEvery=0, posePriorsAt=tuple(range(0, nvars, priorEvery)),
                                  vardims=vardims, sigma_pose_prior=sigma, sigma_pose_pose=sigma, solverParams=sp)


def generateCircularDoors(nposes=2000, N=200, sightEvery=25):
    """BASELINE config 3: Circular poses stepping 2pi/50, four door landmarks, multihypo sightings."""
    fg = initfg(SolverParams(N=N))
    doors = [-2.4, -0.8, 0.8, 2.4]
    for k, th in enumerate(doors):
        addVariable(fg, f"l{k}", Circular)
        addFactor(fg, [f"l{k}"], PriorCircular(Normal(th, 0.01)))
    step = 2 * np.pi / 50
    for i in range(nposes):
        addVariable(fg, f"x{i}", Circular)
        if i == 0:
            addFactor(fg, ["x0"], PriorCircular(Normal(0.0, 0.1)))
        else:
            addFactor(fg, [f"x{i - 1}", f"x{i}"], CircularCircular(Normal(step, 0.05)))
        if i % sightEvery == 0:
            # door sighting with unknown association: the measurement is the bearing difference to the
            # nearest door (so exactly one of the four hypotheses is consistent with the truth)
            xi = (i * step + np.pi) % (2 * np.pi) - np.pi
            dz = min(((d - xi + np.pi) % (2 * np.pi) - np.pi for d in doors), key=abs)
            addFactor(fg, [f"x{i}", "l0", "l1", "l2", "l3"], CircularCircular(Normal(dz, 0.1)),
                      multihypo=[1.0, 0.25, 0.25, 0.25, 0.25])
    return fg


def generateSE2Lattice(rows=50, cols=100, N=200, closeEvery=5):
    """BASELINE config 4: boustrophedon SE(2) lattice, 1 m spacing, loop closures between
    vertically adjacent poses every `closeEvery`-th column."""
    fg = initfg(SolverParams(N=N))
    odo = [0.1, 0.1, 0.01]
    idx = {}
    k = 0
    for r in range(rows):
        cs = range(cols) if r % 2 == 0 else range(cols - 1, -1, -1)
        for c in cs:
            idx[(r, c)] = k
            k += 1
    pose = {}
    for (r, c), i in idx.items():
        heading = 0.0 if r % 2 == 0 else np.pi
        pose[i] = np.array([float(c), float(r), heading])
    n = rows * cols
    for i in range(n):
        addVariable(fg, f"x{i}", SpecialEuclidean2)
    addFactor(fg, ["x0"], ManifoldPrior(pose[0], MvNormal(np.zeros(3), [0.01, 0.01, 0.01])))

    def rel(a, b):
        pa, pb = pose[a], pose[b]
        ca, sa = np.cos(pa[2]), np.sin(pa[2])
        d = pb[:2] - pa[:2]
        th = (pb[2] - pa[2] + np.pi) % (2 * np.pi) - np.pi
        return np.array([ca * d[0] + sa * d[1], -sa * d[0] + ca * d[1], th])

    for i in range(1, n):
        addFactor(fg, [f"x{i - 1}", f"x{i}"], ManifoldFactor(MvNormal(rel(i - 1, i), odo)))
    for r in range(1, rows):
        for c in range(0, cols, closeEvery):
            a, b = idx[(r - 1, c)], idx[(r, c)]
            if abs(a - b) > 1:
                addFactor(fg, [f"x{a}", f"x{b}"], ManifoldFactor(MvNormal(rel(a, b), odo)))
    return fg


def generateMixtureChain(nvars=10000, N=300, priorEvery=500):
    """BASELINE config 5: Euclid(3) chain with Mixture(LinearRelative, (tight, loose), [0.8, 0.2])."""
    fg = initfg(SolverParams(N=N))
    vt = ContinuousEuclid(3)
    mu = np.array([1.0, 0.0, 0.0])
    for i in range(nvars):
        addVariable(fg, f"x{i}", vt)
        if i % priorEvery == 0:
            addFactor(fg, [f"x{i}"], Prior(MvNormal(np.array([float(i), 0.0, 0.0]), 0.1)))
        if i > 0:
            addFactor(fg, [f"x{i - 1}", f"x{i}"],
                      Mixture(LinearRelative, (MvNormal(mu, 0.1), MvNormal(mu, 1.0)), [0.8, 0.2]))
    return fg
