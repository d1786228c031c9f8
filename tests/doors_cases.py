"""Config 3 (Circular poses, four door landmarks, multihypo sightings) -- where the mass goes.  Run on the oracle
(tests/test_doors_mode_share.py) and on the GPU (tests/test_gpu_doors_mode_share.py).

Full size (2000 poses) ends with a median of ~0.25 of the particles at the true pose.  That is not a defect of either
implementation: the four doors are (almost) equally spaced, so a trajectory shifted by one door spacing explains every
sighting equally well, and the only thing that tells the aliases apart is the prior on x0 carried along by odometry, whose
uncertainty grows like 0.05 sqrt(i) rad -- beyond a few hundred poses the exact posterior itself has four comparable
modes.  What CAN be pinned: where odometry is still informative the sighting is resolved correctly."""
import numpy as np

from parity_utils import iif


def wrapdiff(a, b):
    return (a - b + np.pi) % (2 * np.pi) - np.pi


STEP = 2 * np.pi / 50


def share(fg, i):
    return float((np.abs(wrapdiff(fg.getVal(f"x{i}")[:, 0], i * STEP)) < 0.35).mean())


def case_sighting_is_resolved_by_odometry(backend):
    """at a sighting pose the multihypo proposal carries ~25 % per door (exact-match semantics of the recipe: one door
    per particle), its relative siblings carry nullSurplusAdd = 0.3 of spread particles (ApproxConv.jl:255-265), and the
    product puts > 90 % of the mass at the true pose"""
    fg = iif.generateCircularDoors(nposes=6, N=200, sightEvery=3)
    iif.initAll(fg, backend=backend, seed=3)
    (pts, bw), ipc, props = iif.propagateBelief(fg, "x3", backend=backend, seed=5, return_proposals=True)
    at = lambda p: float((np.abs(wrapdiff(p[:, 0], 3 * STEP)) < 0.35).mean())
    by = dict(zip(fg.ls("x3"), [at(p) for p, _ in props]))
    mh = [f for f in by if "l0" in f][0]
    assert 0.15 < by[mh] < 0.4, by            # ~1/4 of the particles per door hypothesis
    for f, v in by.items():
        if f != mh:
            # 1 - nullSurplusAdd of the sibling's particles are solved; the rest keep x3's current value plus spreadNH x the
            # belief's own spread of entropy (EvalFactor.jl:222-231) -- +-1.5 sigma of an initialised belief of sigma ~ 0.1-0.2,
            # i.e. mostly still inside the 0.35 rad window: only the lower bound is informative (0.73 if they were
            # scattered over the whole circle, less if the solved ones went astray)
            assert 0.55 < v <= 1.0, by
    assert at(pts) > 0.9, at(pts)
    assert list(ipc) == [3.0]
    return by, at(pts)


def case_short_chains_keep_the_true_mode(backend):
    out = {}
    for nposes, se in ((6, 3), (26, 25)):
        fg = iif.generateCircularDoors(nposes=nposes, N=200, sightEvery=se)
        iif.solveTree(fg, eliminationOrder=iif.nestedDissectionOrder(fg), backend=backend, seed=1)
        s = [share(fg, i) for i in range(nposes)]
        out[nposes] = (min(s), float(np.median(s)))
        assert min(s) > 0.8, (nposes, min(s))
    return out


def mechanisms(backend, nposes, variants, solves=1, seed=1):
    """Config 3's graph solved `solves` times in a row under each (nullSurplusAdd, Niter) of `variants`; per variant and solve:
    the share of particles at the true pose, pose by pose.  What DESIGN.md 5 argues -- the true mode is lost to the reference's
    nullSurplusAdd = 0.3 (ApproxConv.jl:255-265) and to its under-mixed Niter = 1 product (GraphProductOperations.jl:53-60),
    not to a defect of the restatement -- as a measurement (review r04, item 3)."""
    out = {}
    for nsa, niter in variants:
        fg = iif.generateCircularDoors(nposes=nposes, N=200, sightEvery=25)
        fg.solverParams.nullSurplusAdd = nsa
        fg.solverParams.productNiter = niter
        order = iif.nestedDissectionOrder(fg)
        rows = []
        for k in range(solves):
            fg.solverParams.graphinit = (k == 0)  # a further solve continues from the posteriors of the last
            iif.solveTree(fg, eliminationOrder=order, backend=backend, seed=seed + k)
            rows.append(np.array([share(fg, i) for i in range(nposes)]))
        out[(nsa, niter)] = rows
    return out


def case_true_mode_survives_without_null_surplus_and_with_a_mixed_product(backend, nposes=400):
    """BASELINE.md 5's multimodal criterion (>= 0.6 x nominal at the true pose on >= 90 % of the poses) IS met by the restated
    algorithm once the two reference parameters that lose the mode are taken out: nullSurplusAdd = 0 and Niter = 6 (the
    sampler's stationary distribution is the exact product, tests/analytic_cases.py).  With the reference's own values
    (0.3, 1) the same graph, same seed, keeps the true mode on a minority of the poses."""
    r = mechanisms(backend, nposes, [(0.3, 1), (0.0, 6)])
    ref, fixed = r[(0.3, 1)][0], r[(0.0, 6)][0]
    assert (fixed >= 0.6).mean() >= 0.9 and np.median(fixed) >= 0.9, ((fixed >= 0.6).mean(), np.median(fixed), fixed.min())
    assert (ref >= 0.6).mean() < (fixed >= 0.6).mean() - 0.2, ((ref >= 0.6).mean(), (fixed >= 0.6).mean())
    return {"reference (0.3, 1)": (float((ref >= 0.6).mean()), float(np.median(ref))),
            "nullSurplusAdd 0, Niter 6": (float((fixed >= 0.6).mean()), float(np.median(fixed)), float(fixed.min()))}


def case_full_size_mechanisms(backend, nposes=2000):
    """the same at BASELINE's 2000 poses, three solves in a row (device only: seconds there, a quarter of an hour on the oracle).
    Measured on MI355X (profiles/r05_config3_mechanisms.txt): (0, 6) median 0.99 / min 0.81 after ONE solve; (0, 1) keeps the true
    mode for the first ~800 poses; (0.3, 6) recovers it with further solves (0.38 -> 0.77 -> 0.83 of the poses above 0.8); the
    reference's (0.3, 1) rises 0.06 -> 0.13 -> 0.23 and has its first 200 poses at >= 0.6 on 0.66 -> 0.75 -> 0.81."""
    r = mechanisms(backend, nposes, [(0.3, 1), (0.0, 1), (0.3, 6), (0.0, 6)], solves=3)
    above = lambda s: float((s > 0.8).mean())
    # (i) + (ii): both parameters out -> BASELINE 5's criterion holds everywhere after one solve, and stays
    for s in r[(0.0, 6)]:
        assert (s >= 0.6).mean() >= 0.9 and np.median(s) >= 0.9, ((s >= 0.6).mean(), np.median(s))
    # (i) alone: the true mode survives several hundred poses further than with the surplus
    s01, sref = r[(0.0, 1)][0], r[(0.3, 1)][0]
    assert np.median(s01[:600]) >= 0.9 and np.median(s01) > np.median(sref) + 0.3, (np.median(s01[:600]), np.median(s01), np.median(sref))
    # (ii) alone: a mixed product lets further solves repair what the surplus leaks
    a = [above(s) for s in r[(0.3, 6)]]
    assert a[0] < a[1] < a[2] and a[2] >= 0.7, a
    # (iii) the reference's own parameters: the share of poses above 0.8 rises with every solve; the stretch the x0 prior reaches
    b = [above(s) for s in r[(0.3, 1)]]
    assert b[0] < b[1] < b[2] and b[2] >= 0.15, b
    first = [float((s[:200] >= 0.6).mean()) for s in r[(0.3, 1)]]
    assert first[2] >= 0.7 and np.median(r[(0.3, 1)][2][:200]) >= 0.9, first
    return {f"nullSurplusAdd {k[0]}, Niter {k[1]}": [(round(float(np.median(s)), 3), round(float(s.min()), 3), round(above(s), 3), round(float((s[:200] >= 0.6).mean()), 3))
                                                    for s in v] for k, v in r.items()}
