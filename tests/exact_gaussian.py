"""Analytic known answer for a linear-Gaussian chain: the exact posterior (information form) that the
nonparametric solve approximates.  Shared by the oracle test and the GPU test."""
import numpy as np

from parity_utils import iif


def chain_with_end_priors(n, N=200, sigma=0.1):
    fg = iif.initfg(iif.SolverParams(N=N))
    for i in range(n):
        iif.addVariable(fg, f"x{i}", iif.ContinuousScalar)
    iif.addFactor(fg, ["x0"], iif.Prior(iif.Normal(0.0, sigma)))
    iif.addFactor(fg, [f"x{n - 1}"], iif.Prior(iif.Normal(n - 1.0, sigma)))
    for i in range(n - 1):
        iif.addFactor(fg, [f"x{i}", f"x{i + 1}"], iif.LinearRelative(iif.Normal(1.0, sigma)))
    w = 1.0 / sigma ** 2
    L, eta = np.zeros((n, n)), np.zeros(n)
    for i, m in ((0, 0.0), (n - 1, n - 1.0)):
        L[i, i] += w
        eta[i] += w * m
    for i in range(n - 1):
        L[i, i] += w
        L[i + 1, i + 1] += w
        L[i, i + 1] -= w
        L[i + 1, i] -= w
        eta[i] -= w
        eta[i + 1] += w
    S = np.linalg.inv(L)
    return fg, S @ eta, np.sqrt(np.diag(S))


def check_against_exact(backend, seed):
    n = 41
    fg, mu, sig = chain_with_end_priors(n)
    iif.solveTree(fg, eliminationOrder=iif.nestedDissectionOrder(fg), backend=backend, seed=seed)
    err = np.array([(fg.getVal(f"x{i}")[:, 0].mean() - mu[i]) / sig[i] for i in range(n)])
    rat = np.array([fg.getVal(f"x{i}")[:, 0].std() / sig[i] for i in range(n)])
    # means: Monte-Carlo error of a fraction of the exact sigma, no drift along the chain
    assert np.sqrt((err ** 2).mean()) < 0.45 and np.abs(err).max() < 1.2, (np.sqrt((err ** 2).mean()), np.abs(err).max())
    assert abs(err.mean()) < 0.35, err.mean()
    # widths: the reference's algorithm is mildly over-confident where information arrives through the tree and
    # wide where the down solve leans on pre-solve beliefs ("faithful behaviours" (iii))
    assert 0.6 < np.median(rat) < 1.2 and rat.min() > 0.35 and rat.max() < 3.5, (np.median(rat), rat.min(), rat.max())
    return err, rat
