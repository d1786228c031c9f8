"""Shared helpers for the oracle-vs-HIP parity tests: build the same slots and descriptors on both
backends, run, compare.  Identical RNG streams -> identical results, floating point included (round 6): RTOL = 0; BASELINE.md 5
asked for 1e-9 relative per particle."""
import os

import numpy as np

import iif_amd_loader

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

iif = iif_amd_loader.load()
abi = iif.abi

RTOL = 0.0


def rand_points(rng, manifold, N, center=0.0, spread=1.0):
    D = abi.MANIFOLD_DIM[manifold]
    c = rng.normal(size=(N, D)) * spread + center
    if manifold == abi.SE2:
        th = c[:, 2]
        return np.stack([c[:, 0], c[:, 1], np.cos(th), np.sin(th), -np.sin(th), np.cos(th)], axis=1)
    if manifold == abi.CIRCULAR:
        return (c + np.pi) % (2 * np.pi) - np.pi
    return c


def coords(manifold, pts):
    """host points -> tangent coordinates (for comparisons with wrap-aware differences)"""
    if manifold == abi.SE2:
        return np.stack([pts[:, 0], pts[:, 1], np.arctan2(pts[:, 3], pts[:, 2])], axis=1)
    return pts


def coord_diff(manifold, a, b):
    d = coords(manifold, a) - coords(manifold, b)
    if manifold == abi.CIRCULAR:
        d = (d + np.pi) % (2 * np.pi) - np.pi
    if manifold == abi.SE2:
        d[:, 2] = (d[:, 2] + np.pi) % (2 * np.pi) - np.pi
    return d


def assert_points_close(manifold, a, b, rtol=RTOL, max_bad=0, what=""):
    d = np.abs(coord_diff(manifold, a, b))
    scale = np.maximum(1.0, np.abs(coords(manifold, a)))
    bad = (d > rtol * scale).any(axis=1)
    assert bad.sum() <= max_bad, f"{what}: {bad.sum()} particles differ, max abs diff {d.max():.3e}"


def relative_factor_desc(kind, manifold, nvars, sfidx, var_slots, out_slot, seed, mean, sig, *, multihypo=None,
                         nullhypo=0.0, ncomp=1, comps=None, mhidx_in=-1, mhidx_out=-1, cycles=3, inflation=5.0, partial_mask=0,
                         inflate_cycles=None):
    d = abi.ProposalDesc()
    d.partial_mask = partial_mask
    if inflate_cycles is not None:
        cycles = inflate_cycles
    d.factor_kind, d.manifold, d.nvars, d.sfidx = kind, manifold, nvars, sfidx
    for i, s in enumerate(var_slots):
        d.var_slot[i] = s
    d.out_slot, d.ncomp, d.inflate_cycles = out_slot, ncomp, cycles
    d.mhidx_in, d.mhidx_out = mhidx_in, mhidx_out
    d.inflation, d.spread_nh, d.nullhypo = inflation, 3.0, nullhypo
    if comps is None:
        comps = [(1.0, mean, sig)]
    for c, (w, mu, sg) in enumerate(comps):
        d.comp[c][0] = w
        for i, m in enumerate(mu):
            d.comp[c][1 + i] = m
        for i, s in enumerate(sg):
            d.comp[c][4 + 3 * i + i] = s
    if multihypo is not None:
        d.has_multihypo = 1
        for i, p in enumerate(multihypo):
            d.multihypo[i] = p
    d.seed = seed
    return d


def product_desc(manifold, in_slots, out_slot, seed, labels_out=-1, niter=1):
    return iif.solver.product_desc(manifold, in_slots, out_slot, seed, niter, labels_out)


def both(oracle_factory, hip_factory, N, n_slots, side_ints, setup, run, read):
    """run the same closure sequence on both backends and return both results"""
    out = []
    for fac in (oracle_factory, hip_factory):
        be = fac(N, n_slots, side_ints)
        be.diag(reset=True)  # the oracle's counters are process-wide: start every comparison from zero
        setup(be)
        run(be)
        out.append(read(be))
        be.close()
    return out


def record_parity(line):
    """append one line to gpurun_out/r04_whole_solve_parity.txt (the -m gpu suite's whole-solve figures: shares of
    particle-identical variables, KL medians, mode shares; copied to profiles/ after a GPU run)"""
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "r04_whole_solve_parity.txt"), "a") as f:
        f.write(line.rstrip("\n") + "\n")
