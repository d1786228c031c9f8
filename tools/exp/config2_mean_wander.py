#!/usr/bin/env python3
"""config 2 (1000-variable Euclid(2) chain, priors every 100): how far the NBP posterior MEAN of a pose sits from the truth, in
units of the pose's exact posterior sigma, over solve seeds -- the record behind bench_support.tol_chain and BASELINE.md's
amendment "Config 2, mean band".   usage (GPU box): python tools/exp/config2_mean_wander.py [seeds=20]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import numpy as np
import iif_amd_loader
iif = iif_amd_loader.load()
from bench_support import RankSolve, workloads, chain_exact_sigma
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
wl = workloads(iif)["2"]
rs = RankSolve(iif, wl, 1000, 200, 0, 1, 0, None)
rs.prepare()
sig = chain_exact_sigma(1000)
poses = [v for v in rs.mine if v.startswith("x")]
ratios, band, stds = [], [], []
for k in range(seeds):
    rs.step(k)
    rs.be.synchronize()
    for v in poses:
        i = int(v[1:])
        pts, _ = rs.be.slot_read(rs.main[v], iif.abi.EUCLID2)
        err = np.abs(pts.mean(axis=0) - i)
        ratios.append(err.max() / sig[i])
        band.append(err.max() <= 3 * sig[i] / np.sqrt(200) + 0.01)
        stds.append(pts.std(axis=0).mean() / sig[i])
r = np.array(ratios).reshape(seeds, len(poses))
print(f"config 2, 1000 variables, N = 200, {seeds} solve seeds x {len(poses)} poses (all of them):")
print(f"  |mean - truth| / sigma_post: median {np.median(r):.3f}, 95 % {np.quantile(r, 0.95):.3f}, 99.9 % {np.quantile(r, 0.999):.3f}, max {r.max():.3f}")
print(f"  |mean - truth| (absolute): max {max(ra * sig[int(v[1:])] for row in r for ra, v in zip(row, poses)):.3f}")
print(f"  inside BASELINE.md 5's band 3 sigma_post / sqrt(N) + 0.1 x 0.1: {np.mean(band):.3f} of the (seed, pose) pairs")
print(f"  sample std / sigma_post: median {np.median(stds):.3f}, 5 % {np.quantile(stds, 0.05):.3f}, 95 % {np.quantile(stds, 0.95):.3f}; inside [0.5, 2]: {np.mean((np.array(stds) >= 0.5) & (np.array(stds) <= 2)):.3f}")
print(f"  gate of bench_support.tol_chain (0.1 + 2 sigma_post): worst (err - 0.1) / sigma_post = {max((ra * sig[int(v[1:])] - 0.1) / sig[int(v[1:])] for row in r for ra, v in zip(row, poses)):.3f} (< 2 passes)")
rs.close()
