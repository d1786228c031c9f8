#!/usr/bin/env python3
"""where the last comparisons that are not bit-identical come from (round 6): 1-D deconvolution searches, SE(2) products in big launches"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
from parity_utils import abi, iif, rand_points, relative_factor_desc, both
from oracle.oracle_backend import OracleBackend
import test_gpu_mixed_product_launches as M

ob = lambda n, s, side_ints=0: OracleBackend(n, s, side_ints, threads=16)
hb = lambda n, s, side_ints=0: iif.HipBackend(n, s, side_ints=side_ints)

# ---- deconv, LinearRelative on Euclid(1)
N = 200
kind, manifold, mean, sig = abi.F_LINREL, abi.EUCLID1, [1.0], [0.1]
rng = np.random.default_rng(4000 + 10 * kind + manifold)
a = rand_points(rng, manifold, N, center=0.0, spread=0.3)
b = rand_points(rng, manifold, N, center=1.0, spread=0.3)
d = relative_factor_desc(kind, manifold, 2, 1, [0, 1], 2, 777 + kind, mean, sig)
def setup(be):
    be.slot_write(0, manifold, a); be.slot_write(1, manifold, b)
o, h = both(ob, hb, N, 4, 0, setup, lambda be: be.run_deconv([d], [3]), lambda be: (be.slot_read(2, abi.EUCLID1)[0], be.slot_read(3, abi.EUCLID1)[0], be.diag(reset=True)))
bad = np.nonzero(o[0][:, 0] != h[0][:, 0])[0]
print("deconv linrel-1: differing", len(bad), "of", N, "diag oracle", o[2], "device", h[2])
for i in bad[:5]:
    print(f"  particle {i}: a {a[i,0]!r} b {b[i,0]!r} start {o[1][i,0]!r} (device start {h[1][i,0]!r}) oracle {o[0][i,0]!r} device {h[0][i,0]!r} root {b[i,0]-a[i,0]!r}")

# a third implementation of the same search (numpy float64 = IEEE, one rounding per operation)
def f(z, c): r = np.float64(z) - c; return r * r
def grad(z, c):
    hh = np.float64(6.0554544523933395e-06) * max(1.0, abs(z))
    return (f(z + hh, c) - f(z - hh, c)) / (2.0 * hh)
def bfgs(z, c):
    xc = np.float64(z); fx = f(xc, c); g = grad(xc, c); H = np.float64(1.0)
    for it in range(1000):
        if abs(g) <= 1e-8: break
        s = -H * g
        if s * g >= 0: H = np.float64(1.0); s = -g
        al = np.float64(1.0); dphi0 = g * s; ok = False
        for ls in range(50):
            xn = xc + al * s; fn = f(xn, c)
            if fn <= fx + 1e-4 * al * dphi0: ok = True; break
            aq = -dphi0 * al * al / (2.0 * (fn - fx - dphi0 * al))
            if not (aq >= 0.1 * al): aq = 0.1 * al
            if aq > 0.5 * al: aq = 0.5 * al
            al = aq
        if not ok: break
        gn = grad(xn, c); dx = xn - xc; dg = gn - g
        if dx == 0.0: break
        if dx * dg > 0: H = dx / dg
        xc, fx, g = xn, fn, gn
    return xc
for i in bad[:5]:
    c = np.float64(b[i, 0]) - np.float64(a[i, 0])
    print(f"  particle {i}: third implementation {bfgs(o[1][i,0], c)!r}")

# ---- SE(2) products in a launch that mixes density counts (big: node statistics in the global scratch)
for man, N, nprod, counts in ((abi.SE2, 300, 332, (2, 3, 4)), (abi.SE2, 200, 60, (2, 5, 9))):
    Fs = [counts[2] if i % 9 == 8 else (counts[1] if i % 17 == 3 else counts[0]) for i in range(nprod)]
    keep = [i for i in range(nprod) if Fs[i] != counts[0]][:6] + list(range(4))
    def run(make, keep_):
        nsrc = 16
        be = make(N, nsrc + nprod)
        rng = np.random.default_rng(1)
        for j in range(nsrc):
            be.slot_write(j, man, rand_points(rng, man, N, 0.2 * j, 0.3))
        be.run_bandwidth(list(range(nsrc)), [man] * nsrc)
        descs = [M.product_desc(man, [(3 * i + j) % nsrc for j in range(Fs[i])], nsrc + i, 5 + i) for i in range(nprod)]
        for dd in descs: dd.labels_out = -1
        if keep_ is not None: descs = [descs[i] for i in keep_]
        be.run_products(descs)
        out = {i: be.slot_read(nsrc + i, abi.EUCLID3) for i in (keep_ if keep_ is not None else range(nprod))}
        src = {j: be.slot_read(j, abi.EUCLID3) for j in range(nsrc)}
        be.close()
        return out, src
    dv, sd = run(hb, None)
    ov, so = run(ob, keep)
    print(f"SE(2) N={N} launch of {nprod}: source slots identical: {all(np.array_equal(sd[j][0], so[j][0]) and np.array_equal(sd[j][1], so[j][1]) for j in sd)}")
    for i in keep:
        ne = np.nonzero((dv[i][0] != ov[i][0]).any(axis=1))[0]
        if len(ne):
            for s in ne[:3]:
                print(f"  product {i} ({Fs[i]} densities) sample {s}: device {dv[i][0][s].tolist()!r} oracle {ov[i][0][s].tolist()!r}")
