"""Definition of the golden cases: seeded inputs + descriptor recipes, runnable on any backend."""
import numpy as np

from parity_utils import abi, product_desc, rand_points, relative_factor_desc

# name -> dict(kind, manifold, N, ...)
CASES = {
    "prior_euclid2_n100": dict(op="prior", manifold=abi.EUCLID2, N=100, seed=101, mean=[1.0, -2.0], sig=[0.1, 0.3], nullhypo=0.0),
    "prior_circular_nullhypo_n200": dict(op="prior", manifold=abi.CIRCULAR, N=200, seed=102, mean=[3.0], sig=[0.2], nullhypo=0.2),
    "prior_se2_n200": dict(op="prior", manifold=abi.SE2, N=200, seed=103, mean=[10.0, 10.0, 3.0], sig=[0.1, 0.1, 0.01], nullhypo=0.0),
    "linrel_euclid1_n100": dict(op="rel", kind=abi.F_LINREL, manifold=abi.EUCLID1, N=100, seed=111, mean=[1.0], sig=[0.1], sfidx=1),
    "linrel_euclid2_n200": dict(op="rel", kind=abi.F_LINREL, manifold=abi.EUCLID2, N=200, seed=112, mean=[1.0, 1.0], sig=[0.1, 0.1], sfidx=1),
    "linrel_euclid3_rev_n200": dict(op="rel", kind=abi.F_LINREL, manifold=abi.EUCLID3, N=200, seed=113, mean=[1.0, 0.0, 0.5], sig=[0.1, 0.1, 0.2], sfidx=0),
    "circular_n200": dict(op="rel", kind=abi.F_CIRCULAR, manifold=abi.CIRCULAR, N=200, seed=114, mean=[0.1256], sig=[0.05], sfidx=1),
    "se2_n200": dict(op="rel", kind=abi.F_SE2, manifold=abi.SE2, N=200, seed=115, mean=[1.0, 0.0, 0.0], sig=[0.1, 0.1, 0.01], sfidx=1),
    "se2_rev_n100": dict(op="rel", kind=abi.F_SE2, manifold=abi.SE2, N=100, seed=116, mean=[1.0, 2.0, 0.785], sig=[0.01, 0.01, 0.01], sfidx=0),
    "mixture_euclid3_n300": dict(op="mix", manifold=abi.EUCLID3, N=300, seed=117),
    "multihypo_doors_n200": dict(op="mh", manifold=abi.CIRCULAR, N=200, seed=118, sfidx=0),
    "multihypo_landmark_n200": dict(op="mh", manifold=abi.CIRCULAR, N=200, seed=119, sfidx=2),
    "product_euclid2_f3_n200": dict(op="prod", manifold=abi.EUCLID2, N=200, seed=121, F=3),
    "product_circular_f2_n200": dict(op="prod", manifold=abi.CIRCULAR, N=200, seed=122, F=2),
    "product_se2_f3_n100": dict(op="prod", manifold=abi.SE2, N=100, seed=123, F=3),
    "partial_prior_euclid3_n200": dict(op="pprior", manifold=abi.EUCLID3, N=200, seed=131, mask=5, mean=[2.0, -1.0], sig=[1.0, 0.2], nullhypo=0.1),
    "partial_linrel_euclid2_n100": dict(op="prel", manifold=abi.EUCLID2, N=100, seed=132, mask=2, sfidx=1),
    "partial_product_se2_n200": dict(op="pprod", manifold=abi.SE2, N=200, seed=133, masks=[4, 3, 0]),
    "partial_product_euclid3_uninformed_n100": dict(op="pprod", manifold=abi.EUCLID3, N=100, seed=134, masks=[1, 4]),
}


def run_case(c, make_backend):
    man, N = c["manifold"], c["N"]
    rng = np.random.default_rng(c["seed"])
    D = abi.MANIFOLD_DIM[man]
    if c["op"] == "prior":
        be = make_backend(N, 2, N)
        cur = rand_points(rng, man, N, 0.5, 0.4)
        be.slot_write(0, man, cur)
        d = relative_factor_desc(abi.F_PRIOR, man, 1, 0, [0], 1, c["seed"], c["mean"], c["sig"], nullhypo=c["nullhypo"], mhidx_out=0)
        be.run_proposals([d])
        pts, bw = be.slot_read(1, man)
        out = dict(in0=cur, pts=pts, bw=bw, mhidx=be.side_read(0, N))
    elif c["op"] == "pprior":
        be = make_backend(N, 2, N)
        cur = rand_points(rng, man, N, 0.5, 0.4)
        be.slot_write(0, man, cur)
        d = relative_factor_desc(abi.F_PRIOR, man, 1, 0, [0], 1, c["seed"], c["mean"], c["sig"], nullhypo=c["nullhypo"], mhidx_out=0)
        d.partial_mask = c["mask"]
        be.run_proposals([d])
        pts, bw = be.slot_read(1, man)
        out = dict(in0=cur, pts=pts, bw=bw, mhidx=be.side_read(0, N))
    elif c["op"] == "prel":
        be = make_backend(N, 3, N)
        a, b = rand_points(rng, man, N, 0.0, 0.3), rand_points(rng, man, N, 1.0, 0.3)
        be.slot_write(0, man, a)
        be.slot_write(1, man, b)
        d = relative_factor_desc(abi.F_LINREL, man, 2, c["sfidx"], [0, 1], 2, c["seed"], [10.0], [1.0], mhidx_out=0)
        d.partial_mask = c["mask"]
        be.run_proposals([d])
        pts, bw = be.slot_read(2, man)
        out = dict(in0=a, in1=b, pts=pts, bw=bw, mhidx=be.side_read(0, N))
    elif c["op"] == "pprod":
        from parity_utils import iif
        masks = c["masks"]
        F = len(masks)
        be = make_backend(N, F + 2, N * F)
        ins = [rand_points(rng, man, N, 0.2 * j, 0.5) for j in range(F)]
        old = rand_points(rng, man, N, 4.0, 0.2)
        for j in range(F):
            be.slot_write(j, man, ins[j], np.full(D, 0.15 + 0.03 * j))
        be.slot_write(F, man, old)
        be.run_products([iif.solver.product_desc(man, list(range(F)), F + 1, c["seed"], 1, 0, partials=masks, old_slot=F)])
        pts, bw = be.slot_read(F + 1, man)
        out = dict(pts=pts, bw=bw, labels=be.side_read(0, N * F), old=old, **{f"in{i}": p for i, p in enumerate(ins)})
    elif c["op"] in ("rel", "mix"):
        be = make_backend(N, 3, N)
        a, b = rand_points(rng, man, N, 0.0, 0.3), rand_points(rng, man, N, 1.0, 0.3)
        be.slot_write(0, man, a)
        be.slot_write(1, man, b)
        if c["op"] == "rel":
            d = relative_factor_desc(c["kind"], man, 2, c["sfidx"], [0, 1], 2, c["seed"], c["mean"], c["sig"], mhidx_out=0)
        else:
            comps = [(0.8, [1, 0, 0], [0.1, 0.1, 0.1]), (0.2, [1, 0, 0], [1.0, 1.0, 1.0])]
            d = relative_factor_desc(abi.F_LINREL, man, 2, 1, [0, 1], 2, c["seed"], None, None, ncomp=2, comps=comps, nullhypo=0.1, mhidx_out=0)
        be.run_proposals([d])
        pts, bw = be.slot_read(2, man)
        out = dict(in0=a, in1=b, pts=pts, bw=bw, mhidx=be.side_read(0, N))
    elif c["op"] == "mh":
        be = make_backend(N, 6, N)
        doors = [-2.4, -0.8, 0.8, 2.4]
        ins = [rand_points(rng, man, N, 0.75, 0.2)] + [rand_points(rng, man, N, t, 0.01) for t in doors]
        for i, p in enumerate(ins):
            be.slot_write(i, man, p)
        d = relative_factor_desc(abi.F_CIRCULAR, man, 5, c["sfidx"], [0, 1, 2, 3, 4], 5, c["seed"], [0.0], [0.1],
                                 multihypo=[0.0, 0.25, 0.25, 0.25, 0.25], mhidx_out=0)
        be.run_proposals([d])
        pts, bw = be.slot_read(5, man)
        out = dict(pts=pts, bw=bw, mhidx=be.side_read(0, N), **{f"in{i}": p for i, p in enumerate(ins)})
    else:
        F = c["F"]
        be = make_backend(N, F + 1, N * F)
        ins = [rand_points(rng, man, N, 0.2 * j, 0.5) for j in range(F)]
        bws = [np.full(D, 0.15 + 0.03 * j) for j in range(F)]
        for j in range(F):
            be.slot_write(j, man, ins[j], bws[j])
        be.run_products([product_desc(man, list(range(F)), F, c["seed"], labels_out=0)])
        pts, bw = be.slot_read(F, man)
        out = dict(pts=pts, bw=bw, labels=be.side_read(0, N * F), **{f"in{i}": p for i, p in enumerate(ins)})
    be.close()
    return {k: np.asarray(v) for k, v in out.items()}
