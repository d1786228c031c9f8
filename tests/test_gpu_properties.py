"""-m gpu: size-independent properties at BASELINE sizes and edge sizes (no oracle needed):
determinism, launch-geometry independence, manifold validity, extreme particle counts."""
import numpy as np
import pytest

from parity_utils import (abi, assert_points_close, both, iif, product_desc, rand_points,
                          relative_factor_desc)

pytestmark = pytest.mark.gpu


def test_same_seed_same_result_different_seed_differs(hip_backend):
    def run(seed):
        fg = iif.generateChainEuclid(30, vardims=2, priorEvery=10, N=200)
        iif.solveTree(fg, eliminationOrder=iif.nestedDissectionOrder(fg), backend=hip_backend, seed=seed)
        return np.concatenate([fg.getVal(v).ravel() for v in fg.ls()])

    a, b, c = run(11), run(11), run(12)
    np.testing.assert_array_equal(a, b)  # counter-based RNG + fixed-order reductions: bitwise reproducible
    assert np.abs(a - c).max() > 1e-3


def test_batch_size_does_not_change_results(hip_backend):
    """The host picks the workgroup geometry from the batch size (helper lanes P, sample groups G);
    the same op must give the same answer alone and inside a chip-filling batch."""
    N, man = 200, abi.EUCLID2
    rng = np.random.default_rng(5)
    a, b = rand_points(rng, man, N, 0.0, 0.4), rand_points(rng, man, N, 1.0, 0.4)
    nbig = 1500

    def run(nops):
        be = hip_backend(N, 2 + 3 * nops, 0)
        be.slot_write(0, man, a)
        be.slot_write(1, man, b)
        props, prods = [], []
        for i in range(nops):
            o = 2 + 3 * i
            props.append(relative_factor_desc(abi.F_LINREL, man, 2, 1, [0, 1], o, 900, [1.0, 1.0], [0.1, 0.1]))
            props.append(relative_factor_desc(abi.F_PRIOR, man, 1, 0, [1], o + 1, 901, [1.0, 1.0], [0.3, 0.3]))
            prods.append(product_desc(man, [o, o + 1], o + 2, 902))
        prog = be.program([(abi.STAGE_PROPOSALS, props), (abi.STAGE_PRODUCTS, prods)])
        prog.run()
        be.synchronize()
        out = [be.slot_read(2 + 3 * i + 2, man) for i in (0, nops - 1)]
        prop = be.slot_read(2, man)
        be.close()
        return out, prop

    (s0, s1), sp = run(1)
    (b0, b1), bp = run(nbig)
    for (p, bw) in (b0, b1):
        assert_points_close(man, s0[0], p, rtol=0, what="product in big batch")
        np.testing.assert_allclose(bw, s0[1], rtol=0)
    assert_points_close(man, sp[0], bp[0], rtol=0, what="proposal in big batch")
    np.testing.assert_allclose(bp[1], sp[1], rtol=0)


@pytest.mark.parametrize("N", [8, 64, 65, 512])
def test_extreme_particle_counts_match_oracle(oracle_backend, hip_backend, N):
    man = abi.EUCLID2
    rng = np.random.default_rng(N)
    a, b = rand_points(rng, man, N, 0.0, 0.3), rand_points(rng, man, N, 1.0, 0.3)
    d = relative_factor_desc(abi.F_LINREL, man, 2, 1, [0, 1], 2, 77, [1.0, 1.0], [0.1, 0.1])
    d2 = relative_factor_desc(abi.F_PRIOR, man, 1, 0, [1], 3, 78, [1.0, 1.0], [0.3, 0.3])
    pd = product_desc(man, [2, 3], 4, 79, labels_out=0)

    def setup(be):
        be.slot_write(0, man, a)
        be.slot_write(1, man, b)

    def run(be):
        be.run_proposals([d, d2])
        be.run_products([pd])

    o, h = both(oracle_backend, hip_backend, N, 5, 2 * N, setup, run,
                lambda be: (be.slot_read(2, man), be.slot_read(4, man), be.side_read(0, 2 * N)))
    assert_points_close(man, o[0][0], h[0][0], what=f"conv N={N}")
    np.testing.assert_allclose(h[0][1], o[0][1], rtol=0)
    np.testing.assert_array_equal(o[2], h[2])
    assert_points_close(man, o[1][0], h[1][0], what=f"product N={N}")
    np.testing.assert_allclose(h[1][1], o[1][1], rtol=0)


def test_baseline_config2_full_size_properties(hip_backend):
    """BASELINE config 2 at full size (1000 variables, N = 200): finite, on-manifold, bandwidths
    positive, message count 2(C-1), every variable solved once, posterior means near x_i = (i, i)."""
    fg = iif.generateChainEuclid(1000, vardims=2, priorEvery=100, N=200)
    order = iif.nestedDissectionOrder(fg)
    tree, tm = iif.solveTree(fg, eliminationOrder=order, backend=hip_backend, seed=1, return_timing=True)
    assert tm["messages"] == 2 * (len(tree.cliques) - len(tree.roots))
    assert sorted(v for c in tree.cliques.values() for v in c.frontalIDs) == sorted(fg.ls())
    err = []
    for i in range(1000):
        var = fg.getVariable(f"x{i}")
        assert var.solvedCount == 1 and np.isfinite(var.val).all() and (var.bw > 0).all()
        err.append(np.abs(var.val.mean(axis=0) - i).max())
    assert max(err) < 1.5 and np.mean(err) < 0.4


@pytest.mark.parametrize("manifold,F,N", [(abi.SE2, 3, 200), (abi.EUCLID3, 2, 300), (abi.CIRCULAR, 4, 100), (abi.EUCLID2, 8, 64), (abi.EUCLID2, 2, 200)])
@pytest.mark.parametrize("batch", [1, 64, 256])
def test_product_geometries_match_oracle(oracle_backend, hip_backend, manifold, F, N, batch):
    """The three product-kernel geometries (latency l8, m4, throughput t2 -- picked from the batch size)
    against the oracle: identical labels, points and bandwidths to 1e-9, for the first and the last
    product of the batch.  Sweeps per level: 1, 2 (the second sweep draws from a block of its own) and, on the Euclid(2)
    pair, the maximum of 8."""
    rng = np.random.default_rng(17 * F + N + manifold)
    D = abi.MANIFOLD_DIM[manifold]
    dens = [rand_points(rng, manifold, N, 0.15 * j, 0.5) for j in range(F)]
    niter = 8 if (manifold == abi.EUCLID2 and F == 2) else 1 + (F == 2)

    def run(fac, nops):
        be = fac(N, F + nops, N * F * nops)
        for j, p in enumerate(dens):
            be.slot_write(j, manifold, p, np.full(D, 0.2 + 0.02 * j))
        descs = [product_desc(manifold, list(range(F)), F + i, 4321, labels_out=i * N * F, niter=niter) for i in range(nops)]
        be.run_products(descs)
        out = [(be.slot_read(F + i, manifold), be.side_read(i * N * F, N * F)) for i in (0, nops - 1)]
        be.close()
        return out

    ref = run(oracle_backend, 1)[0]
    for (pts, bw), lab in run(hip_backend, batch):
        np.testing.assert_array_equal(lab, ref[1])
        assert_points_close(manifold, ref[0][0], pts, what=f"product batch {batch}")
        np.testing.assert_allclose(bw, ref[0][1], rtol=0)


@pytest.mark.parametrize("manifold,F,N", [(abi.CIRCULAR, 81, 200), (abi.EUCLID2, 40, 200), (abi.SE2, 24, 100), (abi.EUCLID1, 128, 64)])
def test_products_of_many_densities(oracle_backend, hip_backend, manifold, F, N):
    """A landmark with many sightings (BASELINE config 3: 80 sightings + a prior on each door): the node
    statistics no longer fit the LDS and live in global memory; results still equal the oracle's."""
    rng = np.random.default_rng(F + N)
    D = abi.MANIFOLD_DIM[manifold]
    dens = [rand_points(rng, manifold, N, 0.02 * j, 0.6) for j in range(F)]

    def run(fac, nops):
        be = fac(N, F + nops, N * F * nops)
        for j, p in enumerate(dens):
            be.slot_write(j, manifold, p, np.full(D, 0.5 + 0.01 * j))
        descs = [product_desc(manifold, list(range(F)), F + i, 99, labels_out=i * N * F) for i in range(nops)]
        be.run_products(descs)
        out = [(be.slot_read(F + i, manifold), be.side_read(i * N * F, N * F)) for i in (0, nops - 1)]
        be.close()
        return out

    ref = run(oracle_backend, 1)[0]
    for nops in (1, 3):
        for (pts, bw), lab in run(hip_backend, nops):
            np.testing.assert_array_equal(lab, ref[1])
            assert_points_close(manifold, ref[0][0], pts, what=f"product of {F}")
            np.testing.assert_allclose(bw, ref[0][1], rtol=0)


def test_product_mean_is_unbiased_gpu(hip_backend):
    """size-independent property of the product sampler: no side of the KD splits is favoured (see
    tests/test_product_unbiased.py); 64 products of zero-mean densities in one launch"""
    from parity_utils import product_desc
    N, man, sig, B = 100, abi.EUCLID1, 0.1, 64
    rng = np.random.default_rng(1)
    be = hip_backend(N, 3 * B + 2 * B, 0)
    for s in range(3 * B):
        x = rng.normal(0, sig, (N, 1))
        be.slot_write(s, man, x - x.mean(), np.ones(1))
    be.run_bandwidth(list(range(3 * B)), [man] * (3 * B))
    descs = []
    for b in range(B):
        descs.append(product_desc(man, [3 * b, 3 * b + 1], 3 * B + 2 * b, 5000 + b))
        descs.append(product_desc(man, [3 * b, 3 * b + 1, 3 * b + 2], 3 * B + 2 * b + 1, 6000 + b))
    be.run_products(descs)
    m2 = np.array([be.slot_read(3 * B + 2 * b, man)[0].mean() for b in range(B)])
    m3 = np.array([be.slot_read(3 * B + 2 * b + 1, man)[0].mean() for b in range(B)])
    be.close()
    for m in (m2, m3):
        assert abs(m.mean()) < 0.05 * sig, (m.mean(), m.std() / np.sqrt(B))


@pytest.mark.parametrize("build", ["euclid2", "doors", "se2"])
def test_program_with_new_seeds_is_the_program_compiled_with_them(hip_backend, build):
    """nbp_program_set_seeds (what the native host's plan cache re-seeds a cached level program with): a tree program compiled
    with seed A and handed the seeds of the SAME tree compiled with seed B delivers, from the same initial beliefs, the posteriors
    of the program compiled with B -- bit for bit (graph replay included: the second run of a range is captured, the third replayed)"""
    from parity_utils import iif
    make = {"euclid2": lambda: iif.generateChainEuclid(30, vardims=2, priorEvery=10, N=100),
            "doors": lambda: iif.generateCircularDoors(nposes=20, N=100, sightEvery=5),
            "se2": lambda: iif.generateSE2Lattice(rows=2, cols=4, N=100, closeEvery=2)}[build]
    fg = make()
    iif.initAll(fg, backend=hip_backend, seed=3)
    tree = iif.buildTreeReset(fg, iif.nestedDissectionOrder(fg))
    tpa, tpb = iif.TreeProgram(fg, tree, seed=11), iif.TreeProgram(fg, tree, seed=12)
    from iif_amd.backend import HipProgram
    sa, sb = HipProgram.seeds_of(tpa.stages), HipProgram.seeds_of(tpb.stages)
    assert len(sa) == len(sb) and sa != sb

    def solve(tp, seeds_then=None, runs=1):
        be = hip_backend(fg.solverParams.N, tp.n_slots)
        try:
            prog = be.program(tp.stages, lazy_bandwidth=True)
            assert prog.num_seeds() == len(sa)
            out = None
            for r in range(runs):
                for v in fg.ls():
                    var = fg.getVariable(v)
                    be.belief_write(tp.main[v], var.varType.manifold, var.val, var.bw)
                iif.solver.write_densities(fg, be)
                if seeds_then is not None and r == runs - 1:
                    prog.set_seeds(seeds_then)
                prog.run()
                be.synchronize()
                out = {v: be.slot_read(tp.main[v], fg.getVariable(v).varType.manifold) for v in fg.ls()}
            prog.close()
            return out
        finally:
            be.close()

    want = solve(tpb)
    for runs in (1, 3):  # (3: the re-seeded run is a hipGraph replay)
        got = solve(tpa, seeds_then=sb, runs=runs)
        for v in fg.ls():
            assert np.array_equal(got[v][0], want[v][0]) and np.array_equal(got[v][1], want[v][1]), (build, runs, v)
    back = solve(tpa, seeds_then=sa, runs=2)
    ref = solve(tpa)
    assert all(np.array_equal(back[v][0], ref[v][0]) for v in fg.ls())
