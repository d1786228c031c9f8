"""-m gpu: the HIP kernels against the CPU oracle on identical seeded inputs, through the C ABI.
Floating point AND integers identical to the last bit (round 6: the values that travel are computed by one arithmetic on both
sides); the tolerances that remain are against closed forms or across the host boundary's own
conversions, and say so."""
import numpy as np
import pytest

from parity_utils import (abi, assert_points_close, both, coord_diff, iif, product_desc, rand_points,
                          relative_factor_desc)

pytestmark = pytest.mark.gpu

MANIS = [abi.EUCLID1, abi.EUCLID2, abi.EUCLID3, abi.CIRCULAR, abi.SE2]


@pytest.mark.parametrize("manifold", MANIS)
@pytest.mark.parametrize("N", [100, 200])
def test_bandwidth_lcv(oracle_backend, hip_backend, manifold, N):
    rng = np.random.default_rng(manifold * 1000 + N)
    pts = rand_points(rng, manifold, N, center=0.5, spread=0.7)

    def setup(be):
        be.slot_write(0, manifold, pts)

    o, h = both(oracle_backend, hip_backend, N, 2, 0, setup, lambda be: be.run_bandwidth([0], [manifold]),
                lambda be: be.slot_read(0, manifold))
    assert_points_close(manifold, o[0], h[0], what="slot roundtrip")
    np.testing.assert_allclose(h[1], o[1], rtol=0)
    assert (o[1] > 0).all()


@pytest.mark.parametrize("manifold,kind", [(abi.EUCLID1, abi.F_PRIOR), (abi.EUCLID2, abi.F_PRIOR),
                                           (abi.EUCLID3, abi.F_PRIOR), (abi.CIRCULAR, abi.F_PRIOR),
                                           (abi.SE2, abi.F_PRIOR)])
@pytest.mark.parametrize("nullhypo", [0.0, 0.3])
def test_prior_proposal(oracle_backend, hip_backend, manifold, kind, nullhypo):
    N = 200
    rng = np.random.default_rng(7 + manifold)
    cur = rand_points(rng, manifold, N, center=1.0, spread=0.5)
    D = abi.MANIFOLD_DIM[manifold]
    d = relative_factor_desc(kind, manifold, 1, 0, [0], 1, 4242 + manifold, [0.3, -0.2, 0.5][:D], [0.1, 0.2, 0.05][:D],
                             nullhypo=nullhypo, mhidx_out=0)

    def setup(be):
        be.slot_write(0, manifold, cur)

    o, h = both(oracle_backend, hip_backend, N, 2, N, setup, lambda be: be.run_proposals([d]),
                lambda be: (be.slot_read(1, manifold), be.side_read(0, N)))
    np.testing.assert_array_equal(o[1], h[1])
    assert_points_close(manifold, o[0][0], h[0][0], what="prior proposal")
    np.testing.assert_allclose(h[0][1], o[0][1], rtol=0)
    if nullhypo > 0:
        assert 0 < (o[1] == 0).sum() < N


CASES = [
    (abi.F_LINREL, abi.EUCLID1, [1.0], [0.1]),
    (abi.F_LINREL, abi.EUCLID2, [1.0, 1.0], [0.1, 0.1]),
    (abi.F_LINREL, abi.EUCLID3, [1.0, 0.0, -0.5], [0.1, 0.2, 0.1]),
    (abi.F_CIRCULAR, abi.CIRCULAR, [0.4], [0.05]),
    (abi.F_SE2, abi.SE2, [1.0, 0.2, 0.3], [0.1, 0.1, 0.01]),
    (abi.F_EUCLIDDIST, abi.EUCLID2, [3.0], [0.1]),
]


@pytest.mark.parametrize("kind,manifold,mean,sig", CASES)
@pytest.mark.parametrize("sfidx", [0, 1])
def test_relative_conv(oracle_backend, hip_backend, kind, manifold, mean, sig, sfidx):
    N = 200
    rng = np.random.default_rng(100 * kind + manifold + sfidx)
    a = rand_points(rng, manifold, N, center=0.0, spread=0.3)
    b = rand_points(rng, manifold, N, center=1.0, spread=0.3)
    d = relative_factor_desc(kind, manifold, 2, sfidx, [0, 1], 2, 999 + kind * 7 + sfidx, mean, sig)

    def setup(be):
        be.slot_write(0, manifold, a)
        be.slot_write(1, manifold, b)

    def read(be):
        return be.slot_read(2, manifold), be.slot_read(sfidx, manifold), be.diag(reset=True)

    o, h = both(oracle_backend, hip_backend, N, 3, 0, setup, lambda be: be.run_proposals([d]), read)
    # EuclidDistance has a ring of solutions: Nelder-Mead walks to it from the inflated start, a
    # rare branch flip may move a particle along the ring by more than the tolerance
    max_bad = 2 if kind == abi.F_EUCLIDDIST else 0
    assert_points_close(manifold, o[0][0], h[0][0], rtol=0 if kind == abi.F_EUCLIDDIST else 1e-9, max_bad=max_bad,
                        what="relative conv")
    np.testing.assert_allclose(h[0][1], o[0][1], rtol=0 if kind == abi.F_EUCLIDDIST else 1e-9)
    # mutation contract: the stored belief of the target is untouched (testMultiHypo3Door.jl:74-90)
    assert_points_close(manifold, h[1][0], b if sfidx == 1 else a, what="target belief untouched")
    assert h[2]["solves"] == 3 * N == o[2]["solves"]
    assert h[2]["nan_results"] == 0


def test_mixture_and_nullhypo_conv(oracle_backend, hip_backend):
    N, manifold = 300, abi.EUCLID3
    rng = np.random.default_rng(5)
    a = rand_points(rng, manifold, N)
    b = rand_points(rng, manifold, N, center=1.0)
    comps = [(0.8, [1, 0, 0], [0.1, 0.1, 0.1]), (0.2, [1, 0, 0], [1.0, 1.0, 1.0])]
    d = relative_factor_desc(abi.F_LINREL, manifold, 2, 1, [0, 1], 2, 31337, None, None, ncomp=2, comps=comps,
                             nullhypo=0.25, mhidx_out=0)

    def setup(be):
        be.slot_write(0, manifold, a)
        be.slot_write(1, manifold, b)

    o, h = both(oracle_backend, hip_backend, N, 3, N, setup, lambda be: be.run_proposals([d]),
                lambda be: (be.slot_read(2, manifold), be.side_read(0, N)))
    np.testing.assert_array_equal(o[1], h[1])
    assert_points_close(manifold, o[0][0], h[0][0], what="mixture conv")
    np.testing.assert_allclose(h[0][1], o[0][1], rtol=0)


@pytest.mark.parametrize("sfidx", [0, 1, 3])
def test_multihypo_conv_injected_mhidx(oracle_backend, hip_backend, sfidx):
    """door-sighting pattern (testMultiHypo3Door.jl:57): [x, l0..l3], multihypo=[1,.25,.25,.25,.25]."""
    N, manifold = 200, abi.CIRCULAR
    rng = np.random.default_rng(11 + sfidx)
    doors = [-2.4, -0.8, 0.8, 2.4]
    pts = [rand_points(rng, manifold, N, center=0.7, spread=0.2)] + [rand_points(rng, manifold, N, center=t, spread=0.01) for t in doors]
    mh = [0.0, 0.25, 0.25, 0.25, 0.25]
    if sfidx == 0:
        mhidx = rng.integers(2, 6, size=N).astype(np.int32)
    else:
        mhidx = rng.choice([0, 2, 3, 4, 5], size=N).astype(np.int32)
    d = relative_factor_desc(abi.F_CIRCULAR, manifold, 5, sfidx, [0, 1, 2, 3, 4], 5, 77 + sfidx, [0.0], [0.1],
                             multihypo=mh, mhidx_in=0, mhidx_out=N)

    def setup(be):
        for i, p in enumerate(pts):
            be.slot_write(i, manifold, p)
        be.side_write(0, mhidx)

    o, h = both(oracle_backend, hip_backend, N, 6, 2 * N, setup, lambda be: be.run_proposals([d]),
                lambda be: (be.slot_read(5, manifold), be.side_read(N, N)))
    np.testing.assert_array_equal(h[1], mhidx)
    np.testing.assert_array_equal(o[1], mhidx)
    assert_points_close(manifold, o[0][0], h[0][0], what="multihypo conv")
    np.testing.assert_allclose(h[0][1], o[0][1], rtol=0)


def test_multihypo_sampled_mhidx_identical(oracle_backend, hip_backend):
    N, manifold = 200, abi.CIRCULAR
    rng = np.random.default_rng(3)
    pts = [rand_points(rng, manifold, N, spread=0.3) for _ in range(5)]
    d = relative_factor_desc(abi.F_CIRCULAR, manifold, 5, 2, [0, 1, 2, 3, 4], 5, 555, [0.0], [0.1],
                             multihypo=[0.0, 0.25, 0.25, 0.25, 0.25], mhidx_out=0)

    def setup(be):
        for i, p in enumerate(pts):
            be.slot_write(i, manifold, p)

    o, h = both(oracle_backend, hip_backend, N, 6, N, setup, lambda be: be.run_proposals([d]),
                lambda be: (be.slot_read(5, manifold), be.side_read(0, N)))
    np.testing.assert_array_equal(o[1], h[1])
    assert set(np.unique(o[1])) <= {0, 2, 3, 4, 5} and (o[1] == 0).any()
    assert_points_close(manifold, o[0][0], h[0][0])


def test_msgprior_proposal(oracle_backend, hip_backend):
    N, manifold = 200, abi.SE2
    rng = np.random.default_rng(21)
    cur = rand_points(rng, manifold, N)
    msg = rand_points(rng, manifold, N, center=2.0, spread=0.2)
    d = relative_factor_desc(abi.F_MSGPRIOR, manifold, 1, 0, [0, 1], 2, 8080, [0], [0])

    def setup(be):
        be.slot_write(0, manifold, cur)
        be.slot_write(1, manifold, msg, bw=np.array([0.05, 0.07, 0.02]))

    o, h = both(oracle_backend, hip_backend, N, 3, 0, setup, lambda be: be.run_proposals([d]),
                lambda be: be.slot_read(2, manifold))
    assert_points_close(manifold, o[0], h[0], what="MsgPrior proposal")
    np.testing.assert_allclose(h[1], o[1], rtol=0)


@pytest.mark.parametrize("manifold", MANIS)
@pytest.mark.parametrize("F", [2, 3])
def test_manifold_product(oracle_backend, hip_backend, manifold, F):
    N = 200
    rng = np.random.default_rng(manifold * 10 + F)
    D = abi.MANIFOLD_DIM[manifold]
    dens = [rand_points(rng, manifold, N, center=0.2 * j, spread=0.5) for j in range(F)]
    bws = [np.full(D, 0.15 + 0.03 * j) for j in range(F)]
    d = product_desc(manifold, list(range(F)), F, 2024 + manifold + F, labels_out=0)

    def setup(be):
        for j in range(F):
            be.slot_write(j, manifold, dens[j], bws[j])

    o, h = both(oracle_backend, hip_backend, N, F + 1, N * F, setup, lambda be: be.run_products([d]),
                lambda be: (be.slot_read(F, manifold), be.side_read(0, N * F)))
    lab_o, lab_h = o[1].reshape(N, F), h[1].reshape(N, F)
    same = (lab_o == lab_h).all(axis=1)
    assert same.sum() >= N - 1, f"{N - same.sum()} samples picked different labels"
    assert_points_close(manifold, o[0][0][same], h[0][0][same], what="product samples")
    if same.all():
        np.testing.assert_allclose(h[0][1], o[0][1], rtol=0)


def test_product_passthrough_single_density(oracle_backend, hip_backend):
    N, manifold = 100, abi.EUCLID2
    rng = np.random.default_rng(1)
    p = rand_points(rng, manifold, N)
    d = product_desc(manifold, [0], 1, 5)

    def setup(be):
        be.slot_write(0, manifold, p, np.array([0.3, 0.4]))

    o, h = both(oracle_backend, hip_backend, N, 2, 0, setup, lambda be: be.run_products([d]),
                lambda be: be.slot_read(1, manifold))
    np.testing.assert_array_equal(h[0], p)
    np.testing.assert_array_equal(h[1], [0.3, 0.4])
    np.testing.assert_array_equal(o[0], p)


def test_error_codes(hip_backend):
    be = hip_backend(100, 2, 0)
    d = relative_factor_desc(abi.F_LINREL, abi.EUCLID1, 2, 1, [0, 7], 1, 1, [1.0], [0.1])  # slot 7 out of range
    with pytest.raises(iif.NbpError):
        be.run_proposals([d])
    d2 = relative_factor_desc(99, abi.EUCLID1, 2, 1, [0, 1], 1, 1, [1.0], [0.1])
    with pytest.raises(iif.NbpError):
        be.run_proposals([d2])
    be.close()


# ---- partial-dimension factors ----------------------------------------
@pytest.mark.parametrize("manifold,mask", [(abi.EUCLID2, 1), (abi.EUCLID2, 2), (abi.EUCLID3, 5), (abi.EUCLID3, 2),
                                           (abi.SE2, 3), (abi.SE2, 4)])
@pytest.mark.parametrize("nullhypo", [0.0, 0.4])
def test_partial_prior_proposal(oracle_backend, hip_backend, manifold, mask, nullhypo):
    N = 200
    rng = np.random.default_rng(70 + manifold + mask)
    cur = rand_points(rng, manifold, N, center=1.0, spread=0.5)
    Z = bin(mask).count("1")
    d = relative_factor_desc(abi.F_PRIOR, manifold, 1, 0, [0], 1, 515 + manifold + mask, [0.3, -0.2, 0.5][:Z],
                             [0.1, 0.2, 0.05][:Z], nullhypo=nullhypo, mhidx_out=0)
    d.partial_mask = mask

    def setup(be):
        be.slot_write(0, manifold, cur)

    o, h = both(oracle_backend, hip_backend, N, 2, N, setup, lambda be: be.run_proposals([d]),
                lambda be: (be.slot_read(1, manifold), be.side_read(0, N)))
    np.testing.assert_array_equal(o[1], h[1])
    assert_points_close(manifold, o[0][0], h[0][0], what="partial prior proposal")
    np.testing.assert_allclose(h[0][1], o[0][1], rtol=0)
    # coordinates outside the mask keep the target's current values, exactly
    from parity_utils import coords
    D = abi.MANIFOLD_DIM[manifold]
    c0, c1 = coords(manifold, cur), coords(manifold, h[0][0])
    for k in range(D):
        if not (mask >> k) & 1:
            np.testing.assert_allclose(c1[:, k], c0[:, k], atol=1e-12)  # (through the host form: a heading round-trips via atan2(sin, cos))
        else:
            assert np.abs(c1[:, k] - c0[:, k]).max() > 1e-3


@pytest.mark.parametrize("manifold,mask", [(abi.EUCLID2, 2), (abi.EUCLID2, 1), (abi.EUCLID3, 4), (abi.EUCLID3, 5), (abi.EUCLID3, 6), (abi.EUCLID2, 3)])
@pytest.mark.parametrize("sfidx", [0, 1])
def test_partial_relative_conv(oracle_backend, hip_backend, manifold, mask, sfidx):
    N = 200
    rng = np.random.default_rng(300 + manifold + mask + sfidx)
    a = rand_points(rng, manifold, N, center=0.0, spread=0.3)
    b = rand_points(rng, manifold, N, center=1.0, spread=0.3)
    two = bin(mask).count("1") == 2  # two partial coordinates: n-D BFGS on the pair
    d = relative_factor_desc(abi.F_LINREL, manifold, 2, sfidx, [0, 1], 2, 888 + mask + sfidx, [10.0, -4.0] if two else [10.0],
                             [1.0, 0.5] if two else [1.0])
    d.partial_mask = mask

    def setup(be):
        be.slot_write(0, manifold, a)
        be.slot_write(1, manifold, b)

    o, h = both(oracle_backend, hip_backend, N, 3, 0, setup, lambda be: be.run_proposals([d]),
                lambda be: be.slot_read(2, manifold))
    assert_points_close(manifold, o[0], h[0], what="partial relative conv")
    np.testing.assert_allclose(h[1], o[1], rtol=0)
    ks = [k for k in range(3) if (mask >> k) & 1]
    tgt, oth = (b, a) if sfidx == 1 else (a, b)
    sign = 1.0 if sfidx == 1 else -1.0
    for k, zm in zip(ks, [10.0, -4.0]):
        assert abs((h[0][:, k] - oth[:, k]).mean() - sign * zm) < 0.5
    for kk in range(abi.MANIFOLD_DIM[manifold]):
        if kk not in ks:
            np.testing.assert_allclose(h[0][:, kk], tgt[:, kk], atol=1e-12)


PARTIAL_PRODUCTS = [
    (abi.EUCLID2, [1, 0]),        # partial + full
    (abi.EUCLID2, [1, 2]),        # two partials, disjoint coordinates
    (abi.EUCLID2, [2, 2]),        # coordinate 0 uninformed -> old points
    (abi.EUCLID3, [5, 0, 2]),
    (abi.EUCLID3, [1, 1, 4]),     # coordinate 1 uninformed
    (abi.SE2, [3, 0]),            # (x, y) partial + full
    (abi.SE2, [4, 3, 0]),         # theta partial, xy partial, full
]


@pytest.mark.parametrize("manifold,masks", PARTIAL_PRODUCTS)
@pytest.mark.parametrize("N", [100, 200])
def test_partial_product(oracle_backend, hip_backend, manifold, masks, N):
    F = len(masks)
    rng = np.random.default_rng(900 + manifold + sum(masks) + N)
    dens = [rand_points(rng, manifold, N, center=0.2 * j, spread=0.6) for j in range(F)]
    old = rand_points(rng, manifold, N, center=5.0, spread=0.1)
    out = F + 1
    d = iif.solver.product_desc(manifold, list(range(F)), out, 31337 + N, 1, 0, partials=masks, old_slot=F)

    def setup(be):
        for j, p in enumerate(dens):
            be.slot_write(j, manifold, p)
        be.slot_write(F, manifold, old)
        be.run_bandwidth(list(range(F)), [manifold] * F)

    o, h = both(oracle_backend, hip_backend, N, F + 2, N * F, setup, lambda be: be.run_products([d]),
                lambda be: (be.slot_read(out, manifold), be.side_read(0, N * F)))
    np.testing.assert_array_equal(o[1], h[1])
    assert_points_close(manifold, o[0][0], h[0][0], what="partial product")
    np.testing.assert_allclose(h[0][1], o[0][1], rtol=0)
    from parity_utils import coords
    D = abi.MANIFOLD_DIM[manifold]
    cov = 0
    for m in masks:
        cov |= m if m else (1 << D) - 1
    co, cn = coords(manifold, old), coords(manifold, h[0][0])
    for k in range(D):
        if not (cov >> k) & 1:
            np.testing.assert_allclose(cn[:, k], co[:, k], atol=1e-12)


def test_partial_descriptors_are_validated(hip_backend):
    be = hip_backend(64, 4, 0)
    d = relative_factor_desc(abi.F_LINREL, abi.EUCLID3, 2, 1, [0, 1], 2, 1, [1.0], [1.0])
    d.partial_mask = 7  # three partial coordinates on a relative factor: not supported (one or two are)
    with pytest.raises(iif.NbpError):
        be.run_proposals([d])
    d = relative_factor_desc(abi.F_PRIOR, abi.EUCLID2, 1, 0, [0], 1, 1, [1.0], [1.0])
    d.partial_mask = 4  # coordinate 2 of a 2-D variable
    with pytest.raises(iif.NbpError):
        be.run_proposals([d])
    p = iif.solver.product_desc(abi.EUCLID2, [0, 1], 2, 1, 1, -1, partials=[1, 0], old_slot=99)
    with pytest.raises(iif.NbpError):
        be.run_products([p])
    be.close()


# ---- approxDeconv (DeconvUtils.jl) -------------------------------------------------------------
@pytest.mark.parametrize("kind,manifold,mean,sig", CASES)
def test_deconv(oracle_backend, hip_backend, kind, manifold, mean, sig):
    N = 200
    rng = np.random.default_rng(4000 + 10 * kind + manifold)
    a = rand_points(rng, manifold, N, center=0.0, spread=0.3)
    b = rand_points(rng, manifold, N, center=1.0, spread=0.3)
    d = relative_factor_desc(kind, manifold, 2, 1, [0, 1], 2, 777 + kind, mean, sig)
    Z = len(mean)
    zman = {1: abi.EUCLID1, 2: abi.EUCLID2, 3: abi.EUCLID3}[Z]

    def setup(be):
        be.slot_write(0, manifold, a)
        be.slot_write(1, manifold, b)

    def read(be):
        return be.slot_read(2, zman)[0], be.slot_read(3, zman)[0], be.diag(reset=True)

    o, h = both(oracle_backend, hip_backend, N, 4, 0, setup, lambda be: be.run_deconv([d], [3]), read)
    np.testing.assert_allclose(h[1], o[1], rtol=0, atol=0)  # sampled measurements: same stream
    np.testing.assert_allclose(h[0], o[0], rtol=0, atol=0)    # predicted measurements
    assert h[2]["solves"] == N
    # the prediction zeroes the residual: check against the closed forms
    from parity_utils import coords
    ca, cb = coords(manifold, a), coords(manifold, b)
    if kind == abi.F_LINREL:
        np.testing.assert_allclose(h[0], cb - ca, atol=1e-5 if Z == 1 else 2e-3)  # NelderMead stops at g_tol 1e-8 on f
    elif kind == abi.F_EUCLIDDIST:
        np.testing.assert_allclose(h[0][:, 0], np.linalg.norm(cb - ca, axis=1), atol=1e-5)
    elif kind == abi.F_SE2:
        dx, dy = cb[:, 0] - ca[:, 0], cb[:, 1] - ca[:, 1]
        c, s = np.cos(ca[:, 2]), np.sin(ca[:, 2])
        np.testing.assert_allclose(h[0][:, 0], c * dx + s * dy, atol=2e-3)
        np.testing.assert_allclose(h[0][:, 1], -s * dx + c * dy, atol=2e-3)


@pytest.mark.parametrize("flags,expect", [(1 | 0x80 | (1 << 9), {2}), (1 | 0x80 | (1 << 10), {3}), (1 | 0x80 | (7 << 8), {2, 3})])
def test_multihypo_isinit_suppression(oracle_backend, hip_backend, flags, expect):
    """uninitialised hypotheses get probability 0 in the mhidx draw (ExplicitDiscreteMarginalizations.jl:161-172)"""
    N, man = 200, abi.EUCLID2
    rng = np.random.default_rng(flags)
    pts = [rand_points(rng, man, N, c, 0.2) for c in (0.0, 5.0, -5.0)]
    d = relative_factor_desc(abi.F_LINREL, man, 3, 0, [0, 1, 2], 3, 99, [1.0, 1.0], [0.1, 0.1],
                             multihypo=[0.0, 0.5, 0.5], mhidx_out=0)
    d.has_multihypo = flags

    def setup(be):
        for s, p in enumerate(pts):
            be.slot_write(s, man, p)

    o, h = both(oracle_backend, hip_backend, N, 4, N, setup, lambda be: be.run_proposals([d]),
                lambda be: (be.slot_read(3, man), be.side_read(0, N)))
    np.testing.assert_array_equal(o[1], h[1])
    assert set(h[1].tolist()) == expect
    assert_points_close(man, o[0][0], h[0][0], what="multihypo with isinit flags")


# ---- host-buffer entry points: same results as the slot path (and therefore as the oracle) ----------
@pytest.mark.parametrize("manifold", [abi.EUCLID2, abi.CIRCULAR, abi.SE2])
def test_host_buffer_entry_points(hip_backend, manifold):
    N = 200
    rng = np.random.default_rng(31 + manifold)
    D = abi.MANIFOLD_DIM[manifold]
    a, b = rand_points(rng, manifold, N, 0.0, 0.3), rand_points(rng, manifold, N, 1.0, 0.3)
    kind = {abi.EUCLID2: abi.F_LINREL, abi.CIRCULAR: abi.F_CIRCULAR, abi.SE2: abi.F_SE2}[manifold]
    Z = {abi.F_LINREL: D, abi.F_CIRCULAR: 1, abi.F_SE2: 3}[kind]
    be = hip_backend(N, 6, 3 * N)
    # nbp_kde_bandwidth == slot path
    be.slot_write(5, manifold, a)
    be.run_bandwidth([5], [manifold])
    np.testing.assert_array_equal(be.kde_bandwidth(manifold, a), be.slot_read(5, manifold)[1])
    # nbp_conv == nbp_run_proposals
    d = relative_factor_desc(kind, manifold, 2, 1, [0, 1], 2, 2024, [0.5, 0.2, 0.1][:Z], [0.1, 0.1, 0.05][:Z], nullhypo=0.2, mhidx_out=0)
    be.slot_write(0, manifold, a)
    be.slot_write(1, manifold, b)
    be.run_proposals([d])
    ref_pts, ref_bw = be.slot_read(2, manifold)
    ref_mh = be.side_read(0, N)
    pts, bw, mh = be.conv(d, [a, b], want_mhidx=True)
    np.testing.assert_array_equal(mh, ref_mh)
    np.testing.assert_array_equal(pts, ref_pts)
    np.testing.assert_array_equal(bw, ref_bw)
    # injected mhidx is honoured
    inj = np.where(np.arange(N) % 3 == 0, 0, 1).astype(np.int32)
    _, _, mh2 = be.conv(d, [a, b], mhidx_in=inj, want_mhidx=True)
    np.testing.assert_array_equal(mh2, inj)
    # nbp_manifold_product == nbp_run_products
    dens = [(rand_points(rng, manifold, N, 0.1 * j, 0.5), np.full(D, 0.2 + 0.05 * j)) for j in range(3)]
    for j, (p, w) in enumerate(dens):
        be.slot_write(j, manifold, p, w)
    be.run_products([product_desc(manifold, [0, 1, 2], 4, 777, labels_out=0)])
    ref_pts, ref_bw = be.slot_read(4, manifold)
    ref_lab = be.side_read(0, 3 * N)
    pts, bw, lab = be.manifold_product(manifold, dens, 777, want_labels=True)
    np.testing.assert_array_equal(lab, ref_lab)
    np.testing.assert_array_equal(pts, ref_pts)
    np.testing.assert_array_equal(bw, ref_bw)
    # too few slots -> status, not a crash
    small = hip_backend(N, 2, 0)
    with pytest.raises(iif.NbpError):
        small.manifold_product(manifold, dens, 1)
    small.close()
    be.close()


# ---- useMsgLikelihoods: differential factors (TreeMessageUtils.jl:279-335) ---------------------------
@pytest.mark.parametrize("kind,manifold,mean,sig", [c for c in CASES if c[0] != abi.F_EUCLIDDIST])
def test_differential_factor_program(oracle_backend, hip_backend, kind, manifold, mean, sig):
    """a resident program: deconvolution between two beliefs (NBP_STAGE_DECONV: predicted measurements + their
    manikde! bandwidth), then the relative factor whose measurement is that KDE (meas_kde) convolved both ways,
    then a product -- GPU against the oracle"""
    N = 200
    rng = np.random.default_rng(5000 + 10 * kind + manifold)
    a = rand_points(rng, manifold, N, center=0.3, spread=0.3)
    b = rand_points(rng, manifold, N, center=1.2, spread=0.3)
    dec = relative_factor_desc(kind, manifold, 2, 1, [0, 1], 2, 901 + kind, [0.0] * len(mean), [1.0] * len(mean))
    fwd = relative_factor_desc(kind, manifold, 2, 1, [0, 1], 3, 902 + kind, mean, sig)
    rev = relative_factor_desc(kind, manifold, 2, 0, [0, 1], 4, 903 + kind, mean, sig)
    fwd.meas_kde = rev.meas_kde = 2 + 1
    prod = product_desc(manifold, [3, 1], 5, 904)

    def setup(be):
        be.slot_write(0, manifold, a)
        be.slot_write(1, manifold, b)
        be.run_bandwidth([0, 1], [manifold, manifold])

    def run(be):
        prog = be.program([(abi.STAGE_DECONV, [dec]), (abi.STAGE_PROPOSALS, [fwd, rev]), (abi.STAGE_PRODUCTS, [prod])])
        prog.run()
        be.synchronize() if hasattr(be, "synchronize") else None
        prog.close()

    def read(be):
        return [be.slot_read(s, manifold) for s in (2, 3, 4, 5)]

    o, h = both(oracle_backend, hip_backend, N, 6, 0, setup, run, read)
    assert_points_close(manifold, h[0][0], o[0][0], rtol=0, what="predicted measurements")
    np.testing.assert_allclose(h[0][1], o[0][1], rtol=0)  # the KDE's bandwidth
    for k in (1, 2):  # both convolution directions through the KDE-measurement factor
        assert_points_close(manifold, h[k][0], o[k][0], rtol=0, max_bad=0, what=f"proposal {k}")
    # the forward convolution lands on b's belief
    assert np.abs(coord_diff(manifold, h[1][0], b).mean(axis=0)).max() < 0.2
    assert np.isfinite(h[3][0]).all()


def test_differential_factor_validation(hip_backend):
    be = hip_backend(64, 4, 0)
    d = relative_factor_desc(abi.F_EUCLIDDIST, abi.EUCLID2, 2, 1, [0, 1], 2, 1, [1.0], [1.0])
    d.meas_kde = 4
    with pytest.raises(iif.NbpError):  # the measurement of a distance factor does not live on the variable's manifold
        be.run_proposals([d])
    d = relative_factor_desc(abi.F_LINREL, abi.EUCLID2, 2, 1, [0, 1], 2, 1, [1.0, 1.0], [1.0, 1.0])
    d.meas_kde = 6  # slot 5 of 4
    with pytest.raises(iif.NbpError):
        be.run_proposals([d])
    p = relative_factor_desc(abi.F_PRIOR, abi.EUCLID2, 1, 0, [0], 1, 1, [1.0, 1.0], [1.0, 1.0])
    with pytest.raises(iif.NbpError):  # deconvolution stages take relative factors
        be.program([(abi.STAGE_DECONV, [p])])
    be.close()


def test_stored_measurements_gpu(oracle_backend, hip_backend):
    """needFreshMeasurements = false: `meas_seed` (SolveTree.jl:119) on the device, incl. program re-keying,
    and the same proposals GPU against oracle"""
    from test_stored_measurements import check_stored_measurements
    check_stored_measurements(hip_backend)
    N, man = 200, abi.EUCLID2
    rng = np.random.default_rng(77)
    a, b = rand_points(rng, man, N, 0.0, 0.3), rand_points(rng, man, N, 2.0, 0.3)
    d1 = relative_factor_desc(abi.F_LINREL, man, 2, 1, [0, 1], 2, 31, [1.0, 1.0], [0.3, 0.3])
    d2 = relative_factor_desc(abi.F_LINREL, man, 2, 0, [0, 1], 3, 32, [1.0, 1.0], [0.3, 0.3])
    d2.meas_seed = 31

    def setup(be):
        be.slot_write(0, man, a)
        be.slot_write(1, man, b)

    o, h = both(oracle_backend, hip_backend, N, 4, 0, setup, lambda be: be.run_proposals([d1, d2]),
                lambda be: [be.slot_read(s, man)[0] for s in (2, 3)])
    for k in range(2):
        assert_points_close(man, h[k], o[k], rtol=0, max_bad=0, what=f"proposal {k}")


# ---- manikde! of predicted measurements: a circular coordinate is fitted on its wrapped representative --------------------
@pytest.mark.parametrize("manifold", [abi.CIRCULAR, abi.SE2])
@pytest.mark.parametrize("N", [200, 300])
def test_fit_of_angles_a_deconvolution_leaves_unwrapped(oracle_backend, hip_backend, manifold, N):
    """approxDeconv stores b - a as the search found it: outside [-pi, pi) for one point in ten on a circle.  The fit of such
    a slot is the fit of the wrapped angles (the kernels stage them wrapped), bit for bit the oracle's -- and the fit of the
    same points written wrapped.  (Found by the stage-wise harness with useMsgLikelihoods: profiles/r06_stagewise_other_solver_parameters.txt)"""
    rng = np.random.default_rng(90 + manifold + N)
    D = abi.MANIFOLD_DIM[manifold]
    rows = np.zeros((N, 3))
    rows[:, :D] = rng.normal(size=(N, D)) * 2.0
    rows[:, D - 1] = rng.uniform(-2 * np.pi, 2 * np.pi, size=N)  # the circular coordinate, as raw as a search leaves it
    wrapped = rows.copy()
    wrapped[:, D - 1] = (rows[:, D - 1] + np.pi) % (2 * np.pi) - np.pi

    def setup(be):
        be.slot_write(0, abi.EUCLID3, rows)
        be.slot_write(1, abi.EUCLID3, wrapped)

    def read(be):
        return be.slot_read(0, abi.EUCLID3), be.slot_read(1, abi.EUCLID3)

    o, h = both(oracle_backend, hip_backend, N, 2, 0, setup, lambda be: be.run_bandwidth([0, 1], [manifold, manifold]), read)
    for s in (0, 1):
        assert np.array_equal(o[s][0], h[s][0]) and np.array_equal(np.asarray(o[s][1]), np.asarray(h[s][1]))
    assert np.array_equal(h[0][0], rows)  # the fit does not touch the points
    bw_raw, bw_wrapped = np.asarray(h[0][1]), np.asarray(h[1][1])
    assert np.all(bw_raw[:D] > 0)
    # the same KDE: the two fits can only differ through the rounding of the host-side wrap above (an ulp on a few angles)
    np.testing.assert_allclose(bw_raw, bw_wrapped, rtol=0.05)


# ---- Nelder-Mead with EQUAL vertex values: Optim's sortperm! orders them by their slot -------------------------------------
@pytest.mark.parametrize("manifold", [abi.EUCLID2, abi.EUCLID3])
@pytest.mark.parametrize("N", [200, 300])
def test_searches_whose_vertices_tie(oracle_backend, hip_backend, manifold, N):
    """A search whose objective is symmetric in its coordinates -- both beliefs on the diagonal, a noise-free measurement (1, .., 1),
    no inflation -- starts with vertices 1 and 2 (and 3) at EXACTLY equal values, and meets more ties on its way.  Optim orders
    equal values by their slot in the simplex array (sortperm!, Base's Perm ordering); so does the oracle; the kernels' physically
    sorted simplex has to (nm_cswap, csrc/nbp_device.h): in three dimensions the slots travel with the vertices, in two the search
    notices the tie and runs again with them -- this is the test that makes every 2-D search take that second run.  (Found, in its
    one natural habitat -- a stalled SE(2) search with an objective of 1e8 -- by the fuzz: profiles/r06_fuzz_ops.txt.)"""
    D = abi.MANIFOLD_DIM[manifold]
    rng = np.random.default_rng(40 + D + N)
    a = np.repeat(rng.normal(size=(N, 1)), D, axis=1)
    b = np.repeat(rng.normal(size=(N, 1)) + 2.0, D, axis=1)
    outs = []
    for sfidx in (0, 1):
        d = relative_factor_desc(abi.F_LINREL, manifold, 2, sfidx, [0, 1], 2, 500 + sfidx, [1.0] * D, [0.0] * D, inflation=0.0)
        d.skip_bandwidth = 1

        def setup(be):
            be.slot_write(0, manifold, a)
            be.slot_write(1, manifold, b)

        o, h = both(oracle_backend, hip_backend, N, 3, 0, setup, lambda be: be.run_proposals([d]), lambda be: (be.slot_read(2, manifold)[0], be.diag(reset=True)))
        assert np.array_equal(o[0], h[0]), f"sfidx {sfidx}: {int((o[0] != h[0]).any(axis=1).sum())} of {N} particles differ, by up to {np.abs(o[0] - h[0]).max():.2e}"
        assert o[1]["residual_evals"] == h[1]["residual_evals"]  # (the evaluations counted are those of the search that stands)
        outs.append(o[0])
        # the searches do find the root (x_b = x_a + 1), within Nelder-Mead's own stopping distance
        want = (a + 1.0) if sfidx == 1 else (b - 1.0)
        assert np.abs(o[0] - want).max() < 2e-3
    # ... and the ties DID decide something: a symmetric problem with its ties broken by slot does not stay on the diagonal
    assert (outs[0][:, 0] != outs[0][:, 1]).any()
