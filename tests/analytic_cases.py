"""Stream-independent known answers for every restated third-party algorithm on the path (the
un-vendored KernelDensityEstimate / ApproxManifoldProducts / Optim / Manifolds pieces), written once and run
against both backends: the CPU oracle (`tests/test_analytic_known_answers.py`) and the HIP library on the GPU
(`tests/test_gpu_analytic_known_answers.py`).

None of these compares one implementation with the other, and none depends on anybody's random streams: the
expected values are closed forms or brute-force numpy evaluations of the *definition*:

  a12  manikde! bandwidth      the fitted bandwidth maximises the leave-one-out log likelihood: compared with a
                               brute-force evaluation of that likelihood on a fine grid of bandwidths (5 manifolds)
  a13  manifoldProduct         F = 2: the exact product of two kernel density estimates is a mixture of N^2
                               Gaussians with closed-form weights, means and variances -- its mean, variance and
                               (for bimodal inputs) mode masses are enumerated exactly, wrapped on circular
                               coordinates, restricted to the informed coordinates for partial densities;
                               F = 2..8: mean / variance of the exact product density prod_j p_j(x) evaluated on a
                               dense grid
  a9/a10 per-particle solve    with a noise-free measurement the minimiser of the squared residual is the root of
                               the residual: closed form for every functor, forward and reverse
"""
import numpy as np

from parity_utils import abi, iif, relative_factor_desc

PI, TWO_PI = np.pi, 2 * np.pi


def wrap(a):
    return (np.asarray(a) + PI) % TWO_PI - PI


def circ_coords(man):
    return {abi.CIRCULAR: [0], abi.SE2: [2]}.get(man, [])


def to_points(man, c):
    """tangent coordinates (N x D) -> host points (N x P)"""
    c = np.asarray(c, dtype=float)
    if man == abi.SE2:
        th = c[:, 2]
        return np.stack([c[:, 0], c[:, 1], np.cos(th), np.sin(th), -np.sin(th), np.cos(th)], axis=1)
    if man == abi.CIRCULAR:
        return wrap(c)
    return c


def to_coords(man, p):
    if man == abi.SE2:
        return np.stack([p[:, 0], p[:, 1], np.arctan2(p[:, 3], p[:, 2])], axis=1)
    return p


# ----------------------------------------------------------------------------------------------------------
# a12: leave-one-out likelihood cross validation
# ----------------------------------------------------------------------------------------------------------
def loo_loglik(x, h, circ):
    """mean over i of log( 1/(N-1) sum_{j != i} N(x_i - x_j; 0, h^2) ): the definition, O(N^2) in numpy"""
    d = x[:, None] - x[None, :]
    if circ:
        d = wrap(d)
    K = np.exp(-0.5 * (d / h) ** 2)
    np.fill_diagonal(K, 0.0)
    s = np.maximum(K.sum(axis=1), 1e-300)
    return float(np.mean(np.log(s)) - np.log(h) - 0.5 * np.log(TWO_PI) - np.log(len(x) - 1))


def _datasets(rng, N):
    g = rng.normal(0.0, 1.0, N)
    return {
        "gauss": 2.0 * g + 5.0,
        "narrow": 0.01 * rng.normal(0, 1, N) - 1000.0,  # testBasicGraphs.jl:137-156: offsets of -1000 must not matter
        "bimodal": np.where(rng.random(N) < 0.4, rng.normal(-3.0, 0.5, N), rng.normal(2.0, 1.0, N)),
        "skewed": rng.exponential(1.5, N),
        "uniform": rng.uniform(-1.0, 4.0, N),
    }


def case_lcv_maximises_loo_likelihood(backend, N=200):
    """For every coordinate of every manifold: the likelihood at the fitted bandwidth is within 2e-3 nats per
    point of the best value on a 600-point log grid spanning four decades."""
    rng = np.random.default_rng(12)
    worst = 0.0
    for man in (abi.EUCLID1, abi.EUCLID2, abi.EUCLID3, abi.CIRCULAR, abi.SE2):
        D = abi.MANIFOLD_DIM[man]
        sets = _datasets(rng, N)
        names = list(sets)
        for trial in range(len(names)):
            cols = []
            for k in range(D):
                x = sets[names[(trial + k) % len(names)]].copy()
                if k in circ_coords(man):
                    # circular coordinate: a belief of moderate width straddling the +-pi seam
                    x = wrap(0.25 * (x - np.median(x)) / (np.std(x) + 1e-12) + 3.0)
                cols.append(x)
            c = np.stack(cols, axis=1)
            be = backend(N, 1)
            try:
                be.slot_write(0, man, to_points(man, c), np.ones(D))
                be.run_bandwidth([0], [man])
                pts, bw = be.slot_read(0, man)
            finally:
                be.close()
            assert np.abs(to_coords(man, pts) - (wrap(c) if man == abi.CIRCULAR else c)).max() < 1e-12  # fitting moves no point
            for k in range(D):
                x, circ = c[:, k], k in circ_coords(man)
                span = TWO_PI if circ else x.max() - x.min()
                grid = np.exp(np.linspace(np.log(span * 1e-4), np.log(span), 600))
                ll = np.array([loo_loglik(x, h, circ) for h in grid])
                got = loo_loglik(x, bw[k], circ)
                worst = max(worst, ll.max() - got)
                assert got >= ll.max() - 2e-3, (man, names[(trial + k) % len(names)], k, bw[k], grid[ll.argmax()], ll.max() - got)
    return worst


# ----------------------------------------------------------------------------------------------------------
# a13: products of kernel density estimates
# ----------------------------------------------------------------------------------------------------------
def exact_product_of_two(man, a, ha, b, hb, mask_a=None, mask_b=None):
    """The exact product of two KDEs (diagonal Gaussian kernels on tangent coordinates, wrapped differences on
    circular ones): component (i, j) has weight prod_k N(a_ik - b_jk; 0, ha_k^2 + hb_k^2) over the coordinates
    BOTH densities inform, mean = precision-weighted mean (along the shorter arc on a circle), variance
    1 / (1/ha^2 + 1/hb^2).  Returns the component arrays: w (N, N), mean (N, N, D), var (D)."""
    D = a.shape[1]
    mask_a = mask_a if mask_a else (1 << D) - 1
    mask_b = mask_b if mask_b else (1 << D) - 1
    logw = np.zeros((a.shape[0], b.shape[0]))
    mean = np.zeros((a.shape[0], b.shape[0], D))
    var = np.zeros(D)
    for k in range(D):
        ina, inb = (mask_a >> k) & 1, (mask_b >> k) & 1
        d = b[None, :, k] - a[:, None, k]
        if k in circ_coords(man):
            d = wrap(d)
        if ina and inb:
            s2 = ha[k] ** 2 + hb[k] ** 2
            logw += -0.5 * d * d / s2 - 0.5 * np.log(s2)
            pa, pb = 1 / ha[k] ** 2, 1 / hb[k] ** 2
            mean[:, :, k] = a[:, None, k] + d * pb / (pa + pb)
            var[k] = 1 / (pa + pb)
        elif ina:
            mean[:, :, k] = a[:, None, k] + 0 * d
            var[k] = ha[k] ** 2
        elif inb:
            mean[:, :, k] = b[None, :, k] + 0 * d
            var[k] = hb[k] ** 2
        else:
            mean[:, :, k] = np.nan  # nobody informs this coordinate: the old points stay
    w = np.exp(logw - logw.max())
    return w / w.sum(), mean, var


def _mixture_moments(man, w, mean, var, k):
    m = mean[:, :, k]
    if k in circ_coords(man):
        mu = np.arctan2((w * np.sin(m)).sum(), (w * np.cos(m)).sum())
        dev = wrap(m - mu)
        return mu, float((w * (dev * dev + var[k])).sum())
    mu = float((w * m).sum())
    return mu, float((w * ((m - mu) ** 2 + var[k])).sum())


def _sample_moments(man, c, k):
    x = c[:, k]
    if k in circ_coords(man):
        mu = np.arctan2(np.sin(x).mean(), np.cos(x).mean())
        return mu, float((wrap(x - mu) ** 2).mean())
    return float(x.mean()), float(x.var())


def _fit_and_multiply(backend, man, dens, seeds, partials=None, old=None, N=None, niter=1):
    """write the densities, fit their bandwidths (manikde!), run one product per seed; returns
    (bandwidths per density, [output coordinates per seed])"""
    F = len(dens)
    N = N or dens[0].shape[0]
    D = abi.MANIFOLD_DIM[man]
    be = backend(N, F + 2)
    try:
        for j, c in enumerate(dens):
            be.slot_write(j, man, to_points(man, c), np.ones(D))
        be.run_bandwidth(list(range(F)), [man] * F)
        bws = [be.slot_read(j, man)[1] for j in range(F)]
        old_slot = -1
        if old is not None:
            be.slot_write(F + 1, man, to_points(man, old), np.ones(D))
            old_slot = F + 1
        outs = []
        for s in seeds:
            be.run_products([iif.solver.product_desc(man, list(range(F)), F, s, niter, -1, partials, old_slot)])
            outs.append(to_coords(man, be.slot_read(F, man)[0]))
    finally:
        be.close()
    return bws, outs


def _gauss_cloud(rng, N, mu, sig, man):
    """N points with EXACT sample mean `mu` and standard deviation `sig` per coordinate (so that the only Monte-Carlo
    error left in a comparison is the product sampler's own)"""
    D = len(mu)
    x = rng.normal(size=(N, D))
    x = (x - x.mean(axis=0)) / x.std(axis=0)
    c = x * np.asarray(sig) + np.asarray(mu)
    for k in circ_coords(man):
        c[:, k] = wrap(c[:, k])
    return c


PRODUCT2_CASES = [
    # manifold, (mu_a, sig_a), (mu_b, sig_b)
    (abi.EUCLID1, ([-0.5], [1.0]), ([0.7], [0.6])),
    (abi.EUCLID2, ([0.0, 10.0], [1.0, 0.2]), ([1.0, 10.3], [0.5, 0.4])),
    (abi.EUCLID3, ([0.0, 1.0, -2.0], [1.0, 0.5, 2.0]), ([0.5, 1.2, -1.0], [1.0, 1.0, 1.0])),
    (abi.CIRCULAR, ([3.0], [0.25]), ([-3.05], [0.3])),      # the two beliefs sit on either side of the +-pi seam
    (abi.SE2, ([2.0, -1.0, 3.0], [0.3, 0.3, 0.2]), ([2.2, -1.1, -3.1], [0.4, 0.2, 0.25])),
]


def case_product_of_two_matches_exact_mixture(backend, N=128, nseeds=24):
    """Product of two KDEs vs the exact N^2-component mixture, every manifold: the mean over seeds of the sample
    mean within 0.06 sigma of the exact mean, the mean sample variance within [0.8, 1.25] of the exact variance."""
    rng = np.random.default_rng(5)
    report = []
    for man, (mua, sa), (mub, sb) in PRODUCT2_CASES:
        D = abi.MANIFOLD_DIM[man]
        a, b = _gauss_cloud(rng, N, mua, sa, man), _gauss_cloud(rng, N, mub, sb, man)
        (ha, hb), outs = _fit_and_multiply(backend, man, [a, b], [7000 + s for s in range(nseeds)])
        w, mean, var = exact_product_of_two(man, a, ha, b, hb)
        for k in range(D):
            mu, v = _mixture_moments(man, w, mean, var, k)
            sm = [_sample_moments(man, o, k) for o in outs]
            dm = np.array([m for m, _ in sm]) - mu
            if k in circ_coords(man):
                dm = wrap(dm)
            ratio = np.mean([s for _, s in sm]) / v
            report.append((man, k, dm.mean() / np.sqrt(v), ratio))
            assert abs(dm.mean()) < 0.06 * np.sqrt(v) + 3 * dm.std() / np.sqrt(nseeds), (man, k, dm.mean(), np.sqrt(v))
            assert 0.8 < ratio < 1.25, (man, k, ratio)
    return report


def case_partial_product_matches_exact_mixture(backend, N=128, nseeds=24):
    """Partial densities (AMP.marginal(propBel, pardims), ApproxConv.jl:287-291): a density multiplies in on its own
    coordinates only; a coordinate informed by one density is that density's marginal; a coordinate nobody informs
    keeps the old points (GraphProductOperations.jl:39-45)."""
    rng = np.random.default_rng(6)
    for man, mask_a, mask_b in ((abi.EUCLID2, 0, 1), (abi.EUCLID3, 0, 5), (abi.SE2, 3, 4), (abi.SE2, 0, 4), (abi.EUCLID3, 1, 2)):
        D = abi.MANIFOLD_DIM[man]
        mua = [0.5, -1.0, 2.9][:D]
        mub = [1.0, -0.5, -3.0][:D]
        a, b = _gauss_cloud(rng, N, mua, [0.5] * D, man), _gauss_cloud(rng, N, mub, [0.4] * D, man)
        old = _gauss_cloud(rng, N, [7.0, 8.0, 1.0][:D], [0.1] * D, man)
        (ha, hb), outs = _fit_and_multiply(backend, man, [a, b], [8000 + s for s in range(nseeds)], partials=[mask_a, mask_b], old=old)
        w, mean, var = exact_product_of_two(man, a, ha, b, hb, mask_a, mask_b)
        full = (1 << D) - 1
        for k in range(D):
            informed = (((mask_a or full) | (mask_b or full)) >> k) & 1
            if not informed:
                for o in outs:  # untouched, particle by particle
                    assert np.abs(wrap(o[:, k] - old[:, k]) if k in circ_coords(man) else o[:, k] - old[:, k]).max() < 1e-12, (man, k)
                continue
            mu, v = _mixture_moments(man, w, mean, var, k)
            sm = [_sample_moments(man, o, k) for o in outs]
            dm = np.array([m for m, _ in sm]) - mu
            if k in circ_coords(man):
                dm = wrap(dm)
            ratio = np.mean([s for _, s in sm]) / v
            assert abs(dm.mean()) < 0.08 * np.sqrt(v) + 3 * dm.std() / np.sqrt(nseeds), (man, mask_a, mask_b, k, dm.mean(), np.sqrt(v))
            assert 0.75 < ratio < 1.3, (man, mask_a, mask_b, k, ratio)


def case_bimodal_mode_masses(backend, N=128, nseeds=40):
    """Mode masses of a bimodal product at Niter = 1 against the exact mixture: a two-mode density (weights w / 1-w
    at -2 / +2) times a broad unimodal one centred at `c`.  One Gibbs sweep per level does not reach the stationary
    label distribution: the label of the bimodal density is settled on the coarse levels, where every node is its
    moment-matched Gaussian and the broad density hardly discriminates between the modes -- with the broad density off
    centre (c = 0.8) the exact left mass 0.36 comes out 0.51, the bimodal density's own weight (the sampler of rounds 1-3,
    which handed a label down to a random child instead of drawing it on the point of the level above, read 0.27).  The
    reference's own multihypo test accepts ~33 % where 50 % is exact (testSpecialEuclidean2Mani.jl:628-629).
    Asserted: within 0.16 of the exact mass, and no flip of the dominant mode where the exact masses differ by > 0.3."""
    rng = np.random.default_rng(7)
    report = []
    for wleft, c in ((0.5, 0.0), (0.7, 0.0), (0.3, 0.0), (0.5, 0.8), (0.9, 0.0)):
        nl = int(round(wleft * N))
        a = np.concatenate([rng.normal(-2.0, 0.3, nl), rng.normal(2.0, 0.3, N - nl)])[:, None]
        a = a[rng.permutation(N)]
        b = _gauss_cloud(rng, N, [c], [2.0], abi.EUCLID1)
        (ha, hb), outs = _fit_and_multiply(backend, abi.EUCLID1, [a, b], [9000 + s for s in range(nseeds)])
        w, mean, var = exact_product_of_two(abi.EUCLID1, a, ha, b, hb)
        # mass left of 0 of the exact mixture: sum_ij w_ij Phi(-m_ij / sd)
        from math import erf, sqrt
        Phi = np.vectorize(lambda t: 0.5 * (1 + erf(t / sqrt(2))))
        exact = float((w * Phi(-mean[:, :, 0] / np.sqrt(var[0]))).sum())
        got = float(np.mean([(o[:, 0] < 0).mean() for o in outs]))
        report.append((wleft, c, exact, got))
        assert abs(got - exact) < 0.16, (wleft, c, exact, got)
        if abs(exact - 0.5) > 0.15:
            assert (got > 0.5) == (exact > 0.5), (wleft, c, exact, got)
    return report


def exact_product_moments_on_grid(man, dens, bws, G=None):
    """Mean and variance per coordinate of the EXACT product prod_j p_j(x) of F kernel density estimates, by direct
    evaluation on a dense D-dimensional grid (diagonal Gaussian kernels on tangent coordinates, wrapped distance on
    circular ones): p_j on the grid is sum_i prod_k K(x_k - a_ik; h_jk), an outer-product sum per density."""
    F, (N, D) = len(dens), dens[0].shape
    G = G or {1: 4001, 2: 321, 3: 97}[D]
    # grid window: precision-weighted Gaussian guess, +- 7 of its standard deviations (checked below: no mass at the edge)
    axes = []
    for k in range(D):
        circ = k in circ_coords(man)
        ref = dens[0][0, k]
        prec = np.array([1.0 / (np.var(wrap(d[:, k] - ref) if circ else d[:, k]) + h[k] ** 2) for d, h in zip(dens, bws)])
        cen = np.array([np.mean(wrap(d[:, k] - ref)) if circ else np.mean(d[:, k]) - ref for d in dens])
        v = 1.0 / prec.sum()
        mu = ref + (prec * cen).sum() * v
        half = 7.0 * np.sqrt(v) + 3.0 * max(h[k] for h in bws)
        axes.append(np.linspace(mu - half, mu + half, G))
    logp = np.zeros([G] * D)
    for d, h in zip(dens, bws):
        Ks = []
        for k in range(D):
            diff = axes[k][:, None] - d[None, :, k]
            if k in circ_coords(man):
                diff = wrap(diff)
            Ks.append(np.exp(-0.5 * (diff / h[k]) ** 2))
        if D == 1:
            pj = Ks[0].sum(axis=1)
        elif D == 2:
            pj = Ks[0] @ Ks[1].T
        else:
            pj = np.einsum("ai,bi,ci->abc", Ks[0], Ks[1], Ks[2], optimize=True)
        logp += np.log(np.maximum(pj, 1e-300))
    p = np.exp(logp - logp.max())
    p /= p.sum()
    out = []
    for k in range(D):
        marg = p.sum(axis=tuple(q for q in range(D) if q != k))
        assert marg[0] + marg[-1] < 1e-6 * marg.max() + 1e-12, "grid window too narrow"
        mu = float((marg * axes[k]).sum())
        out.append((wrap(mu) if k in circ_coords(man) else mu, float((marg * (axes[k] - mu) ** 2).sum())))
    return out


def case_product_of_many_densities(backend, N=128, nseeds=16, check=True):
    """F = 2 ... 8 densities on every manifold against the exact product (dense-grid evaluation of prod_j p_j(x)).
    Two settings of the number of Gibbs sweeps per level:
      Niter = 6  the sampler must reproduce the exact product (its stationary distribution): mean over seeds of the
                 sample mean within 0.1 sigma (+ its own Monte-Carlo error), mean sample variance within [0.8, 1.2];
      Niter = 1  what the reference runs (GraphProductOperations.jl:53-60): one sweep per level leaves the labels
                 under-mixed -- the product comes out ~9 % too wide on average and up to 1.9x for eight 3-D densities --
                 so the band is [0.7, 2.0] and 0.25 sigma.
    (The exact product of F bumpy N-point KDEs is NOT the product of the moment-matched Gaussians; the sampler is
    tested against the former.)"""
    rng = np.random.default_rng(8)
    report = []
    for man in (abi.EUCLID1, abi.EUCLID2, abi.EUCLID3, abi.CIRCULAR, abi.SE2):
        D = abi.MANIFOLD_DIM[man]
        for F in (2, 3, 4, 5, 6, 8):
            mus = rng.uniform(-0.4, 0.4, (F, D))
            sig = rng.uniform(0.5, 1.5, (F, D))
            for k in circ_coords(man):
                mus[:, k] = rng.uniform(-0.1, 0.1, F) + 3.1  # around the seam
                sig[:, k] = rng.uniform(0.15, 0.3, F)
            dens = [_gauss_cloud(rng, N, mus[j], sig[j], man) for j in range(F)]
            exact = None
            for niter, mtol, lo, hi in ((6, 0.1, 0.8, 1.2), (1, 0.25, 0.7, 2.0)):
                bws, outs = _fit_and_multiply(backend, man, dens, [10000 + 100 * F + s for s in range(nseeds)], niter=niter)
                exact = exact or exact_product_moments_on_grid(man, dens, bws)
                for k in range(D):
                    mu, v = exact[k]
                    sm = [_sample_moments(man, o, k) for o in outs]
                    dm = np.array([m for m, _ in sm]) - mu
                    if k in circ_coords(man):
                        dm = wrap(dm)
                    ratio = np.mean([s for _, s in sm]) / v
                    report.append((man, F, niter, k, round(dm.mean() / np.sqrt(v), 3), round(ratio, 3)))
                    if check:
                        assert abs(dm.mean()) < mtol * np.sqrt(v) + 3 * dm.std() / np.sqrt(nseeds), (man, F, niter, k, dm.mean(), np.sqrt(v))
                        assert lo < ratio < hi, (man, F, niter, k, ratio)
    return report


# ----------------------------------------------------------------------------------------------------------
# a9 / a10: per-particle minimisation of the squared residual
# ----------------------------------------------------------------------------------------------------------
def _rot(th):
    return np.cos(th), np.sin(th)


def root_of(kind, man, z, other, solve_b):
    """closed-form root of the residual functor: the point x with r(z, a, b) = 0 where (a, b) = (other, x) when
    solving for the second variable and (x, other) when solving for the first"""
    z, o = np.asarray(z, float), np.asarray(other, float)
    if kind == abi.F_LINREL:  # r = z - (b - a)
        return o + z if solve_b else o - z
    if kind == abi.F_CIRCULAR:  # r = wrap(a + z - b)
        return wrap(o + z) if solve_b else wrap(o - z)
    if kind == abi.F_SE2:  # b = a o (z_t, z_theta)
        if solve_b:
            c, s = _rot(o[:, 2])
            return np.stack([o[:, 0] + c * z[:, 0] - s * z[:, 1], o[:, 1] + s * z[:, 0] + c * z[:, 1], wrap(o[:, 2] + z[:, 2])], axis=1)
        th = wrap(o[:, 2] - z[:, 2])
        c, s = _rot(th)
        return np.stack([o[:, 0] - (c * z[:, 0] - s * z[:, 1]), o[:, 1] - (s * z[:, 0] + c * z[:, 1]), th], axis=1)
    raise ValueError(kind)


ROOT_CASES = [
    # factor kind, manifold, noise-free measurement
    (abi.F_LINREL, abi.EUCLID1, [1.7]),
    (abi.F_LINREL, abi.EUCLID2, [10.0, -3.0]),
    (abi.F_LINREL, abi.EUCLID3, [0.5, 2.0, -7.0]),
    (abi.F_CIRCULAR, abi.CIRCULAR, [2.5]),
    (abi.F_SE2, abi.SE2, [1.0, 0.5, 0.8]),
    (abi.F_SE2, abi.SE2, [10.0, -4.0, -2.9]),
]


def case_solver_finds_the_residual_root(backend, N=128):
    """NelderMead (D >= 2) and BFGS (D = 1) against the closed-form root, forward (solve the second variable) and
    reverse (solve the first), with a noise-free measurement.  Tolerance: Optim stops NelderMead when the spread of
    the objective over the simplex is <= 1e-8, i.e. at residuals of ~1e-4, BFGS at |gradient| <= 1e-8."""
    rng = np.random.default_rng(9)
    worst = {}
    for kind, man, z in ROOT_CASES:
        D = abi.MANIFOLD_DIM[man]
        o = rng.normal(0.0, 2.0, (N, D))
        x0 = rng.normal(0.0, 2.0, (N, D))  # the target's current belief: only the start of the search
        for k in circ_coords(man):
            o[:, k], x0[:, k] = wrap(o[:, k]), wrap(x0[:, k])
        for solve_b in (1, 0):
            be = backend(N, 3)
            try:
                slots = [0, 1]
                be.slot_write(0, man, to_points(man, o if solve_b else x0), np.ones(D))
                be.slot_write(1, man, to_points(man, x0 if solve_b else o), np.ones(D))
                d = relative_factor_desc(kind, man, 2, 1 if solve_b else 0, slots, 2, 1234 + solve_b, z, [0.0] * len(z))
                d.skip_bandwidth = 1
                be.run_proposals([d])
                got = to_coords(man, be.slot_read(2, man)[0])
            finally:
                be.close()
            want = root_of(kind, man, np.tile(z, (N, 1)), o, solve_b)
            err = got - want
            for k in circ_coords(man):
                err[:, k] = wrap(err[:, k])
            e = np.abs(err).max()
            worst[(kind, man, solve_b)] = e
            assert e < (1e-6 if D == 1 else 1.5e-3), (kind, man, solve_b, e)
            assert np.sqrt((err ** 2).mean()) < (1e-6 if D == 1 else 3e-4), (kind, man, solve_b)
    return worst


def case_euclid_distance_ring(backend, N=128):
    """EuclidDistance r = z - ||b - a||: the solutions form a ring; every solved particle sits on it"""
    rng = np.random.default_rng(10)
    out = {}
    for man in (abi.EUCLID2, abi.EUCLID3):
        D = abi.MANIFOLD_DIM[man]
        o = rng.normal(0.0, 1.0, (N, D))
        x0 = o + rng.normal(0.0, 3.0, (N, D))
        for solve_b in (1, 0):
            be = backend(N, 3)
            try:
                be.slot_write(0, man, o if solve_b else x0, np.ones(D))
                be.slot_write(1, man, x0 if solve_b else o, np.ones(D))
                d = relative_factor_desc(abi.F_EUCLIDDIST, man, 2, 1 if solve_b else 0, [0, 1], 2, 4321 + solve_b, [5.0], [0.0])
                d.skip_bandwidth = 1
                be.run_proposals([d])
                got = be.slot_read(2, man)[0]
            finally:
                be.close()
            r = np.abs(np.linalg.norm(got - o, axis=1) - 5.0)
            out[(man, solve_b)] = r.max()
            assert np.median(r) < 2e-4 and (r < 2e-3).mean() > 0.97, (man, solve_b, np.median(r), r.max())
    return out


def case_partial_relative_over_two_coordinates(backend, N=128):
    """A `.partial` relative factor over two of three coordinates (BFGS on the pair: NumericalCalculations.jl:108,424):
    the partial coordinates land on the root of the residual, the third keeps the target's own value bit for bit (it gets
    neither entropy nor a solve, EvalFactor.jl:184-198)."""
    rng = np.random.default_rng(11)
    man = abi.EUCLID3
    out = {}
    for mask, coords in ((0b101, (0, 2)), (0b011, (0, 1)), (0b110, (1, 2))):
        o = rng.normal(0.0, 2.0, (N, 3))
        x0 = rng.normal(0.0, 2.0, (N, 3))
        z = [4.0, -1.5]
        for solve_b in (1, 0):
            be = backend(N, 3)
            try:
                be.slot_write(0, man, o if solve_b else x0, np.ones(3))
                be.slot_write(1, man, x0 if solve_b else o, np.ones(3))
                d = relative_factor_desc(abi.F_LINREL, man, 2, 1 if solve_b else 0, [0, 1], 2, 555 + solve_b, z, [0.0, 0.0])
                d.partial_mask, d.skip_bandwidth = mask, 1
                be.run_proposals([d])
                got, _, ipc = be.belief_read(2, man)
            finally:
                be.close()
            sign = 1.0 if solve_b else -1.0
            for i, k in enumerate(coords):
                err = np.abs(got[:, k] - (o[:, k] + sign * z[i])).max()
                out[(mask, solve_b, k)] = err
                assert err < 1e-6, (mask, solve_b, k, err)
            free = [k for k in range(3) if k not in coords][0]
            assert np.array_equal(got[:, free], x0[:, free])
            assert list(ipc) == [float((mask >> k) & 1) for k in range(3)]  # ones on `.partial`, EvalFactor.jl:383-391
    return out


# ----------------------------------------------------------------------------------------------------------
# a13, pinned by ENUMERATION: the labels a product draws, against the exact weights of every label tuple
# ----------------------------------------------------------------------------------------------------------
def exact_label_weights(dens, bws):
    """The exact product of F kernel density estimates with diagonal Gaussian kernels is a mixture with one component per
    label tuple (i_1 .. i_F); its weight is the integral of prod_j N(x; a_j[i_j], h_j^2) -- per coordinate
    exp(-0.5 (sum_j r_j m_j^2 - (sum_j r_j m_j)^2 / sum_j r_j)), r_j = 1 / h_j^2, times a constant the tuples share.
    Returns the normalised weights as an array of shape (N,) * F."""
    F, (N, D) = len(dens), dens[0].shape
    logw = np.zeros((N,) * F)
    for k in range(D):
        r = np.array([1.0 / h[k] ** 2 for h in bws])
        s2, s1 = np.zeros((N,) * F), np.zeros((N,) * F)
        for j in range(F):
            shape = [1] * F
            shape[j] = N
            m = dens[j][:, k].reshape(shape)
            s2 = s2 + r[j] * m * m
            s1 = s1 + r[j] * m
        logw += -0.5 * (s2 - s1 * s1 / r.sum())
    w = np.exp(logw - logw.max())
    return w / w.sum()


def case_product_labels_match_enumeration(backend, N=8, nprod=4000):
    """Small mixtures, every label tuple enumerated.  The samples of a product are independent draws (one chain and one
    random stream per sample), so the counts of the label tuples over nprod x N samples are multinomial and Pearson's
    statistic is chi-square distributed IF the sampler draws from the exact product: the bound is the 1 - 1e-6 quantile of
    that distribution -- a tolerance that comes from Monte-Carlo error alone, not from what was observed.  Held for the
    converged sampler (Niter = 8 sweeps per level, the ABI's maximum); the reference's Niter = 1 is REPORTED beside it:
    its excess over the same bound is the under-mixing the reference accepts (GraphProductOperations.jl:53-60), measured
    rather than folded into a tolerance.  Also per sample moments: mean and variance of the output against the exact
    mixture's, within 5 standard errors."""
    from scipy.stats import chi2
    rng = np.random.default_rng(77)
    report = []
    for man, F, shape in ((abi.EUCLID1, 2, "overlapping"), (abi.EUCLID1, 3, "overlapping"), (abi.EUCLID2, 2, "overlapping"), (abi.EUCLID1, 2, "doors")):
        D = abi.MANIFOLD_DIM[man]
        if shape == "overlapping":
            dens = [rng.normal(0.3 * j, 1.0, size=(N, D)) for j in range(F)]
            bws = [np.full(D, 0.6 + 0.1 * j) for j in range(F)]
        else:
            # the scenario of config 3's sightings, in small (N = 16): an odometry-propagated belief with an eighth of its
            # particles one door spacing off, times a sighting with four equally likely doors -- the case in which rounds 1-3's
            # label hand-down tripled the minority mode (profiles/r04_sampler_variants.txt)
            N = 16
            dens = [np.concatenate([rng.normal(0.0, 0.1, 14), rng.normal(1.6, 0.1, 2)])[:, None],
                    np.concatenate([rng.normal(c, 0.1, 4) for c in (-1.6, 0.0, 1.6, 3.2)])[:, None]]
            dens = [d[rng.permutation(N)] for d in dens]
            bws = [np.full(1, 0.15), np.full(1, 0.12)]
        w = exact_label_weights(dens, bws)
        # exact moments of the mixture (per coordinate): component mean = precision-weighted mean of the selected kernels
        rr = np.array([[1.0 / h[k] ** 2 for h in bws] for k in range(D)])  # D x F
        for niter in (8, 1):
            be = backend(N, F + nprod, N * F * nprod)
            try:
                for j in range(F):
                    be.slot_write(j, man, to_points(man, dens[j]), bws[j])
                descs = [iif.solver.product_desc(man, list(range(F)), F + i, 5000 + i, niter, i * N * F) for i in range(nprod)]
                be.run_products(descs)
                lab = np.concatenate([be.side_read(i * N * F, N * F).reshape(N, F) for i in range(nprod)])
                pts = np.concatenate([to_coords(man, be.slot_read(F + i, man)[0]) for i in range(0, nprod, 8)])
            finally:
                be.close()
            n = lab.shape[0]
            counts = np.zeros((N,) * F)
            np.add.at(counts, tuple(lab[:, j] for j in range(F)), 1)
            e = (w * n).ravel()
            c = counts.ravel()
            big = e >= 5.0
            ee, cc = np.append(e[big], e[~big].sum()), np.append(c[big], c[~big].sum())
            if ee[-1] < 5.0:  # the pooled small cells join the smallest big one
                ee[-2] += ee[-1]; cc[-2] += cc[-1]; ee, cc = ee[:-1], cc[:-1]
            stat, dof = float(((cc - ee) ** 2 / ee).sum()), len(ee) - 1
            bound = float(chi2.ppf(1 - 1e-6, dof))
            # moments
            zs = []
            for k in range(D):
                s1 = sum(rr[k, j] * dens[j][:, k].reshape([N if q == j else 1 for q in range(F)]) for j in range(F))
                mean_c = s1 / rr[k].sum()
                mu = float((w * mean_c).sum())
                var = float((w * ((mean_c - mu) ** 2 + 1.0 / rr[k].sum())).sum())
                x = pts[:, k]
                se_m = np.sqrt(var / len(x))
                m4 = float((w * (3 * (1.0 / rr[k].sum()) ** 2 + 6 * (1.0 / rr[k].sum()) * (mean_c - mu) ** 2 + (mean_c - mu) ** 4)).sum())
                se_v = np.sqrt(max(m4 - var * var, 1e-300) / len(x))
                zs.append((float((x.mean() - mu) / se_m), float((((x - mu) ** 2).mean() - var) / se_v)))
            extra = ()
            if shape == "doors":  # mass of the minority mode (the off-by-one-door particles): exact against drawn
                minority = dens[0][:, 0] > 0.8
                extra = (round(float(w[minority, :].sum()), 4), round(float(minority[lab[:, 0]].mean()), 4))
            report.append((man, F, shape, niter, n, dof, round(stat, 1), round(bound, 1), [(round(a, 2), round(b, 2)) for a, b in zs]) + extra)
            if niter == 8 and shape == "overlapping":
                assert stat < bound, (man, F, "label tuples of the converged sampler against the enumerated product", stat, bound, dof)
                for zm, zv in zs:
                    assert abs(zm) < 5 and abs(zv) < 5, (man, F, zs)
            if shape == "doors":
                # Separated modes: the sweeps cannot carry a sample from one mode to another (both labels would have to change
                # at once), so the MASS of a mode is whatever the coarse levels -- moment-matched Gaussians -- gave it: the
                # published algorithm's approximation, not Monte-Carlo error (exact 0.139; 0.064 with eight sweeps per level,
                # 0.096 with the reference's one: more sweeps equilibrate to the COARSE level's distribution).  Held: within
                # 0.1 of exact and not amplified (rounds 1-3's hand-down: 0.33).  WITHIN the majority mode the sampler is
                # exact again: the label tuples there against the enumerated conditional weights, chi-square as above.
                exact_min, got_min = extra
                assert abs(got_min - exact_min) < 0.1 and got_min < exact_min + 4 * np.sqrt(exact_min / n), (niter, exact_min, got_min)
                inmode = (~minority)[:, None] & (np.abs(dens[1][:, 0]) < 0.8)[None, :]
                sel = inmode[lab[:, 0], lab[:, 1]]
                wc = np.where(inmode, w, 0.0)
                wc /= wc.sum()
                cnt = np.zeros_like(w)
                np.add.at(cnt, (lab[sel, 0], lab[sel, 1]), 1)
                e2, c2 = (wc * sel.sum())[inmode], cnt[inmode]
                big2 = e2 >= 5.0
                e3, c3 = np.append(e2[big2], e2[~big2].sum()), np.append(c2[big2], c2[~big2].sum())
                if e3[-1] < 5.0:
                    e3, c3 = e3[:-1], c3[:-1]
                stat2, dof2 = float(((c3 - e3) ** 2 / e3).sum()), len(e3) - 1
                report[-1] = report[-1] + (("within the majority mode", int(sel.sum()), dof2, round(stat2, 1), round(float(chi2.ppf(1 - 1e-6, dof2)), 1)),)
                if niter == 8:
                    assert stat2 < chi2.ppf(1 - 1e-6, dof2), ("label tuples within the majority mode", stat2, dof2)
    return report
