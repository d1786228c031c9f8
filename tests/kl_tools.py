"""Symmetric Kullback-Leibler divergence between two particle beliefs -- the BASELINE.md 5 parity figure:
"Symmetric KL(GPU || CPU-restatement) <= 0.05 nats per variable, on a KDE evaluated at the pooled particles".

Definition used here.  Both particle sets get a kernel density estimate with the SAME normal-reference bandwidth
(computed from the pooled particles, so that neither side's own bandwidth selector enters the figure); the
densities are evaluated at the pooled particles x_k, k = 1..2N (leave-one-out, see _log_kde), and

    symKL = 1/2 [ KL(p || q) + KL(q || p) ] = 1/2 Int (p - q) log(p / q)
          ~ 1/2 * mean_k [ (p(x_k) - q(x_k)) / m(x_k) * log(p(x_k) / q(x_k)) ],     m = (p + q) / 2,

the pooled particles being draws from the mixture m.  Every term is >= 0, so sampling noise biases the plug-in figure
UP: two independent N = 200 samples of the same density read 0.02 (1-D), 0.06 (2-D), 0.11 (3-D) -- more than the bound
itself.  The reported figure is therefore the plug-in value minus its permutation baseline (the same statistic on
random halvings of the pooled particles), i.e. the divergence in excess of what two samples of one density show;
tests/test_kl_tools.py pins it to closed forms and to ~0 for equal densities.  Circular coordinates use the wrapped
difference.
"""
import numpy as np

PI = np.pi


def wrap(a):
    return (np.asarray(a) + PI) % (2 * PI) - PI


def _coords(manifold, pts, abi):
    pts = np.asarray(pts, dtype=float)
    if manifold == abi.SE2:
        return np.stack([pts[:, 0], pts[:, 1], np.arctan2(pts[:, 3], pts[:, 2])], axis=1), [False, False, True]
    return pts, [manifold == abi.CIRCULAR] * pts.shape[1]


def _spread(x, circ):
    if circ:
        mu = np.arctan2(np.sin(x).mean(), np.cos(x).mean())
        x = wrap(x - mu)
    s = x.std()
    q75, q25 = np.percentile(x, [75, 25])
    r = (q75 - q25) / 1.349
    return min(s, r) if r > 0 else s


def _log_kde(src, at, h, circ, loo=True):
    """log of the KDE of `src` (n x D, bandwidth h per coordinate) at the points `at`.  Leave-one-out in the form that
    stays symmetric when the two particle sets (nearly) coincide: a source point closer than 1e-3 bandwidths to the
    evaluation point -- the point itself, or its twin in the other set when both solves drew the same streams -- is
    left out of the estimate, whichever set it belongs to."""
    n, D = src.shape
    e = np.zeros((at.shape[0], n))
    same = np.ones((at.shape[0], n), dtype=bool)
    for k in range(D):
        d = at[:, None, k] - src[None, :, k]
        if circ[k]:
            d = wrap(d)
        e += -0.5 * (d / h[k]) ** 2
        same &= np.abs(d) < 1e-3 * h[k]
    if loo:
        e[same] = -np.inf
    cnt = n - (same.sum(axis=1) if loo else 0)
    m = e.max(axis=1, keepdims=True)
    m = np.where(np.isfinite(m), m, 0.0)
    s = np.exp(e - m).sum(axis=1)
    norm = np.log(np.maximum(cnt, 1)) + 0.5 * D * np.log(2 * PI) + np.log(h).sum()
    return np.log(np.maximum(s, 1e-300)) + m[:, 0] - norm


def _raw_symkl(a, b, circ, h):
    pool = np.concatenate([a, b])
    lpa, lpb = _log_kde(a, pool, h, circ), _log_kde(b, pool, h, circ)
    pa, pb = np.exp(lpa), np.exp(lpb)
    m = 0.5 * (pa + pb)
    ok = m > 0
    return float(0.5 * np.mean(((pa - pb) / np.where(ok, m, 1.0) * (lpa - lpb))[ok]))


def symmetric_kl_coords(a, b, circ, splits=8, return_raw=False):
    """a, b: (N x D) tangent coordinates.  Returns the plug-in estimate minus its same-distribution floor: the mean of
    the same statistic over `splits` random halvings of the pooled particles (a permutation baseline: what two
    samples of ONE density read at this N, D and bandwidth), clipped at 0."""
    a, b = np.asarray(a, float), np.asarray(b, float)
    D = a.shape[1]
    pool = np.concatenate([a, b])
    n = 0.5 * (len(a) + len(b))
    h = np.array([max(_spread(pool[:, k] if not circ[k] else wrap(pool[:, k] - a[0, k]), False), 1e-9) for k in range(D)])
    h = h * (4.0 / (D + 2)) ** (1.0 / (D + 4)) * n ** (-1.0 / (D + 4))  # normal-reference rule
    raw = _raw_symkl(a, b, circ, h)
    if raw < 1e-12 or not splits:
        return (raw, raw, 0.0) if return_raw else raw
    rng = np.random.default_rng(20260928)
    floor = []
    for _ in range(splits):
        perm = rng.permutation(len(pool))
        floor.append(_raw_symkl(pool[perm[:len(a)]], pool[perm[len(a):]], circ, h))
    est = max(0.0, raw - float(np.mean(floor)))
    return (est, raw, float(np.mean(floor))) if return_raw else est


def symmetric_kl(abi, manifold, pts_a, pts_b):
    """host points (N x P) of two beliefs of one variable -> symmetric KL in nats"""
    a, circ = _coords(manifold, pts_a, abi)
    b, _ = _coords(manifold, pts_b, abi)
    return symmetric_kl_coords(a, b, circ)


def kl_table(abi, fa, fb):
    """{variable: symKL} between two solved graphs with the same variables"""
    return {v: symmetric_kl(abi, fa.getVariable(v).varType.manifold, fa.getVal(v), fb.getVal(v)) for v in fa.ls()}


def symmetric_kl_to_gaussian(x, mu, sigma):
    """1-D particles against an exact Gaussian N(mu, sigma^2): the same estimator with q known in closed form and
    evaluated at the particles and at an equal number of quadrature draws of q (deterministic: Gaussian quantiles)"""
    from statistics import NormalDist
    x = np.asarray(x, float).ravel()
    n = len(x)
    q = np.array([NormalDist(mu, sigma).inv_cdf((i + 0.5) / n) for i in range(n)])
    h = np.array([max(_spread(x, False), 1e-9) * 1.06 * n ** -0.2])
    pool = np.concatenate([x, q])
    lpa = _log_kde(x[:, None], pool[:, None], h, [False])
    # compare like with like: the exact density smoothed by the same kernel
    s2 = sigma ** 2 + h[0] ** 2
    lpb = -0.5 * (pool - mu) ** 2 / s2 - 0.5 * np.log(2 * PI * s2)
    pa, pb = np.exp(lpa), np.exp(lpb)
    m = 0.5 * (pa + pb)
    return float(0.5 * np.mean((pa - pb) / m * (lpa - lpb)))
