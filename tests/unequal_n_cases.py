"""Beliefs whose particle count differs from the solver's N (row "+" of the scope table: "multinomial resampling for
N-particle beliefs", `_getindex_anyn`), written once and run on the oracle (tests/test_unequal_particle_counts.py) and
on the GPU (tests/test_gpu_unequal_particle_counts.py).  What the reference does with such a belief:

  propagateBelief          oldPoints = the belief's points topped up with sample(oldBel, N - Npts)   GraphProductOperations.jl:37-45
                           (more than N points: the first N)
  _beforeSolveCCW!         the scratch copy of the target is resized to N, new entries = point default  CalcFactor.jl:555-565
  CalcFactorNormSq         an operand shorter than the particle index is read at a random element     NumericalCalculations.jl:377-381
  MsgPrior / KDE sampling  samples among the points the density has                                     Factors/MsgPrior.jl:27-30
  manikde!                 fits the points there are
"""
import numpy as np

from analytic_cases import loo_loglik
from parity_utils import abi, iif, relative_factor_desc

N = 128


def case_count_round_trip(backend):
    rng = np.random.default_rng(1)
    be = backend(N, 3)
    try:
        for man, n in ((abi.EUCLID2, 50), (abi.SE2, 31), (abi.CIRCULAR, N), (abi.EUCLID3, N + 40)):
            D, P = abi.MANIFOLD_DIM[man], abi.MANIFOLD_P[man]
            c = rng.normal(size=(n, D))
            pts = c if man != abi.SE2 else np.stack([c[:, 0], c[:, 1], np.cos(c[:, 2]), np.sin(c[:, 2]), -np.sin(c[:, 2]), np.cos(c[:, 2])], axis=1)
            if man == abi.CIRCULAR:
                pts = (pts + np.pi) % (2 * np.pi) - np.pi
            be.belief_write(0, man, pts, np.full(D, 0.3), np.arange(1.0, D + 1))
            got, bw, ipc = be.belief_read(0, man)
            assert got.shape == (min(n, N), P)  # more than N points: the first N are kept
            np.testing.assert_allclose(got, pts[:min(n, N)], atol=1e-12)
            np.testing.assert_allclose(bw, 0.3)
            np.testing.assert_allclose(ipc, np.arange(1.0, D + 1))
    finally:
        be.close()


def case_shorter_operand_is_read_at_a_random_element(backend):
    """x_b = x_a + z with a noise-free z and an `a` that holds 40 points: particle n < 40 lands on a_n + z, every other
    particle on a_i + z for some i < 40 -- and not always the same i"""
    rng = np.random.default_rng(2)
    cnt = 40
    a = rng.normal(0.0, 3.0, (cnt, 2))
    b = rng.normal(0.0, 1.0, (N, 2))
    z = [5.0, -2.0]
    be = backend(N, 3)
    try:
        be.belief_write(0, abi.EUCLID2, a, np.full(2, 0.2))
        be.slot_write(1, abi.EUCLID2, b, np.ones(2))
        d = relative_factor_desc(abi.F_LINREL, abi.EUCLID2, 2, 1, [0, 1], 2, 777, z, [0.0, 0.0])
        d.skip_bandwidth = 1
        be.run_proposals([d])
        out = be.slot_read(2, abi.EUCLID2)[0]
    finally:
        be.close()
    src = out - np.array(z)
    assert np.abs(src[:cnt] - a).max() < 2e-3
    dist = np.linalg.norm(src[cnt:, None, :] - a[None, :, :], axis=2)
    assert dist.min(axis=1).max() < 2e-3                       # each one sits on some element of `a`
    assert len(set(dist.argmin(axis=1))) > 10                  # ... drawn at random, not a fixed one
    return out


def case_shorter_target_is_filled_with_the_point_default(backend):
    """solving for a variable that holds 20 points: all N particles are produced (the search of the new ones starts at
    the identity), each at other_n - z"""
    rng = np.random.default_rng(3)
    a = rng.normal(0.0, 1.0, (20, 1))
    b = rng.normal(10.0, 2.0, (N, 1))
    be = backend(N, 3)
    try:
        be.belief_write(0, abi.EUCLID1, a, np.full(1, 0.2))
        be.slot_write(1, abi.EUCLID1, b, np.ones(1))
        d = relative_factor_desc(abi.F_LINREL, abi.EUCLID1, 2, 0, [0, 1], 2, 778, [3.0], [0.0])
        d.skip_bandwidth = 1
        be.run_proposals([d])
        out, _, _ = be.belief_read(2, abi.EUCLID1)
        kept, _, _ = be.belief_read(0, abi.EUCLID1)
    finally:
        be.close()
    assert out.shape == (N, 1) and np.abs(out - (b - 3.0)).max() < 1e-6
    assert kept.shape == (20, 1) and np.abs(kept - a).max() == 0.0  # the stored belief is untouched and keeps its count
    return out


def case_message_with_fewer_points(backend):
    """a MsgPrior samples the KDE of a message that holds 25 points: every draw is one of those points plus bandwidth noise"""
    rng = np.random.default_rng(4)
    msg = np.concatenate([rng.normal(-4.0, 0.05, (12, 2)), rng.normal(6.0, 0.05, (13, 2))])
    be = backend(N, 3)
    try:
        be.slot_write(0, abi.EUCLID2, np.zeros((N, 2)), np.ones(2))
        be.belief_write(1, abi.EUCLID2, msg, np.array([0.1, 0.1]))
        d = abi.ProposalDesc()
        d.factor_kind, d.manifold, d.nvars, d.sfidx = abi.F_MSGPRIOR, abi.EUCLID2, 1, 0
        d.var_slot[0], d.var_slot[1], d.out_slot, d.ncomp = 0, 1, 2, 1
        d.comp[0][0] = 1.0
        d.mhidx_in = d.mhidx_out = -1
        d.inflate_cycles, d.inflation, d.spread_nh, d.seed = 3, 5.0, 3.0, 99
        be.run_proposals([d])
        out, bw = be.slot_read(2, abi.EUCLID2)
    finally:
        be.close()
    near = np.linalg.norm(out[:, None, :] - msg[None, :, :], axis=2).min(axis=1)
    assert near.max() < 0.6 and (bw > 0).all()
    left = (out[:, 0] < 1.0).mean()
    assert 0.3 < left < 0.66  # 12 of 25 kernels on the left
    return out, bw


def case_bandwidth_of_a_shorter_belief(backend):
    """manikde! of 60 points in a slot of capacity N: the leave-one-out optimum over those 60 points"""
    rng = np.random.default_rng(5)
    x = rng.normal(2.0, 1.5, (60, 1))
    be = backend(N, 1)
    try:
        be.belief_write(0, abi.EUCLID1, x, np.ones(1))
        be.run_bandwidth([0], [abi.EUCLID1])
        pts, bw, _ = be.belief_read(0, abi.EUCLID1)
    finally:
        be.close()
    assert pts.shape == (60, 1)
    span = x.max() - x.min()
    grid = np.exp(np.linspace(np.log(span * 1e-3), np.log(span), 400))
    best = max(loo_loglik(x[:, 0], h, False) for h in grid)
    assert loo_loglik(x[:, 0], bw[0], False) >= best - 2e-3
    return bw


def case_resample_tops_up_to_n(backend):
    """sample(oldBel, N - Npts): the points the belief has stay, the new ones are draws of its KDE"""
    rng = np.random.default_rng(6)
    cnt = 32
    means, vars_ = [], []
    for man in (abi.EUCLID2, abi.CIRCULAR):
        D = abi.MANIFOLD_DIM[man]
        for seed in range(20):
            x = rng.normal(0.5, 0.4, (cnt, D))
            h = np.full(D, 0.3)
            be = backend(N, 1)
            try:
                be.belief_write(0, man, x, h)
                be.run_resample([0], [man], seed)
                pts, bw, _ = be.belief_read(0, man)
            finally:
                be.close()
            assert pts.shape == (N, D) and np.abs(pts[:cnt] - x).max() < 1e-12 and np.allclose(bw, 0.3)
            new = pts[cnt:]
            near = np.abs(new[:, None, :] - x[None, :, :]).max(axis=2).min(axis=1)
            assert near.max() < 5 * 0.3
            means.append(float((new.mean(axis=0) - x.mean(axis=0)).mean()))
            vars_.append(float((new.var(axis=0) - (x.var(axis=0) + 0.09)).mean()))  # KDE variance = sample variance + h^2
    assert abs(np.mean(means)) < 0.05 and abs(np.mean(vars_)) < 0.04, (np.mean(means), np.mean(vars_))


def case_old_points_of_a_partial_product(backend):
    """a product whose inputs inform coordinate 0 only, on a variable that holds 48 points: the uninformed coordinate of
    the N output samples = the 48 old values followed by draws from the old belief's KDE"""
    rng = np.random.default_rng(7)
    cnt = 48
    old = np.stack([rng.normal(0, 1, cnt), rng.normal(50.0, 0.5, cnt)], axis=1)
    a = np.stack([rng.normal(1.0, 0.5, N), np.zeros(N)], axis=1)
    b = np.stack([rng.normal(1.4, 0.5, N), np.zeros(N)], axis=1)
    be = backend(N, 4)
    try:
        be.slot_write(0, abi.EUCLID2, a, np.ones(2))
        be.slot_write(1, abi.EUCLID2, b, np.ones(2))
        be.run_bandwidth([0, 1], [abi.EUCLID2] * 2)
        be.belief_write(3, abi.EUCLID2, old, np.array([0.3, 0.2]))
        be.run_products([iif.solver.product_desc(abi.EUCLID2, [0, 1], 3, 4242, 1, -1, [1, 1], 3)])  # in place, like a variable update
        out, bw, ipc = be.belief_read(3, abi.EUCLID2)
    finally:
        be.close()
    assert out.shape == (N, 2) and (bw > 0).all() and np.all(ipc == 2.0)
    assert np.abs(out[:cnt, 1] - old[:, 1]).max() < 1e-12
    assert np.abs(out[cnt:, 1] - 50.0).max() < 3.0 and out[cnt:, 1].std() > 0.2
    assert 0.7 < out[:, 0].mean() < 1.7  # the informed coordinate is the product of the two densities
    return out
