"""-m gpu: the fused variable-update kernel (csrc/nbp_fused.h: proposals, their bandwidth fits, KD trees, product and the
fit of the result in one workgroup per variable) against the three-launch form of the same round (proposal kernel ->
prep kernel -> product kernel).  Both run the same device functions with the same random streams; they differ in the
launch geometry (one lane per particle throughout vs. the geometries the host picks per launch), i.e. in the order of
floating-point sums -- the comparison is the one `test_batch_size_does_not_change_results` makes between geometries:
points, bandwidths, infoPerCoord to 1e-9, and the oracle for the fused round itself."""
import os

import numpy as np
import pytest

from parity_utils import abi, assert_points_close, iif, product_desc, rand_points, relative_factor_desc

pytestmark = pytest.mark.gpu

N, MAN = 200, abi.EUCLID2


@pytest.fixture
def fused_min():
    """smallest stage that runs fused, for contexts created inside the test"""
    old = os.environ.get("NBP_FUSED_MIN")

    def set_(n):
        os.environ["NBP_FUSED_MIN"] = str(n)

    yield set_
    if old is None:
        os.environ.pop("NBP_FUSED_MIN", None)
    else:
        os.environ["NBP_FUSED_MIN"] = old


def _round(nops, F):
    """nops updates, each the product of F proposals on its own target: relatives from slot 0 / 1 and a prior"""
    props, prods = [], []
    stride = F + 1
    for i in range(nops):
        o = 4 + stride * i
        ins = []
        for j in range(F):
            if j == F - 1 and F > 1:  # the last input of every product with several inputs is a prior
                props.append(relative_factor_desc(abi.F_PRIOR, MAN, 1, 0, [2], o + j, 900 + 7 * i + j, [1.0, 1.0], [0.3, 0.3]))
            else:
                props.append(relative_factor_desc(abi.F_LINREL, MAN, 2, 1, [j % 2, 2], o + j, 900 + 7 * i + j, [1.0 - j, 1.0 - j], [0.1, 0.1]))
            ins.append(o + j)
        prods.append(product_desc(MAN, ins, o + F, 5000 + i))
    return props, prods, stride


def _run_round(hip_backend, nops, F, fused, lazy=False, read=(0, -1)):
    rng = np.random.default_rng(5)
    a, b, c = rand_points(rng, MAN, N, 0.0, 0.4), rand_points(rng, MAN, N, 2.0, 0.4), rand_points(rng, MAN, N, 1.0, 0.6)
    props, prods, stride = _round(nops, F)
    be = hip_backend(N, 4 + stride * nops, 0)
    for s, p in enumerate((a, b, c)):
        be.slot_write(s, MAN, p)
    prog = be.program([(abi.STAGE_PROPOSALS, props), (abi.STAGE_PRODUCTS, prods)], lazy_bandwidth=lazy, fused_updates=fused)
    nf = prog.num_fused()
    prog.run()
    be.synchronize()
    out = []
    for i in read:
        i = i % nops
        out.append(be.belief_read(4 + stride * i + F, MAN))
    prop0 = be.belief_read(4, MAN)  # the first proposal's own slot: written when the program ends with it still there
    diag = be.diag()
    prog.close()
    be.close()
    return nf, out, prop0, diag


@pytest.mark.parametrize("F", [1, 2, 3, 4])
def test_fused_round_equals_three_launch_round(hip_backend, fused_min, F):
    fused_min(256)
    nops = 1100
    nf, fo, fp, fd = _run_round(hip_backend, nops, F, True)
    nu, uo, up, ud = _run_round(hip_backend, nops, F, False)
    assert nf == 1 and nu == 0
    for (p, bw, ipc), (q, bw2, ipc2) in zip(fo, uo):
        assert_points_close(MAN, q, p, rtol=0, what=f"fused product, F = {F}")
        np.testing.assert_allclose(bw, bw2, rtol=0)
        np.testing.assert_array_equal(ipc, ipc2)
    # proposals that are still in their slots when the program ends are written there by the fused kernel too
    assert_points_close(MAN, up[0], fp[0], rtol=0, what="proposal slot")
    if F > 1:  # (a lone proposal's fit travels with the pass-through product: its own slot keeps no bandwidth)
        np.testing.assert_allclose(fp[1], up[1], rtol=0)
    for k in ("solves", "nonconverged", "nan_results", "residual_evals"):
        assert fd[k] == ud[k], k


def test_small_rounds_keep_the_three_launch_form(hip_backend, fused_min):
    fused_min(256)
    nf, _, _, _ = _run_round(hip_backend, 100, 2, True)
    assert nf == 0


def test_fused_round_against_the_oracle(oracle_backend, hip_backend, fused_min):
    fused_min(16)
    nops, F = 24, 3
    rng = np.random.default_rng(9)
    pts = [rand_points(rng, MAN, N, c, 0.4) for c in (0.0, 2.0, 1.0)]
    props, prods, stride = _round(nops, F)
    res = []
    for make in (oracle_backend, hip_backend):
        be = make(N, 4 + stride * nops, 0)
        for s, p in enumerate(pts):
            be.slot_write(s, MAN, p)
        prog = be.program([(abi.STAGE_PROPOSALS, props), (abi.STAGE_PRODUCTS, prods)])
        if make is hip_backend:
            assert prog.num_fused() == 1
        prog.run()
        be.synchronize()
        res.append([be.slot_read(4 + stride * i + F, MAN) for i in range(nops)])
        prog.close()
        be.close()
    for (p, bw), (q, bw2) in zip(*res):
        assert_points_close(MAN, p, q, rtol=0, what="fused update vs oracle")
        np.testing.assert_allclose(bw2, bw, rtol=0)


def test_later_readers_of_a_proposal_slot_see_it(hip_backend, fused_min):
    """a stage behind the fused pair that reads a proposal from its arena slot (here: a slot copy) gets the proposal"""
    fused_min(64)
    nops, F = 300, 2
    rng = np.random.default_rng(2)
    pts = [rand_points(rng, MAN, N, c, 0.4) for c in (0.0, 2.0, 1.0)]
    props, prods, stride = _round(nops, F)
    extra = 4 + stride * nops
    outs = []
    for fused in (True, False):
        be = hip_backend(N, extra + 2, 0)
        for s, p in enumerate(pts):
            be.slot_write(s, MAN, p)
        copies = [abi.CopyDesc(4 + stride * 7, extra), abi.CopyDesc(4 + stride * 7 + 1, extra + 1)]
        prog = be.program([(abi.STAGE_PROPOSALS, props), (abi.STAGE_PRODUCTS, prods), (abi.STAGE_COPIES, copies)], fused_updates=fused)
        assert prog.num_fused() == (1 if fused else 0)
        prog.run()
        be.synchronize()
        outs.append([be.slot_read(extra, MAN), be.slot_read(extra + 1, MAN)])
        prog.close()
        be.close()
    for (p, bw), (q, bw2) in zip(*outs):
        assert_points_close(MAN, q, p, rtol=0, what="copied proposal")
        np.testing.assert_allclose(bw, bw2, rtol=0)
        assert np.abs(p).max() > 0


def test_a_range_that_splits_a_fused_pair_runs_it_in_three_launches(hip_backend, fused_min):
    """nbp_program_run(first, last) with a range that ends or starts between the two stages of a fused pair (legal when the
    program was finalized; round 4 refused it with NBP_ERR_ARG): the pair runs in the three-launch form -- the proposals
    to their arena slots with their fits at the end of the range, then KD builds, products and the fits of the results.
    The same particles and bandwidths as a program with fused updates off, run in the same two halves."""
    fused_min(64)
    nops, F = 200, 2
    props, prods, stride = _round(nops, F)
    rng = np.random.default_rng(5)
    pts = [rand_points(rng, MAN, N, c, 0.4) for c in (0.0, 2.0, 1.0)]
    res = []
    for fused in (True, False):
        be = hip_backend(N, 4 + stride * nops, 0)
        for s, p_ in enumerate(pts):
            be.slot_write(s, MAN, p_)
        prog = be.program([(abi.STAGE_PROPOSALS, props), (abi.STAGE_PRODUCTS, prods)], fused_updates=fused)
        assert prog.num_fused() == (1 if fused else 0)
        prog.run(0, 1)
        be.synchronize()
        mid = [be.belief_read(4 + stride * i + j, MAN) for i in (0, 77, nops - 1) for j in range(F)]  # the proposals, fitted
        prog.run(1, 2)
        be.synchronize()
        out = [be.belief_read(4 + stride * i + F, MAN) for i in (0, 77, nops - 1)]
        prog.close()
        be.close()
        res.append((mid, out))
    for a, b in zip(res[0][0] + res[0][1], res[1][0] + res[1][1]):
        assert_points_close(MAN, a[0], b[0], rtol=0, what="split fused pair vs three-launch program")
        np.testing.assert_allclose(a[1], b[1], rtol=0)  # bandwidths
        np.testing.assert_allclose(a[2], b[2])             # infoPerCoord
        assert np.all(a[1] > 0)


def test_whole_solve_with_fused_rounds(hip_backend, fused_min):
    """a chain solved with every round of >= 16 updates fused against the same solve in the three-launch form: same
    random streams, geometry-level rounding differences only -- most variables stay particle-identical, all stay at the
    truth"""
    fused_min(16)

    def solve(fused):
        if not fused:
            os.environ["NBP_NO_FUSED_UPDATE"] = "1"
        try:
            fg = iif.generateChainEuclid(160, vardims=2, priorEvery=20, N=100)
            iif.solveTree(fg, eliminationOrder=iif.nestedDissectionOrder(fg), backend=hip_backend, seed=3)
        finally:
            os.environ.pop("NBP_NO_FUSED_UPDATE", None)
        return {v: fg.getVal(v) for v in fg.ls()}

    a, b = solve(True), solve(False)
    same = 0
    for i, v in enumerate(sorted(a, key=lambda s: int(s[1:]))):
        assert np.abs(a[v].mean(axis=0) - i).max() < 0.6, (v, a[v].mean(axis=0))
        same += np.allclose(a[v], b[v], rtol=0, atol=0)
    print(f"fused vs three-launch solve: {same} of {len(a)} variables particle-identical")
    assert same >= len(a) // 2
