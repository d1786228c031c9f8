"""Chip-filling launches of "simple" Euclidean proposals (LinearRelative between two variables, no multihypo, no nullhypo;
priors, message priors and pass-through densities ride along) run one WAVE per proposal (nbp_proposal_wave_kernel_lin2 /
_lin3 / _lin3n5, launch_proposals in nbp_api.hip; selected from NBP_PROPOSAL_WAVE_MIN proposals per launch on, 3000 by
default).  Lane l of the wave owns the particles l, l + 64, ...; the spread statistics add the chunks of 64 in the order
the workgroup kernels add their wave partials, every particle's search is the same sequence of operations.  The particles
must not depend on the geometry: compared with the workgroup instance of the same class, BIT FOR BIT -- as many
residual evaluations in every search, the same particles and bandwidths to the last bit.  (Since round 5 libnbp is built with
-ffp-contract=on: a multiply-add is contracted inside one source statement only, so an inlined device function is the same
arithmetic in every kernel it lands in.  With the compiler's default the two geometries differed in one coordinate of a few
hundred at 3e-12: profiles/r05_fp_contract_on_vs_fast.txt.)"""
import os

import numpy as np
import pytest

from parity_utils import abi, iif, rand_points, relative_factor_desc

pytestmark = pytest.mark.gpu


def _batch(man, dim, N, rng, be):
    # slots 0..5: beliefs; slot 3 holds fewer points than N (a target copy resized to N / an operand read through
    # _getindex_anyn); slot 5: a density for the pass-through proposal
    for j in range(6):
        pts = rand_points(rng, man, N if j != 3 else N - 37, 1.5 * j, 0.4)
        be.belief_write(j, man, pts, np.full(dim, 0.3))
    mean, sig = [1.0, -0.5, 0.25][:dim], [0.1, 0.2, 0.15][:dim]
    descs = [
        relative_factor_desc(abi.F_LINREL, man, 2, 1, [0, 1], 8, 11, mean, sig),              # solve the second variable
        relative_factor_desc(abi.F_LINREL, man, 2, 0, [2, 4], 9, 12, mean, sig),              # solve the first
        relative_factor_desc(abi.F_LINREL, man, 2, 1, [3, 2], 10, 13, mean, sig),             # operand with fewer points
        relative_factor_desc(abi.F_LINREL, man, 2, 0, [3, 1], 11, 14, mean, sig, cycles=2),   # target with fewer points
        relative_factor_desc(abi.F_PRIOR, man, 1, 0, [0], 12, 15, mean, sig),
        relative_factor_desc(abi.F_MSGPRIOR, man, 1, 0, [0, 1], 13, 16, [0], [0]),
        relative_factor_desc(abi.F_MSGPRIOR, man, 1, 0, [2, 3], 14, 17, [0], [0]),            # message with fewer points
        relative_factor_desc(abi.F_LINREL, man, 2, 1, [4, 0], 15, 18, mean, sig, mhidx_out=0),
    ]
    pt = relative_factor_desc(abi.F_PASSTHROUGH, man, 1, 0, [1, 5], 16, 19, [0], [0])
    descs.append(pt)
    return descs, list(range(8, 17))


@pytest.mark.parametrize("man,dim,N", [(abi.EUCLID2, 2, 200), (abi.EUCLID2, 2, 150), (abi.EUCLID2, 2, 256), (abi.EUCLID3, 3, 200),
                                       (abi.EUCLID3, 3, 300)])
def test_wave_geometry_equals_workgroup_geometry(man, dim, N):
    out = {}
    for mode in ("workgroup", "wave"):
        if mode == "wave":
            os.environ["NBP_PROPOSAL_WAVE_MIN"] = "1"
        else:
            os.environ.pop("NBP_PROPOSAL_WAVE_MIN", None)
        try:
            be = iif.HipBackend(N, 24, N)
        finally:
            os.environ.pop("NBP_PROPOSAL_WAVE_MIN", None)
        descs, outs = _batch(man, dim, N, np.random.default_rng(5), be)
        be.diag(reset=True)
        be.run_proposals(descs)
        dg = be.diag()
        out[mode] = ([be.slot_read(s, man) for s in outs], np.array(be.side_read(0, N)), dg)
        be.close()
    (wg, mh_wg, dg_wg), (wv, mh_wv, dg_wv) = out["workgroup"], out["wave"]
    assert np.array_equal(mh_wg, mh_wv) and (mh_wv == 1).all()
    # the same searches were run: as many solves and residual evaluations
    assert dg_wg["solves"] == dg_wv["solves"] and dg_wg["residual_evals"] == dg_wv["residual_evals"], (dg_wg, dg_wv)
    worst = 0.0
    for (pa, ba), (pb, bb) in zip(wg, wv):
        assert pa.shape == pb.shape
        worst = max(worst, float(np.abs(pa - pb).max()))
        assert np.array_equal(pb, pa) and np.array_equal(bb, ba), f"max |difference| {np.abs(pa - pb).max():.3e}"
    print(f"wave vs workgroup, manifold {man} N {N}: max |difference| {worst:.3e}")
