"""-m gpu: the axes BASELINE's configurations never vary -- the particle count of a shape and the number of densities of a
product -- against the oracle, bit for bit.

Every BASELINE configuration runs at ONE particle count (200 or 300), and its products multiply two to four densities; the
kernels pick their geometries from exactly those two numbers (waves per row, helper lanes per sample, LDS or scratch for the
node statistics, chunk sums per lane).  Round 5's two shipped defects lived on these axes -- a scratch stride in launches that
mix density counts at N >= ~260, a fit that double-counted pairs for beliefs shorter than their slot -- and four rounds of
green parity never saw them, because every parity launch was uniform and every shape ran at its own N.  These cases were
builder-run scripts then (tools/exp/stagewise_any_n.py, tools/exp/many_density_products.py); they are the driver's now.

(1) every BASELINE shape (reduced graphs), EVERY STAGE of its tree program on the oracle's state, at N = 64, 150, 257, 300, 500:
    one wave, a ragged third wave, 4k + 1 waves (the five-waves-per-SIMD instances), BASELINE's other count, eight waves.
(2) products of 8 / 32 / 128 densities in ONE launch beside two- and three-density ones, five manifolds, N = 100 / 200 / 300.
(3) every BASELINE shape with the SOLVER PARAMETERS its configurations leave at their defaults set otherwise: joint messages
    (useMsgLikelihoods -> deconvolution stages and message-likelihood factors), the number of product iterations and Gibbs
    sweeps, the inflation cycles and their spread, limitfixeddown, no null-hypothesis surplus.  The first run of this sweep
    (tools/exp/stagewise_other_params.sh, profiles/r06_stagewise_other_solver_parameters.txt) found the two places where
    checker and harness did not treat an UNWRAPPED angle the way the kernels do: the fit of a deconvolution's output, and the
    hand-back of one through the circular manifold.
What is asserted is np.array_equal on stored coordinates and bandwidths (tests/test_gpu_stagewise_parity.py says why that holds)."""
import numpy as np
import pytest

import test_gpu_stagewise_parity as stagewise
from parity_utils import abi, iif
from test_gpu_mixed_product_launches import run

pytestmark = pytest.mark.gpu

SHAPES = {
    "config2_shape": lambda N: iif.generateChainEuclid(48, vardims=2, priorEvery=12, N=N),
    "config3_shape": lambda N: iif.generateCircularDoors(nposes=30, N=N, sightEvery=10),
    "config4_shape": lambda N: iif.generateSE2Lattice(rows=3, cols=6, N=N, closeEvery=2),
    "config5_shape": lambda N: iif.generateMixtureChain(nvars=24, N=N, priorEvery=8),
}


@pytest.mark.parametrize("N", [64, 150, 257, 300, 500])
@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_every_stage_at_particle_counts_baseline_does_not_use(oracle_backend, hip_backend, shape, N):
    # ("full_size" in the name: initialised on the device, the oracle on up to 64 host threads -- test_gpu_stagewise_parity)
    name = f"{shape}_full_size_probe_N{N}"
    stagewise.FULL[name] = lambda: SHAPES[shape](N)
    try:
        stagewise.test_every_stage_of_the_tree_program_on_the_oracles_state(oracle_backend, hip_backend, name)
    finally:
        del stagewise.FULL[name]


@pytest.mark.parametrize("N", [100, 200, 300])
@pytest.mark.parametrize("man", [abi.EUCLID1, abi.EUCLID2, abi.EUCLID3, abi.CIRCULAR, abi.SE2])
def test_products_of_many_densities_beside_small_ones(hip_backend, man, N):
    from oracle.oracle_backend import OracleBackend
    import os
    nthreads = max(8, min(64, os.cpu_count() or 8))
    Fs = [2, 8, 2, 3, 32, 2, 2, 128, 2, 3, 2, 16, 2, 2, 2, 2, 64, 2, 2, 2]
    keep = [i for i, f in enumerate(Fs) if f > 3] + [0, 3, len(Fs) - 1]
    d = run(hip_backend, N, man, Fs, nsrc=40)
    o = run(lambda n, s, side_ints=0: OracleBackend(n, s, side_ints, threads=nthreads), N, man, Fs, keep=keep, nsrc=40)
    assert all(np.isfinite(v).all() for v in d.values())
    for i in keep:
        assert np.array_equal(d[i], o[i]), (f"product {i} of the launch ({Fs[i]} densities, manifold {man}, N = {N}): "
                                           f"{int((d[i] != o[i]).any(axis=1).sum())} of {N} samples differ, by up to {np.abs(d[i] - o[i]).max():.2e}")


SETTINGS = [{"useMsgLikelihoods": True}, {"productNiter": 2}, {"productNiter": 8}, {"inflateCycles": 1}, {"inflateCycles": 5},
            {"gibbsIters": 1}, {"gibbsIters": 5}, {"limitfixeddown": True}, {"spreadNH": 1.0, "inflation": 2.0}, {"nullSurplusAdd": 0.0}]


@pytest.mark.parametrize("setting", SETTINGS, ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
@pytest.mark.parametrize("shape", sorted(SHAPES))
def test_every_stage_with_solver_parameters_baseline_leaves_at_their_defaults(oracle_backend, hip_backend, shape, setting):
    name = f"{shape}_full_size_probe_" + "_".join(setting)

    def make():
        fg = SHAPES[shape](200)
        for k, v in setting.items():
            assert hasattr(fg.solverParams, k), k
            setattr(fg.solverParams, k, v)
        return fg

    stagewise.FULL[name] = make
    try:
        stagewise.test_every_stage_of_the_tree_program_on_the_oracles_state(oracle_backend, hip_backend, name)
    finally:
        del stagewise.FULL[name]
