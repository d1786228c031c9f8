"""-m gpu: the C ABI turns bad input into a status + message (the shim maps it to `error()`,
which fails the clique task like the reference's monitorCSMs path) -- never into a crash or a hang."""
import ctypes as C

import numpy as np
import pytest

from parity_utils import abi, iif, product_desc, rand_points, relative_factor_desc

pytestmark = pytest.mark.gpu


def test_context_arguments():
    for N in (0, 7, 513):
        with pytest.raises(iif.NbpError):
            iif.HipBackend(N, 4)
    with pytest.raises(iif.NbpError):
        iif.HipBackend(100, 0)
    with pytest.raises(iif.NbpError):
        iif.HipBackend(100, 4, device=99)
    lib = abi.load_library()
    assert lib.nbp_arena_bytes(200, 10) == 10 * (3 * 200 + 8) * 8
    assert b"" != lib.nbp_last_error()


def test_caller_arena_too_small():
    import torch
    t = torch.zeros(100, dtype=torch.float64, device="cuda")
    with pytest.raises(iif.NbpError, match="arena"):
        iif.HipBackend(200, 4, arena_ptr=t.data_ptr(), arena_bytes=t.numel() * 8)


@pytest.mark.parametrize("field,value", [("manifold", 0), ("manifold", 6), ("factor_kind", 0), ("factor_kind", 7), ("nvars", 0), ("nvars", 7),
                                          ("sfidx", 2), ("ncomp", 0), ("ncomp", 5), ("inflate_cycles", 9), ("out_slot", 4), ("out_slot", -1),
                                          ("mhidx_in", 10), ("mhidx_out", 10)])
def test_bad_proposal_descriptors(hip_backend, field, value):
    N = 64
    be = hip_backend(N, 4, N)
    rng = np.random.default_rng(0)
    for s in range(3):
        be.slot_write(s, abi.EUCLID2, rand_points(rng, abi.EUCLID2, N))
    d = relative_factor_desc(abi.F_LINREL, abi.EUCLID2, 2, 1, [0, 1], 2, 1, [1.0, 1.0], [0.1, 0.1])
    be.run_proposals([d])  # the template itself is fine
    setattr(d, field, value)
    with pytest.raises(iif.NbpError):
        be.run_proposals([d])
    with pytest.raises(iif.NbpError):  # and a program refuses it at add_stage time
        be.program([(abi.STAGE_PROPOSALS, [d])])
    # the context stays usable after an error
    d2 = relative_factor_desc(abi.F_LINREL, abi.EUCLID2, 2, 1, [0, 1], 2, 1, [1.0, 1.0], [0.1, 0.1])
    be.run_proposals([d2])
    assert np.isfinite(be.slot_read(2, abi.EUCLID2)[0]).all()
    be.close()


def test_bad_variable_slot_and_factor_manifold_mismatch(hip_backend):
    be = hip_backend(64, 3, 0)
    d = relative_factor_desc(abi.F_LINREL, abi.EUCLID2, 2, 1, [0, 7], 2, 1, [1.0, 1.0], [0.1, 0.1])
    with pytest.raises(iif.NbpError):
        be.run_proposals([d])
    for kind, man in ((abi.F_CIRCULAR, abi.EUCLID1), (abi.F_SE2, abi.EUCLID3), (abi.F_LINREL, abi.SE2)):
        d = relative_factor_desc(kind, man, 2, 1, [0, 1], 2, 1, [0.1], [0.1])
        with pytest.raises(iif.NbpError):
            be.run_proposals([d])
    d = relative_factor_desc(abi.F_PRIOR, abi.EUCLID2, 2, 0, [0, 1], 2, 1, [0.1, 0.1], [0.1, 0.1])  # binary prior
    with pytest.raises(iif.NbpError):
        be.run_proposals([d])
    be.close()


@pytest.mark.parametrize("field,value", [("manifold", 9), ("nfactors", 0), ("nfactors", 129), ("niter", 0), ("niter", 9), ("out_slot", 99),
                                          ("labels_out", 5)])
def test_bad_product_descriptors(hip_backend, field, value):
    N = 64
    be = hip_backend(N, 5, N)  # N side ints: labels of a 2-density product (2N) do not fit
    rng = np.random.default_rng(0)
    for s in range(2):
        be.slot_write(s, abi.EUCLID2, rand_points(rng, abi.EUCLID2, N), np.full(2, 0.2))
    d = product_desc(abi.EUCLID2, [0, 1], 2, 1)
    be.run_products([d])
    setattr(d, field, value)
    with pytest.raises(iif.NbpError):
        be.run_products([d])
    be.close()


def test_slot_and_side_ranges(hip_backend):
    be = hip_backend(64, 2, 8)
    with pytest.raises(iif.NbpError):
        be.slot_write(2, abi.EUCLID1, np.zeros((64, 1)))
    with pytest.raises(iif.NbpError):
        be.slot_read(-1, abi.EUCLID1)
    with pytest.raises(iif.NbpError):
        be.side_write(4, np.zeros(8, dtype=np.int32))
    with pytest.raises(iif.NbpError):
        be.run_bandwidth([0], [42])
    with pytest.raises(iif.NbpError):
        be.run_copies([abi.CopyDesc(0, 5)])
    with pytest.raises(iif.NbpError):
        be.run_deconv([relative_factor_desc(abi.F_PRIOR, abi.EUCLID1, 1, 0, [0], 1, 1, [0.0], [1.0])])
    be.close()


def test_program_misuse(hip_backend):
    be = hip_backend(64, 4, 0)
    lib = be.lib
    p = C.c_void_p()
    assert lib.nbp_program_create(be._ctx, C.byref(p)) == 0
    assert lib.nbp_program_run(p, 0, -1) != 0          # not finalized
    assert lib.nbp_program_add_stage(p, 99, None, 0) != 0  # unknown stage kind
    assert lib.nbp_program_finalize(p) == 0
    assert lib.nbp_program_add_stage(p, abi.STAGE_COPIES, None, 0) != 0  # already finalized
    assert lib.nbp_program_run(p, 0, -1) == 0          # an empty program runs
    assert lib.nbp_program_destroy(p) == 0
    be.close()


@pytest.mark.timeout(120)
def test_degenerate_inputs_terminate_and_stay_finite(hip_backend):
    """identical points, far-apart clusters, huge offsets, a NaN particle (tests/degenerate_inputs.py)"""
    from degenerate_inputs import run_degenerate
    run_degenerate(hip_backend)
