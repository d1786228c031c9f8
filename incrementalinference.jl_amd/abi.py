"""ctypes mirror of include/nbp.h -- the C ABI of libnbp.

The structures here are byte-for-byte the ones declared in ``include/nbp.h``; the same
descriptors are consumed by the HIP library (product) and, in tests only, by the CPU oracle.
"""
import ctypes as C
import os

MAXV, MAXF, MAXD, MAXC, MAXN, COMP_STRIDE = 6, 128, 3, 4, 512, 13

# enum nbp_manifold
EUCLID1, EUCLID2, EUCLID3, CIRCULAR, SE2 = 1, 2, 3, 4, 5
# enum nbp_factor
F_PRIOR, F_MSGPRIOR, F_LINREL, F_CIRCULAR, F_SE2, F_EUCLIDDIST, F_PASSTHROUGH = 1, 2, 3, 4, 5, 6, 7
DIST_GAUSSIAN, DIST_UNIFORM, DIST_RAYLEIGH, DIST_TABLE = 0, 1, 2, 3  # enum nbp_dist: family of a scalar measurement component (comp[c][12])
STAGE_PROPOSALS, STAGE_PRODUCTS, STAGE_COPIES, STAGE_DECONV, STAGE_COPY_POINTS = 1, 2, 3, 4, 5
OPT_LAZY_BANDWIDTH = 1
OPT_GRAPH_REPLAY = 2
OPT_FUSED_UPDATES = 3

MANIFOLD_DIM = {EUCLID1: 1, EUCLID2: 2, EUCLID3: 3, CIRCULAR: 1, SE2: 3}
MANIFOLD_P = {EUCLID1: 1, EUCLID2: 2, EUCLID3: 3, CIRCULAR: 1, SE2: 6}


class ProposalDesc(C.Structure):
    _fields_ = [
        ("factor_kind", C.c_int32),
        ("manifold", C.c_int32),
        ("nvars", C.c_int32),
        ("sfidx", C.c_int32),
        ("var_slot", C.c_int32 * MAXV),
        ("out_slot", C.c_int32),
        ("ncomp", C.c_int32),
        ("has_multihypo", C.c_int32),
        ("inflate_cycles", C.c_int32),
        ("mhidx_in", C.c_int32),
        ("mhidx_out", C.c_int32),
        ("skip_bandwidth", C.c_int32),
        ("partial_mask", C.c_int32),
        ("meas_kde", C.c_int32),
        ("keep_count", C.c_int32),
        ("multihypo", C.c_double * MAXV),
        ("nullhypo", C.c_double),
        ("inflation", C.c_double),
        ("spread_nh", C.c_double),
        ("comp", (C.c_double * COMP_STRIDE) * MAXC),
        ("seed", C.c_uint64),
        ("meas_seed", C.c_uint64),
    ]


class ProductDesc(C.Structure):
    _fields_ = [
        ("manifold", C.c_int32),
        ("nfactors", C.c_int32),
        ("niter", C.c_int32),
        ("out_slot", C.c_int32),
        ("in_slot", C.c_int32 * MAXF),
        ("labels_out", C.c_int32),
        ("old_slot", C.c_int32),
        ("in_partial", C.c_uint8 * MAXF),
        ("seed", C.c_uint64),
    ]


class CopyDesc(C.Structure):
    _fields_ = [("src_slot", C.c_int32), ("dst_slot", C.c_int32)]


class Xfer(C.Structure):
    _fields_ = [("peer", C.c_int32), ("slot", C.c_int32)]


COMM_ID_BYTES = 128


class Diag(C.Structure):
    _fields_ = [
        ("solves", C.c_int64),
        ("nonconverged", C.c_int64),
        ("nan_results", C.c_int64),
        ("residual_evals", C.c_int64),
        ("lcv_evals", C.c_int64),
        ("lcv_evals_f32", C.c_int64),
    ]


def slot_stride(N):
    return 3 * N + 8


_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, "csrc", "libnbp.so")

# every symbol include/nbp.h declares (tests check the .so exports all of them)
EXPORTS = [
    "nbp_arena_bytes", "nbp_slot_stride_doubles", "nbp_ctx_create", "nbp_ctx_destroy",
    "nbp_last_error", "nbp_synchronize", "nbp_arena_ptr", "nbp_stream_ptr", "nbp_ctx_particles", "nbp_ctx_slots", "nbp_ctx_device",
    "nbp_ctx_reserve_resident", "nbp_ctx_resident", "nbp_belief_write_batch_async", "nbp_belief_read_batch_begin", "nbp_belief_read_batch_end",
    "nbp_run_copies_async", "nbp_program_retire",
    "nbp_slot_write", "nbp_slot_read", "nbp_belief_write", "nbp_belief_read", "nbp_belief_write_batch", "nbp_belief_read_batch", "nbp_run_resample", "nbp_side_write", "nbp_side_read",
    "nbp_run_proposals", "nbp_run_bandwidth", "nbp_run_products", "nbp_run_copies", "nbp_run_deconv", "nbp_kde_bandwidth", "nbp_conv", "nbp_manifold_product",
    "nbp_program_create", "nbp_program_add_stage", "nbp_program_set_option", "nbp_program_finalize", "nbp_program_run",
    "nbp_program_reseed", "nbp_program_num_seeds", "nbp_program_set_seeds", "nbp_ctx_attach", "nbp_ctx_attached", "nbp_program_num_stages", "nbp_program_num_fused", "nbp_program_num_two_stream", "nbp_program_destroy",
    "nbp_timing_enable", "nbp_timing_read", "nbp_timing_read_n", "nbp_diag_read",
    "nbp_comm_unique_id", "nbp_comm_create", "nbp_comm_destroy", "nbp_comm_info", "nbp_exchange", "nbp_math_eval",
]

_lib = None


def load_library(path=None):
    """dlopen libnbp.so and declare argument types.  Fails loudly when the HIP extension has
    not been built -- there is no CPU fallback in the product path."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("NBP_LIB_OVERRIDE") or LIB_PATH  # override: kernel experiments (tools/exp)
    if not os.path.exists(path):
        raise RuntimeError(
            f"libnbp.so not found at {path}: build it with `python __graft_entry__.py` "
            "(hipcc --offload-arch=gfx950). The product path has no CPU fallback."
        )
    # One HIP runtime per process.  PyTorch ships its own libamdhip64 / libhsa-runtime64 with the same
    # SONAMEs as /opt/rocm; if libnbp pulled in the system copies first, a later `import torch` (the
    # multi-GPU path: torch-owned arena + torch.distributed) would bring up a second runtime that cannot
    # open the GPU ("No HIP GPUs are available").  Importing torch first makes libnbp bind to the copies
    # torch loaded.  Without PyTorch (e.g. the Julia shim) the system runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    vp, i32, i64, dp = C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_double)
    ip = C.POINTER(C.c_int32)
    lib.nbp_arena_bytes.restype = i64
    lib.nbp_arena_bytes.argtypes = [i32, i32]
    lib.nbp_slot_stride_doubles.restype = i64
    lib.nbp_slot_stride_doubles.argtypes = [i32]
    lib.nbp_ctx_create.argtypes = [i32, i32, i32, vp, i64, i32, C.POINTER(vp)]
    lib.nbp_ctx_destroy.argtypes = [vp]
    lib.nbp_last_error.restype = C.c_char_p
    lib.nbp_synchronize.argtypes = [vp]
    lib.nbp_arena_ptr.restype = vp
    lib.nbp_arena_ptr.argtypes = [vp]
    lib.nbp_stream_ptr.restype = vp
    lib.nbp_stream_ptr.argtypes = [vp]
    lib.nbp_ctx_particles.argtypes = [vp]
    lib.nbp_ctx_slots.argtypes = [vp]
    lib.nbp_ctx_device.argtypes = [vp]
    lib.nbp_slot_write.argtypes = [vp, i32, i32, dp, dp]
    lib.nbp_slot_read.argtypes = [vp, i32, i32, dp, dp]
    lib.nbp_belief_write.argtypes = [vp, i32, i32, dp, i32, dp, dp]
    lib.nbp_belief_read.argtypes = [vp, i32, i32, dp, ip, dp, dp]
    lib.nbp_belief_write_batch.argtypes = [vp, i32, ip, ip, C.POINTER(dp), ip, C.POINTER(dp), C.POINTER(dp)]
    lib.nbp_belief_read_batch.argtypes = [vp, i32, ip, ip, C.POINTER(dp), ip, C.POINTER(dp), C.POINTER(dp)]
    lib.nbp_run_resample.argtypes = [vp, ip, ip, i32, C.c_uint64]
    lib.nbp_side_write.argtypes = [vp, i32, ip, i32]
    lib.nbp_side_read.argtypes = [vp, i32, ip, i32]
    lib.nbp_run_proposals.argtypes = [vp, C.POINTER(ProposalDesc), i32]
    lib.nbp_run_bandwidth.argtypes = [vp, ip, ip, i32]
    lib.nbp_run_deconv.argtypes = [vp, C.POINTER(ProposalDesc), ip, i32]
    dpp = C.POINTER(dp)
    lib.nbp_kde_bandwidth.argtypes = [vp, i32, dp, dp]
    lib.nbp_conv.argtypes = [vp, C.POINTER(ProposalDesc), dpp, dpp, ip, dp, dp, ip]
    lib.nbp_manifold_product.argtypes = [vp, i32, i32, dpp, dpp, C.POINTER(C.c_uint8), dp, i32, C.c_uint64, dp, dp, ip]
    lib.nbp_run_products.argtypes = [vp, C.POINTER(ProductDesc), i32]
    lib.nbp_run_copies.argtypes = [vp, C.POINTER(CopyDesc), i32]
    lib.nbp_program_create.argtypes = [vp, C.POINTER(vp)]
    lib.nbp_program_add_stage.argtypes = [vp, i32, vp, i32]
    lib.nbp_program_set_option.argtypes = [vp, i32, i32]
    lib.nbp_program_finalize.argtypes = [vp]
    lib.nbp_program_run.argtypes = [vp, i32, i32]
    lib.nbp_program_reseed.argtypes = [vp, C.c_uint64]
    lib.nbp_program_num_seeds.argtypes = [vp, ip]
    lib.nbp_program_set_seeds.argtypes = [vp, C.POINTER(C.c_uint64), i32]
    lib.nbp_program_num_stages.argtypes = [vp, ip]
    lib.nbp_program_destroy.argtypes = [vp]
    lib.nbp_comm_unique_id.argtypes = [vp]
    lib.nbp_comm_create.argtypes = [vp, i32, i32, vp, C.POINTER(vp)]
    lib.nbp_comm_destroy.argtypes = [vp]
    lib.nbp_comm_info.argtypes = [vp, ip, ip]
    lib.nbp_math_eval.argtypes = [vp, i32, dp, dp, dp, dp, i64]
    lib.nbp_exchange.argtypes = [vp, vp, C.POINTER(Xfer), i32, C.POINTER(Xfer), i32]
    lib.nbp_timing_enable.argtypes = [vp, i32]
    lib.nbp_timing_read.argtypes = [vp, dp, C.POINTER(i64)]
    lib.nbp_timing_read_n.argtypes = [vp, dp, C.POINTER(i64), i32]
    lib.nbp_program_num_fused.argtypes = [vp, C.POINTER(i32)]
    lib.nbp_program_num_two_stream.argtypes = [vp, C.POINTER(i32)]
    lib.nbp_diag_read.argtypes = [vp, C.POINTER(Diag), i32]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int:
            fn.restype = C.c_int32
    if path == LIB_PATH:
        _lib = lib
    return lib
