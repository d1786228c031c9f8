"""CPU oracle backend -- TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py's
cpu_baseline leg).  Implements the same backend interface as iif_amd.backend.HipBackend on top of
oracle/liboracle.so (plain C restatement, oracle/nbp_oracle.c).  Never imported by the package."""
import ctypes as C
import os
import subprocess

import numpy as np

import iif_amd_loader

abi = iif_amd_loader.load().abi

_DIR = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_DIR, "liboracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-C", _DIR, "-s"])


def use_native_build(out):
    """bench.py's cpu_baseline leg: the same source compiled for the timing host (-O3 -march=native) into `out`, loaded
    in place of the stock library.  -> the flags used, or None when the build failed (the stock -O2 library stays)"""
    global _SO, _lib
    try:
        subprocess.check_call(["make", "-C", _DIR, "-s", "native", f"OUT={out}"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except Exception:  # noqa: BLE001
        return None
    _SO, _lib = out, None
    return "-O3 -march=native -ffp-contract=off -fopenmp"


def lib():
    global _lib
    if _lib is None:
        if _SO.startswith(_DIR) and (not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_DIR, "nbp_oracle.c"))):
            build()
        L = C.CDLL(_SO)
        dp, ip, i32 = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int32
        L.orc_slot_stride.restype = C.c_int64
        L.orc_slot_write.argtypes = [dp, i32, i32, i32, dp, dp]
        L.orc_slot_write.restype = None
        L.orc_slot_read.argtypes = [dp, i32, i32, i32, dp, dp]
        L.orc_slot_read.restype = None
        L.orc_belief_write.argtypes = [dp, i32, i32, i32, dp, i32, dp]
        L.orc_belief_write.restype = None
        L.orc_slot_count.argtypes = [dp, i32, i32]
        L.orc_slot_count.restype = i32
        L.orc_run_resample.argtypes = [dp, i32, ip, ip, i32, C.c_uint64]
        L.orc_run_resample.restype = None
        L.orc_slot_ipc_write.argtypes = [dp, i32, i32, i32, dp]
        L.orc_slot_ipc_write.restype = None
        L.orc_slot_ipc_read.argtypes = [dp, i32, i32, i32, dp]
        L.orc_slot_ipc_read.restype = None
        L.orc_run_proposals.argtypes = [dp, i32, ip, C.POINTER(abi.ProposalDesc), i32]
        L.orc_run_products.argtypes = [dp, i32, ip, C.POINTER(abi.ProductDesc), i32]
        L.orc_run_deconvs.argtypes = [dp, i32, C.POINTER(abi.ProposalDesc), ip, i32]
        L.orc_run_copies.argtypes = [dp, i32, C.POINTER(abi.CopyDesc), i32]
        L.orc_run_copies.restype = None
        L.orc_run_bandwidth.argtypes = [dp, i32, i32, i32]
        L.orc_run_bandwidth.restype = None
        L.orc_lcv_bandwidth_1d.argtypes = [dp, i32, i32]
        L.orc_lcv_bandwidth_1d.restype = C.c_double
        L.orc_wrap.argtypes = [C.c_double]
        L.orc_wrap.restype = C.c_double
        L.orc_std_basic_spread.argtypes = [i32, dp, i32]
        L.orc_std_basic_spread.restype = C.c_double
        L.orc_residual.argtypes = [i32, i32, dp, dp, dp, dp]
        L.orc_solve_particle.argtypes = [i32, i32, dp, dp, i32, dp]
        L.orc_hypo_recipe.argtypes = [i32, dp, i32, i32, C.c_double, ip, i32, ip, ip, ip, ip, ip, ip, ip]
        L.orc_philox.argtypes = [C.c_uint32] * 6 + [C.POINTER(C.c_uint32)]
        L.orc_philox.restype = None
        L.orc_uniform_pair.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, dp, dp]
        L.orc_uniform_pair.restype = None
        L.orc_normal_pair.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, dp, dp]
        L.orc_normal_pair.restype = None
        L.orc_math_eval.argtypes = [i32, dp, dp, dp, dp, C.c_int64]
        L.orc_math_eval.restype = None
        L.orc_mean_geodesic_device_order.argtypes = [dp, i32, i32]
        L.orc_mean_geodesic_device_order.restype = C.c_double
        L.orc_mean_geodesic_walk.argtypes = [dp, i32, i32]
        L.orc_mean_geodesic_walk.restype = C.c_double
        L.orc_set_threads.argtypes = [i32]
        L.orc_set_nested.argtypes = [i32]
        L.orc_set_nested.restype = None
        L.orc_set_threads.restype = None
        L.orc_get_max_threads.restype = i32
        L.orc_diag_read.argtypes = [C.POINTER(C.c_int64 * 5), i32]
        L.orc_diag_read.restype = None
        _lib = L
    return _lib


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def math_eval(fn, a, b=None):
    """the shared elementary functions (include/nbp_math.h) as gcc compiles them -- same numbering as nbp_math_eval:
    0 log, 1 sincos, 2 atan2(a, b), 3 wrap to [-pi, pi), 4 Box-Muller of the uniforms (a, b).  -> (out0, out1)"""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.zeros_like(a) if b is None else np.ascontiguousarray(b, dtype=np.float64)
    o0, o1 = np.empty_like(a), np.empty_like(a)
    lib().orc_math_eval(fn, _dp(a), _dp(b), _dp(o0), _dp(o1), a.size)
    return o0, o1


class OracleBackend:
    name = "oracle"

    def __init__(self, N, n_slots, side_ints=0, threads=1, nested=False, **_):
        """threads: OpenMP team over the independent ops of a stage; nested: a second level inside an op (particles, product
        samples, likelihood rows) on the threads the stage's team leaves idle -- same results bit for bit either way"""
        self.lib = lib()
        self.nested = bool(nested)
        self.N, self.n_slots = int(N), int(n_slots)
        self.arena = np.zeros(self.n_slots * abi.slot_stride(self.N))
        self.side = np.zeros(max(int(side_ints), 1), dtype=np.int32)
        self.threads = threads

    def close(self):
        pass

    def _sidep(self):
        return self.side.ctypes.data_as(C.POINTER(C.c_int32))

    def slot_write(self, slot, manifold, pts, bw=None):
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(self.N, abi.MANIFOLD_P[manifold])
        bwp = None
        if bw is not None:
            bw = np.ascontiguousarray(bw, dtype=np.float64)
            bwp = _dp(bw)
        self.lib.orc_slot_write(_dp(self.arena), self.N, slot, manifold, _dp(pts), bwp)

    def slot_read(self, slot, manifold):
        pts = np.empty((self.N, abi.MANIFOLD_P[manifold]))
        bw = np.empty(abi.MANIFOLD_DIM[manifold])
        self.lib.orc_slot_read(_dp(self.arena), self.N, slot, manifold, _dp(pts), _dp(bw))
        return pts, bw

    def belief_write(self, slot, manifold, pts, bw=None, ipc=None):
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, abi.MANIFOLD_P[manifold])
        bwa = None if bw is None else np.ascontiguousarray(bw, dtype=np.float64)
        self.lib.orc_belief_write(_dp(self.arena), self.N, slot, manifold, _dp(pts), pts.shape[0], _dp(bwa) if bwa is not None else None)
        if ipc is not None:
            self.lib.orc_slot_ipc_write(_dp(self.arena), self.N, slot, manifold, _dp(np.ascontiguousarray(ipc, dtype=np.float64)))

    def belief_read(self, slot, manifold):
        pts, bw = self.slot_read(slot, manifold)
        ipc = np.zeros(abi.MANIFOLD_DIM[manifold])
        self.lib.orc_slot_ipc_read(_dp(self.arena), self.N, slot, manifold, _dp(ipc))
        return pts[:self.lib.orc_slot_count(_dp(self.arena), self.N, slot)], bw, ipc

    def run_resample(self, slots, manifolds, seed=0):
        s = np.ascontiguousarray(slots, dtype=np.int32)
        m = np.ascontiguousarray(manifolds, dtype=np.int32)
        ip = C.POINTER(C.c_int32)
        self.lib.orc_run_resample(_dp(self.arena), self.N, s.ctypes.data_as(ip), m.ctypes.data_as(ip), s.size, C.c_uint64(seed))

    def side_write(self, offset, ints):
        a = np.asarray(ints, dtype=np.int32)
        self.side[offset:offset + a.size] = a

    def side_read(self, offset, n):
        return self.side[offset:offset + n].copy()

    def _arr(self, descs, ctype):
        if isinstance(descs, C.Array):
            return descs, len(descs)
        return (ctype * len(descs))(*descs), len(descs)

    def run_proposals(self, descs):
        self.lib.orc_set_threads(self.threads)
        self.lib.orc_set_nested(1 if self.nested else 0)
        arr, n = self._arr(descs, abi.ProposalDesc)
        rc = self.lib.orc_run_proposals(_dp(self.arena), self.N, self._sidep(), arr, n)
        if rc:
            raise RuntimeError(f"oracle status {rc}")

    def run_products(self, descs):
        self.lib.orc_set_threads(self.threads)
        self.lib.orc_set_nested(1 if self.nested else 0)
        arr, n = self._arr(descs, abi.ProductDesc)
        rc = self.lib.orc_run_products(_dp(self.arena), self.N, self._sidep(), arr, n)
        if rc:
            raise RuntimeError(f"oracle status {rc}")

    def run_deconv(self, descs, meas_slots=None):
        self.lib.orc_set_threads(self.threads)
        self.lib.orc_set_nested(1 if self.nested else 0)
        arr, n = self._arr(descs, abi.ProposalDesc)
        ms = np.ascontiguousarray(meas_slots if meas_slots is not None else [-1] * n, dtype=np.int32)
        rc = self.lib.orc_run_deconvs(_dp(self.arena), self.N, arr, ms.ctypes.data_as(C.POINTER(C.c_int32)), n)
        if rc:
            raise RuntimeError(f"oracle status {rc}")

    def run_copies(self, descs):
        arr, n = self._arr(descs, abi.CopyDesc)
        self.lib.orc_run_copies(_dp(self.arena), self.N, arr, n)

    def run_copy_points(self, descs):
        """NBP_STAGE_COPY_POINTS: points only, the destination's bandwidth stays as it is"""
        S = 3 * self.N + 8
        a = self.arena.reshape(-1, S)
        for d in descs:
            a[d.dst_slot, :3 * self.N] = a[d.src_slot, :3 * self.N]
            a[d.dst_slot, 3 * self.N + 6] = a[d.src_slot, 3 * self.N + 6]  # the particle count belongs to the points

    def run_bandwidth(self, slots, manifolds):
        for s, m in zip(slots, manifolds):
            self.lib.orc_run_bandwidth(_dp(self.arena), self.N, int(s), int(m))

    def synchronize(self):
        pass

    def program(self, stages, lazy_bandwidth=False):
        return OracleProgram(self, stages)  # the oracle always fits every bandwidth

    def diag(self, reset=False):
        d = (C.c_int64 * 5)()
        self.lib.orc_diag_read(C.byref(d), int(reset))
        return dict(zip(["solves", "nonconverged", "nan_results", "residual_evals", "lcv_evals"], list(d)))


class OracleProgram:
    def __init__(self, backend, stages):
        self.backend = backend
        self.stages = [(k, backend._arr(d, {abi.STAGE_PROPOSALS: abi.ProposalDesc,
                                            abi.STAGE_PRODUCTS: abi.ProductDesc,
                                            abi.STAGE_COPIES: abi.CopyDesc,
                                            abi.STAGE_DECONV: abi.ProposalDesc,
                                            abi.STAGE_COPY_POINTS: abi.CopyDesc}[k])[0]) for k, d in stages]
        self.n_stages = len(stages)

    def run(self, first=0, last=-1):
        last = self.n_stages if last < 0 else last
        for kind, arr in self.stages[first:last]:
            if kind == abi.STAGE_PROPOSALS:
                self.backend.run_proposals(arr)
            elif kind == abi.STAGE_PRODUCTS:
                self.backend.run_products(arr)
            elif kind == abi.STAGE_DECONV:  # predicted measurements + manikde! of them
                if len(arr):
                    self.backend.run_deconv(arr)
                    self.backend.run_bandwidth([d.out_slot for d in arr], [d.manifold for d in arr])
            elif kind == abi.STAGE_COPY_POINTS:
                self.backend.run_copy_points(arr)
            else:
                self.backend.run_copies(arr)

    def reseed(self, salt):
        from iif_amd.seeds import mix_seed
        for kind, arr in self.stages:
            if kind not in (abi.STAGE_COPIES, abi.STAGE_COPY_POINTS):
                for d in arr:
                    d.seed = mix_seed(d.seed, salt)
                    if kind != abi.STAGE_PRODUCTS and d.meas_seed:
                        d.meas_seed = mix_seed(d.meas_seed, salt)

    def close(self):
        pass
