"""Device backend: thin object wrapper over the libnbp C ABI (include/nbp.h).

The host-side mirror of the reference API (factorgraph.py / solver.py) talks to a *backend*
through this narrow interface: belief slots in, descriptor batches run, belief slots out.
The product backend is :class:`HipBackend`.  (The CPU oracle implements the same interface in
``oracle/`` -- test infrastructure, never imported from this package.)
"""
import ctypes as C
import weakref

import numpy as np

from . import abi


class NbpError(RuntimeError):
    """Hard error from libnbp (status < 0).  The Julia shim maps this to `error()`, which fails
    the clique Task and makes `monitorCSMs` tear the solve down with a CompositeException
    (reference: CliqStateMachineUtils.jl:184-246, test/testCSMMonitor.jl:51)."""


def _as_array(descs, ctype):
    if isinstance(descs, C.Array):
        return descs, len(descs)
    arr = (ctype * len(descs))(*descs)
    return arr, len(descs)


class HipBackend:
    name = "hip"

    def __init__(self, N, n_slots, side_ints=0, device=0, arena_ptr=None, arena_bytes=0):
        self.lib = abi.load_library()
        self.N, self.n_slots = int(N), int(n_slots)
        self._ctx = C.c_void_p()
        self._programs = weakref.WeakSet()  # live HipPrograms: closed before the context (close())
        self._check(self.lib.nbp_ctx_create(device, self.N, self.n_slots, arena_ptr, arena_bytes,
                                            max(int(side_ints), 1), C.byref(self._ctx)))

    def _check(self, rc):
        if rc != 0:
            msg = self.lib.nbp_last_error()
            raise NbpError(f"libnbp status {rc}: {msg.decode() if msg else ''}")

    def close(self):
        if self._ctx:
            for prog in list(getattr(self, "_programs", ())):  # a program must not outlive its context
                prog.close()
            self.comm_destroy()
            self.lib.nbp_ctx_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- belief I/O -----------------------------------------------------------------------
    def slot_write(self, slot, manifold, pts, bw=None):
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(self.N, abi.MANIFOLD_P[manifold])
        bwp = None
        if bw is not None:
            bw = np.ascontiguousarray(bw, dtype=np.float64)
            bwp = bw.ctypes.data_as(C.POINTER(C.c_double))
        self._check(self.lib.nbp_slot_write(self._ctx, slot, manifold,
                                            pts.ctypes.data_as(C.POINTER(C.c_double)), bwp))

    def slot_read(self, slot, manifold):
        pts = np.empty((self.N, abi.MANIFOLD_P[manifold]))
        bw = np.empty(abi.MANIFOLD_DIM[manifold])
        self._check(self.lib.nbp_slot_read(self._ctx, slot, manifold,
                                           pts.ctypes.data_as(C.POINTER(C.c_double)),
                                           bw.ctypes.data_as(C.POINTER(C.c_double))))
        return pts, bw

    def belief_write(self, slot, manifold, pts, bw=None, ipc=None):
        """the full TreeBelief triple (val, bw, infoPerCoord)"""
        dp = C.POINTER(C.c_double)
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, abi.MANIFOLD_P[manifold])
        bw = None if bw is None else np.ascontiguousarray(bw, dtype=np.float64)
        ipc = None if ipc is None else np.ascontiguousarray(ipc, dtype=np.float64)
        self._check(self.lib.nbp_belief_write(self._ctx, slot, manifold, pts.ctypes.data_as(dp), pts.shape[0],
                                              bw.ctypes.data_as(dp) if bw is not None else None,
                                              ipc.ctypes.data_as(dp) if ipc is not None else None))

    def belief_read(self, slot, manifold):
        dp = C.POINTER(C.c_double)
        pts = np.empty((self.N, abi.MANIFOLD_P[manifold]))
        bw, ipc = np.empty(abi.MANIFOLD_DIM[manifold]), np.empty(abi.MANIFOLD_DIM[manifold])
        n = C.c_int32(0)
        self._check(self.lib.nbp_belief_read(self._ctx, slot, manifold, pts.ctypes.data_as(dp), C.byref(n),
                                             bw.ctypes.data_as(dp), ipc.ctypes.data_as(dp)))
        return pts[:n.value], bw, ipc

    def beliefs_write(self, slots, manifolds, beliefs):
        """many TreeBeliefs in one call (nbp_belief_write_batch): beliefs = [(pts, bw or None, ipc or None), ...]"""
        dp, n = C.POINTER(C.c_double), len(slots)
        keep, P, B, I, cnt = [], (dp * n)(), (dp * n)(), (dp * n)(), (C.c_int32 * n)()
        for i, (pts, bw, ipc) in enumerate(beliefs):
            pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, abi.MANIFOLD_P[manifolds[i]])
            keep.append(pts)
            P[i], cnt[i] = pts.ctypes.data_as(dp), pts.shape[0]
            for arr, tab in ((bw, B), (ipc, I)):
                if arr is not None:
                    arr = np.ascontiguousarray(arr, dtype=np.float64)
                    keep.append(arr)
                    tab[i] = arr.ctypes.data_as(dp)
        self._check(self.lib.nbp_belief_write_batch(self._ctx, n, (C.c_int32 * n)(*slots), (C.c_int32 * n)(*manifolds), P, cnt, B, I))
        # (packed into the staging buffer before the call returns: `keep` may go)

    def beliefs_read(self, slots, manifolds):
        """-> [(pts, bw, ipc), ...] (nbp_belief_read_batch)"""
        dp, n = C.POINTER(C.c_double), len(slots)
        P, B, I, cnt = (dp * n)(), (dp * n)(), (dp * n)(), (C.c_int32 * n)()
        out = []
        for i, m in enumerate(manifolds):
            pts, bw, ipc = np.empty((self.N, abi.MANIFOLD_P[m])), np.empty(abi.MANIFOLD_DIM[m]), np.empty(abi.MANIFOLD_DIM[m])
            out.append((pts, bw, ipc))
            P[i], B[i], I[i] = pts.ctypes.data_as(dp), bw.ctypes.data_as(dp), ipc.ctypes.data_as(dp)
        self._check(self.lib.nbp_belief_read_batch(self._ctx, n, (C.c_int32 * n)(*slots), (C.c_int32 * n)(*manifolds), P, cnt, B, I))
        return [(p[:cnt[i]], b, q) for i, (p, b, q) in enumerate(out)]

    def side_write(self, offset, ints):
        a = np.ascontiguousarray(ints, dtype=np.int32)
        self._check(self.lib.nbp_side_write(self._ctx, offset, a.ctypes.data_as(C.POINTER(C.c_int32)), a.size))

    def side_read(self, offset, n):
        a = np.empty(n, dtype=np.int32)
        self._check(self.lib.nbp_side_read(self._ctx, offset, a.ctypes.data_as(C.POINTER(C.c_int32)), n))
        return a

    # ---- op batches -----------------------------------------------------------------------
    def run_proposals(self, descs):
        arr, n = _as_array(descs, abi.ProposalDesc)
        self._check(self.lib.nbp_run_proposals(self._ctx, arr, n))

    def run_products(self, descs):
        arr, n = _as_array(descs, abi.ProductDesc)
        self._check(self.lib.nbp_run_products(self._ctx, arr, n))

    def run_copies(self, descs):
        arr, n = _as_array(descs, abi.CopyDesc)
        self._check(self.lib.nbp_run_copies(self._ctx, arr, n))

    def run_deconv(self, descs, meas_slots=None):
        arr, n = _as_array(descs, abi.ProposalDesc)
        ms = np.ascontiguousarray(meas_slots if meas_slots is not None else [-1] * n, dtype=np.int32)
        self._check(self.lib.nbp_run_deconv(self._ctx, arr, ms.ctypes.data_as(C.POINTER(C.c_int32)), n))

    def run_bandwidth(self, slots, manifolds):
        s = np.ascontiguousarray(slots, dtype=np.int32)
        m = np.ascontiguousarray(manifolds, dtype=np.int32)
        ip = C.POINTER(C.c_int32)
        self._check(self.lib.nbp_run_bandwidth(self._ctx, s.ctypes.data_as(ip), m.ctypes.data_as(ip), s.size))

    def run_resample(self, slots, manifolds, seed=0):
        """sample(oldBel, N - Npts): top beliefs with fewer than N points up to N, in place"""
        s = np.ascontiguousarray(slots, dtype=np.int32)
        m = np.ascontiguousarray(manifolds, dtype=np.int32)
        ip = C.POINTER(C.c_int32)
        self._check(self.lib.nbp_run_resample(self._ctx, s.ctypes.data_as(ip), m.ctypes.data_as(ip), s.size, C.c_uint64(seed)))

    # ---- host-buffer entry points (one call per reference function) -------------------------------
    def kde_bandwidth(self, manifold, pts):
        pts = np.ascontiguousarray(pts, dtype=np.float64)
        bw = np.zeros(abi.MANIFOLD_DIM[manifold])
        dp = C.POINTER(C.c_double)
        self._check(self.lib.nbp_kde_bandwidth(self._ctx, manifold, pts.ctypes.data_as(dp), bw.ctypes.data_as(dp)))
        return bw

    def conv(self, desc, var_pts, var_bw=None, mhidx_in=None, want_mhidx=False, want_bw=True):
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        man = desc.manifold
        pts = [np.ascontiguousarray(p, dtype=np.float64) for p in var_pts]
        bws = [None if (var_bw is None or b is None) else np.ascontiguousarray(b, dtype=np.float64) for b in (var_bw or [None] * len(pts))]
        ppts = (dp * len(pts))(*[p.ctypes.data_as(dp) for p in pts])
        pbw = (dp * len(pts))(*[(b.ctypes.data_as(dp) if b is not None else C.cast(None, dp)) for b in bws])
        out = np.zeros((self.N, abi.MANIFOLD_P[man]))
        obw = np.zeros(abi.MANIFOLD_DIM[man])
        mh_in = None if mhidx_in is None else np.ascontiguousarray(mhidx_in, dtype=np.int32)
        mh_out = np.zeros(self.N, dtype=np.int32) if want_mhidx else None
        self._check(self.lib.nbp_conv(self._ctx, C.byref(desc), ppts, pbw,
                                      mh_in.ctypes.data_as(ip) if mh_in is not None else C.cast(None, ip),
                                      out.ctypes.data_as(dp), obw.ctypes.data_as(dp) if want_bw else C.cast(None, dp),
                                      mh_out.ctypes.data_as(ip) if want_mhidx else C.cast(None, ip)))
        return (out, obw, mh_out) if want_mhidx else (out, obw)

    def manifold_product(self, manifold, dens, seed, niter=1, partial_masks=None, old_pts=None, want_labels=False):
        """dens: list of (pts N x P, bw D)"""
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int32)
        F = len(dens)
        pts = [np.ascontiguousarray(p, dtype=np.float64) for p, _ in dens]
        bws = [np.ascontiguousarray(b, dtype=np.float64) for _, b in dens]
        ppts = (dp * F)(*[p.ctypes.data_as(dp) for p in pts])
        pbw = (dp * F)(*[b.ctypes.data_as(dp) for b in bws])
        masks = None if partial_masks is None else np.ascontiguousarray(partial_masks, dtype=np.uint8)
        old = None if old_pts is None else np.ascontiguousarray(old_pts, dtype=np.float64)
        out = np.zeros((self.N, abi.MANIFOLD_P[manifold]))
        obw = np.zeros(abi.MANIFOLD_DIM[manifold])
        lab = np.zeros(self.N * F, dtype=np.int32) if want_labels else None
        self._check(self.lib.nbp_manifold_product(
            self._ctx, manifold, F, ppts, pbw,
            masks.ctypes.data_as(C.POINTER(C.c_uint8)) if masks is not None else C.cast(None, C.POINTER(C.c_uint8)),
            old.ctypes.data_as(dp) if old is not None else C.cast(None, dp), niter, C.c_uint64(seed),
            out.ctypes.data_as(dp), obw.ctypes.data_as(dp), lab.ctypes.data_as(ip) if want_labels else C.cast(None, ip)))
        return (out, obw, lab) if want_labels else (out, obw)

    def synchronize(self):
        self._check(self.lib.nbp_synchronize(self._ctx))

    # ---- separator exchange between ranks: RCCL point-to-point from C, on the library stream -----------------------------
    def comm_unique_id(self):
        buf = (C.c_char * abi.COMM_ID_BYTES)()
        self._check(self.lib.nbp_comm_unique_id(buf))
        return bytes(buf)

    def comm_create(self, world, rank, uid):
        h = C.c_void_p()
        buf = (C.c_char * abi.COMM_ID_BYTES).from_buffer_copy(uid)
        self._check(self.lib.nbp_comm_create(self._ctx, world, rank, buf, C.byref(h)))
        self._comm = h
        return h

    def comm_info(self):
        """(nranks, rank) as RCCL reports them for the library's communicator"""
        n, r = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.nbp_comm_info(self._comm, C.byref(n), C.byref(r)))
        return n.value, r.value

    def math_eval(self, fn, a, b=None):
        """the shared elementary functions (include/nbp_math.h) evaluated on the device: (out0, out1)"""
        dp = C.POINTER(C.c_double)
        a = np.ascontiguousarray(a, dtype=np.float64)
        b = None if b is None else np.ascontiguousarray(b, dtype=np.float64)
        o0, o1 = np.empty_like(a), np.empty_like(a)
        self._check(self.lib.nbp_math_eval(self._ctx, fn, a.ctypes.data_as(dp), b.ctypes.data_as(dp) if b is not None else C.cast(None, dp),
                                           o0.ctypes.data_as(dp), o1.ctypes.data_as(dp), a.size))
        return o0, o1

    def comm_destroy(self):
        if getattr(self, "_comm", None):
            self.lib.nbp_comm_destroy(self._comm)
            self._comm = None

    def exchange(self, sends, recvs):
        """one grouped ncclSend / ncclRecv of whole slots: sends / recvs = [(peer rank, slot)]; asynchronous (stream-ordered)"""
        sx = (abi.Xfer * max(1, len(sends)))(*[abi.Xfer(p, s) for p, s in sends])
        rx = (abi.Xfer * max(1, len(recvs)))(*[abi.Xfer(p, s) for p, s in recvs])
        self._check(self.lib.nbp_exchange(self._ctx, self._comm, sx, len(sends), rx, len(recvs)))

    # ---- resident programs (clique seam) -----------------------------------------------------
    def program(self, stages, lazy_bandwidth=False, fused_updates=True):
        return HipProgram(self, stages, lazy_bandwidth, fused_updates)

    def timing_enable(self, on=True):
        self._check(self.lib.nbp_timing_enable(self._ctx, int(on)))

    KERNELS = ("nbp_proposal_kernel", "nbp_prep_kernel", "nbp_product_kernel", "nbp_bandwidth_kernel", "nbp_update_kernel")

    def timing_read(self):
        """{kernel: (total ms, launches)} measured with HIP events on the library stream"""
        ms = (C.c_double * 5)()
        nl = (C.c_int64 * 5)()
        self._check(self.lib.nbp_timing_read_n(self._ctx, ms, nl, 5))
        return {k: (ms[i], nl[i]) for i, k in enumerate(self.KERNELS)}

    def diag(self, reset=False):
        d = abi.Diag()
        self._check(self.lib.nbp_diag_read(self._ctx, C.byref(d), int(reset)))
        return {k: getattr(d, k) for k, _ in abi.Diag._fields_}

    def arena_ptr(self):
        return self.lib.nbp_arena_ptr(self._ctx)

    def stream_ptr(self):
        return self.lib.nbp_stream_ptr(self._ctx)


_STAGE_CTYPE = {abi.STAGE_PROPOSALS: abi.ProposalDesc, abi.STAGE_PRODUCTS: abi.ProductDesc,
                abi.STAGE_COPIES: abi.CopyDesc, abi.STAGE_DECONV: abi.ProposalDesc,
                abi.STAGE_COPY_POINTS: abi.CopyDesc}


class HipProgram:
    """A device-resident schedule: list of (kind, descriptor-array) stages uploaded once."""

    def __init__(self, backend, stages, lazy_bandwidth=False, fused_updates=True):
        self.backend, lib = backend, backend.lib
        self._p = C.c_void_p()
        backend._check(lib.nbp_program_create(backend._ctx, C.byref(self._p)))
        backend._programs.add(self)
        if lazy_bandwidth:  # whole-solve programs: intermediate bandwidths nobody reads are not fitted
            backend._check(lib.nbp_program_set_option(self._p, abi.OPT_LAZY_BANDWIDTH, 1))
        if not fused_updates:  # every round as three launches (proposal, prep, product)
            backend._check(lib.nbp_program_set_option(self._p, abi.OPT_FUSED_UPDATES, 0))
        for kind, descs in stages:
            arr, n = _as_array(descs, _STAGE_CTYPE[kind])
            backend._check(lib.nbp_program_add_stage(self._p, kind, C.cast(arr, C.c_void_p), n))
        backend._check(lib.nbp_program_finalize(self._p))
        self.n_stages = len(stages)

    def num_fused(self):
        """rounds (PROPOSALS + PRODUCTS stage pairs) that run as one launch of the fused update kernel"""
        n = C.c_int32(0)
        self.backend._check(self.backend.lib.nbp_program_num_fused(self._p, C.byref(n)))
        return n.value

    def num_two_stream(self):
        """rounds whose two halves run on two streams one launch apart (NBP_PIPELINE_MIN)"""
        n = C.c_int32(0)
        self.backend._check(self.backend.lib.nbp_program_num_two_stream(self._p, C.byref(n)))
        return n.value

    def run(self, first=0, last=-1):
        self.backend._check(self.backend.lib.nbp_program_run(self._p, first, last))

    def reseed(self, salt):
        self.backend._check(self.backend.lib.nbp_program_reseed(self._p, C.c_uint64(salt)))

    def num_seeds(self):
        n = C.c_int32(0)
        self.backend._check(self.backend.lib.nbp_program_num_seeds(self._p, C.byref(n)))
        return n.value

    def set_seeds(self, seeds):
        """new seeds for every op, in stage order: per proposal / deconv descriptor its seed and (where the descriptor named a
        stored measurement at finalize) its meas_seed behind it; per product descriptor its seed (nbp_program_set_seeds)"""
        arr = (C.c_uint64 * len(seeds))(*[int(x) for x in seeds])
        self.backend._check(self.backend.lib.nbp_program_set_seeds(self._p, arr, len(seeds)))

    @staticmethod
    def seeds_of(stages):
        """the seed list set_seeds() takes, read off a list of (kind, descriptors) stages"""
        out = []
        for kind, descs in stages:
            if kind in (abi.STAGE_PROPOSALS, abi.STAGE_DECONV):
                for d in descs:
                    out.append(d.seed)
                    if d.meas_seed:
                        out.append(d.meas_seed)
            elif kind == abi.STAGE_PRODUCTS:
                out.extend(d.seed for d in descs)
        return out

    def close(self):
        if self._p:
            self.backend.lib.nbp_program_destroy(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
