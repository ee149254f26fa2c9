"""Thin object wrapper over the C ABI (include/hsd.h): one Engine = one hsd_ctx = one GPU.

numpy arrays go in and out by pointer; the caller-allocates-outputs convention of
the ABI is kept (this module allocates the numpy outputs on the caller's behalf).
Status codes become the exception types the reference raises on the same
conditions.
"""
from __future__ import annotations

import ctypes as C
import threading
import weakref

import numpy as np

from . import _lib as L


class HsdError(RuntimeError):
    pass


def _raise(lib, ctx, status):
    msg = lib.hsd_last_error(ctx).decode() if ctx else lib.hsd_strerror(status).decode()
    if not msg:
        msg = lib.hsd_strerror(status).decode()
    if status == L.ERR_EMPTY_RING:
        raise ValueError("Distribution can't be empty.")               # scipy _validate_distribution
    if status == L.ERR_NOT_NORMALISED:
        raise ValueError("The sum of probability distribution must be 1.0.")   # tools/metrics.py:51
    if status == L.ERR_UNSUPPORTED:
        raise NotImplementedError(msg)                                  # tools/metrics.py:167
    if status == L.ERR_INVALID:
        raise ValueError(msg)
    if status == L.ERR_NOMEM:
        raise MemoryError(msg)
    raise HsdError(f"{lib.hsd_strerror(status).decode()}: {msg}")


class DeviceMatrix:
    """A row block of the distance matrix that stays in HBM (N=100k -> 80 GB)."""

    def __init__(self, engine, rows, cols, ld, row0):
        self.engine, self.rows, self.cols, self.ld, self.row0 = engine, rows, cols, ld, row0
        self.ptr = None
        self.ptr = engine.malloc(rows * ld * 8)
        engine._matrices.add(self)             # (weak: Engine.close() returns the HBM of matrices that are still alive)

    def free(self):
        if self.ptr:
            ptr, self.ptr = self.ptr, None
            self.engine._matrices.discard(self)
            if self.engine._ctx:
                self.engine.free(ptr)

    # A matrix is up to 80 GB of HBM: it goes back when the object dies or leaves a `with` block, not only on free().
    # NOTE: the release is hipFree, a device-wide synchronisation -- a finalizer that the garbage collector runs in the middle
    # of a timed region or between asynchronous launches stalls them; hot paths (bench.py) free explicitly or use `with`.
    def __del__(self):
        try:
            if self.ptr and self.engine._ctx:
                self.free()
        except Exception:       # noqa: BLE001  (interpreter shutdown: the context may already be gone)
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()
        return False

    def rows_to_host(self, a, b):
        """Rows [a, b) of the block (block-local indices) as a numpy array."""
        out = np.empty((b - a, self.ld), dtype=np.float64)
        self.engine.memcpy_d2h(out, self.ptr + a * self.ld * 8, out.nbytes)
        return out[:, :self.cols]

    def to_host(self):
        return self.rows_to_host(0, self.rows)

    def add(self, other):
        """self += other (same shape), on the device."""
        assert (self.rows, self.cols) == (other.rows, other.cols)
        self.engine._call("hsd_matrix_add", C.c_void_p(self.ptr), self.ld, C.c_void_p(other.ptr), other.ld, self.rows, self.cols)
        return self

    def stats(self):
        """dict(sum, max_abs_diag, max_asym, max, min) computed on the device."""
        s = self.engine.matrix_stats(self.ptr, self.rows, self.cols, self.ld, self.row0)
        return dict(sum=s[0], max_abs_diag=s[1], max_asym=s[2], max=s[3], min=s[4])


class Engine:
    def __init__(self, device=0):
        self.lib = L.load()
        self._ctx = C.c_void_p()
        st = self.lib.hsd_create(int(device), C.byref(self._ctx))
        if st != L.OK:
            raise HsdError(f"hsd_create(device={device}) failed: {self.lib.hsd_strerror(st).decode()} "
                           "(is a ROCm GPU visible? hsd_amd has no CPU fallback)")
        self._lock = threading.RLock()      # the ABI is not re-entrant per context (re-entrant lock: a DeviceMatrix finalizer may run inside _call)
        self._matrices = weakref.WeakSet()  # live DeviceMatrix allocations (hsd_malloc is not tracked by the context itself)
        self.n = 0
        self.device = device
        self._owner = None

    # -- plumbing ---------------------------------------------------------
    def close(self):
        if self._ctx:
            for m in list(self._matrices):  # matrices that outlive the engine would keep their HBM until the process exits
                m.free()
            self.lib.hsd_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001
            pass

    def _call(self, name, *args):
        # the message of a failed call is read under the same lock: another thread's call on this context
        # would overwrite hsd_last_error in between
        with self._lock:
            st = getattr(self.lib, name)(self._ctx, *args)
            if st != L.OK:
                _raise(self.lib, self._ctx, st)

    def _call_into(self, matrix, owned, name, *args):
        """_call for the helpers that fill a DeviceMatrix: one they allocated themselves is freed when the
        library call raises (Variant A raises on empty rings AFTER the allocation: 80 GB at N = 100k)."""
        try:
            self._call(name, *args)
        except BaseException:
            if owned:
                matrix.free()
            raise
        return matrix

    @staticmethod
    def _ptr(a):
        return a.ctypes.data_as(C.c_void_p)

    def device_info(self):
        info = L.DeviceInfo()
        self._call("hsd_get_device_info", C.byref(info))
        return dict(name=info.name.decode(), arch=info.arch.decode(), compute_units=info.compute_units,
                    lds_bytes_per_cu=info.lds_bytes_per_cu, hbm_bytes=info.hbm_bytes, clock_khz=info.clock_khz)

    def sync(self):
        self._call("hsd_sync")

    def malloc(self, nbytes):
        p = C.c_void_p()
        self._call("hsd_malloc", C.c_size_t(int(nbytes)), C.byref(p))
        return p.value

    def free(self, ptr):
        self._call("hsd_free", C.c_void_p(ptr))

    def memcpy_d2h(self, host_array, dev_ptr, nbytes):
        self._call("hsd_memcpy_d2h", self._ptr(host_array), C.c_void_p(dev_ptr), C.c_size_t(int(nbytes)))

    def memcpy_h2d(self, dev_ptr, host_array, nbytes):
        self._call("hsd_memcpy_h2d", C.c_void_p(dev_ptr), self._ptr(host_array), C.c_size_t(int(nbytes)))

    def timer_start(self):
        self._call("hsd_timer_start")

    def timer_stop(self):
        ms = C.c_float()
        self._call("hsd_timer_stop", C.byref(ms))
        return ms.value

    # -- graph / K1 -------------------------------------------------------
    def set_graph(self, rowptr, col):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int32)
        n = rowptr.shape[0] - 1
        self._owner = None              # whatever model had its graph here no longer does (hsd_amd.model rebinding)
        self._call("hsd_set_graph", n, self._ptr(rowptr), self._ptr(col))
        self.n = n

    def khop_sizes(self, hop, r0=0, r1=None):
        r1 = self.n if r1 is None else r1
        out = np.empty((r1 - r0, hop + 1), dtype=np.int32)
        self._call("hsd_khop_sizes", hop, r0, r1, self._ptr(out))
        return out

    def khop_members(self, hop, r0=0, r1=None):
        """(offsets int64[(r1-r0)*(hop+1)+1], members int32) -- members ascending per ring."""
        r1 = self.n if r1 is None else r1
        sizes = self.khop_sizes(hop, r0, r1)
        offsets = np.zeros(sizes.size + 1, dtype=np.int64)
        np.cumsum(sizes.reshape(-1), out=offsets[1:])
        members = np.empty(max(int(offsets[-1]), 1), dtype=np.int32)
        self._call("hsd_khop_members", hop, r0, r1, self._ptr(offsets), self._ptr(members))
        return offsets, members[:int(offsets[-1])]

    # -- K2 ---------------------------------------------------------------
    def signature_build_degree(self, hop, r0=0, r1=None):
        r1 = self.n if r1 is None else r1
        self._call("hsd_signature_build_degree", hop, r0, r1)

    def signature_build(self, hop, node_bin, support, r0=0, r1=None):
        r1 = self.n if r1 is None else r1
        node_bin = np.ascontiguousarray(node_bin, dtype=np.int32)
        support = np.ascontiguousarray(support, dtype=np.float64)
        if node_bin.shape[0] != self.n:
            raise ValueError("node_bin must have one entry per node")
        self._call("hsd_signature_build", hop, self._ptr(node_bin), support.shape[0], self._ptr(support), r0, r1)

    def signature_dims(self):
        hop, nb = C.c_int(), C.c_int()
        self._call("hsd_signature_dims", C.byref(hop), C.byref(nb))
        return hop.value, nb.value

    def signature_support(self):
        _, nb = self.signature_dims()
        out = np.empty(nb, dtype=np.float64)
        self._call("hsd_signature_support", self._ptr(out))
        return out

    def signature_counts(self, r0=0, r1=None):
        r1 = self.n if r1 is None else r1
        hop, nb = self.signature_dims()
        out = np.empty((r1 - r0, hop + 1, nb), dtype=np.uint32)
        self._call("hsd_signature_get_counts", r0, r1, self._ptr(out))
        return out

    def signature_set_counts(self, r0, r1, counts):
        counts = np.ascontiguousarray(counts, dtype=np.uint32)
        hop, nb = self.signature_dims()
        if counts.shape != (r1 - r0, hop + 1, nb):
            raise ValueError(f"counts must have shape {(r1 - r0, hop + 1, nb)}")
        self._call("hsd_signature_set_counts", r0, r1, self._ptr(counts))

    # -- C1 ---------------------------------------------------------------
    def comm_unique_id(self):
        buf = (C.c_char * 128)()
        st = self.lib.hsd_comm_unique_id(buf)
        if st != L.OK:
            _raise(self.lib, None, st)
        return bytes(buf)

    def comm_init(self, world, rank, unique_id):
        buf = C.create_string_buffer(unique_id, 128) if unique_id is not None else None
        self._call("hsd_comm_init", world, rank, buf)

    def comm_shard(self):
        a, b = C.c_int64(), C.c_int64()
        self._call("hsd_comm_shard", C.byref(a), C.byref(b))
        return a.value, b.value

    def signature_allgather(self):
        self._call("hsd_signature_allgather")

    # -- K3 ---------------------------------------------------------------
    def pair_prepare(self, metric, align, layer_lo, layer_hi, kl_alpha=0.5, kl_sorted=False):
        self._call("hsd_pair_prepare", metric, align, layer_lo, layer_hi, C.c_double(kl_alpha), int(bool(kl_sorted)))

    def pair_shard(self, world, rank):
        """Rows [r0, r1) of rank `rank` when `world` ranks share the prepared distance in row blocks of equal planned cost."""
        a, b = C.c_int64(), C.c_int64()
        self._call("hsd_pair_shard", int(world), int(rank), C.byref(a), C.byref(b))
        return a.value, b.value

    def pair_operand_dims(self):
        a, b = C.c_int(), C.c_int()
        self._call("hsd_pair_operand_dims", C.byref(a), C.byref(b))
        return a.value, b.value

    def pair_work(self):
        """operand columns x pairs evaluated by the last pair_rows call (synchronises)."""
        out = C.c_double()
        self._call("hsd_pair_work", C.byref(out))
        return out.value

    def pair_rows_host(self, r0=0, r1=None):
        r1 = self.n if r1 is None else r1
        out = np.empty((r1 - r0, self.n), dtype=np.float64)
        if r1 > r0:
            self._call("hsd_pair_rows", r0, r1, self._ptr(out), self.n, 0)
        return out

    def pair_topk(self, k, r0=0, r1=None):
        """(idx int32[(r1-r0), k], dist float64[(r1-r0), k]): the k nearest other nodes of every row,
        ascending by (distance, index); the N x N matrix is never written."""
        r1 = self.n if r1 is None else r1
        idx = np.empty((r1 - r0, k), dtype=np.int32)
        dist = np.empty((r1 - r0, k), dtype=np.float64)
        if r1 > r0:
            self._call("hsd_pair_topk", k, r0, r1, self._ptr(idx), self._ptr(dist))
        return idx, dist

    def pair_rows_device(self, r0=0, r1=None, matrix=None):
        """Asynchronous: D[r0:r1, :] into a resident DeviceMatrix (allocated if not given)."""
        r1 = self.n if r1 is None else r1
        owned = matrix is None
        if owned:
            ld = (self.n + 15) // 16 * 16       # rows start on 128-byte lines: every store of the pair kernel is a whole line
            matrix = DeviceMatrix(self, r1 - r0, self.n, ld, r0)
        return self._call_into(matrix, owned, "hsd_pair_rows", r0, r1, C.c_void_p(matrix.ptr), matrix.ld, 1)

    # -- K4 / heat-coefficient signatures -----------------------------------
    def set_laplacian(self, rowptr, col, val):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val = np.ascontiguousarray(val, dtype=np.float64)
        self._call("hsd_set_laplacian", rowptr.shape[0] - 1, self._ptr(rowptr), self._ptr(col), self._ptr(val))

    def heat_wavelets_host(self, coeff, lmax, threshold, c0=0, c1=None):
        """Rows [c0, c1) of psi as a numpy array (threshold < 0: no thresholding)."""
        c1 = self.n if c1 is None else c1
        coeff = np.ascontiguousarray(coeff, dtype=np.float64)
        out = np.empty((c1 - c0, self.n), dtype=np.float64)
        if c1 > c0:
            self._call("hsd_heat_wavelets", self._ptr(coeff), coeff.shape[0] - 1, C.c_double(lmax), C.c_double(threshold),
                       c0, c1, self._ptr(out), self.n, 0)
        return out

    def heat_wavelets_device(self, coeff, lmax, threshold, c0=0, c1=None):
        """Rows [c0, c1) of psi resident on the device (DeviceMatrix, ld = n)."""
        c1 = self.n if c1 is None else c1
        coeff = np.ascontiguousarray(coeff, dtype=np.float64)
        m = DeviceMatrix(self, c1 - c0, self.n, self.n, c0)
        return self._call_into(m, True, "hsd_heat_wavelets", self._ptr(coeff), coeff.shape[0] - 1, C.c_double(lmax),
                               C.c_double(threshold), c0, c1, C.c_void_p(m.ptr), m.ld, 1)

    def heat_wavelets_multi_device(self, coeffs, lmax, threshold):
        """coeffs (n_scales, order+1) -> list of resident psi matrices, one Chebyshev recurrence for all."""
        coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
        S = coeffs.shape[0]
        mats = [DeviceMatrix(self, self.n, self.n, self.n, 0) for _ in range(S)]
        ptrs = (C.c_void_p * S)(*[m.ptr for m in mats])
        try:
            self._call("hsd_heat_wavelets_multi", self._ptr(coeffs), S, coeffs.shape[1] - 1, C.c_double(lmax), C.c_double(threshold),
                       ptrs, self.n)
        except Exception:
            for m in mats:
                m.free()
            raise
        return mats

    def ring_stats(self, hop, psi_rows, r0=0, r1=None):
        """(sums float64[(r1-r0), hop+1], sizes int32[...]) for psi rows r0..r1 held in `psi_rows`."""
        r1 = self.n if r1 is None else r1
        sums = np.empty((r1 - r0, hop + 1), dtype=np.float64)
        sizes = np.empty((r1 - r0, hop + 1), dtype=np.int32)
        self._call("hsd_ring_stats", hop, C.c_void_p(psi_rows.ptr), psi_rows.ld, r0, r1, self._ptr(sums), self._ptr(sizes))
        return sums, sizes

    def multiscale_ring_stats(self, hop, coeffs, lmax, threshold, r0=0, r1=None):
        """coeffs: (n_scales, order+1).  Returns (sums float64[n_scales, r1-r0, hop+1], sizes int32[r1-r0, hop+1])."""
        r1 = self.n if r1 is None else r1
        coeffs = np.ascontiguousarray(coeffs, dtype=np.float64)
        S, m1 = coeffs.shape
        sums = np.empty((S, r1 - r0, hop + 1), dtype=np.float64)
        sizes = np.empty((r1 - r0, hop + 1), dtype=np.int32)
        self._call("hsd_multiscale_ring_stats", hop, self._ptr(coeffs), S, m1 - 1, C.c_double(lmax), C.c_double(threshold),
                   r0, r1, self._ptr(sums), self._ptr(sizes))
        return sums, sizes

    def heat_signature_build(self, hop, psi):
        self._call("hsd_heat_signature_build", hop, C.c_void_p(psi.ptr), psi.ld)

    def pair_heat_rows_host(self, layer_lo, layer_hi, r0=0, r1=None):
        r1 = self.n if r1 is None else r1
        out = np.empty((r1 - r0, self.n), dtype=np.float64)
        if r1 > r0:
            self._call("hsd_pair_heat_rows", layer_lo, layer_hi, r0, r1, self._ptr(out), self.n, 0)
        return out

    def pair_heat_rows_device(self, layer_lo, layer_hi, r0=0, r1=None, matrix=None):
        """Variant A on the heat-coefficient samples into a resident DeviceMatrix."""
        r1 = self.n if r1 is None else r1
        owned = matrix is None
        if owned:
            matrix = DeviceMatrix(self, r1 - r0, self.n, self.n, r0)
        return self._call_into(matrix, owned, "hsd_pair_heat_rows", layer_lo, layer_hi, r0, r1, C.c_void_p(matrix.ptr), matrix.ld, 1)

    def pair_heat_rowform_device(self, psi, r0=0, r1=None, matrix=None):
        """Variant B as written (both samples from row startIndex of psi) into a resident DeviceMatrix."""
        r1 = self.n if r1 is None else r1
        owned = matrix is None
        if owned:
            matrix = DeviceMatrix(self, r1 - r0, self.n, self.n, r0)
        return self._call_into(matrix, owned, "hsd_pair_heat_rowform", C.c_void_p(psi.ptr), psi.ld, r0, r1,
                               C.c_void_p(matrix.ptr), matrix.ld, 1)

    def pair_heat_rowform_host(self, psi, r0=0, r1=None):
        r1 = self.n if r1 is None else r1
        out = np.empty((r1 - r0, self.n), dtype=np.float64)
        if r1 > r0:
            self._call("hsd_pair_heat_rowform", C.c_void_p(psi.ptr), psi.ld, r0, r1, self._ptr(out), self.n, 0)
        return out

    def sample_distance(self, p, q, metric):
        p = np.ascontiguousarray(p, dtype=np.float64).reshape(-1)
        q = np.ascontiguousarray(q, dtype=np.float64).reshape(-1)
        out = C.c_double()
        self._call("hsd_sample_distance", self._ptr(p), p.size, self._ptr(q), q.size, metric, C.byref(out))
        return out.value

    def upload_matrix(self, a):
        """numpy (rows, n) float64 -> DeviceMatrix (caller-assigned wavelets)."""
        a = np.ascontiguousarray(a, dtype=np.float64)
        m = DeviceMatrix(self, a.shape[0], a.shape[1], a.shape[1], 0)
        return self._call_into(m, True, "hsd_memcpy_h2d", C.c_void_p(m.ptr), self._ptr(a), C.c_size_t(int(a.nbytes)))

    def matrix_stats(self, dev_ptr, rows, cols, ld, row0):
        out = np.empty(5, dtype=np.float64)
        self._call("hsd_matrix_stats", C.c_void_p(dev_ptr), rows, cols, ld, row0, self._ptr(out))
        return out
