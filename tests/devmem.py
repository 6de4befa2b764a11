"""Test helper: drive the device-level C-ABI (mxd_*) directly with library-managed device memory
(mx_dev_malloc / mx_memcpy_*), no torch involved."""
import ctypes as C

import numpy as np

from matrixextra_amd import _lib
from matrixextra_amd._lib import check


class Dev:
    """A numpy array mirrored into device memory for the lifetime of the object."""

    def __init__(self, arr=None, nbytes=None):
        lib = _lib.load()
        self.host = None if arr is None else np.ascontiguousarray(arr)
        self.nbytes = int(nbytes if arr is None else self.host.nbytes)
        self.ptr = C.c_void_p()
        check(lib.mx_dev_malloc(C.byref(self.ptr), C.c_size_t(max(self.nbytes, 16))))
        if self.host is not None and self.nbytes:
            # synchronous copies through the library's transfer engine (what the exports use): hipMemcpyAsync from / to
            # ordinary numpy memory makes the runtime pin caller pages on the fly, and the long fuzz loops of round 3 ended
            # now and then in a GPU fault at a heap address with those in the mix
            check(lib.mx_upload(self.ptr, C.c_void_p(self.host.ctypes.data), C.c_size_t(self.nbytes)))

    def download(self, dtype, shape):
        lib = _lib.load()
        out = np.empty(shape, dtype=dtype)
        if out.nbytes:
            check(lib.mx_stream_sync(None))
            check(lib.mx_download(C.c_void_p(out.ctypes.data), self.ptr, C.c_size_t(out.nbytes)))
        return out

    def __del__(self):
        try:
            _lib.load().mx_dev_free(self.ptr)
        except Exception:
            pass


def spmm_device(p, j, x, B_rowmajor, colmajor, algo, rows_sorted, npanels=0, wg_per_cu=0):
    """C = A @ B through mxd_spmm_csr_dense_ex; returns C as an (m, n) numpy array."""
    lib = _lib.load()
    m, (K, n) = p.size - 1, B_rowmajor.shape
    dt = _lib.MX_F64 if B_rowmajor.dtype == np.float64 else _lib.MX_F32
    dp, dj, dx, dB = Dev(p.astype(np.int32)), Dev(j.astype(np.int32)), Dev(x.astype(np.float64)), Dev(B_rowmajor)
    dC = Dev(nbytes=m * n * B_rowmajor.dtype.itemsize)
    check(lib.mx_dev_memset(dC.ptr, 0xFF, C.c_size_t(dC.nbytes), None))      # poison: every element must be written
    check(lib.mxd_spmm_csr_dense_ex(C.c_int(m), C.c_int(n), C.c_int(K), dp.ptr, dj.ptr, dx.ptr, dB.ptr, C.c_size_t(n),
                                    dC.ptr, C.c_size_t(m if colmajor else n), C.c_int(dt), C.c_int(int(colmajor)),
                                    C.c_int(algo), C.c_int(int(rows_sorted)), C.c_int(npanels), C.c_int(wg_per_cu), None))
    check(lib.mx_stream_sync(None))
    out = dC.download(B_rowmajor.dtype, (n, m) if colmajor else (m, n))
    return out.T if colmajor else out


def spmm_planned_device(p, j, x, B_rowmajor, colmajor, npanels=0, wg_per_cu=0, sync_mode=-1):
    """C = A @ B through mxd_spmm_plan_create / mxd_spmm_plan_run; returns C as an (m, n) numpy array."""
    lib = _lib.load()
    m, (K, n) = p.size - 1, B_rowmajor.shape
    dt = _lib.MX_F64 if B_rowmajor.dtype == np.float64 else _lib.MX_F32
    dp, dj, dx, dB = Dev(p.astype(np.int32)), Dev(j.astype(np.int32)), Dev(x.astype(np.float64)), Dev(B_rowmajor)
    dC = Dev(nbytes=m * n * B_rowmajor.dtype.itemsize)
    check(lib.mx_dev_memset(dC.ptr, 0xFF, C.c_size_t(dC.nbytes), None))
    plan = C.c_void_p()
    check(lib.mxd_spmm_plan_create(C.c_int(m), C.c_int(K), dp.ptr, dj.ptr, dx.ptr, C.c_int(npanels), None, C.byref(plan)))
    try:
        check(lib.mxd_spmm_plan_run(plan, C.c_int(n), dB.ptr, C.c_size_t(n), dC.ptr, C.c_size_t(m if colmajor else n),
                                    C.c_int(dt), C.c_int(int(colmajor)), C.c_int(wg_per_cu), C.c_int(sync_mode), None))
        check(lib.mx_stream_sync(None))
    finally:
        lib.mxd_spmm_plan_destroy(plan)
    out = dC.download(B_rowmajor.dtype, (n, m) if colmajor else (m, n))
    return out.T if colmajor else out


def spmv_device(p, j, x, v, v_dtype, algo=0):
    """y = A @ v through mxd_spmv_csr_dvec_ex (algo 0 auto, 1 lane-group kernel, 2 LDS-panel tile kernel)."""
    lib = _lib.load()
    m, K = p.size - 1, v.size
    dp, dj, dx, dv = Dev(p.astype(np.int32)), Dev(j.astype(np.int32)), Dev(x.astype(np.float64)), Dev(v)
    odt = np.float32 if v_dtype == _lib.MX_F32 else np.float64
    dy = Dev(nbytes=m * np.dtype(odt).itemsize)
    check(lib.mx_dev_memset(dy.ptr, 0xFF, C.c_size_t(dy.nbytes), None))      # poison: every row must be written
    check(lib.mxd_spmv_csr_dvec_ex(C.c_int(m), C.c_int(K), C.c_int64(int(p[-1])), dp.ptr, dj.ptr, dx.ptr, dv.ptr,
                                   C.c_int(v_dtype), dy.ptr, C.c_int(algo), None))
    check(lib.mx_stream_sync(None))
    return dy.download(odt, (m,))


def merge_fused_device(op, p1, j1, x1, p2, j2, x2):
    """A (op) B through the one-pass kernel (mxd_csr_merge_fused); returns (indptr, indices, values)."""
    lib = _lib.load()
    m = p1.size - 1
    logical = op in (_lib.MX_OP_OR, _lib.MX_OP_XOR, _lib.MX_OP_AND)
    vdt = np.int32 if logical else np.float64
    n1, n2 = int(p1[-1]), int(p2[-1])
    bound = min(n1, n2) if op in (_lib.MX_OP_MUL, _lib.MX_OP_AND) else n1 + n2
    d = [Dev(np.asarray(a, dtype=t)) for a, t in ((p1, np.int32), (j1, np.int32), (x1, vdt), (p2, np.int32), (j2, np.int32),
                                                   (x2, vdt))]
    dp, dj, dx = Dev(nbytes=4 * (m + 1)), Dev(nbytes=4 * max(bound, 1)), Dev(nbytes=np.dtype(vdt).itemsize * max(bound, 1))
    ws = Dev(nbytes=int(lib.mxd_merge_fused_workspace_bytes(C.c_int(m))))
    nnz = C.c_int64(-1)
    check(lib.mxd_csr_merge_fused(C.c_int(op), C.c_int(m), d[0].ptr, d[1].ptr, d[2].ptr, C.c_int64(n1), d[3].ptr, d[4].ptr,
                                  d[5].ptr, C.c_int64(n2), dp.ptr, dj.ptr, dx.ptr, ws.ptr, C.byref(nnz), None))
    n = int(nnz.value)
    return dp.download(np.int32, (m + 1,)), dj.download(np.int32, (max(bound, 1),))[:n], dx.download(vdt, (max(bound, 1),))[:n]


def spmv_plan_device(p, j, x, vectors):
    """[A @ v for (v, v_dtype) in vectors] through ONE planned-SpMV plan (mxd_spmv_plan_create / _run)."""
    lib = _lib.load()
    m = p.size - 1
    K = vectors[0][0].size
    dp, dj, dx = Dev(p.astype(np.int32)), Dev(j.astype(np.int32)), Dev(x.astype(np.float64))
    plan = C.c_void_p()
    check(lib.mxd_spmv_plan_create(C.c_int(m), C.c_int(K), dp.ptr, dj.ptr, dx.ptr, None, C.byref(plan)))
    outs = []
    try:
        for v, v_dtype in vectors:
            dv = Dev(v)
            odt = np.float32 if v_dtype == _lib.MX_F32 else np.float64
            dy = Dev(nbytes=max(m, 1) * np.dtype(odt).itemsize)
            check(lib.mx_dev_memset(dy.ptr, 0xFF, C.c_size_t(dy.nbytes), None))
            check(lib.mxd_spmv_plan_run(plan, dv.ptr, C.c_int(v_dtype), dy.ptr, None))
            check(lib.mx_stream_sync(None))
            outs.append(dy.download(odt, (m,)))
    finally:
        lib.mxd_spmv_plan_destroy(plan)
    return outs


def rows_sorted_device(p, j, misalign=0):
    """mxd_csr_rows_sorted on device copies; misalign = int32 elements by which the indices array is shifted off its
    16-byte alignment (0..3).  p may start above 0 (a row-block view that keeps absolute offsets into j)."""
    lib = _lib.load()
    m = p.size - 1
    dp = Dev(p.astype(np.int32))
    jj = np.concatenate([np.full(misalign, -7, dtype=np.int32), j.astype(np.int32)]) if misalign else j.astype(np.int32)
    dj = Dev(jj)
    ws = Dev(nbytes=16)
    flag = C.c_int(-1)
    check(lib.mxd_csr_rows_sorted(C.c_int(m), dp.ptr, C.c_void_p(dj.ptr.value + 4 * misalign), ws.ptr, C.byref(flag), None))
    return bool(flag.value)


def gather_fused_device(p, j, x, rows, capacity, value_dtype):
    """mxd_csr_gather_fused; returns (new_indptr, new_indices[:min(nnz, capacity)], new_values or None, nnz_out)"""
    lib = _lib.load()
    r = rows.size
    dp, dj, dr = Dev(p.astype(np.int32)), Dev(j.astype(np.int32)), Dev(rows.astype(np.int32))
    dx = None if x is None else Dev(x)
    vb = 0 if x is None else x.dtype.itemsize
    op, oj = Dev(nbytes=4 * (r + 1)), Dev(nbytes=4 * max(capacity, 1))
    ox = None if x is None else Dev(nbytes=vb * max(capacity, 1))
    check(lib.mx_dev_memset(oj.ptr, 0xFF, C.c_size_t(oj.nbytes), None))
    ws = Dev(nbytes=int(lib.mxd_gather_workspace_bytes(C.c_int(r))))
    nnz = C.c_int64(-1)
    avg = float(p[-1] - p[0]) / max(p.size - 1, 1)
    check(lib.mxd_csr_gather_fused(C.c_int(r), dp.ptr, dj.ptr, None if dx is None else dx.ptr, dr.ptr, op.ptr, oj.ptr,
                                   None if ox is None else ox.ptr, C.c_int(value_dtype), C.c_int64(capacity), C.c_double(avg),
                                   ws.ptr, C.byref(nnz), None))
    n = int(min(nnz.value, capacity))
    return (op.download(np.int32, (r + 1,)), oj.download(np.int32, (max(capacity, 1),))[:n],
            None if ox is None else ox.download(x.dtype, (max(capacity, 1),))[:n], int(nnz.value))
