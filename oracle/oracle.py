"""ctypes front end of the CPU restatement (oracle/mx_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py — never by matrixextra_amd/.

Functions carry the names and argument order of the reference's Rcpp exports
(src/RcppExports.cpp) and return what the R side receives: a numpy array for
the dense results, a dict(indptr=, indices=, values=) for the list results.
Dense matrices are numpy arrays in Fortran (column-major) order, as R holds them.

Parity status: see the header of mx_oracle.c ("parity unpinned by
reference-run outputs"; pinned by the reference's literal KATs + dense numpy).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

NA_INTEGER = np.int32(-2147483648)
NA_LOGICAL = NA_INTEGER
NA_REAL = np.frombuffer(np.uint64(0x7FF00000000007A2).tobytes(), dtype=np.float64)[0]


def _cpu_stamp() -> str:
    """md5 of this machine's CPU model + ISA flags — the same digest oracle/Makefile computes (a -march=native build only
    runs on the CPU model it was made on)."""
    import hashlib
    model = flags = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if not model and line.startswith("model name"):
                    model = line
                elif not flags and line.startswith("flags"):
                    flags = line
                if model and flags:
                    break
    except OSError:
        pass
    return hashlib.md5((model + flags).encode()).hexdigest()[:12]


def build(force: bool = False) -> str:
    """MXORACLE_SO=<path>: use that build as is (tools/sanitize.sh points it at an ASan + UBSan build).
    Otherwise oracle/_build/libmxoracle-<cpu stamp>.so: one library per CPU model (the build container's file travels to
    the GPU box, whose CPU differs — it is simply not the file that box asks for).  Built when missing or older than the
    source, under an exclusive file lock and through a temporary name (make: `mv` into place), so that processes that
    ask at the same time — pytest workers, the ranks of `bench.py --gpus N` — never load a half-written file."""
    if os.environ.get("MXORACLE_SO"):
        return os.environ["MXORACLE_SO"]
    import fcntl
    bdir = os.path.join(_HERE, "_build")
    so = os.path.join(bdir, f"libmxoracle-{_cpu_stamp()}.so")
    src = os.path.join(_HERE, "mx_oracle.c")

    def fresh():
        return os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(src)
    if force or not fresh():
        os.makedirs(bdir, exist_ok=True)
        with open(os.path.join(bdir, ".lock"), "w") as lk:
            fcntl.flock(lk, fcntl.LOCK_EX)
            try:
                if force or not fresh():                              # (another process may have built it meanwhile)
                    subprocess.check_call(["make", "-B", "-C", _HERE, f"OUT=_build/{os.path.basename(so)}"],
                                          stdout=subprocess.DEVNULL)
            finally:
                fcntl.flock(lk, fcntl.LOCK_UN)
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.mxo_multiply_csr_elemwise.restype = C.c_size_t
        _lib.mxo_add_csr_elemwise.restype = C.c_size_t
        _lib.mxo_copy_csr_rows_size.restype = C.c_size_t
        _lib.mxo_copy_csr_rows_col_seq.restype = C.c_size_t
        _lib.mxo_copy_csr_arbitrary.restype = C.c_size_t
    return _lib


def max_threads() -> int:
    """Threads the OpenMP loops may usefully use: the OpenMP default, capped by the CPU affinity mask and by the cgroup's
    CPU quota (the GPU boxes show 256 logical CPUs but grant 16 CPUs' worth of time: 128 threads there run 8x SLOWER
    than one)."""
    n = int(lib().mxo_max_threads())
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _same_buffer(a, b) -> bool:
    """Pointer identity, the test the reference applies with INTEGER(a) == INTEGER(b)."""
    return (a is b) or (a.size == b.size and a.size > 0 and
                        a.__array_interface__["data"][0] == b.__array_interface__["data"][0])


# ----------------------------------------------------------------------------- SpMM kernels
def gemm_csr_drm_as_drm(m, n, indptr, indices, values, B, ldb, C_out, ldc, nthreads=1, use_fma=False):
    """src/matmul.cpp:118-142. B, C_out flat arrays (float64 or float32)."""
    f = lib().mxo_gemm_csr_drm_as_drm_f64 if B.dtype == np.float64 else lib().mxo_gemm_csr_drm_as_drm_f32
    f(C.c_int(m), C.c_int(n), _p(indptr), _p(indices), _p(values), _p(B), C.c_size_t(ldb),
      _p(C_out), C.c_size_t(ldc), C.c_int(nthreads), C.c_int(int(use_fma)))


def gemm_csr_drm_as_dcm(m, n, indptr, indices, values, B, ldb, C_out, ldc, nthreads=1, use_fma=False):
    """src/matmul.cpp:150-185."""
    f = lib().mxo_gemm_csr_drm_as_dcm_f64 if B.dtype == np.float64 else lib().mxo_gemm_csr_drm_as_dcm_f32
    f(C.c_int(m), C.c_int(n), _p(indptr), _p(indices), _p(values), _p(B), C.c_size_t(ldb),
      _p(C_out), C.c_int(ldc), C.c_int(nthreads), C.c_int(int(use_fma)))


def _dense_in(M, dtype):
    M = np.asarray(M)
    assert M.ndim == 2
    return np.asfortranarray(M, dtype=dtype)


def tcrossprod_csr_dense(X_indptr, X_indices, X_values, Y_colmajor, nthreads=1, use_fma=False):
    """tcrossprod_csr_dense<> src/matmul.cpp:316-343: out(nrow X, nrow Y) col-major."""
    dt = np.float32 if np.asarray(Y_colmajor).dtype == np.float32 else np.float64
    Y = _dense_in(Y_colmajor, dt)
    p, j, x = _i32(X_indptr), _i32(X_indices), _f64(X_values)
    m, n = p.size - 1, Y.shape[0]
    out = np.zeros((m, n), dtype=dt, order="F")
    gemm_csr_drm_as_dcm(m, n, p, j, x, Y.reshape(-1, order="F"), Y.shape[0],
                        out.reshape(-1, order="F"), m, nthreads, use_fma)
    return out


tcrossprod_csr_dense_numeric = tcrossprod_csr_dense
tcrossprod_csr_dense_float32 = tcrossprod_csr_dense


def matmul_dense_csc(X_colmajor, Y_indptr, Y_indices, Y_values, nthreads=1, use_fma=False):
    """matmul_dense_csc<> src/matmul.cpp:188-219: out(nrow X, ncol Y) col-major."""
    dt = np.float32 if np.asarray(X_colmajor).dtype == np.float32 else np.float64
    X = _dense_in(X_colmajor, dt)
    p, i, x = _i32(Y_indptr), _i32(Y_indices), _f64(Y_values)
    nrows_X, ncols_Y = X.shape[0], p.size - 1
    out = np.zeros((nrows_X, ncols_Y), dtype=dt, order="F")
    gemm_csr_drm_as_drm(ncols_Y, nrows_X, p, i, x, X.reshape(-1, order="F"), nrows_X,
                        out.reshape(-1, order="F"), nrows_X, nthreads, use_fma)
    return out


matmul_dense_csc_numeric = matmul_dense_csc
matmul_dense_csc_float32 = matmul_dense_csc


def tcrossprod_dense_csr(X_colmajor, Y_indptr, Y_indices, Y_values, nthreads=1, ncols_Y=0, use_fma=False):
    """tcrossprod_dense_csr<> src/matmul.cpp:254-281: out(nrow X, nrow Y) col-major."""
    dt = np.float32 if np.asarray(X_colmajor).dtype == np.float32 else np.float64
    X = _dense_in(X_colmajor, dt)
    p, j, x = _i32(Y_indptr), _i32(Y_indices), _f64(Y_values)
    nrows_X, nrows_Y = X.shape[0], p.size - 1
    out = np.zeros((nrows_X, nrows_Y), dtype=dt, order="F")
    gemm_csr_drm_as_drm(nrows_Y, nrows_X, p, j, x, X.reshape(-1, order="F"), nrows_X,
                        out.reshape(-1, order="F"), nrows_X, nthreads, use_fma)
    return out


tcrossprod_dense_csr_numeric = tcrossprod_dense_csr
tcrossprod_dense_csr_float32 = tcrossprod_dense_csr


# ----------------------------------------------------------------------------- SpMV
def _dvec(kind, X_indptr, X_indices, X_values, y_dense, nthreads):
    p, j, x = _i32(X_indptr), _i32(X_indices), _f64(X_values)
    m = p.size - 1
    if kind == 3:
        y = np.ascontiguousarray(y_dense, dtype=np.float32)
        out = np.zeros(m, dtype=np.float32)
    elif kind == 0:
        y = _f64(y_dense)
        out = np.zeros(m, dtype=np.float64)
    else:
        y = _i32(y_dense)
        out = np.zeros(m, dtype=np.float64)
    lib().mxo_matmul_csr_dvec(C.c_int(m), _p(p), _p(j), _p(x), _p(y), C.c_int(kind), _p(out), C.c_int(nthreads))
    return out


def matmul_csr_dvec_numeric(p, j, x, y, nthreads=1):
    """src/matmul.cpp:421-435"""
    return _dvec(0, p, j, x, y, nthreads)


def matmul_csr_dvec_integer(p, j, x, y, nthreads=1):
    """src/matmul.cpp:437-451"""
    return _dvec(1, p, j, x, y, nthreads)


def matmul_csr_dvec_logical(p, j, x, y, nthreads=1):
    """src/matmul.cpp:453-467"""
    return _dvec(2, p, j, x, y, nthreads)


def matmul_csr_dvec_float32(p, j, x, y, nthreads=1):
    """src/matmul.cpp:469-483"""
    return _dvec(3, p, j, x, y, nthreads)


# ----------------------------------------------------------------------------- merges
def _r_logical_vec(which, a, b):
    f = lib().mxo_r_logical
    return np.array([f(which, int(u), int(v)) for u, v in zip(a, b)], dtype=np.int32)


def _multiply(indptr1, indptr2, indices1, indices2, values1, values2, logical):
    vdt = np.int32 if logical else np.float64
    # identical-pattern fast path: operators.cpp:104-132 (indptr/indices returned aliased)
    if (indptr1.size == indptr2.size and indices1.size == indices2.size and
            _same_buffer(indptr1, indptr2) and _same_buffer(indices1, indices2)):
        v1, v2 = np.asarray(values1, dtype=vdt), np.asarray(values2, dtype=vdt)
        vals = _r_logical_vec(1, v1, v2) if logical else v1 * v2
        return dict(indptr=indptr1, indices=indices1, values=vals)
    p1, p2, j1, j2 = _i32(indptr1), _i32(indptr2), _i32(indices1), _i32(indices2)
    v1, v2 = np.ascontiguousarray(values1, dtype=vdt), np.ascontiguousarray(values2, dtype=vdt)
    nrows = p1.size - 1
    cap = max(min(j1.size, j2.size), 1)
    op = np.zeros(nrows + 1, dtype=np.int32)
    oj = np.empty(cap, dtype=np.int32)
    ov = np.empty(cap, dtype=vdt)
    nnz = lib().mxo_multiply_csr_elemwise(C.c_int(nrows), _p(p1), _p(p2), _p(j1), _p(j2), _p(v1), _p(v2),
                                          C.c_int(int(logical)), _p(op), _p(oj), _p(ov))
    return dict(indptr=op, indices=oj[:nnz].copy(), values=ov[:nnz].copy())


def multiply_csr_elemwise(indptr1, indptr2, indices1, indices2, values1, values2):
    """src/operators.cpp:209-222"""
    return _multiply(indptr1, indptr2, indices1, indices2, values1, values2, False)


def logicaland_csr_elemwise(indptr1, indptr2, indices1, indices2, values1, values2):
    """src/operators.cpp:224-237"""
    return _multiply(indptr1, indptr2, indices1, indices2, values1, values2, True)


def _add(indptr1, indptr2, indices1, indices2, values1, values2, mode, substract):
    lgl = mode != 0
    vdt = np.int32 if lgl else np.float64
    # same-structure fast paths: operators.cpp:343-395
    if (indices1.size == indices2.size and _same_buffer(indptr1, indptr2) and
            _same_buffer(indices1, indices2)):
        if substract and _same_buffer(np.asarray(values1), np.asarray(values2)):
            # operators.cpp:348-355: zeroed indptr of the same length, EMPTY indices/values
            return dict(indptr=np.zeros(indptr1.size, dtype=np.int32),
                        indices=np.zeros(0, dtype=np.int32), values=np.zeros(0, dtype=np.float64))
        v1, v2 = np.asarray(values1, dtype=vdt), np.asarray(values2, dtype=vdt)
        if not lgl:
            vals = v1 - v2 if substract else v1 + v2
        else:
            vals = _r_logical_vec(2 if mode == 2 else 0, v1, v2)
        return dict(indptr=indptr1, indices=indices1, values=vals)
    p1, p2, j1, j2 = _i32(indptr1), _i32(indptr2), _i32(indices1), _i32(indices2)
    v1, v2 = np.ascontiguousarray(values1, dtype=vdt), np.ascontiguousarray(values2, dtype=vdt)
    nrows = p1.size - 1
    cap = max(j1.size + j2.size, 1)
    op = np.zeros(nrows + 1, dtype=np.int32)
    oj = np.empty(cap, dtype=np.int32)
    ov = np.empty(cap, dtype=vdt)
    nnz = lib().mxo_add_csr_elemwise(C.c_int(nrows), _p(p1), _p(p2), _p(j1), _p(j2), _p(v1), _p(v2),
                                     C.c_int(mode), C.c_int(int(substract)), _p(op), _p(oj), _p(ov))
    return dict(indptr=op, indices=oj[:nnz].copy(), values=ov[:nnz].copy())


def add_csr_elemwise(indptr1, indptr2, indices1, indices2, values1, values2, substract):
    """src/operators.cpp:539-554"""
    return _add(indptr1, indptr2, indices1, indices2, values1, values2, 0, bool(substract))


def logicalor_csr_elemwise(indptr1, indptr2, indices1, indices2, values1, values2, xor_op):
    """src/operators.cpp:556-571"""
    return _add(indptr1, indptr2, indices1, indices2, values1, values2, 2 if xor_op else 1, False)


# ----------------------------------------------------------------------------- gather
def _copy_rows(indptr, indices, values, rows_take, vdt):
    p, j, rows = _i32(indptr), _i32(indices), _i32(rows_take)
    total = lib().mxo_copy_csr_rows_size(_p(p), _p(rows), C.c_size_t(rows.size))
    empty_v = np.zeros(0, dtype=vdt if vdt is not None else np.float64)
    if total == 0:  # slice.cpp:236-240: three empty vectors
        return dict(indptr=np.zeros(0, dtype=np.int32), indices=np.zeros(0, dtype=np.int32), values=empty_v)
    has_values = values is not None and np.asarray(values).size > 0
    v = np.ascontiguousarray(values, dtype=vdt) if has_values else None
    npz = np.zeros(rows.size + 1, dtype=np.int32)
    nj = np.empty(total, dtype=np.int32)
    nv = np.empty(total, dtype=vdt) if has_values else empty_v
    lib().mxo_copy_csr_rows(_p(p), _p(j), _p(v), C.c_int(v.dtype.itemsize if has_values else 0),
                            _p(rows), C.c_size_t(rows.size), _p(npz), _p(nj), _p(nv) if has_values else None)
    return dict(indptr=npz, indices=nj, values=nv)


def copy_csr_rows_numeric(indptr, indices, values, rows_take):
    """src/slice.cpp:276-291"""
    return _copy_rows(indptr, indices, values, rows_take, np.float64)


def copy_csr_rows_logical(indptr, indices, values, rows_take):
    """src/slice.cpp:293-308"""
    return _copy_rows(indptr, indices, values, rows_take, np.int32)


def copy_csr_rows_binary(indptr, indices, rows_take):
    """src/slice.cpp:310-324"""
    return _copy_rows(indptr, indices, None, rows_take, None)


def check_is_seq(indices) -> bool:
    """src/slice.cpp:25-35"""
    a = _i32(indices)
    return bool(lib().mxo_check_is_seq(_p(a), C.c_size_t(a.size)))


def check_is_rev_seq(indices) -> bool:
    """src/slice.cpp:37-47"""
    a = _i32(indices)
    return bool(lib().mxo_check_is_rev_seq(_p(a), C.c_size_t(a.size)))


# ----------------------------------------------------------------------------- sort precondition
def check_indices_are_sorted(indptr, indices) -> bool:
    """check_is_sorted per row, src/misc.cpp:118-128 / :283"""
    p, j = _i32(indptr), _i32(indices)
    return bool(lib().mxo_check_indices_are_sorted(_p(p), _p(j), C.c_int(p.size - 1)))


def sort_sparse_indices(indptr, indices, values=None):
    """sort_sparse_indices_known_ncol, src/misc.cpp:261-298; returns sorted COPIES."""
    p = _i32(indptr)
    j = _i32(indices).copy()
    v = None if values is None else np.array(values, copy=True)
    vb = 0 if v is None else v.dtype.itemsize
    lib().mxo_sort_sparse_indices(_p(p), _p(j), _p(v), C.c_int(vb), C.c_int(p.size - 1))
    return j, v


# ----------------------------------------------------------------------------- column-filtering slices (§8f-2)
def _col_seq(indptr, indices, values, rows_take, cols_take, index1, kind):
    p, j, rows, cols = _i32(indptr), _i32(indices), _i32(rows_take), _i32(cols_take)
    lo, hi = int(cols.min()) - int(bool(index1)), int(cols.max()) - int(bool(index1))
    has_values = values is not None and np.asarray(values).size > 0
    v = None
    if has_values:
        v = np.ascontiguousarray(values, dtype=np.float64 if kind == 1 else np.int32)
    npz = np.zeros(rows.size + 1, dtype=np.int32)
    total = lib().mxo_copy_csr_rows_col_seq(_p(p), _p(j), _p(v), C.c_int(kind if has_values else 0), _p(rows),
                                            C.c_size_t(rows.size), C.c_int(lo), C.c_int(hi), _p(npz), None, None)
    if total == 0:  # slice.cpp:355-359
        return dict(indptr=npz, indices=np.zeros(0, dtype=np.int32), values=np.zeros(0, dtype=np.float64))
    nj = np.empty(total, dtype=np.int32)
    nv = np.empty(total if has_values else 0, dtype=np.float64)
    lib().mxo_copy_csr_rows_col_seq(_p(p), _p(j), _p(v), C.c_int(kind if has_values else 0), _p(rows),
                                    C.c_size_t(rows.size), C.c_int(lo), C.c_int(hi), _p(npz), _p(nj),
                                    _p(nv) if has_values else None)
    return dict(indptr=npz, indices=nj, values=nv)


def copy_csr_rows_col_seq_numeric(indptr, indices, values, rows_take, cols_take, index1):
    """src/slice.cpp:385-403"""
    return _col_seq(indptr, indices, values, rows_take, cols_take, index1, 1)


def copy_csr_rows_col_seq_logical(indptr, indices, values, rows_take, cols_take, index1):
    """src/slice.cpp:405-423"""
    return _col_seq(indptr, indices, values, rows_take, cols_take, index1, 2)


def copy_csr_rows_col_seq_binary(indptr, indices, rows_take, cols_take, index1):
    """src/slice.cpp:425-443"""
    return _col_seq(indptr, indices, None, rows_take, cols_take, index1, 0)


def _arbitrary(indptr, indices, values, rows_take, cols_take, vdt):
    p, j, rows, cols = _i32(indptr), _i32(indices), _i32(rows_take), _i32(cols_take)
    has_values = values is not None and np.asarray(values).size > 0
    v = np.ascontiguousarray(values, dtype=vdt) if has_values else None
    vb = v.dtype.itemsize if has_values else 0
    ncol = int(cols.max()) + 1 if cols.size else 0
    npz = np.zeros(rows.size + 1, dtype=np.int32)
    total = lib().mxo_copy_csr_arbitrary(_p(p), _p(j), _p(v), C.c_int(vb), _p(rows), C.c_size_t(rows.size), _p(cols),
                                         C.c_size_t(cols.size), C.c_int(ncol), _p(npz), None, None)
    nj = np.empty(total, dtype=np.int32)
    nv = np.empty(total, dtype=vdt) if has_values else None
    if total:
        lib().mxo_copy_csr_arbitrary(_p(p), _p(j), _p(v), C.c_int(vb), _p(rows), C.c_size_t(rows.size), _p(cols),
                                     C.c_size_t(cols.size), C.c_int(ncol), _p(npz), _p(nj), _p(nv))
    out = dict(indptr=npz, indices=nj)
    if has_values:
        out["values"] = nv
    return out


def copy_csr_arbitrary_numeric(indptr, indices, values, rows_take, cols_take):
    """src/slice.cpp:580-596"""
    return _arbitrary(indptr, indices, values, rows_take, cols_take, np.float64)


def copy_csr_arbitrary_logical(indptr, indices, values, rows_take, cols_take):
    """src/slice.cpp:598-614"""
    return _arbitrary(indptr, indices, values, rows_take, cols_take, np.int32)


def copy_csr_arbitrary_binary(indptr, indices, rows_take, cols_take):
    """src/slice.cpp:616-632"""
    return _arbitrary(indptr, indices, None, rows_take, cols_take, None)


def _reverse_rows(indptr, indices, values, vdt):
    p, j = _i32(indptr), _i32(indices)
    has_values = values is not None and np.asarray(values).size > 0
    v = np.ascontiguousarray(values, dtype=vdt) if has_values else None
    npz = np.zeros(p.size, dtype=np.int32)
    nj = np.empty(j.size, dtype=np.int32)
    nv = np.empty(j.size, dtype=vdt) if has_values else np.zeros(0, dtype=vdt if vdt is not None else np.float64)
    lib().mxo_reverse_rows(_p(p), _p(j), _p(v), C.c_int(v.dtype.itemsize if has_values else 0), C.c_int(p.size - 1),
                           _p(npz), _p(nj), _p(nv) if has_values else None)
    return dict(indptr=npz, indices=nj, values=nv)


def reverse_rows_numeric(indptr, indices, values):
    """src/slice.cpp:98-110"""
    return _reverse_rows(indptr, indices, values, np.float64)


def reverse_rows_logical(indptr, indices, values):
    """src/slice.cpp:112-124"""
    return _reverse_rows(indptr, indices, values, np.int32)


def reverse_rows_binary(indptr, indices):
    """src/slice.cpp:126-138"""
    return _reverse_rows(indptr, indices, None, None)


def reverse_columns_inplace(indptr, indices, values, ncol):
    """src/slice.cpp:142-170: modifies indices / values (int32 / float64-or-int32 numpy arrays) in place."""
    p = _i32(indptr)
    vb = 0 if values is None else values.dtype.itemsize
    lib().mxo_reverse_columns_inplace(_p(p), _p(indices), _p(values), C.c_int(vb), C.c_int(p.size - 1), C.c_int(int(ncol)))


# ----------------------------------------------------------------------------- cbind / rbind (§8f-3)
def _cbind(Xp, Xj, Xx, Yp, Yj_plus_ncol, Yx, vdt):
    Xp, Xj, Yp, Yj = _i32(Xp), _i32(Xj), _i32(Yp), _i32(Yj_plus_ncol)
    hx = Xx is not None and np.asarray(Xx).size > 0
    hy = Yx is not None and np.asarray(Yx).size > 0
    has_values = hx or hy
    nnz = Xj.size + Yj.size
    nrows = max(Xp.size, Yp.size) - 1
    indptr = np.zeros(nrows + 1, dtype=np.int32)
    indices = np.zeros(nnz, dtype=np.int32)
    values = np.zeros(nnz if has_values else 0, dtype=vdt if vdt is not None else np.float64)
    if nnz == 0:
        return dict(indptr=indptr, indices=indices, values=values)
    xv = np.ascontiguousarray(Xx, dtype=vdt) if has_values else None
    yv = np.ascontiguousarray(Yx, dtype=vdt) if has_values else None
    lib().mxo_cbind_csr(_p(Xp), _p(Xj), _p(xv), C.c_int(Xp.size - 1), _p(Yp), _p(Yj), _p(yv), C.c_int(Yp.size - 1),
                        C.c_int(values.dtype.itemsize if has_values else 0), _p(indptr), _p(indices),
                        _p(values) if has_values else None)
    return dict(indptr=indptr, indices=indices, values=values)


def cbind_csr_numeric(Xp, Xj, Xx, Yp, Yj_plus_ncol, Yx):
    """src/cbind.cpp:101-119"""
    return _cbind(Xp, Xj, Xx, Yp, Yj_plus_ncol, Yx, np.float64)


def cbind_csr_logical(Xp, Xj, Xx, Yp, Yj_plus_ncol, Yx):
    """src/cbind.cpp:121-139"""
    return _cbind(Xp, Xj, Xx, Yp, Yj_plus_ncol, Yx, np.int32)


def cbind_csr_binary(Xp, Xj, Yp, Yj_plus_ncol):
    """src/cbind.cpp:141-157"""
    return _cbind(Xp, Xj, None, Yp, Yj_plus_ncol, None, None)


def concat_indptr2(ptr1, ptr2):
    """src/rbind.cpp:9-21"""
    a, b = _i32(ptr1), _i32(ptr2)
    out = np.empty(a.size + b.size - 1, dtype=np.int32)
    lib().mxo_concat_indptr2(_p(a), C.c_int(a.size), _p(b), C.c_int(b.size), _p(out))
    return out


def concat_csr_batch(objects, out_kind):
    """src/rbind.cpp:24-173.  objects: list of (in_kind, indptr|None, indices, values|None, nrows); kinds as in
    mx_oracle.c (0 dgR, 1 lgR, 2 ngR, 3..6 d/i/l/n sparse vectors with 1-based indices).  out_kind 0 dgR, 1 lgR, 2 ngR."""
    nrows = sum(o[4] if o[0] <= 2 else 1 for o in objects)
    nnz = sum(np.asarray(o[2]).size for o in objects)
    indptr = np.zeros(nrows + 1, dtype=np.int32)
    indices = np.zeros(nnz, dtype=np.int32)
    values = None if out_kind == 2 else np.zeros(nnz, dtype=np.float64 if out_kind == 0 else np.int32)
    row = pos = 0
    for kind, p, j, x, nr in objects:
        j = _i32(j)
        pp = _i32(p) if p is not None else None
        xv = None
        if x is not None:
            xv = np.ascontiguousarray(x, dtype=np.float64 if kind in (0, 3) else np.int32)
        row += lib().mxo_concat_csr_append(C.c_int(kind), _p(pp), _p(j), _p(xv), C.c_int(nr), C.c_int(j.size),
                                           C.c_int(out_kind), C.c_int(row), C.c_int(pos), _p(indptr), _p(indices),
                                           _p(values))
        pos += j.size
    return dict(indptr=indptr, indices=indices, values=values)


# ----------------------------------------------------------------------------- CSR x sparse vector, CSR (.) dense (§8f-4)
def _svec(kind, p, j, x, y_indices_base1, y_values, nthreads):
    p, j, x, yi = _i32(p), _i32(j), _f64(x), _i32(y_indices_base1)
    yv = None
    if kind in (0,):
        yv = _f64(y_values)
    elif kind in (1, 2):
        yv = _i32(y_values)
    elif kind == 4:
        yv = np.ascontiguousarray(y_values, dtype=np.float32)
    out = np.zeros(p.size - 1, dtype=np.float64)
    lib().mxo_matmul_csr_svec(C.c_int(p.size - 1), _p(p), _p(j), _p(x), _p(yi), C.c_size_t(yi.size), _p(yv),
                              C.c_int(kind), _p(out), C.c_int(nthreads))
    return out


def matmul_csr_svec_numeric(p, j, x, yi, yv, nthreads=1):
    """src/matmul.cpp:555-571"""
    return _svec(0, p, j, x, yi, yv, nthreads)


def matmul_csr_svec_integer(p, j, x, yi, yv, nthreads=1):
    """src/matmul.cpp:573-589"""
    return _svec(1, p, j, x, yi, yv, nthreads)


def matmul_csr_svec_logical(p, j, x, yi, yv, nthreads=1):
    """src/matmul.cpp:591-607"""
    return _svec(2, p, j, x, yi, yv, nthreads)


def matmul_csr_svec_binary(p, j, x, yi, nthreads=1):
    """src/matmul.cpp:609-624"""
    return _svec(3, p, j, x, yi, None, nthreads)


def matmul_csr_svec_float32(p, j, x, yi, yv, nthreads=1):
    """src/matmul.cpp:626-641 (y values are float32 bits)"""
    return _svec(4, p, j, x, yi, yv, nthreads)


def _csr_by_dense(kind, p, j, x, dense_mat):
    p, j = _i32(p), _i32(j)
    nrows = p.size - 1
    ddt = {0: np.float64, 1: np.float32, 2: np.int32, 3: np.int32, 4: np.int32}[kind]
    D = np.asfortranarray(dense_mat, dtype=ddt)
    assert D.shape[0] == nrows
    xv = np.ascontiguousarray(x, dtype=np.int32 if kind == 4 else np.float64)
    out = np.empty(xv.size, dtype=np.int32 if kind == 4 else np.float64)
    lib().mxo_multiply_csr_by_dense_elemwise(C.c_int(nrows), _p(p), _p(j), _p(xv), _p(D.reshape(-1, order="F")),
                                             C.c_int(kind), _p(out))
    return out


def multiply_csr_by_dense_elemwise_double(p, j, x, D):
    """src/operators.cpp:289-296"""
    return _csr_by_dense(0, p, j, x, D)


def multiply_csr_by_dense_elemwise_float32(p, j, x, D):
    """src/operators.cpp:298-305"""
    return _csr_by_dense(1, p, j, x, D)


def multiply_csr_by_dense_elemwise_int(p, j, x, D):
    """src/operators.cpp:307-314"""
    return _csr_by_dense(2, p, j, x, D)


def multiply_csr_by_dense_elemwise_bool(p, j, x, D):
    """src/operators.cpp:316-323"""
    return _csr_by_dense(3, p, j, x, D)


def logicaland_csr_by_dense_cpp(p, j, x, D):
    """src/operators.cpp:325-334"""
    return _csr_by_dense(4, p, j, x, D)


DV_OPS = {"multiply": 0, "powerto": 1, "divide": 2, "divrest": 3, "intdiv": 4, "logical_and": 5}


def _csr_by_dvec(p, j, x, dvec, ncols, op, x_is_lhs):
    p, j = _i32(p), _i32(j)
    dt = np.int32 if op == 5 else np.float64
    xv = np.ascontiguousarray(x, dtype=dt)
    dv = np.ascontiguousarray(dvec, dtype=dt)
    out = np.empty(xv.size, dtype=dt)
    if xv.size:
        lib().mxo_csr_by_dvec(C.c_int(p.size - 1), C.c_int(int(ncols)), _p(p), _p(j), _p(xv), _p(dv),
                              C.c_size_t(dv.size), C.c_int(op), C.c_int(1 if x_is_lhs else 0), _p(out))
    return out


def multiply_csr_by_dvec_no_NAs_numeric(p, j, x, dvec, ncols, multiply, powerto, divide, divrest, intdiv, X_is_LHS):
    """src/operators.cpp:2142-2175 (flag precedence of :1620-1632)"""
    if multiply:
        op = 0
    elif powerto:
        op = 1
    elif divide:
        op = 2
    elif divrest:
        op = 3
    elif intdiv:
        op = 4
    else:
        raise ValueError("Internal error. Please file an issue in GitHub.")
    return _csr_by_dvec(p, j, x, dvec, ncols, op, X_is_LHS)


def logicaland_csr_by_dvec_internal(p, j, x, dvec, ncols):
    """src/operators.cpp:2177-2200"""
    return _csr_by_dvec(p, j, x, dvec, ncols, 5, True)


class _DvecNaResult(C.Structure):
    _fields_ = [("indptr", C.POINTER(C.c_int)), ("indices", C.POINTER(C.c_int)), ("values", C.POINTER(C.c_double)),
                ("nnz", C.c_size_t), ("status", C.c_int)]


def multiply_csr_by_dvec_with_NAs(p, j, x, dvec, ncols, multiply, powerto, divide, divrest, intdiv, X_is_LHS):
    """src/operators.cpp:2258-2856 (R/RcppExports.R: multiply_csr_by_dvec_with_NAs): the structure-changing route of
    `CSR op vector` when the vector holds NA / NaN, zeros under / %% %/% ^, negatives under ^ or infinities under *.
    Returns dict(indptr, indices, values, alias_structure); raises ValueError for the reference's two error exits."""
    p, j = _i32(p), _i32(j)
    xv, dv = _f64(x), _f64(dvec).reshape(-1)
    op = 0 if multiply else 1 if powerto else 2 if divide else 3 if divrest else 4 if intdiv else -1
    res = _DvecNaResult()
    lib().mxo_csr_by_dvec_with_NAs(C.c_int(p.size - 1), C.c_int(int(ncols)), _p(p), _p(j), _p(xv), _p(dv), C.c_size_t(dv.size),
                                   C.c_int(op), C.c_int(1 if X_is_LHS else 0), C.byref(res))
    try:
        if res.status == 3:
            raise ValueError("Unexpected error.")
        if res.status == 2:
            raise ValueError("Error: the resulting matrix would have too many entries for a sparse CSR representation (int overflow).")
        n = int(res.nnz)
        vals = np.ctypeslib.as_array(res.values, shape=(n,)).copy() if n else np.zeros(0)
        if res.status == 1:                                          # the reference hands back its input indptr / indices
            return dict(indptr=p, indices=j, values=vals, alias_structure=True)
        return dict(indptr=np.ctypeslib.as_array(res.indptr, shape=(p.size,)).copy(),
                    indices=np.ctypeslib.as_array(res.indices, shape=(n,)).copy() if n else np.zeros(0, dtype=np.int32),
                    values=vals, alias_structure=False)
    finally:
        lib().mxo_free_dvec_na(C.byref(res))


def _remove_zero_valued_csr(indptr, indices, values, remove_NAs, kind):
    p, j = _i32(indptr), _i32(indices)
    vdt = np.float64 if kind == 0 else np.int32
    v = np.ascontiguousarray(values, dtype=vdt)
    n = p.size - 1
    npz = np.zeros(n + 1, dtype=np.int32)
    nj = np.empty(j.size, dtype=np.int32)
    nv = np.empty(j.size, dtype=vdt)
    fn = lib().mxo_remove_zero_valued_csr
    fn.restype = C.c_longlong
    k = fn(C.c_int(n), _p(p), _p(j), _p(v), C.c_int(kind), C.c_int(1 if remove_NAs else 0), _p(npz), _p(nj), _p(nv))
    if k < 0:   # misc.cpp:586-590: the INPUT vectors themselves
        return dict(indptr=indptr, indices=indices, values=values)
    return dict(indptr=npz, indices=nj[:k].copy(), values=nv[:k].copy())


def remove_zero_valued_csr_numeric(indptr, indices, values, remove_NAs):
    """src/misc.cpp:667-682"""
    return _remove_zero_valued_csr(indptr, indices, values, remove_NAs, 0)


def remove_zero_valued_csr_logical(indptr, indices, values, remove_NAs):
    """src/misc.cpp:684-698"""
    return _remove_zero_valued_csr(indptr, indices, values, remove_NAs, 1)


_VALID_CSR_MESSAGES = {1: "Matrix has negative indices.", 2: "Matrix has invalid column indices.",
                       3: "Matrix has indices with missing values.", 4: "Matrix has missing values in the index pointer.",
                       5: "Matrix index pointer is not monotonicaly increasing."}


def check_valid_csr_matrix(indptr, indices, nrows, ncols):
    """src/misc.cpp:970-1016: list(err=...) or an empty list"""
    p, j = _i32(indptr), _i32(indices)
    code = lib().mxo_check_valid_csr_matrix(_p(p), _p(j), C.c_longlong(j.size), C.c_int(nrows), C.c_int(ncols))
    return dict(err=_VALID_CSR_MESSAGES[code]) if code else dict()


def matmul_rowvec_by_csc(rowvec, indptr, indices, values):
    """src/matmul.cpp:643-663"""
    r = np.ascontiguousarray(rowvec, dtype=np.float32).reshape(-1)
    p, j = _i32(indptr), _i32(indices)
    v = None if values is None else _f64(values)
    out = np.zeros((1, p.size - 1), dtype=np.float32)
    lib().mxo_matmul_rowvec_by_csc(_p(r), _p(p), _p(j), _p(v) if v is not None else None, C.c_int(p.size - 1), _p(out))
    return out


def matmul_rowvec_by_cscbin(rowvec, indptr, indices):
    """src/matmul.cpp:665-684"""
    return matmul_rowvec_by_csc(rowvec, indptr, indices, None)
