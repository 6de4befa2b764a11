"""Python twins of the reference's R/RcppExports.R wrappers for the hot path.

Same names, argument order and return shapes as the R side sees them
(R/RcppExports.R:148-150, 360, 388, 564 …): dense results are numpy arrays in
column-major (Fortran) order, list results are dicts with `indptr`, `indices`,
`values`.  Every function goes through the C-ABI of libmxgpu.so
(include/mxgpu.h) — i.e. through the HIP kernels; nothing here computes.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import (MX_F64, MX_LGL, MX_NONE, MX_OP_ADD, MX_OP_AND, MX_OP_MUL, MX_OP_OR,
                   MX_OP_SUB, MX_OP_XOR, ResultInfo, check, ptr)


def _i32(a):
    a = np.asarray(a)
    if a.dtype != np.int32 or not a.flags.c_contiguous:
        a = np.ascontiguousarray(a, dtype=np.int32)
    return a


def _f64(a):
    a = np.asarray(a)
    if a.dtype != np.float64 or not a.flags.c_contiguous:
        a = np.ascontiguousarray(a, dtype=np.float64)
    return a


def _dense(M, dtype):
    M = np.asarray(M)
    if M.ndim != 2:
        raise ValueError("dense operand must be a 2-d matrix")
    if M.dtype != dtype or not M.flags.f_contiguous:
        M = np.asfortranarray(M, dtype=dtype)
    return M


# ----------------------------------------------------------------------------- SpMM
def _tcrossprod_csr_dense(X_csr_indptr, X_csr_indices, X_csr_values, Y_colmajor, nthreads, dtype, fn):
    p, j, x = _i32(X_csr_indptr), _i32(X_csr_indices), _f64(X_csr_values)
    Y = _dense(Y_colmajor, dtype)
    m, n, K = p.size - 1, Y.shape[0], Y.shape[1]
    out = np.empty((m, n), dtype=dtype, order="F")
    check(fn(ptr(p), ptr(j), ptr(x), C.c_int(m), ptr(Y), C.c_int(n), C.c_int(K), C.c_int(int(nthreads)), ptr(out)))
    return out


def tcrossprod_csr_dense_numeric(X_csr_indptr, X_csr_indices, X_csr_values, Y_colmajor, nthreads=1):
    """R/RcppExports.R:148-150 -> src/matmul.cpp:345-359."""
    return _tcrossprod_csr_dense(X_csr_indptr, X_csr_indices, X_csr_values, Y_colmajor, nthreads, np.float64,
                                 _lib.load().mx_tcrossprod_csr_dense_numeric)


def tcrossprod_csr_dense_float32(X_csr_indptr, X_csr_indices, X_csr_values, Y_colmajor, nthreads=1):
    """src/matmul.cpp:361-375 (Y / result are the float32@Data bits; here numpy float32)."""
    return _tcrossprod_csr_dense(X_csr_indptr, X_csr_indices, X_csr_values, Y_colmajor, nthreads, np.float32,
                                 _lib.load().mx_tcrossprod_csr_dense_float32)


def _dense_times_sparse(X_colmajor, indptr, indices, values, nthreads, dtype, fn, extra=None):
    X = _dense(X_colmajor, dtype)
    p, i, x = _i32(indptr), _i32(indices), _f64(values)
    nrows_X, ncols_X, nout = X.shape[0], X.shape[1], p.size - 1
    out = np.empty((nrows_X, nout), dtype=dtype, order="F")
    args = [ptr(X), C.c_int(nrows_X), C.c_int(ncols_X), ptr(p), ptr(i), ptr(x), C.c_int(nout),
            C.c_int(int(nthreads))]
    if extra is not None:
        args.append(C.c_int(int(extra)))
    args.append(ptr(out))
    check(fn(*args))
    return out


def matmul_dense_csc_numeric(X_colmajor, Y_csc_indptr, Y_csc_indices, Y_csc_values, nthreads=1):
    """src/matmul.cpp:221-235."""
    return _dense_times_sparse(X_colmajor, Y_csc_indptr, Y_csc_indices, Y_csc_values, nthreads, np.float64,
                               _lib.load().mx_matmul_dense_csc_numeric)


def matmul_dense_csc_float32(X_colmajor, Y_csc_indptr, Y_csc_indices, Y_csc_values, nthreads=1):
    """src/matmul.cpp:237-251."""
    return _dense_times_sparse(X_colmajor, Y_csc_indptr, Y_csc_indices, Y_csc_values, nthreads, np.float32,
                               _lib.load().mx_matmul_dense_csc_float32)


def tcrossprod_dense_csr_numeric(X_colmajor, Y_csr_indptr, Y_csr_indices, Y_csr_values, nthreads=1, ncols_Y=0):
    """src/matmul.cpp:283-297."""
    return _dense_times_sparse(X_colmajor, Y_csr_indptr, Y_csr_indices, Y_csr_values, nthreads, np.float64,
                               _lib.load().mx_tcrossprod_dense_csr_numeric, extra=ncols_Y)


def tcrossprod_dense_csr_float32(X_colmajor, Y_csr_indptr, Y_csr_indices, Y_csr_values, nthreads=1, ncols_Y=0):
    """src/matmul.cpp:299-313."""
    return _dense_times_sparse(X_colmajor, Y_csr_indptr, Y_csr_indices, Y_csr_values, nthreads, np.float32,
                               _lib.load().mx_tcrossprod_dense_csr_float32, extra=ncols_Y)


# ----------------------------------------------------------------------------- SpMV
def _dvec(X_csr_indptr, X_csr_indices, X_csr_values, y_dense, nthreads, ydt, odt, fn):
    p, j, x = _i32(X_csr_indptr), _i32(X_csr_indices), _f64(X_csr_values)
    y = np.ascontiguousarray(y_dense, dtype=ydt)
    m = p.size - 1
    out = np.empty(m, dtype=odt)
    check(fn(ptr(p), ptr(j), ptr(x), C.c_int(m), ptr(y), C.c_int(y.size), C.c_int(int(nthreads)), ptr(out)))
    return out


def matmul_csr_dvec_numeric(X_csr_indptr, X_csr_indices, X_csr_values, y_dense, nthreads=1):
    """src/matmul.cpp:421-435."""
    return _dvec(X_csr_indptr, X_csr_indices, X_csr_values, y_dense, nthreads, np.float64, np.float64,
                 _lib.load().mx_matmul_csr_dvec_numeric)


def matmul_csr_dvec_integer(X_csr_indptr, X_csr_indices, X_csr_values, y_dense, nthreads=1):
    """src/matmul.cpp:437-451."""
    return _dvec(X_csr_indptr, X_csr_indices, X_csr_values, y_dense, nthreads, np.int32, np.float64,
                 _lib.load().mx_matmul_csr_dvec_integer)


def matmul_csr_dvec_logical(X_csr_indptr, X_csr_indices, X_csr_values, y_dense, nthreads=1):
    """src/matmul.cpp:453-467."""
    return _dvec(X_csr_indptr, X_csr_indices, X_csr_values, y_dense, nthreads, np.int32, np.float64,
                 _lib.load().mx_matmul_csr_dvec_logical)


def matmul_csr_dvec_float32(X_csr_indptr, X_csr_indices, X_csr_values, y_dense, nthreads=1):
    """src/matmul.cpp:469-483."""
    return _dvec(X_csr_indptr, X_csr_indices, X_csr_values, y_dense, nthreads, np.float32, np.float32,
                 _lib.load().mx_matmul_csr_dvec_float32)


# ----------------------------------------------------------------------------- list results
_VDT = {MX_F64: np.float64, MX_LGL: np.int32}


def _finish(res, info, alias_from=None, empty_values_dtype=np.float64):
    lib = _lib.load()
    vdt = _VDT.get(info.values_dtype)
    if info.alias_structure:
        indptr, indices = alias_from          # the INPUT objects themselves, as the reference returns them
        values = np.empty(info.values_len, dtype=vdt)
        check(lib.mx_result_finish(res, None, None, ptr(values)))
        return dict(indptr=indptr, indices=indices, values=values)
    indptr = np.empty(info.indptr_len, dtype=np.int32)
    indices = np.empty(info.nnz, dtype=np.int32)
    values = np.empty(info.values_len if vdt is not None else 0, dtype=vdt if vdt is not None else empty_values_dtype)
    check(lib.mx_result_finish(res, ptr(indptr), ptr(indices), ptr(values) if values.size else None))
    return dict(indptr=indptr, indices=indices, values=values)


def _same(a, b):
    return a is b


def _elemwise(op, indptr1, indptr2, indices1, indices2, values1, values2, vdt):
    lib = _lib.load()
    # keep object identity visible to the C-ABI as pointer identity (operators.cpp:104-108, :343-346)
    p1 = _i32(indptr1)
    p2 = p1 if _same(indptr1, indptr2) else _i32(indptr2)
    j1 = _i32(indices1)
    j2 = j1 if _same(indices1, indices2) else _i32(indices2)
    v1 = np.ascontiguousarray(values1, dtype=vdt)
    v2 = v1 if _same(values1, values2) else np.ascontiguousarray(values2, dtype=vdt)
    res = C.c_void_p()
    info = ResultInfo()
    check(lib.mx_csr_elemwise_begin(C.c_int(op), C.c_int(p1.size - 1), ptr(p1), ptr(p2), ptr(j1), ptr(j2),
                                    ptr(v1), ptr(v2), C.c_int64(j1.size), C.c_int64(j2.size),
                                    C.byref(res), C.byref(info)))
    return _finish(res, info, alias_from=(indptr1, indices1))


def multiply_csr_elemwise(indptr1, indptr2, indices1, indices2, values1, values2):
    """R/RcppExports.R:360 -> src/operators.cpp:209-222."""
    return _elemwise(MX_OP_MUL, indptr1, indptr2, indices1, indices2, values1, values2, np.float64)


def logicaland_csr_elemwise(indptr1, indptr2, indices1, indices2, values1, values2):
    """src/operators.cpp:224-237."""
    return _elemwise(MX_OP_AND, indptr1, indptr2, indices1, indices2, values1, values2, np.int32)


def add_csr_elemwise(indptr1, indptr2, indices1, indices2, values1, values2, substract):
    """R/RcppExports.R:388 -> src/operators.cpp:539-554."""
    return _elemwise(MX_OP_SUB if substract else MX_OP_ADD, indptr1, indptr2, indices1, indices2,
                     values1, values2, np.float64)


def logicalor_csr_elemwise(indptr1, indptr2, indices1, indices2, values1, values2, xor_op):
    """src/operators.cpp:556-571."""
    return _elemwise(MX_OP_XOR if xor_op else MX_OP_OR, indptr1, indptr2, indices1, indices2,
                     values1, values2, np.int32)


def _copy_rows(indptr, indices, values, rows_take, value_dtype, vdt):
    lib = _lib.load()
    p, j, rows = _i32(indptr), _i32(indices), _i32(rows_take)
    v = None if values is None else np.ascontiguousarray(values, dtype=vdt)
    res = C.c_void_p()
    info = ResultInfo()
    check(lib.mx_copy_csr_rows_begin(ptr(p), C.c_int(p.size - 1), ptr(j), ptr(v), C.c_int(value_dtype),
                                     C.c_int64(0 if v is None else v.size), ptr(rows), C.c_int64(rows.size),
                                     C.byref(res), C.byref(info)))
    return _finish(res, info, empty_values_dtype=vdt if vdt is not None else np.float64)


def copy_csr_rows_numeric(indptr, indices, values, rows_take):
    """R/RcppExports.R:564 -> src/slice.cpp:276-291."""
    return _copy_rows(indptr, indices, values, rows_take, MX_F64, np.float64)


def copy_csr_rows_logical(indptr, indices, values, rows_take):
    """src/slice.cpp:293-308."""
    return _copy_rows(indptr, indices, values, rows_take, MX_LGL, np.int32)


def copy_csr_rows_binary(indptr, indices, rows_take):
    """src/slice.cpp:310-324."""
    return _copy_rows(indptr, indices, None, rows_take, MX_NONE, None)


# ----------------------------------------------------------------------------- column-filtering slices (§8f-2)
def _col_seq(indptr, indices, values, rows_take, cols_take, index1, value_dtype, vdt):
    lib = _lib.load()
    p, j, rows, cols = _i32(indptr), _i32(indices), _i32(rows_take), _i32(cols_take)
    v = None if values is None else np.ascontiguousarray(values, dtype=vdt)
    res = C.c_void_p()
    info = ResultInfo()
    check(lib.mx_copy_csr_rows_col_seq_begin(ptr(p), C.c_int(p.size - 1), ptr(j), ptr(v), C.c_int(value_dtype),
                                             C.c_int64(0 if v is None else v.size), ptr(rows), C.c_int64(rows.size),
                                             ptr(cols), C.c_int64(cols.size), C.c_int(int(bool(index1))),
                                             C.byref(res), C.byref(info)))
    return _finish(res, info)          # values are always a numeric (float64) vector, slice.cpp:363


def copy_csr_rows_col_seq_numeric(indptr, indices, values, rows_take, cols_take, index1):
    """R/RcppExports.R:576 -> src/slice.cpp:385-403."""
    return _col_seq(indptr, indices, values, rows_take, cols_take, index1, MX_F64, np.float64)


def copy_csr_rows_col_seq_logical(indptr, indices, values, rows_take, cols_take, index1):
    """src/slice.cpp:405-423 (values come back as numeric, like the reference's NumericVector)."""
    return _col_seq(indptr, indices, values, rows_take, cols_take, index1, MX_LGL, np.int32)


def copy_csr_rows_col_seq_binary(indptr, indices, rows_take, cols_take, index1):
    """src/slice.cpp:425-443."""
    return _col_seq(indptr, indices, None, rows_take, cols_take, index1, MX_NONE, None)


def _arbitrary(indptr, indices, values, rows_take, cols_take, value_dtype, vdt):
    lib = _lib.load()
    p, j, rows, cols = _i32(indptr), _i32(indices), _i32(rows_take), _i32(cols_take)
    v = None if values is None else np.ascontiguousarray(values, dtype=vdt)
    res = C.c_void_p()
    info = ResultInfo()
    check(lib.mx_copy_csr_arbitrary_begin(ptr(p), C.c_int(p.size - 1), ptr(j), ptr(v), C.c_int(value_dtype),
                                          C.c_int64(0 if v is None else v.size), ptr(rows), C.c_int64(rows.size),
                                          ptr(cols), C.c_int64(cols.size), C.byref(res), C.byref(info)))
    out = _finish(res, info, empty_values_dtype=vdt if vdt is not None else np.float64)
    if info.values_dtype == MX_NONE:
        del out["values"]              # the reference's list has no `values` element then (slice.cpp:565)
    return out


def copy_csr_arbitrary_numeric(indptr, indices, values, rows_take, cols_take):
    """src/slice.cpp:580-596."""
    return _arbitrary(indptr, indices, values, rows_take, cols_take, MX_F64, np.float64)


def copy_csr_arbitrary_logical(indptr, indices, values, rows_take, cols_take):
    """src/slice.cpp:598-614."""
    return _arbitrary(indptr, indices, values, rows_take, cols_take, MX_LGL, np.int32)


def copy_csr_arbitrary_binary(indptr, indices, rows_take, cols_take):
    """src/slice.cpp:616-632."""
    return _arbitrary(indptr, indices, None, rows_take, cols_take, MX_NONE, None)


def _reverse_rows(indptr, indices, values, value_dtype, vdt):
    lib = _lib.load()
    p, j = _i32(indptr), _i32(indices)
    v = None if values is None else np.ascontiguousarray(values, dtype=vdt)
    res = C.c_void_p()
    info = ResultInfo()
    check(lib.mx_reverse_rows_begin(ptr(p), C.c_int(p.size - 1), ptr(j), ptr(v), C.c_int(value_dtype),
                                    C.c_int64(0 if v is None else v.size), C.byref(res), C.byref(info)))
    return _finish(res, info, empty_values_dtype=vdt if vdt is not None else np.float64)


def reverse_rows_numeric(indptr, indices, values):
    """src/slice.cpp:98-110."""
    return _reverse_rows(indptr, indices, values, MX_F64, np.float64)


def reverse_rows_logical(indptr, indices, values):
    """src/slice.cpp:112-124."""
    return _reverse_rows(indptr, indices, values, MX_LGL, np.int32)


def reverse_rows_binary(indptr, indices):
    """src/slice.cpp:126-138."""
    return _reverse_rows(indptr, indices, None, MX_NONE, None)


def _reverse_columns_inplace(indptr, indices, values, ncol, value_dtype):
    p = _i32(indptr)
    if not (isinstance(indices, np.ndarray) and indices.dtype == np.int32 and indices.flags.c_contiguous):
        raise TypeError("indices must be a contiguous int32 numpy array (modified in place)")
    check(_lib.load().mx_reverse_columns_inplace(ptr(p), C.c_int(p.size - 1), ptr(indices), ptr(values),
                                                 C.c_int(value_dtype), C.c_int64(0 if values is None else values.size),
                                                 C.c_int(int(ncol))))


def reverse_columns_inplace_numeric(indptr, indices, values, ncol):
    """src/slice.cpp:172-187: modifies `indices` / `values`."""
    _reverse_columns_inplace(indptr, indices, values, ncol, MX_F64)


def reverse_columns_inplace_logical(indptr, indices, values, ncol):
    """src/slice.cpp:189-204."""
    _reverse_columns_inplace(indptr, indices, values, ncol, MX_LGL)


def reverse_columns_inplace_binary(indptr, indices, ncol):
    """src/slice.cpp:206-221."""
    _reverse_columns_inplace(indptr, indices, None, ncol, MX_NONE)


# ----------------------------------------------------------------------------- CSR x sparse vector, CSR (.) dense (§8f-4)
def _svec(kind, X_csr_indptr, X_csr_indices, X_csr_values, y_indices_base1, y_values, nthreads):
    p, j, x, yi = _i32(X_csr_indptr), _i32(X_csr_indices), _f64(X_csr_values), _i32(y_indices_base1)
    yv = None
    if kind == 0:
        yv = _f64(y_values)
    elif kind in (1, 2):
        yv = _i32(y_values)
    elif kind == 4:
        yv = np.ascontiguousarray(y_values, dtype=np.float32)
    out = np.empty(p.size - 1, dtype=np.float64)
    check(_lib.load().mx_matmul_csr_svec(ptr(p), ptr(j), ptr(x), C.c_int(p.size - 1), ptr(yi), C.c_int64(yi.size),
                                         ptr(yv), C.c_int(kind), C.c_int(int(nthreads)), ptr(out)))
    return out


def matmul_csr_svec_numeric(X_csr_indptr, X_csr_indices, X_csr_values, y_indices_base1, y_values, nthreads=1):
    """src/matmul.cpp:555-571."""
    return _svec(0, X_csr_indptr, X_csr_indices, X_csr_values, y_indices_base1, y_values, nthreads)


def matmul_csr_svec_integer(X_csr_indptr, X_csr_indices, X_csr_values, y_indices_base1, y_values, nthreads=1):
    """src/matmul.cpp:573-589."""
    return _svec(1, X_csr_indptr, X_csr_indices, X_csr_values, y_indices_base1, y_values, nthreads)


def matmul_csr_svec_logical(X_csr_indptr, X_csr_indices, X_csr_values, y_indices_base1, y_values, nthreads=1):
    """src/matmul.cpp:591-607."""
    return _svec(2, X_csr_indptr, X_csr_indices, X_csr_values, y_indices_base1, y_values, nthreads)


def matmul_csr_svec_binary(X_csr_indptr, X_csr_indices, X_csr_values, y_indices_base1, nthreads=1):
    """src/matmul.cpp:609-624."""
    return _svec(3, X_csr_indptr, X_csr_indices, X_csr_values, y_indices_base1, None, nthreads)


def matmul_csr_svec_float32(X_csr_indptr, X_csr_indices, X_csr_values, y_indices_base1, y_values, nthreads=1):
    """src/matmul.cpp:626-641."""
    return _svec(4, X_csr_indptr, X_csr_indices, X_csr_values, y_indices_base1, y_values, nthreads)


def _csr_by_dense(kind, indptr, indices, values, dense_mat):
    p, j = _i32(indptr), _i32(indices)
    ddt = {0: np.float64, 1: np.float32, 2: np.int32, 3: np.int32, 4: np.int32}[kind]
    D = _dense(dense_mat, ddt)
    if D.shape[0] != p.size - 1:
        raise ValueError("dense operand must have as many rows as the sparse one")
    xv = np.ascontiguousarray(values, dtype=np.int32 if kind == 4 else np.float64)
    out = np.empty(xv.size, dtype=xv.dtype)
    check(_lib.load().mx_multiply_csr_by_dense_elemwise(ptr(p), ptr(j), ptr(xv), C.c_int(p.size - 1), ptr(D),
                                                        C.c_int64(D.shape[1]), C.c_int(kind), ptr(out)))
    return out


def multiply_csr_by_dense_elemwise_double(indptr, indices, values, dense_mat):
    """src/operators.cpp:289-296."""
    return _csr_by_dense(0, indptr, indices, values, dense_mat)


def multiply_csr_by_dense_elemwise_float32(indptr, indices, values, dense_mat):
    """src/operators.cpp:298-305."""
    return _csr_by_dense(1, indptr, indices, values, dense_mat)


def multiply_csr_by_dense_elemwise_int(indptr, indices, values, dense_mat):
    """src/operators.cpp:307-314."""
    return _csr_by_dense(2, indptr, indices, values, dense_mat)


def multiply_csr_by_dense_elemwise_bool(indptr, indices, values, dense_mat):
    """src/operators.cpp:316-323."""
    return _csr_by_dense(3, indptr, indices, values, dense_mat)


def logicaland_csr_by_dense_cpp(indptr, indices, values, dense_mat):
    """src/operators.cpp:325-334."""
    return _csr_by_dense(4, indptr, indices, values, dense_mat)


# ----------------------------------------------------------------------------- cbind / rbind (§8f-3)
def _cbind(Xp, Xj, Xx, Yp, Yj_plus_ncol, Yx, value_dtype, vdt):
    lib = _lib.load()
    Xp, Xj, Yp, Yj = _i32(Xp), _i32(Xj), _i32(Yp), _i32(Yj_plus_ncol)
    xv = None if Xx is None else np.ascontiguousarray(Xx, dtype=vdt)
    yv = None if Yx is None else np.ascontiguousarray(Yx, dtype=vdt)
    res = C.c_void_p()
    info = ResultInfo()
    check(lib.mx_cbind_csr_begin(ptr(Xp), C.c_int(Xp.size - 1), ptr(Xj), ptr(xv), C.c_int64(0 if xv is None else xv.size),
                                 ptr(Yp), C.c_int(Yp.size - 1), ptr(Yj), ptr(yv), C.c_int64(0 if yv is None else yv.size),
                                 C.c_int(value_dtype), C.byref(res), C.byref(info)))
    return _finish(res, info)


def cbind_csr_numeric(X_csr_indptr, X_csr_indices, X_csr_values, Y_csr_indptr, Y_csr_indices_plus_ncol, Y_csr_values):
    """src/cbind.cpp:101-119."""
    return _cbind(X_csr_indptr, X_csr_indices, X_csr_values, Y_csr_indptr, Y_csr_indices_plus_ncol, Y_csr_values,
                  MX_F64, np.float64)


def cbind_csr_logical(X_csr_indptr, X_csr_indices, X_csr_values, Y_csr_indptr, Y_csr_indices_plus_ncol, Y_csr_values):
    """src/cbind.cpp:121-139."""
    return _cbind(X_csr_indptr, X_csr_indices, X_csr_values, Y_csr_indptr, Y_csr_indices_plus_ncol, Y_csr_values,
                  MX_LGL, np.int32)


def cbind_csr_binary(X_csr_indptr, X_csr_indices, Y_csr_indptr, Y_csr_indices_plus_ncol):
    """src/cbind.cpp:141-157."""
    return _cbind(X_csr_indptr, X_csr_indices, None, Y_csr_indptr, Y_csr_indices_plus_ncol, None, MX_NONE, None)


class _RbindInput(C.Structure):
    _fields_ = [("kind", C.c_int), ("indptr", C.c_void_p), ("indices", C.c_void_p), ("values", C.c_void_p),
                ("nrows", C.c_int), ("nnz", C.c_int64)]


def concat_csr_batch(objects, out_kind):
    """concat_csr_batch (src/rbind.cpp:24-173) over plain arrays: objects = list of
    (kind, indptr|None, indices, values|None, nrows) with kind 0 dgR, 1 lgR, 2 ngR, 3/4/5/6 d/i/l/n sparseVector
    (1-based indices, one row); out_kind 0 dgR, 1 lgR, 2 ngR.  Returns dict(indptr, indices, values)."""
    lib = _lib.load()
    keep, arr = [], (_RbindInput * max(len(objects), 1))()
    for k, (kind, p, j, x, nr) in enumerate(objects):
        jj = _i32(j)
        pp = None if p is None else _i32(p)
        xx = None if x is None else np.ascontiguousarray(x, dtype=np.float64 if kind in (0, 3) else np.int32)
        keep += [jj, pp, xx]
        arr[k] = _RbindInput(kind, None if pp is None else pp.ctypes.data, jj.ctypes.data,
                             None if xx is None else xx.ctypes.data, int(nr), jj.size)
    res = C.c_void_p()
    info = ResultInfo()
    check(lib.mx_concat_csr_batch_begin(arr, C.c_int(len(objects)), C.c_int(out_kind), C.byref(res), C.byref(info)))
    out = _finish(res, info)
    if out_kind == 2:
        out["values"] = None
    return out


def check_is_seq(indices) -> bool:
    """src/slice.cpp:25-35."""
    a = _i32(indices)
    r = C.c_int(0)
    check(_lib.load().mx_check_is_seq(ptr(a), C.c_int64(a.size), C.byref(r)))
    return bool(r.value)


def check_is_rev_seq(indices) -> bool:
    """src/slice.cpp:37-47."""
    a = _i32(indices)
    r = C.c_int(0)
    check(_lib.load().mx_check_is_rev_seq(ptr(a), C.c_int64(a.size), C.byref(r)))
    return bool(r.value)


# ----------------------------------------------------------------------------- sort precondition (§8f-1)
def check_indices_are_sorted(indptr, indices) -> bool:
    """Per-row check_is_sorted, src/misc.cpp:118-128."""
    p, j = _i32(indptr), _i32(indices)
    r = C.c_int(0)
    check(_lib.load().mx_check_indices_are_sorted(ptr(p), ptr(j), C.c_int(p.size - 1), C.byref(r)))
    return bool(r.value)


def sort_sparse_indices_inplace(indptr, indices, values=None):
    """sort_sparse_indices_{numeric,logical}_known_ncol / _binary (src/misc.cpp:333-378): sorts the given
    int32 `indices` (and `values`) arrays IN PLACE, like the R-side call does."""
    p = _i32(indptr)
    if not (isinstance(indices, np.ndarray) and indices.dtype == np.int32 and indices.flags.c_contiguous):
        raise TypeError("indices must be a contiguous int32 numpy array (sorted in place)")
    if values is None:
        vd = MX_NONE
    elif values.dtype == np.float64:
        vd = MX_F64
    elif values.dtype == np.int32:
        vd = MX_LGL
    else:
        raise TypeError("values must be float64 or int32 (R logical)")
    check(_lib.load().mx_sort_sparse_indices(ptr(p), ptr(indices), ptr(values), C.c_int(vd), C.c_int(p.size - 1)))


def multiply_csr_by_dvec_no_NAs_numeric(indptr, indices, values, dvec, ncols, multiply, powerto, divide, divrest,
                                        intdiv, X_is_LHS):
    """src/operators.cpp:2142-2175 (R/RcppExports.R:480-482): values-only `X op v` / `v op X` with R's recycling."""
    p, j = _i32(indptr), _i32(indices)
    xv = np.ascontiguousarray(values, dtype=np.float64)
    dv = np.ascontiguousarray(dvec, dtype=np.float64).reshape(-1)
    out = np.empty(xv.size, dtype=np.float64)
    check(_lib.load().mx_multiply_csr_by_dvec_no_NAs_numeric(
        ptr(p), ptr(j), ptr(xv), C.c_int(p.size - 1), ptr(dv), C.c_int64(dv.size), C.c_int(int(ncols)),
        C.c_int(bool(multiply)), C.c_int(bool(powerto)), C.c_int(bool(divide)), C.c_int(bool(divrest)),
        C.c_int(bool(intdiv)), C.c_int(bool(X_is_LHS)), ptr(out)))
    return out


def multiply_csr_by_dvec_with_NAs(indptr, indices, values, dvec, ncols, multiply, powerto, divide, divrest, intdiv, X_is_LHS):
    """src/operators.cpp:2258-2856: the structure-changing route of `CSR op vector` (NA / NaN in the vector, zeros under
    / %% %/% ^, negatives under ^, infinities under *).  Returns dict(indptr, indices, values); when no cell is added the
    INPUT indptr / indices objects come back, as in the reference (:2639-2647)."""
    p, j = _i32(indptr), _i32(indices)
    xv = np.ascontiguousarray(values, dtype=np.float64)
    dv = np.ascontiguousarray(dvec, dtype=np.float64).reshape(-1)
    res = C.c_void_p()
    info = ResultInfo()
    check(_lib.load().mx_multiply_csr_by_dvec_with_NAs_begin(
        ptr(p), ptr(j), ptr(xv), C.c_int(p.size - 1), ptr(dv), C.c_int64(dv.size), C.c_int(int(ncols)),
        C.c_int(bool(multiply)), C.c_int(bool(powerto)), C.c_int(bool(divide)), C.c_int(bool(divrest)), C.c_int(bool(intdiv)),
        C.c_int(bool(X_is_LHS)), C.byref(res), C.byref(info)))
    return _finish(res, info, alias_from=(indptr, indices))


def logicaland_csr_by_dvec_internal(indptr, indices, values, dvec, ncols):
    """src/operators.cpp:2177-2200 (R/RcppExports.R:484-486): R logicals in, R logicals out."""
    p, j = _i32(indptr), _i32(indices)
    xv = np.ascontiguousarray(values, dtype=np.int32)
    dv = np.ascontiguousarray(dvec, dtype=np.int32).reshape(-1)
    out = np.empty(xv.size, dtype=np.int32)
    check(_lib.load().mx_logicaland_csr_by_dvec_internal(ptr(p), ptr(j), ptr(xv), C.c_int(p.size - 1), ptr(dv),
                                                         C.c_int64(dv.size), C.c_int(int(ncols)), ptr(out)))
    return out


def _remove_zero_valued_csr(indptr, indices, values, remove_NAs, value_dtype, vdt):
    lib = _lib.load()
    p, j = _i32(indptr), _i32(indices)
    v = np.ascontiguousarray(values, dtype=vdt)
    res = C.c_void_p()
    info = ResultInfo()
    check(lib.mx_remove_zero_valued_csr_begin(ptr(p), ptr(j), ptr(v), C.c_int(value_dtype), C.c_int(p.size - 1),
                                              C.c_int(1 if remove_NAs else 0), C.byref(res), C.byref(info)))
    if info.alias_structure == 2:          # nothing to remove: the INPUT objects themselves (misc.cpp:586-590)
        check(lib.mx_result_discard(res))
        return dict(indptr=indptr, indices=indices, values=values)
    return _finish(res, info)


def remove_zero_valued_csr_numeric(indptr, indices, values, remove_NAs):
    """R/RcppExports.R `remove_zero_valued_csr_numeric` -> src/misc.cpp:667-682."""
    return _remove_zero_valued_csr(indptr, indices, values, remove_NAs, MX_F64, np.float64)


def remove_zero_valued_csr_logical(indptr, indices, values, remove_NAs):
    """R/RcppExports.R `remove_zero_valued_csr_logical` -> src/misc.cpp:684-698."""
    return _remove_zero_valued_csr(indptr, indices, values, remove_NAs, MX_LGL, np.int32)


def check_valid_csr_matrix(indptr, indices, nrows, ncols):
    """R/RcppExports.R `check_valid_csr_matrix` -> src/misc.cpp:970-1016: dict(err=...) or an empty dict."""
    lib = _lib.load()
    p, j = _i32(indptr), _i32(indices)
    code = C.c_int(0)
    msg = C.c_char_p()
    check(lib.mx_check_valid_csr_matrix(ptr(p), ptr(j), C.c_int64(j.size), C.c_int(int(nrows)), C.c_int(int(ncols)),
                                        C.byref(code), C.byref(msg)))
    return dict(err=msg.value.decode()) if code.value else dict()


def matmul_rowvec_by_csc(rowvec, indptr, indices, values):
    """R/RcppExports.R `matmul_rowvec_by_csc` -> src/matmul.cpp:643-663: float32 row vector x CSC, (1, ncols) float32."""
    lib = _lib.load()
    r = np.ascontiguousarray(rowvec, dtype=np.float32).reshape(-1)
    p, j = _i32(indptr), _i32(indices)
    v = None if values is None else _f64(values)
    out = np.zeros((1, p.size - 1), dtype=np.float32)
    check(lib.mx_matmul_rowvec_by_csc(ptr(r), C.c_int(r.size), ptr(p), ptr(j), ptr(v), C.c_int(p.size - 1), ptr(out)))
    return out


def matmul_rowvec_by_cscbin(rowvec, indptr, indices):
    """R/RcppExports.R `matmul_rowvec_by_cscbin` -> src/matmul.cpp:665-684 (pattern matrix: every stored entry is 1)."""
    return matmul_rowvec_by_csc(rowvec, indptr, indices, None)
