"""Host-side mirror of R/matmul.R for the <CSR, dense> products (the north-star path).

Function names, argument meaning, checks, error messages and result shapes
follow the reference's R glue; the numerics happen in the HIP kernels behind
matrixextra_amd.exports.  Reference call stacks: SURVEY.md §3.1-3.2.
"""
from __future__ import annotations

import numpy as np

from . import exports
from .matrices import (DenseMatrix, RsparseMatrix, as_csr_matrix, check_valid_matrix, dgCMatrix,
                       dimnames_of, float32, options, stop)


def _nthreads():
    return max(int(options.get("MatrixExtra.nthreads", 1)), 1)


def _nrow(x):
    return x.Dim[0] if hasattr(x, "Dim") else (x.Data.shape[0] if isinstance(x, float32) else x.shape[0])


def _ncol(x):
    return x.Dim[1] if hasattr(x, "Dim") else (x.Data.shape[1] if isinstance(x, float32) else x.shape[1])


def check_dimensions_match(x, y, matmult=False, crossprod=False, tcrossprod=False):
    """R/matmul.R:130-146."""
    if matmult:
        inner_x, inner_y = _ncol(x), _nrow(y)
    elif crossprod:
        inner_x, inner_y = _nrow(x), _nrow(y)
    elif tcrossprod:
        inner_x, inner_y = _ncol(x), _ncol(y)
    else:
        stop("Unexpected error. Please open an issue in GitHub explaining what you were doing.")
    if inner_x != inner_y:
        stop("Matrix dimensions do not match.")


def set_dimnames(res, x, y, matmult=False, crossprod=False, tcrossprod=False):
    """R/matmul.R:148-167."""
    dx, dy = dimnames_of(x), dimnames_of(y)
    if matmult:
        rnames, cnames = dx[0], dy[1]
    elif crossprod:
        rnames, cnames = dx[1], dy[1]
    else:
        rnames, cnames = dx[0], dy[0]
    if isinstance(res, float32):
        res.Dimnames = [rnames, cnames]
        return res
    return DenseMatrix(res, [rnames, cnames])


def _as_double_matrix(y):
    y = np.asarray(y)
    if y.ndim != 2:
        stop("Matrix dimensions do not match.")
    return y if y.dtype == np.float64 else y.astype(np.float64)      # mode(y) <- "double"  R/matmul.R:443


# ---- CSR x dense ---------------------------------------------------------------------------------
def tcrossprod_csr_dense(x, y):
    """tcrossprod(RsparseMatrix, matrix) — R/matmul.R:436-457."""
    check_dimensions_match(x, y, tcrossprod=True)
    nthreads = _nthreads()
    y_names = dimnames_of(y)
    yd = _as_double_matrix(y)
    x = as_csr_matrix(x)
    check_valid_matrix(x)
    res = exports.tcrossprod_csr_dense_numeric(x.p, x.j, x.x, yd, nthreads)
    return set_dimnames(res, x, DenseMatrix(yd, y_names), tcrossprod=True)


def gemm_csr_dense(x, y):
    """RsparseMatrix %*% matrix = tcrossprod_csr_dense(x, t(y)) — R/matmul.R:463-465."""
    y_names = dimnames_of(y)
    yt = DenseMatrix(np.asarray(y).T, [y_names[1], y_names[0]])      # base-R t(y): a dense transpose
    return tcrossprod_csr_dense(x, yt)


def tcrossprod_csr_f32(x, y):
    """tcrossprod(RsparseMatrix, float32) — R/matmul.R:514-536."""
    check_dimensions_match(x, y, tcrossprod=True)
    nthreads = _nthreads()
    x = as_csr_matrix(x)
    check_valid_matrix(x)
    res = float32(exports.tcrossprod_csr_dense_float32(x.p, x.j, x.x, y.Data, nthreads))
    return set_dimnames(res, x, y, tcrossprod=True)


def gemm_csr_f32(x, y):
    """RsparseMatrix %*% float32 — R/matmul.R:471-512 (transposes y with float's t())."""
    yt = float32(y.Data.T, [dimnames_of(y)[1], dimnames_of(y)[0]])
    return tcrossprod_csr_f32(x, yt)


# ---- dense x CSC / dense x t(CSR) ----------------------------------------------------------------
def gemm_dense_csc(x, y):
    """matrix %*% CsparseMatrix — R/matmul.R:171-198."""
    check_dimensions_match(x, y, matmult=True)
    nthreads = _nthreads()
    x_names = dimnames_of(x)
    xd = _as_double_matrix(x)
    check_valid_matrix(y)
    res = exports.matmul_dense_csc_numeric(xd, y.p, y.i, y.x, nthreads)
    return set_dimnames(res, DenseMatrix(xd, x_names), y, matmult=True)


def gemm_f32_csc(x, y):
    """float32 %*% CsparseMatrix — R/matmul.R:202-279."""
    if x.is_vector:                       # R/matmul.R:209-262: a float32 VECTOR is a row vector unless y has one row
        check_valid_matrix(y)
        if y.nrow() == 1:                 # (outer product -> dgCMatrix, matmul_colvec_by_scolvecascsr_f32: stays on the CPU)
            stop("float32 vector %*% one-row CsparseMatrix is not on the GPU path.")
        if y.nrow() != x.Data.size:
            stop("(row) vector-Matrix multiplication dimensions do not match.")
        return float32(exports.matmul_rowvec_by_csc(x.Data, y.p, y.i, y.x))
    check_dimensions_match(x, y, matmult=True)
    check_valid_matrix(y)
    res = float32(exports.matmul_dense_csc_float32(x.Data, y.p, y.i, y.x, _nthreads()))
    return set_dimnames(res, x, y, matmult=True)


def crossprod_dense_csc(x, y):
    """crossprod(matrix, CsparseMatrix) = gemm_dense_csc(t(x), y) — R/matmul.R:387-391."""
    x_names = dimnames_of(x)
    return gemm_dense_csc(DenseMatrix(np.asarray(x).T, [x_names[1], x_names[0]]), y)


def tcrossprod_dense_csr(x, y):
    """tcrossprod(matrix, RsparseMatrix) — R/matmul.R:283-305."""
    check_dimensions_match(x, y, tcrossprod=True)
    nthreads = _nthreads()
    x_names = dimnames_of(x)
    xd = _as_double_matrix(x)
    y = as_csr_matrix(y)
    check_valid_matrix(y)
    res = exports.tcrossprod_dense_csr_numeric(xd, y.p, y.j, y.x, nthreads, y.Dim[1])
    return set_dimnames(res, DenseMatrix(xd, x_names), y, tcrossprod=True)


def tcrossprod_f32_csr(x, y):
    """tcrossprod(float32, RsparseMatrix) — R/matmul.R:309-383."""
    check_dimensions_match(x, y, tcrossprod=True)
    y = as_csr_matrix(y)
    check_valid_matrix(y)
    res = float32(exports.tcrossprod_dense_csr_float32(x.Data, y.p, y.j, y.x, _nthreads(), y.Dim[1]))
    return set_dimnames(res, x, y, tcrossprod=True)


# ---- CSR x dense vector ------------------------------------------------------------------------------
def gemv_csr_vec(x, y):
    """RsparseMatrix %*% numeric/integer/logical/float32 vector — R/matmul.R:545-657 (dense branch).
    Returns an (nrow, 1) matrix like `matrix(res, ncol=1)`; float32 input -> float32 result."""
    is_f32 = isinstance(y, float32)
    yv = y.Data.reshape(-1) if is_f32 else np.asarray(y)
    if yv.ndim != 1:
        stop("Matrix-vector dimensions do not match.")
    if x.Dim[1] != yv.size:
        stop("Matrix-vector dimensions do not match.")
    nthreads = options.get("MatrixExtra.nthreads", 1)
    check_valid_matrix(x)
    x = as_csr_matrix(x)
    if is_f32:
        res = exports.matmul_csr_dvec_float32(x.p, x.j, x.x, yv, nthreads)
    elif yv.dtype == np.float64:
        res = exports.matmul_csr_dvec_numeric(x.p, x.j, x.x, yv, nthreads)
    elif yv.dtype == np.bool_:
        res = exports.matmul_csr_dvec_logical(x.p, x.j, x.x, yv.astype(np.int32), nthreads)
    elif yv.dtype == np.int32:
        # R keeps integer and logical apart by type; an int32 vector is an R integer unless tagged
        if getattr(y, "r_logical", False):
            res = exports.matmul_csr_dvec_logical(x.p, x.j, x.x, yv, nthreads)
        else:
            res = exports.matmul_csr_dvec_integer(x.p, x.j, x.x, yv, nthreads)
    else:
        return gemv_csr_vec(x, yv.astype(np.float64))        # as.numeric(y) fallback, R/matmul.R:589-593
    rn = dimnames_of(x)[0]
    if is_f32:
        return float32(res.reshape(-1, 1), [rn, None])
    return DenseMatrix(res.reshape(-1, 1), [rn, None])


class RLogical(np.ndarray):
    """An int32 vector tagged as an R logical ({0,1,NA_LOGICAL}) so `%*%` picks the logical kernel."""
    r_logical = True

    def __new__(cls, data):
        return np.ascontiguousarray(data, dtype=np.int32).view(cls)


# ---- dispatch (setMethod("%*%"/"tcrossprod"/"crossprod", ...)) ----------------------------------------
def matmul(x, y):
    """`%*%` for the signatures the hot path registers (R/matmul.R:200,281,469,512,755-767)."""
    if isinstance(x, RsparseMatrix):
        if isinstance(y, float32):
            return gemv_csr_vec(x, y) if y.is_vector else gemm_csr_f32(x, y)
        y_arr = np.asarray(y)
        if y_arr.ndim == 1:
            return gemv_csr_vec(x, y)
        return gemm_csr_dense(x, y)
    if isinstance(y, dgCMatrix):
        return gemm_f32_csc(x, y) if isinstance(x, float32) else gemm_dense_csc(x, y)
    stop("Unsupported operand types for %*% in the MI355X hot path.")


def tcrossprod(x, y):
    """tcrossprod for (Rsparse, matrix|float32) and (matrix|float32, Rsparse) — R/matmul.R:307,385,461,538."""
    if isinstance(x, RsparseMatrix):
        return tcrossprod_csr_f32(x, y) if isinstance(y, float32) else tcrossprod_csr_dense(x, y)
    if isinstance(y, RsparseMatrix):
        return tcrossprod_f32_csr(x, y) if isinstance(x, float32) else tcrossprod_dense_csr(x, y)
    stop("Unsupported operand types for tcrossprod in the MI355X hot path.")


def crossprod(x, y):
    """crossprod(matrix, CsparseMatrix) — R/matmul.R:393."""
    if isinstance(y, dgCMatrix) and not isinstance(x, float32):
        return crossprod_dense_csc(x, y)
    stop("Unsupported operand types for crossprod in the MI355X hot path.")
