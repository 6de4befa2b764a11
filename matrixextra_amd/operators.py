"""Host-side mirror of the CSR (+) CSR part of R/operators.R
(multiply_csr_by_csr :43-79, add_csr_matrices_internal :713-776 and their registrations) and of
`CSR op vector` (multiply_csr_by_dvec_elemwise_internal :950-1153)."""
from __future__ import annotations

import warnings

import numpy as np

from . import exports
from .matrices import (NA_REAL, RsparseMatrix, as_csr_matrix, check_valid_matrix, dgRMatrix, lgRMatrix, ngRMatrix,
                       options, sort_sparse_indices, stop)


def _is_same_ngRMatrix(e1, e2):
    """is_same_ngRMatrix (src/misc.cpp:108-116): pointer identity of p and j."""
    return (e1.p.ctypes.data == e2.p.ctypes.data and e1.j.ctypes.data == e2.j.ctypes.data
            and e1.p.size == e2.p.size and e1.j.size == e2.j.size)


def _deepcopy_before_sort(e):
    return e.copy()


def _prepare(e, logical):
    """check_valid_matrix; [deepcopy]; as.csr.matrix; sort_sparse_indices — R/operators.R:54-64, :744-754."""
    if not isinstance(e, RsparseMatrix):
        e = as_csr_matrix(e, logical=logical)
    inplace_sort = bool(options.get("MatrixExtra.inplace_sort", False))
    check_valid_matrix(e)
    if inplace_sort:
        e = _deepcopy_before_sort(e)
    e = as_csr_matrix(e, logical=logical)
    return sort_sparse_indices(e, copy=not inplace_sort)


def _assemble(cls, e1, res):
    out = cls.__new__(cls)
    out.Dim = e1.Dim
    out.Dimnames = list(e1.Dimnames)
    out.p = res["indptr"]
    out.j = res["indices"]
    out.x = res["values"]
    return out


def multiply_csr_by_csr(e1, e2, logical=False):
    """R/operators.R:43-79."""
    if e1.Dim[0] != e2.Dim[0] or e1.Dim[1] != e2.Dim[1]:
        stop("Matrices must have the same dimensions in order to multiply them.")
    if isinstance(e1, ngRMatrix) and isinstance(e2, ngRMatrix) and _is_same_ngRMatrix(e1, e2):
        return e1
    e1 = _prepare(e1, logical)
    e2 = _prepare(e2, logical)
    if not logical:
        res = exports.multiply_csr_elemwise(e1.p, e2.p, e1.j, e2.j, e1.x, e2.x)
        return _assemble(dgRMatrix, e1, res)
    res = exports.logicaland_csr_elemwise(e1.p, e2.p, e1.j, e2.j, e1.x, e2.x)
    return _assemble(lgRMatrix, e1, res)


def add_csr_matrices_internal(e1, e2, is_substraction=False, is_ampersand=False, is_xor=False):
    """R/operators.R:713-776 (`is_ampersand` is the reference's name for the `|` path)."""
    if e1.Dim[0] != e2.Dim[0] or e1.Dim[1] != e2.Dim[1]:
        stop("Matrices must have the same dimensions in order to add/substract them.")
    logical = is_ampersand or is_xor
    if isinstance(e1, ngRMatrix) and isinstance(e2, ngRMatrix) and _is_same_ngRMatrix(e1, e2):
        if not is_substraction and not is_xor:
            return e1
        if is_xor:
            return lgRMatrix(np.zeros(e1.Dim[0] + 1, dtype=np.int32), np.zeros(0, dtype=np.int32),
                             np.zeros(0, dtype=np.int32), e1.Dim, e1.Dimnames)
        # R/operators.R:731-738 (sic: the reference fills 2.0 on this branch)
        return dgRMatrix(e1.p, e1.j, np.full(e1.j.size, 2.0), e1.Dim, e1.Dimnames)
    e1 = _prepare(e1, logical)
    e2 = _prepare(e2, logical)
    if not logical:
        res = exports.add_csr_elemwise(e1.p, e2.p, e1.j, e2.j, e1.x, e2.x, is_substraction)
        return _assemble(dgRMatrix, e1, res)
    res = exports.logicalor_csr_elemwise(e1.p, e2.p, e1.j, e2.j, e1.x, e2.x, bool(is_xor))
    return _assemble(lgRMatrix, e1, res)


def add_csr_matrices(e1, e2, is_substraction=False):
    """R/operators.R:778-780."""
    return add_csr_matrices_internal(e1, e2, is_substraction, False, False)


def logicalor_csr_matrices(e1, e2):
    """R/operators.R:782-784."""
    return add_csr_matrices_internal(e1, e2, False, True, False)


def xor_csr_matrices(e1, e2):
    """R/operators.R:786-788."""
    return add_csr_matrices_internal(e1, e2, False, False, True)


_NOT_ACCELERATED = ("This combination takes the reference's NA / dense route "
                    "(multiply_csr_by_dvec_with_NAs or a CsparseMatrix fallback, R/operators.R:%s), "
                    "which is not on the accelerated path.")


def _as_logical(v):
    """as.logical() for a numeric / bool vector -> R logical (int32 with NA_LOGICAL)."""
    v = np.asarray(v)
    if v.dtype == np.int32:
        return v
    if v.dtype.kind == "f":
        return np.where(np.isnan(v), np.int32(-2147483648), (v != 0).astype(np.int32)).astype(np.int32)
    return (v != 0).astype(np.int32)


def multiply_csr_by_dvec_elemwise_internal(e1, e2, logical=False, X_is_LHS=True, op="*"):
    """R/operators.R:950-1153 for RsparseMatrix `e1`: `e1 op e2` (or `e2 op e1` when X_is_LHS is false) with a dense
    vector (or a same-shape dense matrix read as a vector), R's recycling: values-only result, or — when the vector holds
    NA / NaN, zeros under / %% %/% ^, negatives under ^, infinities under * — the structure-changing route
    multiply_csr_by_dvec_with_NAs (:1112-1131).  The routes the reference sends through a CsparseMatrix (an all-NA
    vector, a scalar that needs the NA route, `v op X` for ^ / %% %/% while NAs are kept) raise: they stay on the CPU."""
    e2 = np.asarray(e2)
    if e2.ndim == 2:                                                          # :952-959
        if e1.Dim[0] != e2.shape[0] or e2.shape[1] != e2.shape[1]:            # (sic: the reference compares ncol(e2) with itself)
            stop("Matrix dimensions do not match. Cannot perform the opertion.")
        e2 = e2.reshape(-1, order="F")
    e2 = e2.reshape(-1)
    if e2.size == 0:                                                          # :961-966
        return np.zeros(0, dtype=np.int32 if logical else np.float64)
    if e2.size > e1.Dim[0] * e1.Dim[1]:
        stop("Vector to multiply with has more entries than matrix.")
    keep_NAs = not bool(options.get("MatrixExtra.ignore_na", False))
    if (not X_is_LHS) and keep_NAs and op in ("^", "/", "%%", "%/%"):         # :973-978
        stop(_NOT_ACCELERATED % "973-978")
    check_valid_matrix(e1)
    # as.numeric(): NA_integer_ / NA (logical) become NA_real_ — the NaN whose low word is 1954, which the NA route tells
    # apart from a plain NaN
    if e2.dtype == np.int32:
        e2f = e2.astype(np.float64)
        e2f[e2 == np.int32(-2147483648)] = NA_REAL
    else:
        e2f = e2.astype(np.float64)
    take_route_NAs = (not logical) and keep_NAs and (
        bool(np.isnan(e2f).any())
        or (op in ("^", "/", "%%", "%/%") and bool((e2f == 0).any()))
        or (op == "*" and bool(np.isinf(e2f).any()))
        or (op == "^" and bool((e2f < 0).any())))                             # :981-988
    if take_route_NAs:                                                        # :995-1013
        if bool(np.isnan(e2f).all()):
            stop(_NOT_ACCELERATED % "997-1005")                               # all NA: the reference goes through a CsparseMatrix
    e2 = _as_logical(e2) if logical else e2f
    e1 = as_csr_matrix(e1, logical=logical)                                   # :1020-1029
    out = type(e1).__new__(type(e1))
    out.p, out.j, out.Dim, out.Dimnames = e1.p, e1.j, e1.Dim, list(e1.Dimnames)
    if e2.size == 1:                                                          # :1031-1108
        if logical:
            if e2[0] == np.int32(-2147483648):
                out.x = np.where(e1.x == np.int32(-2147483648), np.int32(-2147483648), np.int32(0)).astype(np.int32)
                return out
            if e2[0] == 0:
                res = lgRMatrix(np.zeros(e1.Dim[0] + 1, dtype=np.int32), np.zeros(0, dtype=np.int32),
                                np.zeros(0, dtype=np.int32), e1.Dim, e1.Dimnames)
                return res
            return e1
        if op in ("/", "%%", "%/%") and e2[0] == 0 and X_is_LHS:
            warnings.warn("Warning: division by zero.")
        if op == "^" and (not X_is_LHS) and (e2[0] == 1 or e2[0] == 0):
            stop(_NOT_ACCELERATED % "1091-1095")
    if take_route_NAs and e2.size == 1:
        # :1056-1105: a scalar that needs the NA route (x * Inf, x / 0, x ^ 0 ...) is computed on a CsparseMatrix upstream
        stop(_NOT_ACCELERATED % "1056-1105")
    if take_route_NAs:                                                        # :1112-1114: the route needs rows sorted by column
        e1 = sort_sparse_indices(e1, copy=not bool(options.get("MatrixExtra.inplace_sort", False)))
    if e2.size != 1 and e1.Dim[0] % e2.size != 0:                             # :1116-1117
        warnings.warn("Number of elements in vector is not a multiple of matrix dimension.")
    if take_route_NAs:                                                        # :1119-1131
        res = exports.multiply_csr_by_dvec_with_NAs(e1.p, e1.j, e1.x, e2, e1.Dim[1], op == "*", op == "^", op == "/",
                                                    op == "%%", op == "%/%", X_is_LHS)
        out.p, out.j, out.x = res["indptr"], res["indices"], res["values"]
        return out
    if logical:
        out.x = exports.logicaland_csr_by_dvec_internal(e1.p, e1.j, e1.x, e2, e1.Dim[1])
    else:
        out.x = exports.multiply_csr_by_dvec_no_NAs_numeric(e1.p, e1.j, e1.x, e2, e1.Dim[1], op == "*", op == "^",
                                                            op == "/", op == "%%", op == "%/%", X_is_LHS)
    return out


def csr_op_vector(e1, e2, op, X_is_LHS=True):
    """`X * v`, `X / v`, `X ^ v`, `X %% v`, `X %/% v`, `X & v` and their mirrored forms (R/operators.R:1155-1215)."""
    if op == "&":
        return multiply_csr_by_dvec_elemwise_internal(e1, e2, logical=True)
    return multiply_csr_by_dvec_elemwise_internal(e1, e2, logical=False, X_is_LHS=X_is_LHS, op=op)
