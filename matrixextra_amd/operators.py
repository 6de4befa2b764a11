"""Host-side mirror of the CSR (+) CSR part of R/operators.R
(multiply_csr_by_csr :43-79, add_csr_matrices_internal :713-776 and their registrations)."""
from __future__ import annotations

import numpy as np

from . import exports
from .matrices import (RsparseMatrix, as_csr_matrix, check_valid_matrix, dgRMatrix, lgRMatrix, ngRMatrix,
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
