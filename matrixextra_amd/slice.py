"""Host-side mirror of the row-selection part of R/slice.R (`X[i, ]`).

`subset_csr(x, i, j)` takes R-style indices: 1-based integers (repeats allowed),
negative integers = exclusion, logical masks (recycled like base R), dimnames.
Path selection follows R/slice.R:439-565: a contiguous ascending run of rows with
all columns is sliced on the host exactly as the reference does in pure R
(:477-483, never reaches native code there either); everything else goes to the
native routines — row gather, column-range / arbitrary-column slices, reversals.
"""
from __future__ import annotations

import numpy as np

from . import exports
from .matrices import (RsparseMatrix, check_valid_matrix, dgRMatrix, lgRMatrix, ngRMatrix, stop)


def get_indices_integer(i, max_i, index_names):
    """R/slice.R:2-53 for integer / logical / character `i`; returns 1-based int32."""
    i = np.asarray(i)
    if i.dtype.kind in ("U", "S", "O"):
        if index_names is not None:
            lookup = {nm: k + 1 for k, nm in reversed(list(enumerate(index_names)))}
            try:
                i = np.array([lookup[s] for s in i.tolist()], dtype=np.int64)
            except KeyError:
                stop("some of row subset indices are not present in matrix")
        else:
            i = i.astype(np.int64)
    if i.dtype == np.bool_:
        if i.size != max_i:
            if i.size > max_i:
                stop("some of row subset indices are not present in matrix")
            reps = -(-max_i // max(i.size, 1))
            i = np.tile(i, reps)[:max_i]            # seq(1, max_i)[i] recycles the mask
        i = np.flatnonzero(i) + 1
    i = i.astype(np.int64, copy=False).reshape(-1)
    if np.any(i <= 0):
        if np.any(i > 0):
            stop("can't mix positive and negative subscripts")
        keep = np.ones(max_i, dtype=bool)
        drop = -i[i < 0]
        if np.any(drop > max_i):
            drop = drop[drop <= max_i]
        keep[drop - 1] = False
        i = np.flatnonzero(keep) + 1
    if i.size and np.any(i > max_i):
        stop("some of row subset indices are not present in matrix")
    return i.astype(np.int32)


def _empty_like(x, n_row, row_names):
    cls = dgRMatrix if isinstance(x, dgRMatrix) else lgRMatrix if isinstance(x, lgRMatrix) else ngRMatrix
    xv = None if cls is ngRMatrix else np.zeros(0, dtype=cls.value_dtype)
    return cls(np.zeros(n_row + 1, dtype=np.int32), np.zeros(0, dtype=np.int32), xv,
               (n_row, x.Dim[1]), [row_names, x.Dimnames[1]])


def _cls_kind(x):
    return "d" if isinstance(x, dgRMatrix) else "l" if isinstance(x, lgRMatrix) else "n"


def _call3(kind, fn_numeric, fn_logical, fn_binary, p, j, xv, *rest):
    if kind == "d":
        return fn_numeric(p, j, xv, *rest)
    if kind == "l":
        return fn_logical(p, j, xv, *rest)
    return fn_binary(p, j, *rest)


def subset_csr(x, i=None, j=None, drop=False):
    """`x[i, j]` for RsparseMatrix — R/slice.R:293-585 (integer / logical / name / negative indices; NA indices and
    `drop` to vectors are not mirrored).  Branch selection is the reference's (:439-565):
      contiguous ascending rows, all columns      -> host slicing, as the reference does in pure R (:477-483)
      descending rows, all columns                -> host slicing + reverse_rows_*                    (:484-501)
      other rows, all columns                     -> copy_csr_rows_*            (row-gather kernel)   (:502-515)
      contiguous ascending columns                -> copy_csr_rows_col_seq_*                          (:516-529)
      contiguous descending columns               -> copy_csr_rows_col_seq_* + reverse_columns_inplace_* (:530-546)
      anything else                               -> copy_csr_arbitrary_*                             (:547-565)
      full reversal of both                       -> reverse_rows_* + reverse_columns_inplace_*       (:439-469)"""
    check_valid_matrix(x)
    if i is None and j is None:
        return x
    row_names, col_names = x.Dimnames[0], x.Dimnames[1]
    nrow, ncol = x.Dim
    all_i = all_j = i_is_seq = j_is_seq = i_is_rev_seq = j_is_rev_seq = False
    if j is None:
        all_j = True
        j = np.arange(1, ncol + 1, dtype=np.int32)
    else:
        j = get_indices_integer(j, ncol, col_names)
        if j.size == ncol and ncol > 0 and j[0] == 1 and j[-1] == ncol:
            all_j = exports.check_is_seq(j)
        else:
            j_is_seq = exports.check_is_seq(j)
    if i is None:
        i = np.arange(1, nrow + 1, dtype=np.int32)
        all_i = i_is_seq = True
    else:
        i = get_indices_integer(i, nrow, row_names)
        i_is_seq = exports.check_is_seq(i)
        if i_is_seq and i.size == nrow and nrow > 0 and i[0] == 1 and i[-1] == nrow:
            all_i = True
    n_row, n_col = int(i.size), int(j.size)
    if not all_i and not i_is_seq:
        i_is_rev_seq = exports.check_is_rev_seq(i)
    if not all_j and not j_is_seq:
        j_is_rev_seq = exports.check_is_rev_seq(j)
    new_rn = None if row_names is None or not len(row_names) else [row_names[k - 1] for k in i]
    new_cn = None if col_names is None or not len(col_names) else [col_names[k - 1] for k in j]

    if n_row == 0 or n_col == 0 or x.j.size == 0:              # R/slice.R:404-421
        out = _empty_like(x, n_row, new_rn)
        out.Dim = (n_row, n_col)
        out.Dimnames = [new_rn, new_cn]
        return out
    if all_i and all_j:                                         # R/slice.R:423-425
        return x

    kind = _cls_kind(x)
    has_x = x.x is not None
    E = exports

    def finish(indptr, col_indices, x_values):
        res = type(x).__new__(type(x))                          # new(class(x)[1L])  R/slice.R:567
        res.p = indptr
        res.j = col_indices
        if has_x:                                               # res@x <- as.logical(x_values) for lsparse (:571-575)
            if kind == "l" and x_values is not None and x_values.dtype != np.int32:
                x_values = np.where(np.isnan(x_values), np.int32(-2147483648), (x_values != 0)).astype(np.int32)
            res.x = x_values
        else:
            res.x = None
        res.Dim = (n_row, n_col)
        res.Dimnames = [new_rn, new_cn]
        return res

    if i_is_rev_seq and j_is_rev_seq and n_row == nrow and n_col == ncol:          # R/slice.R:439-469
        t = _call3(kind, E.reverse_rows_numeric, E.reverse_rows_logical, E.reverse_rows_binary, x.p, x.j, x.x)
        jj = np.ascontiguousarray(t["indices"])
        xx = np.ascontiguousarray(t["values"]) if has_x else None
        if kind == "d":
            E.reverse_columns_inplace_numeric(t["indptr"], jj, xx, ncol)
        elif kind == "l":
            E.reverse_columns_inplace_logical(t["indptr"], jj, xx, ncol)
        else:
            E.reverse_columns_inplace_binary(t["indptr"], jj, ncol)
        return finish(t["indptr"], jj, xx)

    if i_is_seq and all_j:                                      # R/slice.R:477-483 (pure R in the reference too)
        first, last = int(x.p[i[0] - 1]), int(x.p[i[-1]])
        return finish(x.p[i[0] - 1:i[-1] + 1] - x.p[i[0] - 1], x.j[first:last], x.x[first:last] if has_x else None)
    if i_is_rev_seq and all_j:                                  # R/slice.R:484-501
        first, last = int(x.p[i[-1] - 1]), int(x.p[i[0]])
        indptr = (x.p[i[-1] - 1:i[0] + 1] - x.p[i[-1] - 1]).astype(np.int32)
        t = _call3(kind, E.reverse_rows_numeric, E.reverse_rows_logical, E.reverse_rows_binary,
                   indptr, x.j[first:last], x.x[first:last] if has_x else None)
        return finish(t["indptr"], t["indices"], t["values"] if has_x else None)
    rows0 = (i - 1).astype(np.int32)
    if all_j:                                                   # R/slice.R:502-515 -> gather kernel
        t = _call3(kind, E.copy_csr_rows_numeric, E.copy_csr_rows_logical, E.copy_csr_rows_binary, x.p, x.j, x.x, rows0)
        return finish(t["indptr"], t["indices"], t["values"] if has_x else None)
    if j_is_seq or j_is_rev_seq:                                # R/slice.R:516-546
        t = _call3(kind, E.copy_csr_rows_col_seq_numeric, E.copy_csr_rows_col_seq_logical,
                   E.copy_csr_rows_col_seq_binary, x.p, x.j, x.x, rows0, j, True)
        jj = np.ascontiguousarray(t["indices"])
        xx = np.ascontiguousarray(t["values"]) if has_x else None
        if j_is_rev_seq:
            new_ncol = int(j[0] - j[-1] + 1)
            if kind == "n":
                E.reverse_columns_inplace_binary(t["indptr"], jj, new_ncol)
            else:                                               # values are numeric here in both classes (slice.cpp:363)
                E.reverse_columns_inplace_numeric(t["indptr"], jj, xx if xx is not None and xx.size else None, new_ncol)
        return finish(t["indptr"], jj, xx)
    t = _call3(kind, E.copy_csr_arbitrary_numeric, E.copy_csr_arbitrary_logical, E.copy_csr_arbitrary_binary,
               x.p, x.j, x.x, rows0, (j - 1).astype(np.int32))  # R/slice.R:547-565
    return finish(t["indptr"], t["indices"], t.get("values") if has_x else None)


def getitem_python(x, key):
    """Python-style `X[rows]` / `X[rows, :]` (0-based, negative = from the end, bool masks) on top of subset_csr."""
    def canon(k, n):
        if isinstance(k, slice):
            if k == slice(None):
                return None
            return np.arange(n)[k]
        a = np.asarray(k)
        if a.dtype == np.bool_:
            if a.size != n:
                stop("boolean index has wrong length")
            return np.flatnonzero(a)
        a = a.astype(np.int64).reshape(-1)
        a = np.where(a < 0, a + n, a)
        if a.size and (a.min() < 0 or a.max() >= n):
            stop("some of row subset indices are not present in matrix")
        return a
    kj = None
    if isinstance(key, tuple):
        if len(key) != 2:
            stop("incorrect number of dimensions")
        key, kj = key
    rows, cols = canon(key, x.Dim[0]), canon(kj, x.Dim[1]) if kj is not None else None
    return subset_csr(x, None if rows is None else (rows + 1).astype(np.int32),
                      None if cols is None else (cols + 1).astype(np.int32))
