"""Host-side mirror of the row-selection part of R/slice.R (`X[i, ]`).

`subset_csr(x, i)` takes R-style indices: 1-based integers (repeats allowed),
negative integers = exclusion, logical masks (recycled like base R), row names.
Path selection follows R/slice.R:477-515: a contiguous ascending run is sliced
on the host exactly as the reference does in pure R (:477-483, never reaches
native code there either); everything else goes to copy_csr_rows_* — the
gather kernel.  Column selection (`X[i, j]`) is §8(f) rank 2, not built yet.
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


def subset_csr(x, i=None, drop=False):
    """`x[i, ]` for RsparseMatrix — R/slice.R:293-585, rows-only branches."""
    check_valid_matrix(x)
    if i is None:
        return x
    row_names = x.Dimnames[0]
    i = get_indices_integer(i, x.Dim[0], row_names)
    n_row = int(i.size)
    i_is_seq = exports.check_is_seq(i)
    all_i = bool(i_is_seq and n_row == x.Dim[0] and n_row > 0 and i[0] == 1 and i[-1] == x.Dim[0])
    new_names = None if row_names is None or not len(row_names) else [row_names[k - 1] for k in i]

    if n_row == 0 or x.Dim[1] == 0 or x.j.size == 0:          # R/slice.R:404-421
        return _empty_like(x, n_row, new_names)
    if all_i:                                                   # R/slice.R:423-425
        return x

    has_x = x.x is not None
    if i_is_seq:                                                # R/slice.R:477-483 (pure R in the reference too)
        first, last = int(x.p[i[0] - 1]), int(x.p[i[-1]])
        indptr = x.p[i[0] - 1:i[-1] + 1] - x.p[i[0] - 1]
        col_indices = x.j[first:last]
        x_values = x.x[first:last] if has_x else None
    else:                                                       # R/slice.R:502-515 -> gather kernel
        rows0 = (i - 1).astype(np.int32)
        if isinstance(x, dgRMatrix):
            temp = exports.copy_csr_rows_numeric(x.p, x.j, x.x, rows0)
        elif isinstance(x, lgRMatrix):
            temp = exports.copy_csr_rows_logical(x.p, x.j, x.x, rows0)
        else:
            temp = exports.copy_csr_rows_binary(x.p, x.j, rows0)
        indptr, col_indices = temp["indptr"], temp["indices"]
        x_values = temp["values"] if has_x else None

    res = type(x).__new__(type(x))                              # new(class(x)[1L])  R/slice.R:567
    res.p = indptr                                              # NB: empty when the gather selected no entries
    res.j = col_indices                                         #     (slice.cpp:236-240) — kept as the reference does
    res.x = x_values
    res.Dim = (n_row, x.Dim[1])
    res.Dimnames = [new_names, x.Dimnames[1]]
    return res


def getitem_python(x, key):
    """Python-style `X[rows]` / `X[rows, :]` (0-based, negative = from the end, bool masks) on top of subset_csr."""
    if isinstance(key, tuple):
        if len(key) != 2 or not (isinstance(key[1], slice) and key[1] == slice(None)):
            stop("only row selection X[i, :] is implemented in the MI355X hot path")
        key = key[0]
    n = x.Dim[0]
    if isinstance(key, slice):
        rows = np.arange(n)[key]
    else:
        rows = np.asarray(key)
        if rows.dtype == np.bool_:
            if rows.size != n:
                stop("boolean index has wrong length")
            rows = np.flatnonzero(rows)
        else:
            rows = rows.astype(np.int64).reshape(-1)
            rows = np.where(rows < 0, rows + n, rows)
            if rows.size and (rows.min() < 0 or rows.max() >= n):
                stop("some of row subset indices are not present in matrix")
    if rows.size == 0:
        return _empty_like(x, 0, None)
    return subset_csr(x, (rows + 1).astype(np.int32))
