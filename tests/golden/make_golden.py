#!/usr/bin/env python3
"""Generates tests/golden/hotpath_golden.npz — small input/output vectors for every hot-path export.

Provenance: the reference (R + Rcpp) cannot run in this image, so these are NOT outputs of the reference
itself.  Inputs come from numpy's PCG64 with the seeds below (plus the literal matrices the reference tree
prints: vignette 3x3, test-utilities.R sort KAT); outputs come from the CPU restatement oracle/mx_oracle.c
AFTER it was checked against dense numpy / scipy and the reference's literal known answers
(tests/test_oracle.py).  Dense products are additionally cross-checked here against numpy before saving.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import csr_to_dense, rand_csr  # noqa: E402
from oracle import oracle as O  # noqa: E402

NA = np.int32(-2147483648)
out = {}


def put(case, **arrays):
    for k, v in arrays.items():
        out[f"{case}/{k}"] = np.asarray(v)


def lst(case, prefix, r):
    put(case, **{f"{prefix}_indptr": r["indptr"], f"{prefix}_indices": r["indices"], f"{prefix}_values": r["values"]})


# ---- SpMM: unsorted + duplicate columns, empty rows, NaN/Inf values, 1-row / 1-col shapes, n > m
for name, (m, K, n, dens) in {"spmm_a": (40, 30, 20, 0.3), "spmm_row1": (1, 12, 5, 0.6), "spmm_col1": (9, 1, 3, 1.0),
                              "spmm_wide": (5, 7, 130, 0.5)}.items():
    p, j, x = rand_csr(m, K, dens, seed=len(name) + m, sorted_cols=False, empty_rows=(0,) if m > 3 else ())
    Y = np.asfortranarray(np.random.default_rng(n).normal(size=(n, K)).round(3))
    res = O.tcrossprod_csr_dense_numeric(p, j, x, Y, 1, use_fma=False)
    np.testing.assert_allclose(res, csr_to_dense(p, j, x, K) @ Y.T, rtol=1e-12, atol=1e-12)
    put(name, p=p, j=j, x=x, Y=Y, tcrossprod_csr_dense=res,
        tcrossprod_csr_dense_f32=O.tcrossprod_csr_dense_float32(p, j, x, Y.astype(np.float32), 1))
    X = np.asfortranarray(np.random.default_rng(m).normal(size=(n, K)).round(3))
    put(name, X=X, matmul_dense_csc=O.matmul_dense_csc_numeric(X, p, j, x, 1),
        tcrossprod_dense_csr=O.tcrossprod_dense_csr_numeric(X, p, j, x, 1, K))
p = np.array([0, 3, 3, 5], dtype=np.int32); j = np.array([1, 1, 0, 1, 0], dtype=np.int32)
x = np.array([2.0, 3.0, 1.0, np.inf, np.nan])
Y = np.asfortranarray(np.array([[1.0, 10.0], [2.0, 20.0], [0.0, 0.5]]))
put("spmm_special", p=p, j=j, x=x, Y=Y, tcrossprod_csr_dense=O.tcrossprod_csr_dense_numeric(p, j, x, Y))

# ---- SpMV: four kinds incl. NA entries
p, j, x = rand_csr(30, 12, 0.4, seed=77, sorted_cols=False, empty_rows=(4,))
rng = np.random.default_rng(78)
yd = rng.normal(size=12).round(3)
yi = rng.integers(-9, 9, size=12).astype(np.int32); yi[[2, 7]] = NA
yl = rng.integers(0, 2, size=12).astype(np.int32); yl[[1, 7]] = NA
put("spmv", p=p, j=j, x=x, y_numeric=yd, y_integer=yi, y_logical=yl, y_float32=yd.astype(np.float32),
    out_numeric=O.matmul_csr_dvec_numeric(p, j, x, yd), out_integer=O.matmul_csr_dvec_integer(p, j, x, yi),
    out_logical=O.matmul_csr_dvec_logical(p, j, x, yl), out_float32=O.matmul_csr_dvec_float32(p, j, x, yd.astype(np.float32)))

# ---- merges: general / disjoint / identical pattern (different buffers) / cancellation / special values / logical
cases = {"merge_general": (rand_csr(25, 14, 0.4, seed=5, empty_rows=(3,)), rand_csr(25, 14, 0.5, seed=6, empty_rows=(3, 9))),
         "merge_one_empty": (rand_csr(10, 8, 0.0, seed=1), rand_csr(10, 8, 0.5, seed=2))}
pd1 = rand_csr(12, 20, 0.3, seed=9)
pd2 = (pd1[0].copy(), (pd1[1] + 0).copy(), -pd1[2])                      # identical pattern, values cancel under +
cases["merge_cancel"] = (pd1, pd2)
even = rand_csr(8, 10, 0.5, seed=3); odd = rand_csr(8, 10, 0.5, seed=4)
cases["merge_disjoint"] = ((even[0], (even[1] * 2).astype(np.int32), even[2]), (odd[0], (odd[1] * 2 + 1).astype(np.int32), odd[2]))
cases["merge_vignette"] = ((np.array([0, 2, 3, 4], np.int32), np.array([0, 2, 2, 1], np.int32), np.array([1.0, 2, 3, 4])),) * 2
for name, (a, b) in cases.items():
    (p1, j1, x1), (p2, j2, x2) = a, b
    p2, j2, x2 = p2.copy(), j2.copy(), x2.copy()                          # distinct buffers: general path
    put(name, p1=p1, j1=j1, x1=x1, p2=p2, j2=j2, x2=x2)
    lst(name, "add", O.add_csr_elemwise(p1, p2, j1, j2, x1, x2, False))
    lst(name, "sub", O.add_csr_elemwise(p1, p2, j1, j2, x1, x2, True))
    lst(name, "mul", O.multiply_csr_elemwise(p1, p2, j1, j2, x1, x2))
p = np.array([0, 2, 2, 4], dtype=np.int32); j = np.array([1, 3, 0, 2], dtype=np.int32)
x1 = np.array([1.5, -2.0, np.inf, 0.0]); x2 = np.array([-1.5, 5.0, -np.inf, -0.0])
pb = np.array([0, 1, 3, 3], dtype=np.int32); jb = np.array([0, 1, 2], dtype=np.int32); xb = np.array([0.0, 5.0, -7.0])
put("merge_special", p1=p, j1=j, x1=x1, p2=pb, j2=jb, x2=xb, x1b=x2)
lst("merge_special", "sub", O.add_csr_elemwise(p, pb, j, jb, x1, xb, True))
lst("merge_special", "samepat_add", O.add_csr_elemwise(p, p.copy(), j, j.copy(), x1, x2, False))
lst("merge_special", "samepat_mul", O.multiply_csr_elemwise(p, p.copy(), j, j.copy(), x1, x2))
l1 = rand_csr(20, 9, 0.5, seed=31, dtype="l"); l2 = rand_csr(20, 9, 0.5, seed=32, dtype="l")
put("merge_logical", p1=l1[0], j1=l1[1], x1=l1[2], p2=l2[0], j2=l2[1], x2=l2[2])
lst("merge_logical", "or", O.logicalor_csr_elemwise(l1[0], l2[0], l1[1], l2[1], l1[2], l2[2], False))
lst("merge_logical", "xor", O.logicalor_csr_elemwise(l1[0], l2[0], l1[1], l2[1], l1[2], l2[2], True))
lst("merge_logical", "and", O.logicaland_csr_elemwise(l1[0], l2[0], l1[1], l2[1], l1[2], l2[2]))

# ---- gather: repeats, reversed, empty rows, selection without entries
p, j, x = rand_csr(50, 20, 0.2, seed=41, empty_rows=(10, 11))
rows = np.array([49, 49, 10, 0, 7, 7, 3, 11, 48], dtype=np.int32)
put("gather", p=p, j=j, x=x, rows=rows, rows_none=np.array([10, 11, 10], dtype=np.int32), xl=(x > 0).astype(np.int32))
lst("gather", "numeric", O.copy_csr_rows_numeric(p, j, x, rows))
lst("gather", "logical", O.copy_csr_rows_logical(p, j, (x > 0).astype(np.int32), rows))
lst("gather", "binary", O.copy_csr_rows_binary(p, j, rows))
lst("gather", "none", O.copy_csr_rows_numeric(p, j, x, np.array([10, 11, 10], dtype=np.int32)))

# ---- CSR (op) dense vector: the four recycling branches, both operand orders, special values
p, j, x = rand_csr(12, 7, 0.45, seed=51, empty_rows=(5,))
x = x.copy(); x[:4] = [np.inf, -np.inf, np.nan, -0.0]
rng = np.random.default_rng(52)
vecs = {"len_nrows": 12, "len_full": 84, "len_divides": 4, "len_general": 5, "len_1": 1, "len_between": 30}
put("dvec", p=p, j=j, x=x, xl=rand_csr(12, 7, 0.45, seed=51, empty_rows=(5,), dtype="l")[2])
for vname, ln in vecs.items():
    v = (rng.uniform(0.5, 3.0, size=ln) * rng.choice([-1.0, 1.0], size=ln)).round(3)
    vl = rng.integers(0, 2, size=ln).astype(np.int32)
    if ln > 2:
        vl[1] = NA
    put("dvec", **{f"v_{vname}": v, f"vl_{vname}": vl})
    rr = np.repeat(np.arange(12), np.diff(p)); full = np.resize(v, 84).reshape(7, 12).T[rr, j]
    res = O.multiply_csr_by_dvec_no_NAs_numeric(p, j, x, v, 7, True, False, False, False, False, True)
    np.testing.assert_array_equal(np.isnan(res), np.isnan(x * full)); ok = ~np.isnan(res)
    np.testing.assert_array_equal(res[ok], (x * full)[ok])
    for opname, flags in {"mul": (1, 0, 0, 0, 0), "pow": (0, 1, 0, 0, 0), "div": (0, 0, 1, 0, 0), "mod": (0, 0, 0, 1, 0),
                          "idiv": (0, 0, 0, 0, 1)}.items():
        for lhs in (True, False):
            put("dvec", **{f"{opname}_{int(lhs)}_{vname}": O.multiply_csr_by_dvec_no_NAs_numeric(p, j, x, v, 7, *flags, lhs)})
    put("dvec", **{f"and_{vname}": O.logicaland_csr_by_dvec_internal(p, j, out["dvec/xl"], vl, 7)})

# ---- sort KAT (tests/testthat/test-utilities.R:32-49)
put("sort_kat", p=np.array([0, 1, 4, 5, 6], np.int32), j=np.array([4, 2, 1, 4, 1, 0], np.int32),
    x=np.array([-0.91, 0.14, -0.12, -0.12, 1.1, 0.66]), j_sorted=np.array([4, 1, 2, 4, 1, 0], np.int32),
    x_sorted=np.array([-0.91, -0.12, 0.14, -0.12, 1.1, 0.66]))

# ---- remove_zero_valued_csr (what remove_zeros runs after A - B) and check_valid_csr_matrix codes
p, j, x = rand_csr(25, 9, 0.4, seed=91, empty_rows=(2, 24))
rng = np.random.default_rng(92)
x = x.copy(); x[rng.random(x.size) < 0.3] = 0.0; x[3] = -0.0; x[rng.random(x.size) < 0.15] = np.nan
xl = rng.choice(np.array([0, 1, NA], dtype=np.int32), size=x.size)
put("dropzeros", p=p, j=j, x=x, xl=xl)
for rm in (0, 1):
    lst("dropzeros", f"numeric_{rm}", O.remove_zero_valued_csr_numeric(p, j, x, bool(rm)))
    lst("dropzeros", f"logical_{rm}", O.remove_zero_valued_csr_logical(p, j, xl, bool(rm)))
    keep = (x != 0) & (~np.isnan(x) if rm else True)                 # the numeric predicate, restated with masks
    np.testing.assert_array_equal(out[f"dropzeros/numeric_{rm}_indices"], j[keep])
    np.testing.assert_array_equal(out[f"dropzeros/numeric_{rm}_values"], x[keep])

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hotpath_golden.npz")
np.savez_compressed(path, **out)
print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)} bytes")
