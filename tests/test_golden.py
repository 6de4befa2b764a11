"""Committed golden vectors (tests/golden/hotpath_golden.npz, made by tests/golden/make_golden.py — see its
header for provenance): the CPU restatement must reproduce them (CPU run) and the HIP path must
reproduce them through the C-ABI (GPU run)."""
import os

import numpy as np
import pytest

from oracle import oracle as O

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_golden.npz"))


def g(case, k):
    return G[f"{case}/{k}"]


def eq(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, b.shape, a.dtype, b.dtype)
    if a.dtype.kind == "f":
        np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
        ok = ~np.isnan(a)
        view = np.int64 if a.dtype == np.float64 else np.int32
        np.testing.assert_array_equal(a[ok].view(view), b[ok].view(view))
    else:
        np.testing.assert_array_equal(a, b)


def close(a, b, rtol, atol):
    np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
    ok = ~np.isnan(a)
    np.testing.assert_allclose(a[ok], b[ok], rtol=rtol, atol=atol)


def eq_list(r, case, prefix):
    eq(r["indptr"], g(case, prefix + "_indptr"))
    eq(r["indices"], g(case, prefix + "_indices"))
    eq(r["values"], g(case, prefix + "_values"))


def run_all(M, exact_spmm):
    """M: module with the export names (oracle.oracle or matrixextra_amd.exports)."""
    for case in ("spmm_a", "spmm_row1", "spmm_col1", "spmm_wide"):
        p, j, x, Y, X = (g(case, k) for k in "pjxYX")
        K = Y.shape[1]
        close(M.tcrossprod_csr_dense_numeric(p, j, x, Y, 1), g(case, "tcrossprod_csr_dense"), 1e-12, 1e-13)
        close(M.tcrossprod_csr_dense_float32(p, j, x, Y.astype(np.float32), 1), g(case, "tcrossprod_csr_dense_f32"), 1e-5, 1e-5)
        close(M.matmul_dense_csc_numeric(X, p, j, x, 1), g(case, "matmul_dense_csc"), 1e-12, 1e-13)
        close(M.tcrossprod_dense_csr_numeric(X, p, j, x, 1, K), g(case, "tcrossprod_dense_csr"), 1e-12, 1e-13)
        if exact_spmm:
            eq(M.tcrossprod_csr_dense_numeric(p, j, x, Y, 1), g(case, "tcrossprod_csr_dense"))
    c = "spmm_special"
    close(M.tcrossprod_csr_dense_numeric(g(c, "p"), g(c, "j"), g(c, "x"), g(c, "Y")), g(c, "tcrossprod_csr_dense"), 1e-12, 0)
    c = "spmv"
    p, j, x = g(c, "p"), g(c, "j"), g(c, "x")
    close(M.matmul_csr_dvec_numeric(p, j, x, g(c, "y_numeric")), g(c, "out_numeric"), 1e-12, 1e-13)
    close(M.matmul_csr_dvec_integer(p, j, x, g(c, "y_integer")), g(c, "out_integer"), 1e-12, 1e-13)
    close(M.matmul_csr_dvec_logical(p, j, x, g(c, "y_logical")), g(c, "out_logical"), 1e-12, 1e-13)
    close(M.matmul_csr_dvec_float32(p, j, x, g(c, "y_float32")), g(c, "out_float32"), 1e-5, 1e-6)
    for c in ("merge_general", "merge_one_empty", "merge_cancel", "merge_disjoint", "merge_vignette"):
        a = [g(c, k) for k in ("p1", "p2", "j1", "j2", "x1", "x2")]
        eq_list(M.add_csr_elemwise(*a, False), c, "add")
        eq_list(M.add_csr_elemwise(*a, True), c, "sub")
        eq_list(M.multiply_csr_elemwise(*a), c, "mul")
    c = "merge_special"
    p, j, x1, pb, jb, xb, x1b = (g(c, k) for k in ("p1", "j1", "x1", "p2", "j2", "x2", "x1b"))
    eq_list(M.add_csr_elemwise(p, pb, j, jb, x1, xb, True), c, "sub")
    eq_list(M.add_csr_elemwise(p, p.copy(), j, j.copy(), x1, x1b, False), c, "samepat_add")
    eq_list(M.multiply_csr_elemwise(p, p.copy(), j, j.copy(), x1, x1b), c, "samepat_mul")
    c = "merge_logical"
    a = [g(c, k) for k in ("p1", "p2", "j1", "j2", "x1", "x2")]
    eq_list(M.logicalor_csr_elemwise(*a, False), c, "or")
    eq_list(M.logicalor_csr_elemwise(*a, True), c, "xor")
    eq_list(M.logicaland_csr_elemwise(*a), c, "and")
    c = "gather"
    p, j, x, rows = g(c, "p"), g(c, "j"), g(c, "x"), g(c, "rows")
    eq_list(M.copy_csr_rows_numeric(p, j, x, rows), c, "numeric")
    eq_list(M.copy_csr_rows_logical(p, j, g(c, "xl"), rows), c, "logical")
    eq_list(M.copy_csr_rows_binary(p, j, rows), c, "binary")
    eq_list(M.copy_csr_rows_numeric(p, j, x, g(c, "rows_none")), c, "none")

    c = "dvec"
    p, j, x, xl = g(c, "p"), g(c, "j"), g(c, "x"), g(c, "xl")
    flags = {"mul": (1, 0, 0, 0, 0), "pow": (0, 1, 0, 0, 0), "div": (0, 0, 1, 0, 0), "mod": (0, 0, 0, 1, 0), "idiv": (0, 0, 0, 0, 1)}
    for vname in ("len_nrows", "len_full", "len_divides", "len_general", "len_1", "len_between"):
        v, vl = g(c, "v_" + vname), g(c, "vl_" + vname)
        for opname, f in flags.items():
            for lhs in (True, False):
                got = M.multiply_csr_by_dvec_no_NAs_numeric(p, j, x, v, 7, *f, lhs)
                want = g(c, f"{opname}_{int(lhs)}_{vname}")
                if opname in ("mul", "div"):
                    eq(got, want)                                   # one IEEE operation: bit-exact
                else:                                               # %% %/% ^: long double / libm in the reference
                    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
                    ok = ~np.isnan(want)
                    np.testing.assert_allclose(got[ok], want[ok], rtol=1e-13, atol=1e-15)
        eq(M.logicaland_csr_by_dvec_internal(p, j, xl, vl, 7), g(c, "and_" + vname))


def run_dropzeros(M):
    c = "dropzeros"
    p, j, x, xl = g(c, "p"), g(c, "j"), g(c, "x"), g(c, "xl")
    for rm in (0, 1):
        eq_list(M.remove_zero_valued_csr_numeric(p, j, x, bool(rm)), c, f"numeric_{rm}")
        eq_list(M.remove_zero_valued_csr_logical(p, j, xl, bool(rm)), c, f"logical_{rm}")


def test_oracle_reproduces_golden():
    run_dropzeros(O)
    run_all(O, exact_spmm=True)
    c = "sort_kat"
    js, xs = O.sort_sparse_indices(g(c, "p"), g(c, "j"), g(c, "x"))
    eq(js, g(c, "j_sorted")); eq(xs, g(c, "x_sorted"))


@pytest.mark.gpu
def test_hip_reproduces_golden(gpu):
    from matrixextra_amd import exports
    run_all(exports, exact_spmm=False)
    run_dropzeros(exports)
    c = "sort_kat"
    j, x = g(c, "j").copy(), g(c, "x").copy()
    exports.sort_sparse_indices_inplace(g(c, "p"), j, x)
    eq(j, g(c, "j_sorted")); eq(x, g(c, "x_sorted"))
