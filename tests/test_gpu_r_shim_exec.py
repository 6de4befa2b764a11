"""EXECUTES the `.Call` shim (matrixextra_amd/csrc/r_shim.cpp) on the GPU: the shim is built against a mock of the R C API
(tests/r_mock/rmock.c — tagged-heap SEXPs, a real PROTECT stack, mark & sweep on EVERY allocation = gctorture, Rf_error as a
long-jump to a .Call trampoline that runs the pending R_ExecWithCleanup handlers) and libmxgpu.so, registered through
R_init_mxgpu_r / R_registerRoutines, and EVERY routine of the captured table is called BY NAME the way R's `.Call` does
(reference: src/RcppExports.cpp:16,24 BEGIN_RCPP/END_RCPP glue, :2200-2368 table + R_init_MatrixExtra).

What this exercises is MARSHALLING, not parity: type coercions (Rcpp's input_parameter<>), dims of results, names of the
result lists, the R types of values, alias returns (the INPUT SEXPs themselves, operators.cpp:127-131,390-394), in-place
routines, S4 slot reads, PROTECT discipline under gctorture, error long-jumps with the device handle released.  The values
are nevertheless compared with the oracle (bit for bit for every structure / value copy; 1e-12 for the products) — parity
proper of the kernels behind the C-ABI is tests/test_gpu_parity.py & co."""
import os
import sys

import numpy as np
import pytest

from conftest import rand_csr
from oracle import oracle as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "r_mock"))
import rmock  # noqa: E402
from test_r_shim_syntax import REFERENCE_ARITY  # noqa: E402

pytestmark = pytest.mark.gpu
NA = rmock.NA_INTEGER
INVOKED = set()


@pytest.fixture(scope="module")
def R(gpu):
    r = rmock.runtime()
    r.gctorture(True)
    yield r
    r.gctorture(False)


def call(R, name, *args):
    INVOKED.add(name)
    return R.call(name, *args)


def eq(got, want, what=""):
    """bitwise equality of two arrays (NaN payloads included), same dtype and shape"""
    got, want = np.asarray(got), np.asarray(want)
    assert got.dtype == want.dtype, (what, got.dtype, want.dtype)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert got.tobytes() == want.tobytes(), what


def close(got, want, what="", rtol=1e-12):
    """products: NaN / NA in the same places, the rest to rtol (small operands run the lane-group kernels, whose sums are
    not in storage order)"""
    got, want = np.asarray(got), np.asarray(want)
    assert got.dtype == want.dtype and got.shape == want.shape, what
    assert np.array_equal(np.isnan(got), np.isnan(want)), what
    np.testing.assert_allclose(got[~np.isnan(got)], want[~np.isnan(want)], rtol=rtol, atol=rtol, err_msg=what)


def eq_list(R, s, want, vtype, what=""):
    """a list(indptr=, indices=, values=) result against the oracle's dict; vtype = the R type `values` must have"""
    assert R.typeof(s) == rmock.VECSXP
    assert R.names(s) == ["indptr", "indices", "values"], what
    e = R.elts(s)
    assert R.typeof(e[0]) == rmock.INTSXP and R.typeof(e[1]) == rmock.INTSXP and R.typeof(e[2]) == vtype, what
    g = R.as_py(s)
    eq(g["indptr"], np.asarray(want["indptr"], dtype=np.int32), what + " indptr")
    eq(g["indices"], np.asarray(want["indices"], dtype=np.int32), what + " indices")
    wv = want["values"]
    wv = np.zeros(0, dtype=g["values"].dtype) if wv is None else np.asarray(wv)
    eq(g["values"], wv.astype(g["values"].dtype, copy=False), what + " values")


class Snapshot:
    """inputs must never be mutated (tests/testthat/test-operators.R:21-27)"""

    def __init__(self, R, *sexps):
        self.R, self.s = R, sexps
        self.before = [R.view(x).tobytes() for x in sexps]

    def check(self):
        for x, b in zip(self.s, self.before):
            assert self.R.view(x).tobytes() == b, "an input vector was modified"


@pytest.fixture(scope="module")
def mats():
    p, j, x = rand_csr(61, 37, 0.25, seed=5, empty_rows=(0, 17, 60))
    p2, j2, x2 = rand_csr(61, 37, 0.3, seed=6, empty_rows=(3, 17))
    rng = np.random.default_rng(11)
    x[3] = np.nan
    x[5] = np.inf
    xl = rng.choice(np.array([0, 1, NA], dtype=np.int32), size=x.size)
    xl2 = rng.choice(np.array([0, 1, NA], dtype=np.int32), size=x2.size)
    return dict(p=p, j=j, x=x, p2=p2, j2=j2, x2=x2, xl=xl, xl2=xl2, m=61, K=37, rng=rng)


# ----------------------------------------------------------------------------------------------- registration
def test_registration_table(R):
    t = R.routines()
    ours = {k[len("_MatrixExtra_"):]: v for k, v in t.items() if k.startswith("_MatrixExtra_")}
    assert ours == REFERENCE_ARITY                              # names + arities of the reference's CallEntries[]
    assert t["_mxgpu_set_option"] == 2 and t["_mxgpu_set_devices"] == 1
    assert R.L.rmock_dynamic_symbols() == 0                     # R_useDynamicSymbols(dll, FALSE), RcppExports.cpp:2367
    with pytest.raises(LookupError, match="Incorrect number of arguments"):
        R.call("check_is_seq")
    with pytest.raises(LookupError, match="not in DLL"):
        R.call("no_such_routine", R.integer([1]))


# ----------------------------------------------------------------------------------------------- CSR x dense (6 exports)
def test_spmm_exports(R, mats):
    p, j, x, m, K = mats["p"], mats["j"], np.nan_to_num(mats["x"], nan=0.5, posinf=2.0), mats["m"], mats["K"]
    rng = mats["rng"]
    sp, sj, sx = R.integer(p), R.integer(j), R.real(x)
    n = 9
    Y = rng.normal(size=(n, K))                                  # tcrossprod_csr_dense: Y is n x K, result m x n
    sY = R.matrix(Y)
    snap = Snapshot(R, sp, sj, sx, sY)
    # nthreads as a double: Rcpp's input_parameter<int> coerces (SURVEY §8b Ownership)
    out = call(R, "tcrossprod_csr_dense_numeric", sp, sj, sx, sY, R.real([4.0]))
    got = R.view(out)
    assert R.typeof(out) == rmock.REALSXP and got.shape == (m, n)
    np.testing.assert_allclose(got, O.tcrossprod_csr_dense_numeric(p, j, x, np.asfortranarray(Y)), rtol=1e-12, atol=1e-12)
    # float32: INTSXP bit patterns in, INTSXP bit patterns out, with dims (matmul.cpp:213, R/matmul.R:260,276)
    Y32 = Y.astype(np.float32)
    sY32 = R.matrix(Y32, "float32")
    out32 = call(R, "tcrossprod_csr_dense_float32", sp, sj, sx, sY32, R.integer([1]))
    assert R.typeof(out32) == rmock.INTSXP
    g32 = R.view(out32)
    assert g32.shape == (m, n)
    np.testing.assert_allclose(g32.view(np.float32), O.tcrossprod_csr_dense_float32(p, j, x, np.asfortranarray(Y32)), rtol=1e-5, atol=1e-5)
    eq(R.view(sY32), np.asfortranarray(Y32).view(np.int32), "float32 bits round-trip")
    # a column-index vector handed over as doubles (e.g. after arithmetic in R) is coerced, not reinterpreted
    outd = call(R, "tcrossprod_csr_dense_numeric", R.real(p.astype(np.float64)), R.real(j.astype(np.float64)), sx, sY, R.integer([1]))
    eq(R.view(outd), got, "REALSXP indptr / indices coerced")
    snap.check()

    # dense x CSC: X (nr x nc) %*% Y_csc (nc x ncY): the CSC arrays of Y are the CSR arrays of t(Y)  (matmul.cpp:188-219)
    pc, ic, xc = rand_csr(23, K, 0.3, seed=77)                 # t(Y): 23 "rows" = columns of Y, each indexing [0, K)
    X = rng.normal(size=(7, K))
    sX = R.matrix(X)
    o = call(R, "matmul_dense_csc_numeric", sX, R.integer(pc), R.integer(ic), R.real(xc), R.integer([1]))
    assert R.view(o).shape == (7, 23)
    np.testing.assert_allclose(R.view(o), O.matmul_dense_csc_numeric(np.asfortranarray(X), pc, ic, xc), rtol=1e-12, atol=1e-12)
    X32 = X.astype(np.float32)
    o = call(R, "matmul_dense_csc_float32", R.matrix(X32, "float32"), R.integer(pc), R.integer(ic), R.real(xc), R.integer([1]))
    assert R.typeof(o) == rmock.INTSXP and R.view(o).shape == (7, 23)
    np.testing.assert_allclose(R.view(o).view(np.float32), O.matmul_dense_csc_float32(np.asfortranarray(X32), pc, ic, xc), rtol=1e-5, atol=1e-5)
    # dense x t(CSR)   (matmul.cpp:254-281)
    o = call(R, "tcrossprod_dense_csr_numeric", sX, R.integer(pc), R.integer(ic), R.real(xc), R.integer([2]), R.integer([K]))
    assert R.view(o).shape == (7, 23)
    np.testing.assert_allclose(R.view(o), O.tcrossprod_dense_csr_numeric(np.asfortranarray(X), pc, ic, xc, 1, K), rtol=1e-12, atol=1e-12)
    o = call(R, "tcrossprod_dense_csr_float32", R.matrix(X32, "float32"), R.integer(pc), R.integer(ic), R.real(xc), R.real([2]), R.real([K]))
    assert R.typeof(o) == rmock.INTSXP
    np.testing.assert_allclose(R.view(o).view(np.float32), O.tcrossprod_dense_csr_float32(np.asfortranarray(X32), pc, ic, xc, 1, K), rtol=1e-5, atol=1e-5)


def test_spmm_empty_matrix(R):
    """m = 0 and nnz = 0 (matmul.cpp:160-161): a 0 x n / an all-zero result of the right shape, every cell written"""
    o = call(R, "tcrossprod_csr_dense_numeric", R.integer([0]), R.integer([]), R.real([]), R.matrix(np.ones((3, 5))), R.integer([1]))
    assert R.view(o).shape == (0, 3)
    o = call(R, "tcrossprod_csr_dense_numeric", R.integer([0, 0, 0]), R.integer([]), R.real([]), R.matrix(np.ones((3, 5))), R.integer([1]))
    eq(R.view(o), np.zeros((2, 3), order="F"))                 # fresh R memory is NOT zero (the mock poisons it)


# ----------------------------------------------------------------------------------------------- CSR x dense vector (4)
def test_spmv_exports(R, mats):
    p, j, x, m, K, rng = mats["p"], mats["j"], mats["x"], mats["m"], mats["K"], mats["rng"]
    sp, sj, sx = R.integer(p), R.integer(j), R.real(x)
    v = rng.normal(size=K)
    o = call(R, "matmul_csr_dvec_numeric", sp, sj, sx, R.real(v), R.integer([1]))
    assert R.typeof(o) == rmock.REALSXP and R.view(o).shape == (m,)
    close(R.view(o), O.matmul_csr_dvec_numeric(p, j, x, v), "SpMV numeric")
    vi = rng.integers(-5, 6, size=K).astype(np.int32)
    vi[4] = NA
    o = call(R, "matmul_csr_dvec_integer", sp, sj, sx, R.integer(vi), R.integer([1]))
    close(R.view(o), O.matmul_csr_dvec_integer(p, j, x, vi), "SpMV integer (NA_INTEGER -> NA_real_)")
    vl = rng.choice(np.array([0, 1, NA], dtype=np.int32), size=K)
    o = call(R, "matmul_csr_dvec_logical", sp, sj, sx, R.logical(vl), R.integer([1]))
    close(R.view(o), O.matmul_csr_dvec_logical(p, j, x, vl), "SpMV logical")
    xf = np.nan_to_num(x, nan=0.25, posinf=3.0)
    v32 = v.astype(np.float32)
    o = call(R, "matmul_csr_dvec_float32", sp, sj, R.real(xf), R.float32(v32), R.integer([1]))
    assert R.typeof(o) == rmock.INTSXP
    np.testing.assert_allclose(R.view(o).view(np.float32), O.matmul_csr_dvec_float32(p, j, xf, v32), rtol=1e-5, atol=1e-6)
    # an integer vector given where doubles are expected: coerced copy, NA_integer_ becomes NA_real_
    o = call(R, "matmul_csr_dvec_numeric", sp, sj, sx, R.integer(vi), R.integer([1]))
    vd = vi.astype(np.float64)
    vd[4] = np.nan
    close(R.view(o), O.matmul_csr_dvec_numeric(p, j, x, vd), "INTSXP vector coerced to double")


# ----------------------------------------------------------------------------------------------- CSR (+) CSR (4 exports)
def test_elemwise_exports(R, mats):
    p, j, x, p2, j2, x2, xl, xl2 = (mats[k] for k in ("p", "j", "x", "p2", "j2", "x2", "xl", "xl2"))
    sp, sj, sx, sp2, sj2, sx2 = R.integer(p), R.integer(j), R.real(x), R.integer(p2), R.integer(j2), R.real(x2)
    sxl, sxl2 = R.logical(xl), R.logical(xl2)
    snap = Snapshot(R, sp, sj, sx, sp2, sj2, sx2, sxl, sxl2)
    eq_list(R, call(R, "multiply_csr_elemwise", sp, sp2, sj, sj2, sx, sx2), O.multiply_csr_elemwise(p, p2, j, j2, x, x2), rmock.REALSXP, "mul")
    for sub in (False, True):
        eq_list(R, call(R, "add_csr_elemwise", sp, sp2, sj, sj2, sx, sx2, R.logical([int(sub)])),
                O.add_csr_elemwise(p, p2, j, j2, x, x2, sub), rmock.REALSXP, f"add sub={sub}")
    eq_list(R, call(R, "logicaland_csr_elemwise", sp, sp2, sj, sj2, sxl, sxl2), O.logicaland_csr_elemwise(p, p2, j, j2, xl, xl2), rmock.LGLSXP, "and")
    for xor in (False, True):
        eq_list(R, call(R, "logicalor_csr_elemwise", sp, sp2, sj, sj2, sxl, sxl2, R.logical([int(xor)])),
                O.logicalor_csr_elemwise(p, p2, j, j2, xl, xl2, xor), rmock.LGLSXP, f"or xor={xor}")
    snap.check()


def test_elemwise_alias_fast_paths_return_the_input_sexps(R, mats):
    """same indptr / indices OBJECTS on both sides: the result's indptr / indices ARE those objects
    (operators.cpp:127-131 multiply, :390-394 add) — pointer identity, observable from R"""
    p, j, x, x2 = mats["p"], mats["j"], mats["x"], mats["rng"].normal(size=mats["x"].size)
    sp, sj, sx, sx2 = R.integer(p), R.integer(j), R.real(x), R.real(x2)
    for name, extra, want in [("multiply_csr_elemwise", (), O.multiply_csr_elemwise(p, p, j, j, x, x2)),
                              ("add_csr_elemwise", (R.logical([0]),), O.add_csr_elemwise(p, p, j, j, x, x2, False)),
                              ("add_csr_elemwise", (R.logical([1]),), O.add_csr_elemwise(p, p, j, j, x, x2, True))]:
        out = call(R, name, sp, sp, sj, sj, sx, sx2, *extra)
        e = R.elts(out)
        assert e[0] == sp and e[1] == sj, f"{name}: the input SEXPs must come back (alias fast path)"
        assert e[2] not in (sx, sx2)
        eq(R.view(e[2]), want["values"], name)
    sxl = R.logical(mats["xl"])
    out = call(R, "logicaland_csr_elemwise", sp, sp, sj, sj, sxl, sxl)
    assert R.elts(out)[0] == sp and R.elts(out)[1] == sj
    # X - X with the same values object: the all-empty result (operators.cpp:348-355)
    out = call(R, "add_csr_elemwise", sp, sp, sj, sj, sx, sx, R.logical([1]))
    eq_list(R, out, O.add_csr_elemwise(p, p, j, j, x, x, True), rmock.REALSXP, "X - X")
    # equal CONTENTS in different objects is not the alias path: fresh vectors
    out = call(R, "multiply_csr_elemwise", sp, R.integer(p), sj, R.integer(j), sx, sx2)
    assert R.elts(out)[0] != sp and R.elts(out)[1] != sj
    eq_list(R, out, O.multiply_csr_elemwise(p, p.copy(), j, j.copy(), x, x2), rmock.REALSXP, "equal pattern, other objects")


# ----------------------------------------------------------------------------------------------- X[rows, ] (3) + seq checks (2)
def test_row_gather_exports(R, mats):
    p, j, x, xl, m, rng = mats["p"], mats["j"], mats["x"], mats["xl"], mats["m"], mats["rng"]
    sp, sj, sx, sxl = R.integer(p), R.integer(j), R.real(x), R.logical(xl)
    snap = Snapshot(R, sp, sj, sx, sxl)
    for rows in (rng.integers(0, m, size=40).astype(np.int32), np.array([17, 0, 60], dtype=np.int32), np.zeros(0, dtype=np.int32)):
        sr = R.integer(rows)
        eq_list(R, call(R, "copy_csr_rows_numeric", sp, sj, sx, sr), O.copy_csr_rows_numeric(p, j, x, rows), rmock.REALSXP, "gather d")
        eq_list(R, call(R, "copy_csr_rows_logical", sp, sj, sxl, sr), O.copy_csr_rows_logical(p, j, xl, rows), rmock.LGLSXP, "gather l")
        eq_list(R, call(R, "copy_csr_rows_binary", sp, sj, sr), O.copy_csr_rows_binary(p, j, rows), rmock.REALSXP, "gather n")
    # 1-based row numbers arriving as doubles (`i - 1` in R gives a double unless written 1L): coerced
    rows = np.array([5, 5, 2], dtype=np.int32)
    eq_list(R, call(R, "copy_csr_rows_numeric", sp, sj, sx, R.real(rows.astype(np.float64))), O.copy_csr_rows_numeric(p, j, x, rows), rmock.REALSXP)
    snap.check()


def test_check_is_seq_exports(R):
    for v in ([3, 4, 5, 6], [3, 4, 6], [9, 8, 7], [9, 8, 8], [4], []):
        a = np.asarray(v, dtype=np.int32)
        o = call(R, "check_is_seq", R.integer(a))
        assert R.typeof(o) == rmock.LGLSXP and R.view(o).tolist() == [int(O.check_is_seq(a))]
        o = call(R, "check_is_rev_seq", R.integer(a))
        assert R.typeof(o) == rmock.LGLSXP and R.view(o).tolist() == [int(O.check_is_rev_seq(a))]


# ----------------------------------------------------------------------------------------------- X[rows, cols] (6) + reversals (6)
def test_column_slice_exports(R, mats):
    p, j, x, xl, m, K, rng = mats["p"], mats["j"], mats["x"], mats["xl"], mats["m"], mats["K"], mats["rng"]
    sp, sj, sx, sxl = R.integer(p), R.integer(j), R.real(x), R.logical(xl)
    rows = rng.integers(0, m, size=30).astype(np.int32)
    sr = R.integer(rows)
    for cols, index1 in ((np.arange(5, 21, dtype=np.int32), True), (np.arange(30, 12, -1, dtype=np.int32), False)):
        sc, si = R.integer(cols), R.logical([int(index1)])
        eq_list(R, call(R, "copy_csr_rows_col_seq_numeric", sp, sj, sx, sr, sc, si), O.copy_csr_rows_col_seq_numeric(p, j, x, rows, cols, index1), rmock.REALSXP)
        # the reference's template returns a NumericVector for the logical kind too (slice.cpp:363): NA_LOGICAL -> -2147483648.0
        eq_list(R, call(R, "copy_csr_rows_col_seq_logical", sp, sj, sxl, sr, sc, si), O.copy_csr_rows_col_seq_logical(p, j, xl, rows, cols, index1), rmock.REALSXP)
        eq_list(R, call(R, "copy_csr_rows_col_seq_binary", sp, sj, sr, sc, si), O.copy_csr_rows_col_seq_binary(p, j, rows, cols, index1), rmock.REALSXP)
    cols = np.array([20, 3, 3, 11, 0, 36, 3], dtype=np.int32)
    sc = R.integer(cols)
    eq_list(R, call(R, "copy_csr_arbitrary_numeric", sp, sj, sx, sr, sc), O.copy_csr_arbitrary_numeric(p, j, x, rows, cols), rmock.REALSXP)
    eq_list(R, call(R, "copy_csr_arbitrary_logical", sp, sj, sxl, sr, sc), O.copy_csr_arbitrary_logical(p, j, xl, rows, cols), rmock.LGLSXP)
    # the pattern-matrix form has NO `values` element (slice.cpp:565)
    o = call(R, "copy_csr_arbitrary_binary", sp, sj, sr, sc)
    w = O.copy_csr_arbitrary_binary(p, j, rows, cols)
    assert R.names(o) == ["indptr", "indices"]
    g = R.as_py(o)
    eq(g["indptr"], w["indptr"])
    eq(g["indices"], w["indices"])
    # ... and so has the numeric form of a matrix without entries (`if (values.size())`, slice.cpp:566); the row gather of
    # the same matrix returns three EMPTY vectors, indptr included (slice.cpp:236-240)
    pe, je, xe = np.zeros(5, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros(0)
    re = np.array([1, 2], dtype=np.int32)
    o = call(R, "copy_csr_arbitrary_numeric", R.integer(pe), R.integer(je), R.real(xe), R.integer(re), R.integer([0, 3]))
    w = O.copy_csr_arbitrary_numeric(pe, je, xe, re, np.array([0, 3], dtype=np.int32))
    assert R.names(o) == list(w) == ["indptr", "indices"]
    eq(R.as_py(o)["indptr"], w["indptr"])
    eq_list(R, call(R, "copy_csr_rows_numeric", R.integer(pe), R.integer(je), R.real(xe), R.integer(re)),
            O.copy_csr_rows_numeric(pe, je, xe, re), rmock.REALSXP, "entry-less gather")
    eq_list(R, call(R, "copy_csr_rows_logical", R.integer(pe), R.integer(je), R.logical([]), R.integer(re)),
            O.copy_csr_rows_logical(pe, je, np.zeros(0, dtype=np.int32), re), rmock.LGLSXP, "entry-less gather, logical(0)")


def test_reversal_exports(R, mats):
    p, j, x, xl, K = mats["p"], mats["j"], mats["x"], mats["xl"], mats["K"]
    sp, sj, sx, sxl = R.integer(p), R.integer(j), R.real(x), R.logical(xl)
    snap = Snapshot(R, sp, sj, sx, sxl)
    eq_list(R, call(R, "reverse_rows_numeric", sp, sj, sx), O.reverse_rows_numeric(p, j, x), rmock.REALSXP)
    eq_list(R, call(R, "reverse_rows_logical", sp, sj, sxl), O.reverse_rows_logical(p, j, xl), rmock.LGLSXP)
    eq_list(R, call(R, "reverse_rows_binary", sp, sj), O.reverse_rows_binary(p, j), rmock.REALSXP)
    snap.check()
    # in place on the caller's vectors, NULL result (slice.cpp:172-221)
    for name, vals, mk in (("reverse_columns_inplace_numeric", x, R.real), ("reverse_columns_inplace_logical", xl, R.logical),
                           ("reverse_columns_inplace_binary", None, None)):
        jj, vv = j.copy(), None if vals is None else vals.copy()
        O.reverse_columns_inplace(p, jj, vv, K)
        tj = R.integer(j)
        tv = R.nil if vals is None else mk(vals)
        assert call(R, name, sp, tj, tv, R.integer([K])) is None          # void export: R_NilValue
        eq(R.view(tj), jj, name)
        if vals is not None:
            eq(R.view(tv), vv, name)


# ----------------------------------------------------------------------------------------------- CSR op vector (3)
def test_csr_by_dvec_exports(R, mats):
    p, j, x, xl, m, K, rng = mats["p"], mats["j"], mats["x"], mats["xl"], mats["m"], mats["K"], mats["rng"]
    xx = np.nan_to_num(x, nan=0.5, posinf=2.0)
    sp, sj, sx = R.integer(p), R.integer(j), R.real(xx)
    d = rng.normal(size=m) + 3.0
    T, F = R.logical([1]), R.logical([0])
    flags = {"multiply": (T, F, F, F, F), "divide": (F, F, T, F, F), "divrest": (F, F, F, T, F), "intdiv": (F, F, F, F, T)}
    for op, fl in flags.items():
        o = call(R, "multiply_csr_by_dvec_no_NAs_numeric", sp, sj, sx, R.real(d), R.integer([K]), *fl, T)
        b = [op == k for k in ("multiply", "powerto", "divide", "divrest", "intdiv")]
        eq(R.view(o), O.multiply_csr_by_dvec_no_NAs_numeric(p, j, xx, d, K, *b, True), op)
    dl = rng.choice(np.array([0, 1, NA], dtype=np.int32), size=m)
    o = call(R, "logicaland_csr_by_dvec_internal", sp, sj, R.logical(xl), R.logical(dl), R.integer([K]))
    assert R.typeof(o) == rmock.LGLSXP
    eq(R.view(o), O.logicaland_csr_by_dvec_internal(p, j, xl, dl, K))
    # with NAs: cells are added -> fresh structure
    dn = d.copy()
    dn[[2, 17, 40]] = np.nan
    o = call(R, "multiply_csr_by_dvec_with_NAs", sp, sj, sx, R.real(dn), R.integer([K]), T, F, F, F, F, T)
    w = O.multiply_csr_by_dvec_with_NAs(p, j, xx, dn, K, True, False, False, False, False, True)
    assert not w["alias_structure"]
    eq_list(R, o, w, rmock.REALSXP, "dvec with NAs")
    assert R.elts(o)[0] != sp and R.elts(o)[1] != sj
    # no cell added (a full row under NA keeps its pattern): the INPUT indptr / indices objects (operators.cpp:2639-2647)
    pf = np.arange(0, 4 * 5 + 1, 5, dtype=np.int32)
    jf = np.tile(np.arange(5, dtype=np.int32), 4)
    xf = rng.normal(size=20)
    df = np.array([1.0, np.nan, 2.0, 3.0])
    spf, sjf = R.integer(pf), R.integer(jf)
    w = O.multiply_csr_by_dvec_with_NAs(pf, jf, xf, df, 5, True, False, False, False, False, True)
    o = call(R, "multiply_csr_by_dvec_with_NAs", spf, sjf, R.real(xf), R.real(df), R.integer([5]), T, F, F, F, F, T)
    if w["alias_structure"]:
        assert R.elts(o)[0] == spf and R.elts(o)[1] == sjf, "alias return of the input SEXPs"
    eq_list(R, o, w, rmock.REALSXP, "dvec with NAs, unchanged structure")


# ----------------------------------------------------------------------------------------------- cbind (3) / rbind (1)
def test_cbind_exports(R, mats):
    p, j, x, p2, j2, x2, xl, xl2, K = (mats[k] for k in ("p", "j", "x", "p2", "j2", "x2", "xl", "xl2", "K"))
    j2s = (j2 + K).astype(np.int32)
    a = [R.integer(p), R.integer(j), R.real(x), R.integer(p2), R.integer(j2s), R.real(x2)]
    eq_list(R, call(R, "cbind_csr_numeric", *a), O.cbind_csr_numeric(p, j, x, p2, j2s, x2), rmock.REALSXP)
    eq_list(R, call(R, "cbind_csr_logical", a[0], a[1], R.logical(xl), a[3], a[4], R.logical(xl2)), O.cbind_csr_logical(p, j, xl, p2, j2s, xl2), rmock.LGLSXP)
    eq_list(R, call(R, "cbind_csr_binary", a[0], a[1], a[3], a[4]), O.cbind_csr_binary(p, j, p2, j2s), rmock.REALSXP)


def test_concat_csr_batch_fills_the_slots_of_out_in_place(R, mats):
    """rbind.cpp:35-38,171: `out` is an S4 object whose @p / @j / @x the R caller pre-sized (R/rbind.R:79-97); they are
    filled IN PLACE and `out` itself comes back.  Inputs are S4 objects read through their slots."""
    p, j, x, p2, j2, xl2 = mats["p"], mats["j"], mats["x"], mats["p2"], mats["j2"], mats["xl2"]
    K = mats["K"]
    dim = lambda nr: R.integer([nr, K])                                                        # noqa: E731
    A = R.s4("dgRMatrix", p=R.integer(p), j=R.integer(j), x=R.real(x), Dim=dim(61))
    B = R.s4("lgRMatrix", p=R.integer(p2), j=R.integer(j2), x=R.logical(xl2), Dim=dim(61))
    Cn = R.s4("ngRMatrix", p=R.integer(p2), j=R.integer(j2), Dim=dim(61))
    dv = R.s4("dsparseVector", i=R.integer([2, 5]), x=R.real([1.5, np.nan]), length=R.integer([K]))
    iv = R.s4("isparseVector", i=R.integer([1, 3]), x=R.integer([NA, 7]), length=R.integer([K]))
    lv = R.s4("lsparseVector", i=R.integer([3, 4]), x=R.logical([1, NA]), length=R.integer([K]))
    nv = R.s4("nsparseVector", i=R.integer([9]), length=R.integer([K]))
    objs_py = [(0, p, j, x, 61), (1, p2, j2, xl2, 61), (2, p2, j2, None, 61), (3, None, np.array([2, 5], np.int32), np.array([1.5, np.nan]), 1),
               (4, None, np.array([1, 3], np.int32), np.array([NA, 7], np.int32), 1), (5, None, np.array([3, 4], np.int32), np.array([1, NA], np.int32), 1),
               (6, None, np.array([9], np.int32), None, 1)]
    objects = R.list([A, B, Cn, dv, iv, lv, nv])
    nrows = 61 * 3 + 4
    nnz = j.size + 2 * j2.size + 2 + 2 + 2 + 1
    for out_kind, cls in ((0, "dgRMatrix"), (1, "lgRMatrix"), (2, "ngRMatrix")):
        slots = dict(p=R.integer(np.full(nrows + 1, -7)), j=R.integer(np.full(nnz, -7)), Dim=R.integer([nrows, K]))
        if out_kind == 0:
            slots["x"] = R.real(np.full(nnz, -7.0))
        elif out_kind == 1:
            slots["x"] = R.logical(np.full(nnz, -7))
        out = R.s4(cls, **slots)
        got = call(R, "concat_csr_batch", objects, out)
        assert got == out, "concat_csr_batch returns `out` itself"
        w = O.concat_csr_batch(objs_py, out_kind)
        assert R.slot(out, "p") == slots["p"] and R.slot(out, "j") == slots["j"]             # same slot OBJECTS, filled in place
        eq(R.view(slots["p"]), w["indptr"], cls)
        eq(R.view(slots["j"]), w["indices"], cls)
        if out_kind != 2:
            eq(R.view(slots["x"]), w["values"], cls)
    # errors raised BEFORE a device handle exists: a class the routine does not know (rbind.cpp:131-135), short slots
    bad = R.s4("zsparseVector", i=R.integer([1]), x=R.real([1.0]))
    out = R.s4("dgRMatrix", p=R.integer(np.zeros(3)), j=R.integer([0]), x=R.real([0.0]), Dim=R.integer([2, K]))
    with pytest.raises(rmock.RError, match="Invalid vector type"):
        call(R, "concat_csr_batch", R.list([A, bad]), out)
    with pytest.raises(rmock.RError, match="shorter than the result"):
        call(R, "concat_csr_batch", R.list([A, B]), out)
    with pytest.raises(rmock.RError, match="no slot of name"):
        call(R, "concat_csr_batch", R.list([A]), R.s4("dgRMatrix", p=R.integer([0]), j=R.integer([])))


# ----------------------------------------------------------------------------------------------- CSR x sparse vector (5), CSR (.) dense (5)
def test_csr_svec_exports(R, mats):
    p, j, x, m, K, rng = mats["p"], mats["j"], np.nan_to_num(mats["x"], nan=0.5, posinf=2.0), mats["m"], mats["K"], mats["rng"]
    sp, sj, sx = R.integer(p), R.integer(j), R.real(x)
    yi = (np.sort(rng.permutation(K)[:12]) + 1).astype(np.int32)
    yv = rng.normal(size=12)
    yint = rng.integers(-4, 5, size=12).astype(np.int32)
    yint[2] = NA
    ylg = rng.integers(0, 2, size=12).astype(np.int32)
    ylg[1] = NA
    nt = R.integer([1])
    for name, syv, want in (("matmul_csr_svec_numeric", R.real(yv), O.matmul_csr_svec_numeric(p, j, x, yi, yv)),
                            ("matmul_csr_svec_integer", R.integer(yint), O.matmul_csr_svec_integer(p, j, x, yi, yint)),
                            ("matmul_csr_svec_logical", R.logical(ylg), O.matmul_csr_svec_logical(p, j, x, yi, ylg)),
                            ("matmul_csr_svec_float32", R.float32(yv.astype(np.float32)), O.matmul_csr_svec_float32(p, j, x, yi, yv.astype(np.float32)))):
        o = call(R, name, sp, sj, sx, R.integer(yi), syv, nt)
        g = R.view(o)
        assert R.typeof(o) == rmock.REALSXP and np.array_equal(np.isnan(g), np.isnan(want)), name
        np.testing.assert_allclose(g[~np.isnan(g)], want[~np.isnan(want)], rtol=1e-11, atol=1e-12)
    o = call(R, "matmul_csr_svec_binary", sp, sj, sx, R.integer(yi), nt)
    np.testing.assert_allclose(R.view(o), O.matmul_csr_svec_binary(p, j, x, yi), rtol=1e-11, atol=1e-12)


def test_csr_by_dense_exports(R, mats):
    p, j, x, xl, m, K, rng = mats["p"], mats["j"], mats["x"], mats["xl"], mats["m"], mats["K"], mats["rng"]
    sp, sj, sx = R.integer(p), R.integer(j), R.real(x)
    D = rng.normal(size=(m, K))
    eq(R.view(call(R, "multiply_csr_by_dense_elemwise_double", sp, sj, sx, R.matrix(D))), O.multiply_csr_by_dense_elemwise_double(p, j, x, D))
    D32 = D.astype(np.float32)
    eq(R.view(call(R, "multiply_csr_by_dense_elemwise_float32", sp, sj, sx, R.matrix(D32, "float32"))), O.multiply_csr_by_dense_elemwise_float32(p, j, x, D32))
    Di = rng.integers(-3, 4, size=(m, K)).astype(np.int32)
    Di[5, :] = NA
    eq(R.view(call(R, "multiply_csr_by_dense_elemwise_int", sp, sj, sx, R.matrix(Di, "integer"))), O.multiply_csr_by_dense_elemwise_int(p, j, x, Di))
    Dl = rng.choice(np.array([0, 1, NA], np.int32), size=(m, K))
    eq(R.view(call(R, "multiply_csr_by_dense_elemwise_bool", sp, sj, sx, R.matrix(Dl, "logical"))), O.multiply_csr_by_dense_elemwise_bool(p, j, x, Dl))
    o = call(R, "logicaland_csr_by_dense_cpp", sp, sj, R.logical(xl), R.matrix(Dl, "logical"))
    assert R.typeof(o) == rmock.LGLSXP
    eq(R.view(o), O.logicaland_csr_by_dense_cpp(p, j, xl, Dl))


# ----------------------------------------------------------------------------------------------- sortedness / sort (6)
def test_sort_exports(R):
    p, j, x = rand_csr(50, 40, 0.3, seed=9, sorted_cols=False)
    rng = np.random.default_rng(3)
    xl = rng.choice(np.array([0, 1, NA], dtype=np.int32), size=x.size)
    sp = R.integer(p)
    o = call(R, "check_indices_are_unsorted", sp, R.integer(j))
    assert R.typeof(o) == rmock.LGLSXP and R.view(o).tolist() == [int(O.check_indices_are_sorted(p, j))] == [0]
    js, _ = O.sort_sparse_indices(p, j, x)
    assert R.view(call(R, "check_indices_are_unsorted", sp, R.integer(js))).tolist() == [1]
    for name, vals, mk, extra in (("sort_sparse_indices_numeric", x, R.real, ()), ("sort_sparse_indices_logical", xl, R.logical, ()),
                                  ("sort_sparse_indices_numeric_known_ncol", x, R.real, (R.integer([40]),)),
                                  ("sort_sparse_indices_logical_known_ncol", xl, R.logical, (R.integer([40]),)),
                                  ("sort_sparse_indices_binary", None, None, ())):
        wj, wv = O.sort_sparse_indices(p, j, vals)
        tj = R.integer(j)
        args = [sp, tj] + ([] if vals is None else [mk(vals)]) + list(extra)
        assert call(R, name, *args) is None                     # void export: R_NilValue
        eq(R.view(tj), wj, name)                                 # sorted IN PLACE on the caller's vector
        if vals is not None:
            eq(R.view(args[2]), wv, name)
    # the wrong type is an R error, not a reinterpretation of the bytes
    with pytest.raises(rmock.RError, match="numeric values required"):
        call(R, "sort_sparse_indices_numeric", sp, R.integer(j), R.integer(xl))


# ----------------------------------------------------------------------------------------------- drop zeros (2), validity (1), rowvec x CSC (2)
def test_remove_zero_valued_exports(R, mats):
    p, j, x, xl = mats["p"], mats["j"], mats["x"].copy(), mats["xl"]
    x[::4] = 0.0
    sp, sj, sx, sxl = R.integer(p), R.integer(j), R.real(x), R.logical(xl)
    for na in (False, True):
        eq_list(R, call(R, "remove_zero_valued_csr_numeric", sp, sj, sx, R.logical([int(na)])), O.remove_zero_valued_csr_numeric(p, j, x, na), rmock.REALSXP)
        eq_list(R, call(R, "remove_zero_valued_csr_logical", sp, sj, sxl, R.logical([int(na)])), O.remove_zero_valued_csr_logical(p, j, xl, na), rmock.LGLSXP)
    # nothing to remove: the caller's three vectors themselves (misc.cpp:586-590)
    ones = np.ones(j.size)
    s1 = R.real(ones)
    o = call(R, "remove_zero_valued_csr_numeric", sp, sj, s1, R.logical([0]))
    assert R.elts(o) == [sp, sj, s1] and R.names(o) == ["indptr", "indices", "values"]


def test_check_valid_csr_matrix_export(R, mats):
    p, j, m, K = mats["p"], mats["j"], mats["m"], mats["K"]
    o = call(R, "check_valid_csr_matrix", R.integer(p), R.integer(j), R.integer([m]), R.integer([K]))
    assert R.typeof(o) == rmock.VECSXP and R.as_py(o) == [] and O.check_valid_csr_matrix(p, j, m, K) == {}
    jb = j.copy()
    jb[7] = K + 3
    o = call(R, "check_valid_csr_matrix", R.integer(p), R.integer(jb), R.integer([m]), R.integer([K]))
    assert R.names(o) == ["err"] and R.as_py(o) == {"err": [O.check_valid_csr_matrix(p, jb, m, K)["err"]]}


def test_rowvec_by_csc_exports(R, mats):
    p, j, x, m, K, rng = mats["p"], mats["j"], np.nan_to_num(mats["x"], nan=0.5, posinf=2.0), mats["m"], mats["K"], mats["rng"]
    rv = rng.normal(size=K).astype(np.float32)              # a CSC with m columns over K rows = our CSR arrays
    o = call(R, "matmul_rowvec_by_csc", R.float32(rv), R.integer(p), R.integer(j), R.real(x))
    assert R.typeof(o) == rmock.INTSXP and R.view(o).shape == (1, m)
    np.testing.assert_allclose(R.view(o).view(np.float32), O.matmul_rowvec_by_csc(rv, p, j, x), rtol=1e-5, atol=1e-6)
    o = call(R, "matmul_rowvec_by_cscbin", R.float32(rv), R.integer(p), R.integer(j))
    np.testing.assert_allclose(R.view(o).view(np.float32), O.matmul_rowvec_by_cscbin(rv, p, j), rtol=1e-5, atol=1e-6)


# ----------------------------------------------------------------------------------------------- errors: long-jump, handle released
def _live_blocks(gpu):
    import ctypes as C
    v = C.c_int64()
    gpu.check(gpu.load().mx_get_option(b"pool_live_blocks", C.byref(v)))
    return v.value


def test_r_allocation_failure_longjumps_with_the_device_handle_released(R, gpu, mats):
    """Between mx_*_begin and mx_result_finish the shim allocates the R result vectors; R's allocator can long-jump
    ("cannot allocate vector of size ...").  The handle must be discarded by the R_ExecWithCleanup handler: the count of
    device blocks handed out (mx_get_option("pool_live_blocks")) is what it was, the protect stack is balanced, nothing is
    left on the R_PreserveObject list (rmock.call asserts the last two)."""
    # operands of ~3 MB each: the regular path, whose results wait in pooled device blocks between begin and finish (the
    # small path of calls under ~2 MiB keeps its result in the handle itself and is covered by the second half below)
    p, j, x = rand_csr(8000, 3000, 0.011, seed=21)
    p2, j2, x2 = rand_csr(8000, 3000, 0.011, seed=22)
    assert min(12 * j.size, 12 * j2.size) > (5 << 19)
    sp, sj, sx, sp2, sj2, sx2 = R.integer(p), R.integer(j), R.real(x), R.integer(p2), R.integer(j2), R.real(x2)
    rows = R.integer(np.array([4, 4, 9], dtype=np.int32))
    call(R, "add_csr_elemwise", sp, sp2, sj, sj2, sx, sx2, R.logical([0]))           # warm: cache entries, scratch
    call(R, "copy_csr_rows_numeric", sp, sj, sx, rows)
    base = _live_blocks(gpu)
    hit = 0
    small = tuple(R.integer(mats[k]) if k[0] in "pj" else R.real(mats[k]) for k in ("p", "p2", "j", "j2", "x", "x2"))
    for name, args in (("add_csr_elemwise", (sp, sp2, sj, sj2, sx, sx2, R.logical([0]))), ("copy_csr_rows_numeric", (sp, sj, sx, rows)),
                       ("add_csr_elemwise", small + (R.logical([0]),)), ("copy_csr_rows_numeric", (small[0], small[2], small[4], rows))):
        for k in range(8):                                        # the k-th R allocation of the call fails
            R.L.rmock_fail_alloc_at(k)
            try:
                call(R, name, *args)
            except rmock.RError as e:
                assert "cannot allocate" in str(e)
                hit += 1
            finally:
                R.L.rmock_fail_alloc_at(-1)
            assert _live_blocks(gpu) == base, f"{name}: device blocks leaked when R allocation {k} failed"
    assert hit >= 6
    # and the routines still work afterwards
    eq_list(R, call(R, "add_csr_elemwise", sp, sp2, sj, sj2, sx, sx2, R.logical([0])), O.add_csr_elemwise(p, p2, j, j2, x, x2, False), rmock.REALSXP)


def test_library_error_becomes_an_r_error(R, gpu, mats):
    """a non-zero status of the C-ABI -> Rf_error(mx_last_error()) (BEGIN_RCPP / END_RCPP, RcppExports.cpp:16,24)"""
    p, j, x = mats["p"], mats["j"], mats["x"]
    base = _live_blocks(gpu)
    with pytest.raises(rmock.RError):                            # row count mismatch between the two operands
        call(R, "add_csr_elemwise", R.integer(p), R.integer(p[:-3]), R.integer(j), R.integer(j), R.real(x), R.real(x), R.logical([0]))
    with pytest.raises(rmock.RError, match="unknown option"):
        call(R, "_mxgpu_set_option", R.string("no_such_option"), R.integer([1]))
    with pytest.raises(rmock.RError, match="cannot coerce"):        # a list where a vector is expected: R's own coercion error
        call(R, "check_is_seq", R.list([]))
    assert _live_blocks(gpu) == base


def test_control_routines(R, gpu):
    import ctypes as C
    assert call(R, "_mxgpu_set_option", R.string("spmv_algo"), R.integer([3])) is None
    v = C.c_int64()
    gpu.check(gpu.load().mx_get_option(b"spmv_algo", C.byref(v)))
    assert v.value == 3
    call(R, "_mxgpu_set_option", R.string("spmv_algo"), R.real([0]))
    assert call(R, "_mxgpu_set_devices", R.integer([])) is None
    assert call(R, "_mxgpu_set_devices", R.integer([0])) is None
    call(R, "_mxgpu_set_devices", R.integer([]))


def test_offload_gate_large_operands_reach_the_gpu_small_ones_matrixextras_routine(R, gpu):
    """with MatrixExtra's DLL beside the shim (the mock resolves _MatrixExtra_* to a stub returning "host"): a product of 6e4
    entries is served by the GPU (a real matrix, equal to the oracle), the reference's own test size (100 x 50,
    tests/testthat/test-matmul.R:108-114) by the host routine; the gather stays on the host up to 1e7 entries"""
    p, j, x = rand_csr(3000, 100, 0.2, seed=9)
    assert j.size >= 50_000
    Y = np.random.default_rng(2).normal(size=(7, 100))
    ps, js, xs = rand_csr(100, 50, 0.4, seed=10)
    Ys = np.random.default_rng(3).normal(size=(20, 50))
    R.L.rmock_set_host_routines(1)
    try:
        n0 = R.L.rmock_host_calls()
        out = call(R, "tcrossprod_csr_dense_numeric", R.integer(p), R.integer(j), R.real(x), R.matrix(Y), R.integer([1]))
        assert R.typeof(out) == rmock.REALSXP and R.L.rmock_host_calls() == n0
        np.testing.assert_allclose(R.view(out), O.tcrossprod_csr_dense_numeric(p, j, x, np.asfortranarray(Y)), rtol=1e-12, atol=1e-12)
        out = R.call("tcrossprod_csr_dense_numeric", R.integer(ps), R.integer(js), R.real(xs), R.matrix(Ys), R.integer([1]))
        assert R.as_py(out) == ["host"] and R.L.rmock_host_calls() == n0 + 1
        out = R.call("copy_csr_rows_numeric", R.integer(p), R.integer(j), R.real(x), R.integer([5, 1]))
        assert R.as_py(out) == ["host"] and R.L.rmock_host_calls() == n0 + 2
    finally:
        R.L.rmock_set_host_routines(0)
    R.check_clean()


def test_every_registered_routine_was_invoked(R):
    """runs last in this file: each of the reference-named routines went through the trampoline at least once"""
    names = {k[len("_MatrixExtra_"):] for k in R.routines() if k.startswith("_MatrixExtra_")}
    assert names - INVOKED == set(), sorted(names - INVOKED)
    assert len(names) == len(REFERENCE_ARITY) and R.calls >= len(names)
    R.check_clean()
