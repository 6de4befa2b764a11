"""The CPU restatement (oracle/) against (1) the literal known answers the reference tree holds and
(2) an independent dense numpy / scipy evaluation on seeded inputs.  CPU-only.

The reference publishes no golden vectors for these routines (its tests draw inputs from R's RNG,
SURVEY.md §4); the KATs below are the ones that do exist in the tree:
  * vignettes/Introducing_MatrixExtra.Rmd:164-166 3x3 matrix, with X+X (:186), Xr[1:2,] (:306), Xr*Xr (:316,
    printed in inst/doc/Introducing_MatrixExtra.html:648)
  * tests/testthat/test-utilities.R:32-49 index-sort KAT
"""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import csr_to_dense, rand_csr
from oracle import oracle as O

NA = int(O.NA_INTEGER)

# vignette 3x3: i=(1,1,2,3), j=(1,3,3,2), x=(1,2,3,4)  ->  CSR
VIG_P = np.array([0, 2, 3, 4], dtype=np.int32)
VIG_J = np.array([0, 2, 2, 1], dtype=np.int32)
VIG_X = np.array([1.0, 2.0, 3.0, 4.0])


def test_kat_vignette_add():
    # X + X  = [[2,.,4],[.,.,6],[.,8,.]]
    r = O.add_csr_elemwise(VIG_P, VIG_P.copy(), VIG_J, VIG_J.copy(), VIG_X, VIG_X.copy(), False)
    assert r["indptr"].tolist() == [0, 2, 3, 4]
    assert r["indices"].tolist() == [0, 2, 2, 1]
    assert r["values"].tolist() == [2.0, 4.0, 6.0, 8.0]


def test_kat_vignette_multiply():
    # Xr * Xr = [[1,.,4],[.,.,9],[.,16,.]]
    r = O.multiply_csr_elemwise(VIG_P, VIG_P.copy(), VIG_J, VIG_J.copy(), VIG_X, VIG_X.copy())
    assert r["indptr"].tolist() == [0, 2, 3, 4]
    assert r["indices"].tolist() == [0, 2, 2, 1]
    assert r["values"].tolist() == [1.0, 4.0, 9.0, 16.0]


def test_kat_vignette_rows():
    # Xr[1:2,] is a contiguous run in R (pure-R branch); the gather of rows (0,1) must agree with it,
    # and a non-sequential pick exercises copy_csr_rows proper
    r = O.copy_csr_rows_numeric(VIG_P, VIG_J, VIG_X, np.array([0, 1], dtype=np.int32))
    assert r["indptr"].tolist() == [0, 2, 3] and r["indices"].tolist() == [0, 2, 2]
    assert r["values"].tolist() == [1.0, 2.0, 3.0]
    r = O.copy_csr_rows_numeric(VIG_P, VIG_J, VIG_X, np.array([2, 0, 2], dtype=np.int32))
    assert r["indptr"].tolist() == [0, 1, 3, 4] and r["indices"].tolist() == [1, 0, 2, 1]
    assert r["values"].tolist() == [4.0, 1.0, 2.0, 4.0]


def test_kat_sort_indices():
    # tests/testthat/test-utilities.R:32-49
    p = np.array([0, 1, 4, 5, 6], dtype=np.int32)
    j = np.array([4, 2, 1, 4, 1, 0], dtype=np.int32)
    x = np.array([-0.91, 0.14, -0.12, -0.12, 1.1, 0.66])
    assert not O.check_indices_are_sorted(p, j)
    js, xs = O.sort_sparse_indices(p, j, x)
    assert js.tolist() == [4, 1, 2, 4, 1, 0]
    assert xs.tolist() == [-0.91, -0.12, 0.14, -0.12, 1.1, 0.66]
    assert j.tolist() == [4, 2, 1, 4, 1, 0]          # copy=TRUE leaves the input alone
    assert O.check_indices_are_sorted(p, js)


def test_readme_example_matrix_times_dense():
    # the 3 x 4 CSR matrix the README builds (README.md:17-27: p = [0,2,3,3], j = [2,3,1], x = [2,1,3], i.e.
    # [[0,0,2,1],[0,3,0,0],[0,0,0,0]] with an empty last row).  The README prints no product: this is a fixed small
    # case against dense numpy, not a known answer held by the reference.
    p = np.array([0, 2, 3, 3], dtype=np.int32)
    j = np.array([2, 3, 1], dtype=np.int32)
    x = np.array([2.0, 1.0, 3.0])
    X = np.array([[0.0, 0, 2, 1], [0, 3, 0, 0], [0, 0, 0, 0]])
    B = np.arange(12, dtype=np.float64).reshape(4, 3) - 5
    out = O.tcrossprod_csr_dense_numeric(p, j, x, B.T.copy())
    np.testing.assert_array_equal(out, X @ B)
    np.testing.assert_array_equal(O.matmul_csr_dvec_numeric(p, j, x, B[:, 0].copy()), X @ B[:, 0])


@pytest.mark.parametrize("use_fma", [False, True])
@pytest.mark.parametrize("shape", [(100, 50, 20), (7, 5, 1), (1, 9, 4), (30, 40, 130)])
def test_spmm_vs_dense(shape, use_fma):
    m, K, n = shape
    p, j, x = rand_csr(m, K, 0.4, seed=m * 31 + n, sorted_cols=False, empty_rows=(0,) if m > 2 else ())
    Ad = csr_to_dense(p, j, x, K)
    rng = np.random.default_rng(5)
    B = rng.normal(size=(K, n))
    expect = Ad @ B
    got = O.tcrossprod_csr_dense_numeric(p, j, x, np.asfortranarray(B.T), nthreads=2, use_fma=use_fma)
    np.testing.assert_allclose(got, expect, rtol=1e-12, atol=1e-12)
    assert got.flags.f_contiguous and got.shape == (m, n)
    # float32 dense, f64 CSR values
    got32 = O.tcrossprod_csr_dense_float32(p, j, x, np.asfortranarray(B.T.astype(np.float32)), use_fma=use_fma)
    assert got32.dtype == np.float32
    np.testing.assert_allclose(got32, expect, rtol=2e-4, atol=2e-4)
    # dense %*% CSC  (CSC of Y == CSR of t(Y)): X(n x m) %*% t(A)(m... use A as CSR of t(Y), Y = t(A): K x m
    Xd = rng.normal(size=(n, K))
    # tcrossprod(X, A) = X %*% t(A) -> (n x m)
    got_t = O.tcrossprod_dense_csr_numeric(np.asfortranarray(Xd), p, j, x, nthreads=1, ncols_Y=K, use_fma=use_fma)
    np.testing.assert_allclose(got_t, Xd @ Ad.T, rtol=1e-12, atol=1e-12)
    # matmul_dense_csc: Y is (K' x ncolY) CSC; reuse (p,j,x) as CSC of Y with ncol(Y)=m, nrow(Y)=K -> Y = t(A)
    got_c = O.matmul_dense_csc_numeric(np.asfortranarray(Xd), p, j, x, use_fma=use_fma)
    np.testing.assert_allclose(got_c, Xd @ Ad.T, rtol=1e-12, atol=1e-12)


def test_spmm_duplicates_accumulate_and_empty():
    p = np.array([0, 3, 3], dtype=np.int32)
    j = np.array([1, 1, 0], dtype=np.int32)
    x = np.array([2.0, 3.0, 1.0])
    B = np.array([[1.0, 2.0], [10.0, 20.0]])
    out = O.tcrossprod_csr_dense_numeric(p, j, x, np.asfortranarray(B.T))
    np.testing.assert_array_equal(out, np.array([[51.0, 102.0], [0.0, 0.0]]))
    # all-empty matrix: early return keeps zeros (matmul.cpp:160-161)
    out = O.tcrossprod_csr_dense_numeric(np.zeros(4, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros(0),
                                         np.asfortranarray(B.T))
    assert out.shape == (3, 2) and not out.any()


def test_spmv_kinds():
    p, j, x = rand_csr(60, 25, 0.3, seed=3, empty_rows=(5,))
    Ad = csr_to_dense(p, j, x, 25)
    rng = np.random.default_rng(1)
    y = rng.normal(size=25)
    np.testing.assert_allclose(O.matmul_csr_dvec_numeric(p, j, x, y, 2), Ad @ y, rtol=1e-12, atol=1e-12)
    yi = rng.integers(-5, 5, size=25).astype(np.int32)
    np.testing.assert_allclose(O.matmul_csr_dvec_integer(p, j, x, yi), Ad @ yi, rtol=1e-12, atol=1e-12)
    yl = rng.integers(0, 2, size=25).astype(np.int32)
    np.testing.assert_allclose(O.matmul_csr_dvec_logical(p, j, x, yl * 7), Ad @ yl, rtol=1e-12, atol=1e-12)
    yf = y.astype(np.float32)
    r = O.matmul_csr_dvec_float32(p, j, x, yf)
    assert r.dtype == np.float32
    np.testing.assert_allclose(r, Ad @ yf.astype(np.float64), rtol=1e-5, atol=1e-5)
    # NA_INTEGER / NA_LOGICAL -> NA_REAL in every row that touches the entry
    yi2 = yi.copy(); yi2[3] = NA
    r = O.matmul_csr_dvec_integer(p, j, x, yi2)
    touched = Ad[:, 3] != 0
    assert np.isnan(r[touched]).all() and not np.isnan(r[~touched]).any()
    r = O.matmul_csr_dvec_logical(p, j, x, yi2)
    assert np.isnan(r[touched]).all() and not np.isnan(r[~touched]).any()


def test_r_logical_tables():
    # R: NA&TRUE=NA NA&FALSE=FALSE NA&NA=NA NA|TRUE=TRUE NA|FALSE=NA NA|NA=NA (operators.cpp:8-15)
    f = O.lib().mxo_r_logical
    OR, AND, XOR = 0, 1, 2
    assert f(AND, NA, 1) == NA and f(AND, NA, 0) == 0 and f(AND, NA, NA) == NA
    assert f(AND, 1, NA) == NA and f(AND, 0, NA) == 0
    assert f(OR, NA, 1) == 1 and f(OR, NA, 0) == NA and f(OR, NA, NA) == NA
    assert f(OR, 1, NA) == 1 and f(OR, 0, NA) == NA
    assert f(XOR, NA, 1) == NA and f(XOR, 1, 0) == 1 and f(XOR, 1, 1) == 0 and f(XOR, 0, 0) == 0
    assert f(OR, 2, 0) == 1 and f(AND, 2, 3) == 1


def _dense_union(p1, j1, x1, p2, j2, x2, K, op):
    A, B = csr_to_dense(p1, j1, x1, K), csr_to_dense(p2, j2, x2, K)
    return op(A, B)


@pytest.mark.parametrize("dens", [(0.4, 0.6), (0.05, 0.9), (0.0, 0.5), (0.5, 0.0)])
def test_add_sub_mul_vs_dense_and_structure(dens):
    m, K = 100, 35
    p1, j1, x1 = rand_csr(m, K, dens[0], seed=11, empty_rows=(3,))
    p2, j2, x2 = rand_csr(m, K, dens[1], seed=12, empty_rows=(3, 4))
    for sub in (False, True):
        r = O.add_csr_elemwise(p1, p2, j1, j2, x1, x2, sub)
        D = csr_to_dense(r["indptr"], r["indices"], r["values"], K)
        np.testing.assert_array_equal(D, _dense_union(p1, j1, x1, p2, j2, x2, K, np.subtract if sub else np.add))
        # structure = union of patterns, sorted, unique
        pat = (csr_to_dense(p1, j1, None, K) + csr_to_dense(p2, j2, None, K)) > 0
        assert r["indptr"].tolist() == [0] + np.cumsum(pat.sum(axis=1)).tolist()
        assert r["indices"].tolist() == np.nonzero(pat)[1].tolist()
    r = O.multiply_csr_elemwise(p1, p2, j1, j2, x1, x2)
    pat = (csr_to_dense(p1, j1, None, K) * csr_to_dense(p2, j2, None, K)) > 0
    assert r["indptr"].tolist() == [0] + np.cumsum(pat.sum(axis=1)).tolist()
    assert r["indices"].tolist() == np.nonzero(pat)[1].tolist()
    np.testing.assert_array_equal(csr_to_dense(r["indptr"], r["indices"], r["values"], K),
                                  _dense_union(p1, j1, x1, p2, j2, x2, K, np.multiply))


def test_add_keeps_explicit_zero_on_cancellation():
    p = np.array([0, 2], dtype=np.int32)
    j = np.array([1, 3], dtype=np.int32)
    x = np.array([1.5, -2.0])
    r = O.add_csr_elemwise(p, p.copy(), j, j.copy(), x, x.copy(), True)     # different buffers: general path
    assert r["indptr"].tolist() == [0, 2] and r["indices"].tolist() == [1, 3] and r["values"].tolist() == [0.0, 0.0]
    r = O.add_csr_elemwise(p, p, j, j, x, x, True)                          # same buffers: all-empty fast path
    assert r["indptr"].tolist() == [0, 0] and r["indices"].size == 0 and r["values"].size == 0
    r = O.add_csr_elemwise(p, p, j, j, x, x.copy(), False)                  # same structure: aliased
    assert r["indptr"] is p and r["indices"] is j and r["values"].tolist() == [3.0, -4.0]
    r = O.multiply_csr_elemwise(p, p, j, j, x, x)
    assert r["indptr"] is p and r["values"].tolist() == [2.25, 4.0]


def test_sub_negates_b_only_entries_including_zero_sign():
    p1 = np.array([0, 1], dtype=np.int32); j1 = np.array([0], dtype=np.int32); x1 = np.array([1.0])
    p2 = np.array([0, 2], dtype=np.int32); j2 = np.array([1, 2], dtype=np.int32); x2 = np.array([0.0, 5.0])
    r = O.add_csr_elemwise(p1, p2, j1, j2, x1, x2, True)
    assert r["indices"].tolist() == [0, 1, 2]
    assert r["values"][2] == -5.0 and np.signbit(r["values"][1])          # -(0.0) = -0.0 (operators.cpp:434)


def test_logical_or_xor_and():
    m, K = 40, 17
    p1, j1, x1 = rand_csr(m, K, 0.5, seed=21, dtype="l")
    p2, j2, x2 = rand_csr(m, K, 0.5, seed=22, dtype="l")
    f = O.lib().mxo_r_logical

    def dense_l(p, j, x):
        D = np.full((m, K), -1, dtype=np.int64)          # -1 = structurally absent
        for r in range(m):
            D[r, j[p[r]:p[r + 1]]] = x[p[r]:p[r + 1]]
        return D
    D1, D2 = dense_l(p1, j1, x1), dense_l(p2, j2, x2)
    for xor in (False, True):
        r = O.logicalor_csr_elemwise(p1, p2, j1, j2, x1, x2, xor)
        Dr = dense_l(r["indptr"], r["indices"], r["values"])
        for a in range(m):
            for b in range(K):
                u, v = D1[a, b], D2[a, b]
                if u == -1 and v == -1:
                    assert Dr[a, b] == -1
                elif u == -1:
                    assert Dr[a, b] == v                  # one-sided entries are copied verbatim
                elif v == -1:
                    assert Dr[a, b] == u
                else:
                    assert Dr[a, b] == f(2 if xor else 0, int(u), int(v))
    r = O.logicaland_csr_elemwise(p1, p2, j1, j2, x1, x2)
    Dr = dense_l(r["indptr"], r["indices"], r["values"])
    both = (D1 != -1) & (D2 != -1)
    assert ((Dr != -1) == both).all()
    for a, b in zip(*np.nonzero(both)):
        assert Dr[a, b] == f(1, int(D1[a, b]), int(D2[a, b]))


def test_gather_rows():
    p, j, x = rand_csr(1000, 500, 0.1, seed=7, empty_rows=(10, 11))
    rng = np.random.default_rng(9)
    rows = rng.integers(0, 1000, size=300).astype(np.int32)
    rows[:4] = [999, 999, 10, 0]
    r = O.copy_csr_rows_numeric(p, j, x, rows)
    A = sp.csr_matrix((x, j, p), shape=(1000, 500))
    E = A[rows]
    assert r["indptr"].tolist() == E.indptr.tolist()
    assert r["indices"].tolist() == E.indices.tolist()
    assert r["values"].tolist() == E.data.tolist()
    rb = O.copy_csr_rows_binary(p, j, rows)
    assert rb["indices"].tolist() == E.indices.tolist() and rb["values"].size == 0
    xl = (x > 0).astype(np.int32)
    rl = O.copy_csr_rows_logical(p, j, xl, rows)
    assert rl["values"].dtype == np.int32 and rl["values"].tolist() == sp.csr_matrix((xl, j, p), shape=(1000, 500))[rows].data.tolist()
    # selection with no entries at all: three EMPTY vectors (slice.cpp:236-240)
    e = O.copy_csr_rows_numeric(p, j, x, np.array([10, 11, 10], dtype=np.int32))
    assert e["indptr"].size == 0 and e["indices"].size == 0 and e["values"].size == 0


def test_check_is_seq():
    assert O.check_is_seq([]) and O.check_is_seq([5]) and O.check_is_seq([3, 4, 5])
    assert not O.check_is_seq([3, 5, 5]) and not O.check_is_seq([3, 5, 4, 6]) and not O.check_is_seq([5, 4, 3])
    assert O.check_is_rev_seq([5, 4, 3]) and O.check_is_rev_seq([1]) and not O.check_is_rev_seq([3, 4, 5])
    assert not O.check_is_rev_seq([6, 4, 5, 3])


# ----------------------------------------------------------------------------- §8(f) rank 2: column-filtering slices
def test_col_seq_and_arbitrary_vs_scipy():
    p, j, x = rand_csr(200, 80, 0.15, seed=17, empty_rows=(5, 6))
    A = sp.csr_matrix((x, j, p), shape=(200, 80))
    rng = np.random.default_rng(3)
    rows = rng.integers(0, 200, size=60).astype(np.int32)
    cols_seq = np.arange(10, 41, dtype=np.int32)                     # R passes 1-based j with index1=TRUE
    r = O.copy_csr_rows_col_seq_numeric(p, j, x, rows, cols_seq + 1, True)
    E = A[rows][:, 10:41]
    E.sort_indices()
    assert r["indptr"].tolist() == E.indptr.tolist() and r["indices"].tolist() == E.indices.tolist()
    assert r["values"].tolist() == E.data.tolist()
    rb = O.copy_csr_rows_col_seq_binary(p, j, rows, cols_seq, False)
    assert rb["indices"].tolist() == E.indices.tolist() and rb["values"].size == 0
    rl = O.copy_csr_rows_col_seq_logical(p, j, (x > 0).astype(np.int32), rows, cols_seq, False)
    assert rl["values"].dtype == np.float64                          # NumericVector in the reference
    e = O.copy_csr_rows_col_seq_numeric(p, j, x, np.array([5, 6], dtype=np.int32), cols_seq, False)
    assert e["indptr"].tolist() == [0, 0, 0] and e["indices"].size == 0
    # arbitrary columns: unsorted, with a repeated column
    cols = np.array([70, 3, 3, 41, 0, 79, 12], dtype=np.int32)
    r = O.copy_csr_arbitrary_numeric(p, j, x, rows, cols)
    E = A[rows][:, cols]
    E.sort_indices()
    assert r["indptr"].tolist() == E.indptr.tolist() and r["indices"].tolist() == E.indices.tolist()
    assert r["values"].tolist() == E.data.tolist()
    cs = np.sort(cols)
    r = O.copy_csr_arbitrary_numeric(p, j, x, rows, cs)
    E = A[rows][:, cs]
    E.sort_indices()
    assert r["indices"].tolist() == E.indices.tolist() and r["values"].tolist() == E.data.tolist()
    assert "values" not in O.copy_csr_arbitrary_binary(p, j, rows, cols)


def test_reverse_rows_and_columns():
    p, j, x = rand_csr(30, 17, 0.3, seed=23, empty_rows=(0, 29))
    A = sp.csr_matrix((x, j, p), shape=(30, 17))
    r = O.reverse_rows_numeric(p, j, x)
    E = A[::-1]
    assert r["indptr"].tolist() == E.indptr.tolist() and r["indices"].tolist() == E.indices.tolist()
    assert r["values"].tolist() == E.data.tolist()
    jj, xx = j.copy(), x.copy()
    O.reverse_columns_inplace(p, jj, xx, 17)
    E = sp.csr_matrix(A[:, ::-1])
    E.sort_indices()
    assert jj.tolist() == E.indices.tolist() and xx.tolist() == E.data.tolist()


# ----------------------------------------------------------------------------- §8(f) rank 3: cbind / rbind
def test_cbind_rbind_vs_scipy():
    p1, j1, x1 = rand_csr(40, 13, 0.3, seed=51, empty_rows=(2,))
    p2, j2, x2 = rand_csr(40, 9, 0.4, seed=52, empty_rows=(2, 3))
    A, B = sp.csr_matrix((x1, j1, p1), shape=(40, 13)), sp.csr_matrix((x2, j2, p2), shape=(40, 9))
    r = O.cbind_csr_numeric(p1, j1, x1, p2, j2 + 13, x2)
    E = sp.hstack([A, B]).tocsr(); E.sort_indices()
    assert r["indptr"].tolist() == E.indptr.tolist() and r["indices"].tolist() == E.indices.tolist()
    assert r["values"].tolist() == E.data.tolist()
    rb = O.cbind_csr_binary(p1, j1, p2, j2 + 13)
    assert rb["indices"].tolist() == E.indices.tolist() and rb["values"].size == 0
    # unequal row counts (cbind.cpp:33-38, 75-97): the shorter operand contributes nothing past its end
    r = O.cbind_csr_numeric(p1, j1, x1, p2[:21], j2[:p2[20]] + 13, x2[:p2[20]])
    E = sp.hstack([A, sp.vstack([B[:20], sp.csr_matrix((20, 9))])]).tocsr(); E.sort_indices()
    assert r["indptr"].tolist() == E.indptr.tolist() and r["values"].tolist() == E.data.tolist()
    assert O.concat_indptr2(p1, p2).tolist() == np.concatenate([p1, p1[-1] + p2[1:]]).tolist()
    # rbind batch: dgR + lgR (with NA) + ngR + d/i/l/n sparse vectors -> dgRMatrix
    xl = np.where(np.arange(j2.size) % 7 == 0, NA, (x2 > 0).astype(np.int32)).astype(np.int32)
    objs = [(0, p1, j1, x1, 40), (1, p2, j2, xl, 40), (2, p2, j2, None, 40),
            (3, None, np.array([2, 5], np.int32), np.array([1.5, -2.0]), 1),
            (4, None, np.array([1], np.int32), np.array([NA], np.int32), 1),
            (5, None, np.array([3, 4], np.int32), np.array([1, NA], np.int32), 1),
            (6, None, np.array([9], np.int32), None, 1)]
    r = O.concat_csr_batch(objs, 0)
    assert r["indptr"].size == 40 * 3 + 4 + 1 and r["indptr"][-1] == r["indices"].size
    np.testing.assert_array_equal(r["values"][: x1.size], x1)
    seg = r["values"][x1.size: x1.size + xl.size]
    np.testing.assert_array_equal(np.isnan(seg), xl == NA)
    np.testing.assert_array_equal(seg[~np.isnan(seg)], xl[xl != NA].astype(float))
    assert (r["values"][x1.size + xl.size: x1.size + 2 * xl.size] == 1.0).all()
    assert r["indices"][-6:].tolist() == [1, 4, 0, 2, 3, 8]                    # 1-based -> 0-based
    tail = r["values"][-6:]
    assert tail[0] == 1.5 and tail[1] == -2.0 and np.isnan(tail[2]) and tail[3] == 1.0 and np.isnan(tail[4]) and tail[5] == 1.0
    rl = O.concat_csr_batch(objs, 1)
    assert rl["values"].dtype == np.int32 and rl["values"][0] == int(x1[0] != 0)


# ----------------------------------------------------------------------------- §8(f) rank 4
def test_csr_svec_and_csr_by_dense():
    p, j, x = rand_csr(80, 60, 0.2, seed=61, empty_rows=(1,))
    Ad = csr_to_dense(p, j, x, 60)
    rng = np.random.default_rng(62)
    yi = np.sort(rng.permutation(60)[:20]).astype(np.int32) + 1            # 1-based sorted
    yv = rng.normal(size=20)
    ydense = np.zeros(60); ydense[yi - 1] = yv
    np.testing.assert_allclose(O.matmul_csr_svec_numeric(p, j, x, yi, yv, 2), Ad @ ydense, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(O.matmul_csr_svec_binary(p, j, x, yi), Ad @ (ydense != 0), rtol=1e-12, atol=1e-12)
    yint = rng.integers(1, 5, size=20).astype(np.int32)
    yd2 = np.zeros(60); yd2[yi - 1] = yint
    np.testing.assert_allclose(O.matmul_csr_svec_integer(p, j, x, yi, yint), Ad @ yd2, rtol=1e-12, atol=1e-12)
    yint[3] = NA
    r = O.matmul_csr_svec_integer(p, j, x, yi, yint)
    touched = Ad[:, yi[3] - 1] != 0
    assert np.isnan(r[touched]).all() and not np.isnan(r[~touched]).any()
    assert not O.matmul_csr_svec_numeric(p, j, x, np.zeros(0, np.int32), np.zeros(0)).any()
    D = rng.normal(size=(80, 60))
    out = O.multiply_csr_by_dense_elemwise_double(p, j, x, D)
    rows = np.repeat(np.arange(80), np.diff(p))
    np.testing.assert_array_equal(out, x * D[rows, j])
    Di = rng.integers(-2, 3, size=(80, 60)).astype(np.int32); Di[0, :] = NA
    oi = O.multiply_csr_by_dense_elemwise_int(p, j, x, Di)
    assert np.isnan(oi[rows == 0]).all() and np.array_equal(oi[rows != 0], (x * Di[rows, j])[rows != 0])
    xl = rng.choice(np.array([0, 1, NA], np.int32), size=x.size)
    Dl = rng.choice(np.array([0, 1, NA], np.int32), size=(80, 60))
    ol = O.logicaland_csr_by_dense_cpp(p, j, xl, Dl)
    f = O.lib().mxo_r_logical
    assert ol.tolist() == [f(1, int(a), int(b)) for a, b in zip(xl, Dl[rows, j])]


def test_csr_by_dvec_against_numpy():
    """multiply_csr_by_dvec_no_NAs (src/operators.cpp:1604-2140): each of the reference's four vector-length branches
    equals base-R recycling of the vector down the columns; * and / are single IEEE operations, %% %/% ^ follow R's
    arithmetic (numpy's mod / floor_divide / power agree on finite inputs)."""
    p, j, x = rand_csr(48, 11, 0.3, seed=8, empty_rows=(2,))
    x = (x * 6).round(2); x[x == 0] = 2.5
    r = np.repeat(np.arange(48), np.diff(p))
    rng = np.random.default_rng(9)
    for ln in (48, 48 * 11, 12, 1, 7, 100):
        v = (rng.uniform(0.5, 3.0, size=ln) * rng.choice([-1.0, 1.0], size=ln)).round(2)
        d = np.resize(v, 48 * 11).reshape(11, 48).T[r, j]
        f = lambda *fl: O.multiply_csr_by_dvec_no_NAs_numeric(p, j, x, v, 11, *fl)   # noqa: E731
        np.testing.assert_array_equal(f(1, 0, 0, 0, 0, True), x * d)
        np.testing.assert_array_equal(f(1, 0, 0, 0, 0, False), x * d)
        np.testing.assert_array_equal(f(0, 0, 1, 0, 0, True), x / d)
        np.testing.assert_array_equal(f(0, 0, 1, 0, 0, False), d / x)
        np.testing.assert_allclose(f(0, 0, 0, 1, 0, True), np.mod(x, d), rtol=1e-12, atol=1e-14)
        np.testing.assert_allclose(f(0, 0, 0, 1, 0, False), np.mod(d, x), rtol=1e-12, atol=1e-14)
        np.testing.assert_array_equal(f(0, 0, 0, 0, 1, True), np.floor_divide(x, d))
        np.testing.assert_array_equal(f(0, 0, 0, 0, 1, False), np.floor_divide(d, x))
        with np.errstate(invalid="ignore"):
            np.testing.assert_allclose(f(0, 1, 0, 0, 0, True), np.power(x, d), rtol=1e-14, equal_nan=True)
            np.testing.assert_allclose(f(0, 1, 0, 0, 0, False), np.power(d, x), rtol=1e-14, equal_nan=True)
    # R's documented corner cases of ^ %% %/% (?Arithmetic): x^0 = 1 and 1^y = 1 even for NaN, x %% 0 = NaN,
    # x %/% 0 = +-Inf, the sign of %% follows the divisor, -Inf^odd = -Inf
    one = lambda a, b, *fl: O.multiply_csr_by_dvec_no_NAs_numeric(np.array([0, 1], np.int32), np.array([0], np.int32),  # noqa: E731
                                                                 np.array([a]), np.array([b]), 1, *fl, True)[0]
    assert one(np.nan, 0.0, 0, 1, 0, 0, 0) == 1.0 and one(1.0, np.nan, 0, 1, 0, 0, 0) == 1.0
    assert np.isnan(one(5.0, 0.0, 0, 0, 0, 1, 0)) and one(5.0, 0.0, 0, 0, 0, 0, 1) == np.inf
    assert one(-5.0, 3.0, 0, 0, 0, 1, 0) == 1.0 and one(5.0, -3.0, 0, 0, 0, 1, 0) == -1.0
    assert one(-5.0, 3.0, 0, 0, 0, 0, 1) == -2.0
    assert one(-np.inf, 3.0, 0, 1, 0, 0, 0) == -np.inf and one(-np.inf, 2.0, 0, 1, 0, 0, 0) == np.inf
    assert np.isnan(one(-8.0, 1.0 / 3.0, 0, 1, 0, 0, 0))


def test_logicaland_csr_by_dvec():
    """logicaland_csr_by_dvec_internal (src/operators.cpp:2177-2200) against R's truth table, NA included."""
    p, j, xl = rand_csr(30, 9, 0.4, seed=21, dtype="l")
    r = np.repeat(np.arange(30), np.diff(p))
    NA = O.NA_INTEGER
    for ln in (30, 270, 10, 4):
        vl = np.random.default_rng(ln).integers(0, 2, size=ln).astype(np.int32)
        vl[::3] = NA
        d = np.resize(vl, 270).reshape(9, 30).T[r, j]
        want = np.where((xl == 0) | (d == 0), 0, np.where((xl == NA) | (d == NA), NA, 1)).astype(np.int32)
        np.testing.assert_array_equal(O.logicaland_csr_by_dvec_internal(p, j, xl, vl, 9), want)


def test_oracle_csr_by_dvec_with_NAs_matches_dense_r_arithmetic():
    """The oracle's restatement of multiply_csr_by_dvec_with_NAs (src/operators.cpp:2258-2856) against what R's dense
    arithmetic gives for `as.matrix(X) op recycled(v)` (the reference's own tests check it that way, test-operators.R): same
    NaN pattern, same finite values, pattern = X's cells + every cell the vector makes special, rows sorted."""
    p, j, x = rand_csr(40, 9, 0.3, seed=5, empty_rows=(4,))
    x = (x * 3).round(1)
    x[x == 0] = 2.0
    dense = csr_to_dense(p, j, x, 9)
    held = csr_to_dense(p, j, np.ones_like(x), 9) != 0
    NA = np.frombuffer(np.uint64(0x7FF00000000007A2).tobytes(), dtype=np.float64)[0]
    rng = np.random.default_rng(8)
    ops = {"mul": ((1, 0, 0, 0, 0), lambda a, b: a * b), "div": ((0, 0, 1, 0, 0), lambda a, b: a / b),
           "pow": ((0, 1, 0, 0, 0), None)}
    for ln in (40, 20, 360, 7, 43, 120):
        for name, (flags, f) in ops.items():
            v = rng.uniform(0.5, 3.0, size=ln).round(1)
            sp = rng.random(ln) < 0.15
            sp[rng.integers(ln)] = True
            v[sp] = rng.choice(np.array([NA, np.nan, np.inf] if name == "mul" else [NA, 0.0] if name == "div" else [0.0, -1.0, NA]),
                               size=int(sp.sum()))
            r = O.multiply_csr_by_dvec_with_NAs(p, j, x, v, 9, *flags, True)
            rec = np.resize(v, 40 * 9).reshape(9, 40).T                        # R's recycling runs down the columns
            got = np.zeros((40, 9))
            present = np.zeros((40, 9), dtype=bool)
            for i in range(40):
                cols = r["indices"][r["indptr"][i]:r["indptr"][i + 1]]
                assert np.all(np.diff(cols) > 0)
                got[i, cols] = r["values"][r["indptr"][i]:r["indptr"][i + 1]]
                present[i, cols] = True
            if name == "pow":
                special = np.isnan(rec) | (rec <= 0)
                with np.errstate(all="ignore"):
                    exp = np.where(rec == 0, 1.0, np.power(dense, rec))
                    exp = np.where((dense == 0) & (rec < 0), np.inf, exp)
            else:
                special = np.isnan(rec) | (np.isinf(rec) if name == "mul" else rec == 0)
                with np.errstate(all="ignore"):
                    exp = f(dense, rec)
            if not r["alias_structure"]:
                np.testing.assert_array_equal(present, held | special)
            np.testing.assert_array_equal(np.isnan(got), np.isnan(exp), err_msg=f"{name} len={ln}")
            ok = ~np.isnan(exp)
            np.testing.assert_allclose(got[ok], exp[ok], rtol=1e-14)


def _drop_reference(p, j, x, remove_NAs, logical):
    """remove_zero_valued_csr restated with numpy masks (src/misc.cpp:553-664), incl. its first scan and its quirk"""
    NA = np.int32(-2147483648)
    if logical:
        dirty = (x == 0) | ((x == NA) if remove_NAs else False)
        keep = (x != NA) if remove_NAs else (x != 0)             # (with remove_NAs the zeros STAY: misc.cpp:636-647)
    else:
        dirty = (x == 0) | (np.isnan(x) if remove_NAs else False)
        keep = (x != 0) & (~np.isnan(x) if remove_NAs else True)
    if not np.any(dirty):
        return None
    rows = np.repeat(np.arange(p.size - 1), np.diff(p))
    counts = np.bincount(rows[keep], minlength=p.size - 1)
    return np.concatenate([[0], np.cumsum(counts)]).astype(np.int32), j[keep], x[keep]


def test_remove_zero_valued_csr_follows_the_reference_loop():
    NA = -2147483648
    for seed in range(6):
        p, j, x = rand_csr(60, 40, 0.2, seed=seed, empty_rows=(3, 59))
        x = x.copy()
        rng = np.random.default_rng(seed)
        x[rng.random(x.size) < 0.2] = 0.0
        x[rng.random(x.size) < 0.05] = -0.0
        x[rng.random(x.size) < 0.1] = np.nan
        xl = rng.choice(np.array([0, 1, NA], dtype=np.int32), size=x.size)
        for rm in (False, True):
            for vals, logical, fn in ((x, False, O.remove_zero_valued_csr_numeric), (xl, True, O.remove_zero_valued_csr_logical)):
                got = fn(p, j, vals, rm)
                want = _drop_reference(p, j, vals, rm, logical)
                assert want is not None
                np.testing.assert_array_equal(got["indptr"], want[0])
                np.testing.assert_array_equal(got["indices"], want[1])
                np.testing.assert_array_equal(got["values"], want[2])        # (NaN == NaN for assert_array_equal)
    # nothing to remove: the INPUT objects themselves (misc.cpp:586-590)
    p, j, x = rand_csr(20, 10, 0.3, seed=9)
    r = O.remove_zero_valued_csr_numeric(p, j, x, True)
    assert r["indptr"] is p and r["indices"] is j and r["values"] is x
    xn = x.copy(); xn[0] = np.nan
    assert O.remove_zero_valued_csr_numeric(p, j, xn, False)["values"] is xn     # a NaN alone is nothing to remove ...
    assert O.remove_zero_valued_csr_numeric(p, j, xn, True)["values"].size == x.size - 1   # ... unless NAs are to leave
    # R logicals, remove_NAs: zeros trigger the rebuild but stay
    xl = np.ones(x.size, dtype=np.int32); xl[1] = 0
    r = O.remove_zero_valued_csr_logical(p, j, xl, True)
    assert r["values"] is not xl and np.array_equal(r["values"], xl) and np.array_equal(r["indptr"], p)


def test_check_valid_csr_matrix_messages_in_the_reference_order():
    p, j, _ = rand_csr(30, 12, 0.3, seed=2)
    assert O.check_valid_csr_matrix(p, j, 30, 12) == {}
    assert O.check_valid_csr_matrix(p, j, 30, int(j.max()))["err"] == "Matrix has invalid column indices."
    jn = j.copy(); jn[5] = -1
    assert O.check_valid_csr_matrix(p, jn, 30, 12)["err"] == "Matrix has negative indices."
    jn[5] = -2147483648                                              # NA among the indices is "negative" first
    assert O.check_valid_csr_matrix(p, jn, 30, 12)["err"] == "Matrix has negative indices."
    pn = p.copy(); pn[7] = -2147483648
    assert O.check_valid_csr_matrix(pn, j, 30, 12)["err"] == "Matrix has missing values in the index pointer."
    pd = p.copy(); pd[10] = pd[11] + 1
    assert O.check_valid_csr_matrix(pd, j, 30, 12)["err"] == "Matrix index pointer is not monotonicaly increasing."
    jb = j.copy(); jb[0] = 99                                        # both wrong: the index check comes first
    assert O.check_valid_csr_matrix(pd, jb, 30, 12)["err"] == "Matrix has invalid column indices."


def test_matmul_rowvec_by_csc_against_dense():
    """src/matmul.cpp:643-684: float32 row vector x CSC (the CSC arrays of Y = the CSR arrays of t(Y))"""
    p, j, x = rand_csr(40, 25, 0.3, seed=5, empty_rows=(7,))         # 40 columns of Y, 25 rows
    v = np.random.default_rng(6).normal(size=25).astype(np.float32)
    Yt = csr_to_dense(p, j, x, 25)                                   # t(Y): 40 x 25
    got = O.matmul_rowvec_by_csc(v, p, j, x)
    assert got.shape == (1, 40) and got.dtype == np.float32
    np.testing.assert_allclose(got[0], Yt @ v.astype(np.float64), rtol=2e-6, atol=2e-6)
    np.testing.assert_array_equal(got[0], O.matmul_csr_dvec_float32(p, j, x, v))       # the same float accumulation (matmul.cpp:403)
    gb = O.matmul_rowvec_by_cscbin(v, p, j)
    np.testing.assert_allclose(gb[0], (Yt != 0) @ v.astype(np.float64), rtol=2e-6, atol=2e-6)
    np.testing.assert_array_equal(gb, O.matmul_rowvec_by_csc(v, p, j, np.ones(j.size)))
