"""The reference's own R tests re-expressed against the Python mirror of the R glue (matrixextra_amd.matrices /
matmul / operators / slice) — same shapes, classes and assertions; expectations are dense numpy / scipy results
as the R tests use dense base-R results.  Everything computes on the GPU through the C-ABI."""
import numpy as np
import pytest
import scipy.sparse as sp

import matrixextra_amd as mx
from matrixextra_amd import matmul as MM

pytestmark = pytest.mark.gpu
NA_L = int(mx.NA_LOGICAL)


def rsparse(m, n, density, seed):
    rng = np.random.default_rng(seed)
    mask = rng.random((m, n)) < density
    vals = np.round(rng.normal(size=(m, n)), 3)
    vals[vals == 0] = 0.5
    A = sp.csr_matrix(np.where(mask, vals, 0.0))
    return mx.from_scipy(A), A


# ---- tests/testthat/test-matmul.R ------------------------------------------------------------------------------
def test_matmult_csr_dense(gpu):                                  # test-matmul.R:108-114
    A, As = rsparse(100, 50, 0.4, 1)
    B = np.random.default_rng(1).normal(size=(50, 20))
    A.Dimnames = [[f"r{k}" for k in range(100)], None]
    res = A @ mx.DenseMatrix(B, [None, [f"c{k}" for k in range(20)]])
    np.testing.assert_allclose(np.asarray(res), As.toarray() @ B, rtol=1e-10, atol=1e-12)
    assert res.Dimnames[0][3] == "r3" and res.Dimnames[1][5] == "c5"      # set_dimnames, R/matmul.R:148-167
    res32 = A @ mx.float32(B.astype(np.float32))
    assert isinstance(res32, mx.float32)
    np.testing.assert_allclose(res32.Data, As.toarray() @ B, rtol=1e-4, atol=1e-4)


def test_tcrossprod_csr_dense_and_dense_csr(gpu):                 # test-matmul.R:116-123, 53-96
    A, As = rsparse(100, 50, 0.4, 2)
    B = np.random.default_rng(2).normal(size=(20, 50))
    np.testing.assert_allclose(np.asarray(mx.tcrossprod(A, B)), As.toarray() @ B.T, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(np.asarray(mx.tcrossprod(B, A)), B @ As.toarray().T, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(mx.tcrossprod(mx.float32(B), A).Data, B @ As.toarray().T, rtol=1e-4, atol=1e-4)
    Bi = np.random.default_rng(3).integers(-3, 4, size=(20, 50))                  # mode(y) <- "double"
    np.testing.assert_allclose(np.asarray(mx.tcrossprod(A, Bi)), As.toarray() @ Bi.T, rtol=1e-10, atol=1e-12)


def test_dense_csc_and_crossprod(gpu):                            # test-matmul.R:12-51, 98-106
    As = sp.csc_matrix(rsparse(50, 30, 0.3, 4)[1])
    Y = mx.dgCMatrix(As.indptr, As.indices, As.data, As.shape)
    X = np.random.default_rng(4).normal(size=(7, 50))
    np.testing.assert_allclose(np.asarray(X @ Y), X @ As.toarray(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(np.asarray(mx.crossprod(X.T.copy(), Y)), X @ As.toarray(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose((mx.float32(X) @ Y).Data, X @ As.toarray(), rtol=1e-4, atol=1e-4)
    for shape in ((1, 50), (50, 1)):                              # 1-row / 1-col edge shapes (:24-26)
        Xs = np.random.default_rng(5).normal(size=(shape[0], 50))
        np.testing.assert_allclose(np.asarray(Xs @ Y), Xs @ As.toarray(), rtol=1e-10, atol=1e-12)
    # A1 (1 x 50) and B1 (50 x 1) of test-matmul.R:16-17,24-26: a sparse operand with ONE column, results 7 x 1 and 1 x 1
    B1s = sp.csc_matrix(rsparse(50, 1, 0.4, 7)[1])
    Y1 = mx.dgCMatrix(B1s.indptr, B1s.indices, B1s.data, B1s.shape)
    r71 = np.asarray(X @ Y1)
    assert r71.shape == (7, 1)
    np.testing.assert_allclose(r71, X @ B1s.toarray(), rtol=1e-10, atol=1e-12)
    X1 = np.random.default_rng(8).normal(size=(1, 50))
    r11 = np.asarray(X1 @ Y1)
    assert r11.shape == (1, 1)
    np.testing.assert_allclose(r11, X1 @ B1s.toarray(), rtol=1e-10, atol=1e-12)
    # tcrossprod with one-row operands on either side (test-matmul.R:57-58,63-66)
    A1, A1s = rsparse(1, 50, 0.4, 9)
    Bd = np.random.default_rng(9).normal(size=(20, 50))
    np.testing.assert_allclose(np.asarray(mx.tcrossprod(A1, Bd)), A1s.toarray() @ Bd.T, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(np.asarray(mx.tcrossprod(Bd, A1)), Bd @ A1s.toarray().T, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(np.asarray(mx.tcrossprod(Bd[:1], A1)), Bd[:1] @ A1s.toarray().T, rtol=1e-10, atol=1e-12)


def test_matmult_csr_vectors(gpu):                                # test-matmul.R:134-165
    A, As = rsparse(100, 50, 0.3, 6)
    rng = np.random.default_rng(6)
    v = rng.normal(size=50)
    r = A @ v
    assert r.shape == (100, 1)                                    # matrix(res, ncol=1)
    np.testing.assert_allclose(np.asarray(r)[:, 0], As.toarray() @ v, rtol=1e-10, atol=1e-12)
    vi = rng.integers(-4, 5, size=50).astype(np.int32)
    np.testing.assert_allclose(np.asarray(A @ vi)[:, 0], As.toarray() @ vi, rtol=1e-10, atol=1e-12)
    vb = rng.random(50) < 0.5
    np.testing.assert_allclose(np.asarray(A @ vb)[:, 0], As.toarray() @ vb, rtol=1e-10, atol=1e-12)
    vl = MM.RLogical(vb.astype(np.int32)); vl[3] = NA_L           # NA case (commented out upstream, :161-164)
    rl = np.asarray(A @ vl)[:, 0]
    touched = As.toarray()[:, 3] != 0
    assert np.isnan(rl[touched]).all() and not np.isnan(rl[~touched]).any()
    rf = A @ mx.float32(v.astype(np.float32))
    assert isinstance(rf, mx.float32) and rf.Data.shape == (100, 1)
    np.testing.assert_allclose(rf.Data[:, 0], As.toarray() @ v, rtol=1e-4, atol=1e-4)


# ---- tests/testthat/test-operators.R ---------------------------------------------------------------------------
def test_csr_csr_operators(gpu):                                  # test-operators.R:31-215
    M1, S1 = rsparse(100, 35, 0.4, 10)
    M2, S2 = rsparse(100, 35, 0.6, 11)
    E = mx.dgRMatrix(np.zeros(101, dtype=np.int32), [], [], (100, 35))
    snap = [a.copy() for a in (M1.p, M1.j, M1.x, M2.p, M2.j, M2.x)]
    for a, sa in ((M1, S1), (E, sp.csr_matrix((100, 35)))):
        for b, sb in ((M2, S2), (E, sp.csr_matrix((100, 35)))):
            r = a + b
            assert isinstance(r, mx.dgRMatrix)
            np.testing.assert_allclose(r.toarray(), (sa + sb).toarray(), rtol=0, atol=0)
            np.testing.assert_allclose((a - b).toarray(), (sa - sb).toarray(), rtol=0, atol=0)
            np.testing.assert_allclose((a * b).toarray(), sa.multiply(sb).toarray(), rtol=0, atol=0)
            ro, ra = a | b, a & b
            assert isinstance(ro, mx.lgRMatrix) and isinstance(ra, mx.lgRMatrix)
            np.testing.assert_array_equal(ro.toarray() != 0, (sa.toarray() != 0) | (sb.toarray() != 0))
            np.testing.assert_array_equal(ra.toarray() != 0, (sa.toarray() != 0) & (sb.toarray() != 0))
    # both orders, and an operand with itself (test-operators.R:36,58,80,102,118): the identical-pattern fast paths of
    # operators.cpp:104-132, :343-395 (aliased structure; X - X is the all-empty matrix)
    np.testing.assert_allclose((M2 + M1).toarray(), (S2 + S1).toarray(), rtol=0, atol=0)
    np.testing.assert_allclose((M2 - M1).toarray(), (S2 - S1).toarray(), rtol=0, atol=0)
    np.testing.assert_allclose((M2 * M1).toarray(), S2.multiply(S1).toarray(), rtol=0, atol=0)
    same_add, same_sub, same_mul = M1 + M1, M1 - M1, M1 * M1
    for r_ in (same_add, same_sub, same_mul):
        assert isinstance(r_, mx.dgRMatrix) and r_.Dim == (100, 35)
    np.testing.assert_allclose(same_add.toarray(), 2 * S1.toarray(), rtol=0, atol=0)
    np.testing.assert_allclose(same_sub.toarray(), np.zeros((100, 35)), rtol=0, atol=0)
    np.testing.assert_allclose(same_mul.toarray(), S1.toarray() ** 2, rtol=0, atol=0)
    for before, after in zip(snap, (M1.p, M1.j, M1.x, M2.p, M2.j, M2.x)):        # expect_unmodified (:21-27)
        np.testing.assert_array_equal(before, after)
    # unsorted operand: the glue sorts a copy, the input stays as it was (R/operators.R:742-754)
    U = M1.copy()
    for r in range(100):
        s, e = U.p[r], U.p[r + 1]
        U.j[s:e] = U.j[s:e][::-1]; U.x[s:e] = U.x[s:e][::-1]
    ju = U.j.copy()
    np.testing.assert_allclose((U + M2).toarray(), (S1 + S2).toarray(), rtol=0, atol=0)
    np.testing.assert_array_equal(U.j, ju)
    # mixed classes (:314-371): binary / logical operands are expanded by as.csr.matrix
    N = mx.ngRMatrix(M2.p, M2.j, None, M2.Dim)
    np.testing.assert_allclose((M1 + N).toarray(), S1.toarray() + (S2.toarray() != 0), rtol=0, atol=0)
    np.testing.assert_allclose((M1 * N).toarray(), S1.toarray() * (S2.toarray() != 0), rtol=0, atol=0)


# ---- tests/testthat/test-slice.R -------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def slice_fixture():
    A, As = rsparse(1000, 500, 0.1, 20)                           # test-slice.R:6-16
    A.Dimnames = [[f"r{k}" for k in range(1, 1001)], [f"c{k}" for k in range(1, 501)]]
    return A, As.tocsr()


def _chk(res, expect, cls):
    assert isinstance(res, cls) and res.Dim == expect.shape
    np.testing.assert_allclose(res.toarray(), expect.toarray() if sp.issparse(expect) else expect, rtol=0, atol=0)


def test_slice_rows(gpu, slice_fixture):                          # test-slice.R:61-83, 120-132, 243-292
    A, As = slice_fixture
    _chk(mx.subset_csr(A, [1, 3, 3, 1000]), As[[0, 2, 2, 999]], mx.dgRMatrix)
    _chk(mx.subset_csr(A, np.arange(10, 21)), As[9:20], mx.dgRMatrix)                 # contiguous
    _chk(mx.subset_csr(A, np.arange(20, 9, -1)), As[np.arange(19, 8, -1)], mx.dgRMatrix)  # reversed sequence
    _chk(mx.subset_csr(A, [-1, -1000]), As[1:999], mx.dgRMatrix)                      # negative = exclusion
    _chk(mx.subset_csr(A, [1000] * 5 + [1]), As[[999] * 5 + [0]], mx.dgRMatrix)       # nrow+1 repeated-row cases
    r = mx.subset_csr(A, ["r5", "r2"])
    _chk(r, As[[4, 1]], mx.dgRMatrix)
    assert r.Dimnames[0] == ["r5", "r2"]
    mask = np.zeros(1000, dtype=bool); mask[[3, 500, 999]] = True                     # logical masks (:134-158)
    _chk(mx.subset_csr(A, mask), As[[3, 500, 999]], mx.dgRMatrix)
    with pytest.raises(mx.MatrixExtraError):                                          # out of bounds (:127-128)
        mx.subset_csr(A, [1, 1001])
    _chk(A[[5, 2, 2]], As[[5, 2, 2]], mx.dgRMatrix)                                   # python-style 0-based


def test_slice_rows_and_columns(gpu, slice_fixture):              # test-slice.R:18-83, 243-280
    A, As = slice_fixture
    rows = [7, 3, 3, 999, 1]
    r0 = np.array(rows) - 1
    _chk(mx.subset_csr(A, rows, np.arange(100, 201)), As[r0][:, 99:200], mx.dgRMatrix)            # j contiguous
    _chk(mx.subset_csr(A, rows, np.arange(200, 99, -1)), As[r0][:, np.arange(199, 98, -1)], mx.dgRMatrix)
    cols = [400, 2, 2, 77, 500, 1]
    res = mx.subset_csr(A, rows, cols)                                                            # arbitrary j
    _chk(res, As[r0][:, np.array(cols) - 1], mx.dgRMatrix)
    assert res.Dimnames == [["r7", "r3", "r3", "r999", "r1"], ["c400", "c2", "c2", "c77", "c500", "c1"]]
    _chk(mx.subset_csr(A, None, [3, 1]), As[:, [2, 0]], mx.dgRMatrix)                             # all rows
    _chk(mx.subset_csr(A, np.arange(1000, 0, -1), np.arange(500, 0, -1)), As[::-1][:, ::-1], mx.dgRMatrix)
    e = mx.subset_csr(A, [5, 6], np.zeros(0, dtype=np.int32))                                     # empty (:85-104)
    assert e.Dim == (2, 0) and e.p.tolist() == [0, 0, 0]
    # binary and logical classes keep their class
    N = mx.ngRMatrix(A.p, A.j, None, A.Dim)
    _chk(mx.subset_csr(N, rows, cols), (As[r0][:, np.array(cols) - 1] != 0).astype(float), mx.ngRMatrix)
    L = mx.lgRMatrix(A.p, A.j, (A.x > 0).astype(np.int32), A.Dim)
    Ls = sp.csr_matrix(((A.x > 0).astype(float), A.j, A.p), shape=A.Dim)
    _chk(mx.subset_csr(L, rows, np.arange(100, 201)), Ls[r0][:, 99:200], mx.lgRMatrix)
    _chk(A[np.array([5, 2]), 10:20], As[[5, 2]][:, 10:20], mx.dgRMatrix)
    # the literal index vectors of test-slice.R:65-81 ("non sequential", "repeated"), by position and by name
    for ri, ci in (([5, 2, 1, 7, 4], [5, 2, 1, 7, 4, 10, 100]), ([2, 2, 2, 1, 1, 3], [3, 3, 4, 4, 1, 1, 1]),
                   ([5, 2, 1, 7, 4, 1, 5], [5, 2, 1, 7, 4, 1, 10, 100, 5])):
        want = As[np.array(ri) - 1][:, np.array(ci) - 1]
        _chk(mx.subset_csr(A, ri, ci), want, mx.dgRMatrix)
        _chk(mx.subset_csr(A, [f"r{k}" for k in ri], [f"c{k}" for k in ci]), want, mx.dgRMatrix)
    # "subset empty" (test-slice.R:85-104): no rows, with and without a column selection
    e0 = mx.subset_csr(A, np.zeros(0, dtype=np.int32), [3, 3, 4, 4, 1, 1, 1])
    assert e0.Dim == (0, 7) and e0.p.tolist() == [0] and e0.j.size == 0
    e1 = mx.subset_csr(A, np.zeros(0, dtype=np.int32), np.arange(3, 11))
    assert e1.Dim == (0, 8) and e1.p.tolist() == [0]
    e2 = mx.subset_csr(A, np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int32))
    assert e2.Dim == (0, 0) and e2.p.tolist() == [0]
    e3 = mx.subset_csr(A, [2, 2, 2, 1, 1, 3], np.zeros(0, dtype=np.int32))
    assert e3.Dim == (6, 0) and e3.p.tolist() == [0] * 7


def test_sort_sparse_indices_kat(gpu):                            # tests/testthat/test-utilities.R:32-49
    X = mx.dgRMatrix([0, 1, 4, 5, 6], [4, 2, 1, 4, 1, 0], [-0.91, 0.14, -0.12, -0.12, 1.1, 0.66], (4, 5))
    indices = X.j
    Xn = mx.sort_sparse_indices(X, copy=True)
    assert Xn.j.tolist() == [4, 1, 2, 4, 1, 0] and indices.tolist() == [4, 2, 1, 4, 1, 0]
    mx.sort_sparse_indices(X, copy=False)
    assert X.j.tolist() == [4, 1, 2, 4, 1, 0] and X.j is indices


def test_csr_by_vector_operators(gpu):                            # test-operators.R:373-893 (vector / same-shape matrix operands)
    """`X * v`, `v * X`, `X / v`, `X ^ v`, `X %% v`, `X %/% v`, `X & v` keep X's pattern and follow R's recycling;
    the reference's NA / dense routes raise instead of silently doing something else."""
    import scipy.sparse as sp
    from matrixextra_amd import matrices as M
    rng = np.random.default_rng(5)
    A = sp.random(60, 17, 0.25, format="csr", random_state=3); A.sort_indices()
    A.data = (A.data * 8 - 4).round(2); A.data[A.data == 0] = 1.25
    X = mx.dgRMatrix(A.indptr, A.indices, A.data, A.shape, [[f"r{k}" for k in range(60)], None])
    D = A.toarray(); r, c = A.nonzero()
    for ln in (60, 60 * 17, 20, 1, 7):
        v = (rng.uniform(0.5, 3.0, size=ln) * rng.choice([-1.0, 1.0], size=ln)).round(2)
        full = np.resize(v, 60 * 17).reshape(17, 60).T
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for res, want in [(X * v, D * full), (v * X, D * full), (X / v, D / full)]:
                assert isinstance(res, mx.dgRMatrix) and res.Dim == (60, 17) and res.Dimnames[0][3] == "r3"
                assert res.p is X.p and res.j is X.j                # values-only transform (R/operators.R:1134)
                np.testing.assert_array_equal(res.x, want[r, c])
            np.testing.assert_allclose((X % v).x, np.mod(D, full)[r, c], rtol=1e-12, atol=1e-14)
            np.testing.assert_allclose((X // v).x, np.floor_divide(D, full)[r, c])
        vp = np.abs(v) + 1.0                                        # ^ with a negative exponent takes the NA route
        fullp = np.resize(vp, 60 * 17).reshape(17, 60).T
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            with np.errstate(invalid="ignore"):
                np.testing.assert_allclose((X ** vp).x, np.power(D, fullp)[r, c], rtol=1e-13, equal_nan=True)
    same_shape = rng.uniform(1, 2, size=(60, 17))
    np.testing.assert_array_equal((X * same_shape).x, (D * same_shape)[r, c])
    L = mx.lgRMatrix(A.indptr, A.indices, (A.data > 0).astype(np.int32), A.shape)
    vl = rng.integers(0, 2, size=60).astype(np.int32)
    res = L & vl
    assert isinstance(res, mx.lgRMatrix)
    np.testing.assert_array_equal(res.x, (L.x != 0) & (vl[r] != 0))
    assert (L & np.array([0], dtype=np.int32)).j.size == 0          # R/operators.R:1039-1046
    # the structure-changing NA route (R/operators.R:981-1131 -> multiply_csr_by_dvec_with_NAs, round 3): the pattern grows
    # by every cell the vector makes special and the result is what R's dense arithmetic gives
    Dm = X.toarray()
    with np.errstate(all="ignore"):
        cases = ((X * np.array([1.0, np.nan] * 30), Dm * np.array([1.0, np.nan] * 30)[:, None]),
                 (X / np.zeros(60), Dm / np.zeros(60)[:, None]),
                 (X * np.full(60, np.inf), Dm * np.inf),
                 (X ** (-np.ones(60)), np.where(Dm == 0, np.inf, np.power(np.where(Dm == 0, 1.0, Dm), -1.0))))
    for res, exp in cases:
        assert isinstance(res, mx.dgRMatrix) and res.Dim == X.Dim
        got = res.toarray()
        np.testing.assert_array_equal(np.isnan(got), np.isnan(exp))
        np.testing.assert_allclose(got[~np.isnan(exp)], exp[~np.isnan(exp)], rtol=1e-13)
        assert res.p[-1] > X.p[-1]
    # routes that stay on the reference's CPU code (it goes through a CsparseMatrix there)
    for bad in (lambda: X * np.full(60, np.nan), lambda: X * np.array([np.inf]), lambda: np.ones(60) / X):
        with pytest.raises(M.MatrixExtraError):
            bad()
    with pytest.raises(M.MatrixExtraError, match="more entries than matrix"):
        X * np.ones(60 * 17 + 1)
    assert (X * np.zeros(0)).size == 0                              # R/operators.R:961-966


def to_dense(X):
    """as.matrix(X): stored entries summed into a dense array (NaN where a stored value is missing)"""
    out = np.zeros(X.Dim)
    for r in range(X.Dim[0]):
        for k in range(X.p[r], X.p[r + 1]):
            out[r, X.j[k]] += X.x[k]
    return out


def test_removing_zeros(gpu):                                     # tests/testthat/test-utilities.R:6-30
    rng = np.random.default_rng(1)
    D = np.round(rng.normal(size=(10, 5)), 2) * (rng.random((10, 5)) < 0.75)
    X = mx.as_csr_matrix(sp.csr_matrix(D))
    x = X.x.copy()
    x[4:8] = 0.0
    X = mx.dgRMatrix(X.p, X.j, x, X.Dim)
    dense = to_dense(X)
    Xr = mx.remove_sparse_zeros(X)
    np.testing.assert_array_equal(to_dense(Xr), dense)
    assert (Xr.x == 0).sum() == 0 and Xr.j.size == X.j.size - 4
    x[7:10] = np.nan
    X = mx.dgRMatrix(X.p, X.j, x, X.Dim)
    Xr = mx.remove_sparse_zeros(X, na_rm=True)
    want = to_dense(X)
    want[np.isnan(want)] = 0
    np.testing.assert_array_equal(to_dense(Xr), want)
    assert not np.isnan(Xr.x).any()
    assert np.isnan(mx.remove_sparse_zeros(X).x).sum() == 3        # without na.rm the missing values stay
    P = mx.ngRMatrix(X.p, X.j, None, X.Dim)
    assert mx.remove_sparse_zeros(P) is P


def test_checking_indices(gpu):                                   # tests/testthat/test-utilities.R:51-80
    def fresh(p=(0, 1, 4, 5, 6), j=(4, 2, 1, 4, 1, 0)):
        return mx.dgRMatrix(list(p), list(j), [-0.91, 0.14, -0.12, -0.12, 1.1, 0.66], (4, 5))
    X = fresh()
    Xc = mx.check_sparse_matrix(X)
    assert Xc.j.tolist() == [4, 1, 2, 4, 1, 0] and X.j.tolist() == [4, 2, 1, 4, 1, 0]
    for p in ((0, 1, 4, 5, 100), (0, 5, 4, 5, 6), (0, 1, -2147483648, 5, 6), (0, -1, 4, 5, 6)):
        with pytest.raises(mx.MatrixExtraError):
            mx.check_sparse_matrix(fresh(p=p))
    with pytest.raises(mx.MatrixExtraError, match="invalid column indices"):
        mx.check_sparse_matrix(fresh(j=(4, 1, 2, 4, 1, 10)))
    with pytest.raises(mx.MatrixExtraError, match="negative indices"):
        mx.check_sparse_matrix(fresh(j=(4, 1, 2, 4, -1, 0)))
    with pytest.raises(mx.MatrixExtraError, match="not monotonicaly increasing"):
        mx.check_sparse_matrix(fresh(p=(0, 5, 4, 5, 6)))
    mx.check_sparse_matrix(fresh())
    Z = mx.dgRMatrix([0, 2, 3], [3, 0, 1], [0.0, 2.0, 0.0], (2, 4))  # zeros leave first, then the sort (in place: nnz changed)
    Zc = mx.check_sparse_matrix(Z)
    assert Zc.j.tolist() == [0] and Zc.x.tolist() == [2.0] and Zc.p.tolist() == [0, 1, 1]


def test_float32_vector_times_csc(gpu):                           # R/matmul.R:243-259 (float32 row vector %*% CsparseMatrix)
    As = rsparse(30, 12, 0.3, seed=4)[1].tocsc()
    Y = mx.dgCMatrix(As.indptr, As.indices, As.data, As.shape)
    v = np.random.default_rng(5).normal(size=30).astype(np.float32)
    res = mx.float32(v) @ Y
    assert isinstance(res, mx.float32) and res.Data.shape == (1, 12)
    np.testing.assert_allclose(res.Data[0], v.astype(np.float64) @ As.toarray(), rtol=1e-5, atol=1e-5)
    with pytest.raises(mx.MatrixExtraError, match="dimensions do not match"):
        mx.float32(v[:7]) @ Y
