"""The small-call path of the exports (csrc/api.hip SmallStage: operands + result within ~2 MiB (SpMM 6, merges 3) go up in one pinned block,
the SAME kernels run, the result comes down in one copy, one synchronisation): the reference's own test sizes
(tests/testthat/test-matmul.R:108-114: 100 x 50 density .4 times 50 x 20; test-slice.R:6-16: 1000 x 500) take it —
mx_get_option("small_calls") counts — and return what the regular path returns: bit for bit for every structure / value
copy and for the products (same kernels, same order of additions), checked against the oracle as everywhere else."""
import ctypes as C

import numpy as np
import pytest

from conftest import rand_csr
from matrixextra_amd import _lib, exports as G
from oracle import oracle as O

pytestmark = pytest.mark.gpu
NA = -2147483648


def _count(lib):
    v = C.c_int64()
    _lib.check(lib.mx_get_option(b"small_calls", C.byref(v)))
    return v.value


def _same(g, o):
    for k in ("indptr", "indices", "values"):
        a, b = np.asarray(g[k]), np.asarray(o[k])
        assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes(), k


@pytest.mark.parametrize("m,K,dens,n", [(100, 50, 0.4, 20), (1000, 500, 0.02, 8), (37, 3000, 0.05, 3), (1, 7, 1.0, 1)])
def test_small_products_take_the_small_path(gpu, m, K, dens, n):
    lib = gpu.load()
    p, j, x = rand_csr(m, K, dens, seed=m + n, sorted_cols=False, empty_rows=(0,) if m > 3 else ())
    rng = np.random.default_rng(n)
    Y = np.asfortranarray(rng.normal(size=(n, K)))
    c0 = _count(lib)
    got = G.tcrossprod_csr_dense_numeric(p, j, x, Y, 1)
    assert _count(lib) == c0 + 1
    # rows of up to 32 entries (mean): the row-wave kernel — the storage-order FMA chain bit for bit; longer rows go to the
    # row-split kernel even in tiny products (round 5: the row-wave kernel walks a row one dependent read after the other),
    # whose narrow lane groups regroup the sum
    bitwise = j.size <= 32 * m

    def same(a, b, tol):
        if bitwise:
            np.testing.assert_array_equal(a, b)
        else:
            np.testing.assert_allclose(a, b, rtol=tol, atol=tol * float(np.abs(b).max()))
    same(got, O.tcrossprod_csr_dense_numeric(p, j, x, Y, 1, use_fma=True), 1e-12)
    Y32 = Y.astype(np.float32)
    same(G.tcrossprod_csr_dense_float32(p, j, x, Y32, 1), O.tcrossprod_csr_dense_float32(p, j, x, Y32, 1, use_fma=True), 1e-5)
    X = np.asfortranarray(rng.normal(size=(n, K)))              # dense (n x K) %*% CSC whose columns are our rows
    same(G.matmul_dense_csc_numeric(X, p, j, x, 1), O.matmul_dense_csc_numeric(X, p, j, x, 1, use_fma=True), 1e-12)
    same(G.tcrossprod_dense_csr_numeric(X, p, j, x, 1, K), O.tcrossprod_dense_csr_numeric(X, p, j, x, 1, K, use_fma=True), 1e-12)
    assert _count(lib) == c0 + 4
    # SpMV, the four kinds
    v = rng.normal(size=K)
    vi = rng.integers(-4, 5, size=K).astype(np.int32)
    vi[K // 2] = NA
    vl = rng.choice(np.array([0, 1, NA], dtype=np.int32), size=K)
    for gf, of, vec, tol in ((G.matmul_csr_dvec_numeric, O.matmul_csr_dvec_numeric, v, 1e-12), (G.matmul_csr_dvec_integer, O.matmul_csr_dvec_integer, vi, 1e-12),
                             (G.matmul_csr_dvec_logical, O.matmul_csr_dvec_logical, vl, 1e-12),
                             (G.matmul_csr_dvec_float32, O.matmul_csr_dvec_float32, v.astype(np.float32), 1e-5)):
        g, o = gf(p, j, x, vec, 1), of(p, j, x, vec)
        assert g.dtype == o.dtype and np.array_equal(np.isnan(g), np.isnan(o))
        np.testing.assert_allclose(g[~np.isnan(g)], o[~np.isnan(o)], rtol=tol, atol=tol)
    assert _count(lib) == c0 + 8


@pytest.mark.parametrize("m,K,d1,d2", [(100, 50, 0.4, 0.3), (1000, 500, 0.01, 0.012), (64, 64, 0.0, 0.2), (5, 2000, 0.5, 0.5)])
def test_small_merges_take_the_small_path(gpu, m, K, d1, d2):
    lib = gpu.load()
    p1, j1, x1 = rand_csr(m, K, d1, seed=3 * m, empty_rows=(1,) if m > 3 else ())
    p2, j2, x2 = rand_csr(m, K, d2, seed=5 * m, empty_rows=(2,) if m > 3 else ())
    if x1.size > 3:
        x1[1], x1[2] = np.nan, np.inf
    rng = np.random.default_rng(m)
    l1 = rng.choice(np.array([0, 1, NA], dtype=np.int32), size=j1.size)
    l2 = rng.choice(np.array([0, 1, NA], dtype=np.int32), size=j2.size)
    c0 = _count(lib)
    _same(G.multiply_csr_elemwise(p1, p2, j1, j2, x1, x2), O.multiply_csr_elemwise(p1, p2, j1, j2, x1, x2))
    for sub in (False, True):
        _same(G.add_csr_elemwise(p1, p2, j1, j2, x1, x2, sub), O.add_csr_elemwise(p1, p2, j1, j2, x1, x2, sub))
    _same(G.logicaland_csr_elemwise(p1, p2, j1, j2, l1, l2), O.logicaland_csr_elemwise(p1, p2, j1, j2, l1, l2))
    for xor in (False, True):
        _same(G.logicalor_csr_elemwise(p1, p2, j1, j2, l1, l2, xor), O.logicalor_csr_elemwise(p1, p2, j1, j2, l1, l2, xor))
    assert _count(lib) == c0 + 6


def test_small_row_gathers_take_the_small_path_or_fall_back(gpu):
    lib = gpu.load()
    p, j, x = rand_csr(1000, 500, 0.02, seed=9, empty_rows=(10, 11))
    rng = np.random.default_rng(1)
    xl = rng.choice(np.array([0, 1, NA], dtype=np.int32), size=j.size)
    c0 = _count(lib)
    n_small = 0
    for rows in (rng.integers(0, 1000, size=300).astype(np.int32), np.array([10, 11, 10], dtype=np.int32), np.array([5], dtype=np.int32)):
        _same(G.copy_csr_rows_numeric(p, j, x, rows), O.copy_csr_rows_numeric(p, j, x, rows))
        _same(G.copy_csr_rows_logical(p, j, xl, rows), O.copy_csr_rows_logical(p, j, xl, rows))
        _same(G.copy_csr_rows_binary(p, j, rows), O.copy_csr_rows_binary(p, j, rows))
        n_small += 3
    assert _count(lib) == c0 + n_small
    # a selection whose RESULT does not fit the block (one long row taken 2000 times): the regular path, same answer
    pl = np.array([0, 900, 905], dtype=np.int32)
    jl = np.concatenate([np.arange(900), np.arange(5)]).astype(np.int32)
    xv = rng.normal(size=905)
    rows = np.zeros(2000, dtype=np.int32)
    c1 = _count(lib)
    _same(G.copy_csr_rows_numeric(pl, jl, xv, rows), O.copy_csr_rows_numeric(pl, jl, xv, rows))
    assert _count(lib) == c1


def test_operands_above_the_limit_take_the_regular_path(gpu):
    lib = gpu.load()
    p, j, x = rand_csr(6000, 4000, 0.009, seed=2)            # ~216k entries: 2.6 MB of CSR (SpMV's limit: 2 MiB)
    assert 12 * j.size > (2 << 20)
    v = np.random.default_rng(0).normal(size=4000)
    c0 = _count(lib)
    g = G.matmul_csr_dvec_numeric(p, j, x, v, 1)
    assert _count(lib) == c0
    np.testing.assert_allclose(g, O.matmul_csr_dvec_numeric(p, j, x, v), rtol=1e-12, atol=1e-12)
    # just under the limit: the block's offsets are 256-byte aligned, the last result element must still arrive
    p2, j2, x2 = rand_csr(5000, 3000, 0.011, seed=4)
    assert (2 << 20) - (96 << 10) < 12 * j2.size + 4 * 5001 + 8 * 3000 + 8 * 5000 < (2 << 20)
    g2 = G.matmul_csr_dvec_numeric(p2, j2, x2, v[:3000], 1)
    assert _count(lib) == c0 + 1
    np.testing.assert_allclose(g2, O.matmul_csr_dvec_numeric(p2, j2, x2, v[:3000]), rtol=1e-12, atol=1e-12)
