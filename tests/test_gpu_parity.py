"""GPU parity: the HIP path (through the C-ABI) against the CPU restatement on the same seeded inputs.

Bars (BASELINE.json north_star): bit-exact for every integer / index output and for merge / gather
values; <= 1e-6 relative for f64 SpMM / SpMV (we assert 1e-12, the kernels keep the reference's
summation order), 1e-5 for f32.  NaN payloads are compared as NaN-ness only.
"""
import numpy as np
import pytest

from conftest import rand_csr
from matrixextra_amd import exports as G
from matrixextra_amd import _lib, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
NA = int(O.NA_INTEGER)
F64_RTOL = 1e-12      # well inside the 1e-6 bar
F32_RTOL = 1e-5


def assert_list_equal(g, o, values_exact=True):
    assert g["indptr"].dtype == np.int32 and g["indices"].dtype == np.int32
    np.testing.assert_array_equal(g["indptr"], o["indptr"])
    np.testing.assert_array_equal(g["indices"], o["indices"])
    assert g["values"].dtype == o["values"].dtype and g["values"].shape == o["values"].shape
    if g["values"].dtype.kind == "f":
        nan_g, nan_o = np.isnan(g["values"]), np.isnan(o["values"])
        np.testing.assert_array_equal(nan_g, nan_o)
        # bit-exact incl. the sign of zero for everything that is not NaN
        np.testing.assert_array_equal(g["values"][~nan_g].view(np.int64), o["values"][~nan_o].view(np.int64))
    else:
        np.testing.assert_array_equal(g["values"], o["values"])


# ----------------------------------------------------------------------------- SpMM
SPMM_SHAPES = [
    (100, 50, 20, 0.4),     # test-matmul.R:108-114 "matmult CSR-dense" shape
    (1, 50, 7, 0.5), (50, 1, 3, 1.0), (33, 70, 1, 0.3),   # 1-row / 1-col edge shapes (test-matmul.R:24-26)
    (257, 300, 128, 0.05),  # n = 128 (headline slab width), partial row tile
    (64, 90, 130, 0.1),     # n just past one 128-column slab
    (31, 40, 256, 0.2), (40, 64, 65, 0.2), (10, 12, 300, 0.5),
    (5, 3, 9, 0.7),         # n > m  (the reference's scratch overflow case, not copied)
]


@pytest.mark.parametrize("m,K,n,dens", SPMM_SHAPES)
def test_tcrossprod_csr_dense_numeric(gpu, m, K, n, dens):
    p, j, x = rand_csr(m, K, dens, seed=m + 7 * n, sorted_cols=False, empty_rows=(0,) if m > 3 else ())
    Y = np.asfortranarray(np.random.default_rng(n).normal(size=(n, K)))
    got = G.tcrossprod_csr_dense_numeric(p, j, x, Y, 4)
    ref = O.tcrossprod_csr_dense_numeric(p, j, x, Y, 2, use_fma=False)
    assert got.shape == (m, n) and got.flags.f_contiguous and got.dtype == np.float64
    np.testing.assert_allclose(got, ref, rtol=F64_RTOL, atol=1e-13)
    # same summation order + FMA on both sides -> bitwise
    np.testing.assert_array_equal(got, O.tcrossprod_csr_dense_numeric(p, j, x, Y, 1, use_fma=True))


@pytest.mark.parametrize("m,K,n,dens", SPMM_SHAPES)
def test_tcrossprod_csr_dense_float32(gpu, m, K, n, dens):
    p, j, x = rand_csr(m, K, dens, seed=m + 3 * n, sorted_cols=False)
    Y = np.asfortranarray(np.random.default_rng(n).normal(size=(n, K)).astype(np.float32))
    got = G.tcrossprod_csr_dense_float32(p, j, x, Y, 1)
    ref = O.tcrossprod_csr_dense_float32(p, j, x, Y, 1, use_fma=False)
    assert got.dtype == np.float32 and got.flags.f_contiguous
    np.testing.assert_allclose(got, ref, rtol=F32_RTOL, atol=1e-5)
    np.testing.assert_array_equal(got, O.tcrossprod_csr_dense_float32(p, j, x, Y, 1, use_fma=True))


@pytest.mark.parametrize("m,K,n,dens", SPMM_SHAPES[:7])
def test_dense_csc_and_dense_csr(gpu, m, K, n, dens):
    # (p, j, x) read as CSC of Y (K x m) for matmul_dense_csc and as CSR of Y (m x K) for tcrossprod_dense_csr
    p, j, x = rand_csr(m, K, dens, seed=2 * m + n, sorted_cols=False)
    rng = np.random.default_rng(m)
    X = np.asfortranarray(rng.normal(size=(n, K)))
    for gf, of in [(G.matmul_dense_csc_numeric, O.matmul_dense_csc_numeric),
                   (lambda *a: G.tcrossprod_dense_csr_numeric(*a, K), lambda *a: O.tcrossprod_dense_csr_numeric(*a, K))]:
        got, ref = gf(X, p, j, x, 1), of(X, p, j, x, 1)
        assert got.shape == (n, m) and got.flags.f_contiguous
        np.testing.assert_allclose(got, ref, rtol=F64_RTOL, atol=1e-13)
    X32 = np.asfortranarray(X.astype(np.float32))
    np.testing.assert_allclose(G.matmul_dense_csc_float32(X32, p, j, x, 1), O.matmul_dense_csc_float32(X32, p, j, x, 1),
                               rtol=F32_RTOL, atol=1e-5)
    np.testing.assert_allclose(G.tcrossprod_dense_csr_float32(X32, p, j, x, 1, K),
                               O.tcrossprod_dense_csr_float32(X32, p, j, x, 1, K), rtol=F32_RTOL, atol=1e-5)


def test_spmm_special_cases(gpu):
    # duplicates accumulate; unsorted columns; all-empty matrix; NaN / Inf propagate
    p = np.array([0, 3, 3, 5], dtype=np.int32)
    j = np.array([1, 1, 0, 1, 0], dtype=np.int32)
    x = np.array([2.0, 3.0, 1.0, np.inf, np.nan])
    Y = np.asfortranarray(np.array([[1.0, 10.0], [2.0, 20.0], [0.0, 0.5]]))       # n=3, K=2
    got, ref = G.tcrossprod_csr_dense_numeric(p, j, x, Y), O.tcrossprod_csr_dense_numeric(p, j, x, Y)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(ref))
    np.testing.assert_array_equal(got[~np.isnan(got)], ref[~np.isnan(ref)])
    z = G.tcrossprod_csr_dense_numeric(np.zeros(5, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros(0), Y)
    assert z.shape == (4, 3) and not z.any()
    # long rows (> one wavefront of entries) and many rows
    pl, jl, xl = rand_csr(70, 400, 0.6, seed=99)
    Yl = np.asfortranarray(np.random.default_rng(1).normal(size=(128, 400)))
    np.testing.assert_allclose(G.tcrossprod_csr_dense_numeric(pl, jl, xl, Yl), O.tcrossprod_csr_dense_numeric(pl, jl, xl, Yl),
                               rtol=F64_RTOL, atol=1e-12)


def test_spmm_cfg1_shape(gpu):
    # BASELINE.json configs[0]: 10k x 10k, 16/row, dense 10k x 64 f64
    p, j, x = synth.csr_fixed(10_000, 10_000, 16)
    B = synth.dense_normal(10_000, 64)
    Y = np.asfortranarray(B.T)
    got = G.tcrossprod_csr_dense_numeric(p, j, x, Y, 8)
    ref = O.tcrossprod_csr_dense_numeric(p, j, x, Y, O.max_threads())
    np.testing.assert_allclose(got, ref, rtol=F64_RTOL, atol=1e-12)


# ----------------------------------------------------------------------------- SpMV
@pytest.mark.parametrize("m,K,dens", [(60, 25, 0.3), (1, 10, 0.9), (500, 700, 0.02), (300, 40, 1.0), (64, 2000, 0.2)])
def test_spmv_all_kinds(gpu, m, K, dens):
    p, j, x = rand_csr(m, K, dens, seed=m + K, sorted_cols=False, empty_rows=(0,) if m > 2 else ())
    rng = np.random.default_rng(K)
    y = rng.normal(size=K)
    np.testing.assert_allclose(G.matmul_csr_dvec_numeric(p, j, x, y), O.matmul_csr_dvec_numeric(p, j, x, y),
                               rtol=1e-11, atol=1e-12)
    yi = rng.integers(-9, 9, size=K).astype(np.int32)
    yl = rng.integers(0, 2, size=K).astype(np.int32)
    for k in rng.integers(0, K, size=max(1, K // 10)):
        yi[k] = NA
        yl[k] = NA
    for gf, of, v in [(G.matmul_csr_dvec_integer, O.matmul_csr_dvec_integer, yi),
                      (G.matmul_csr_dvec_logical, O.matmul_csr_dvec_logical, yl)]:
        got, ref = gf(p, j, x, v), of(p, j, x, v)
        np.testing.assert_array_equal(np.isnan(got), np.isnan(ref))      # NA_REAL where an NA entry is touched
        np.testing.assert_allclose(got[~np.isnan(got)], ref[~np.isnan(ref)], rtol=1e-11, atol=1e-12)
        na_rows = np.isnan(got)
        if na_rows.any():                                                  # the NA payload (low word 1954) is kept
            assert (got[na_rows].view(np.uint64) & 0xFFFFFFFF == 1954).all()
    yf = y.astype(np.float32)
    got, ref = G.matmul_csr_dvec_float32(p, j, x, yf), O.matmul_csr_dvec_float32(p, j, x, yf)
    assert got.dtype == np.float32
    np.testing.assert_allclose(got, ref, rtol=F32_RTOL, atol=1e-5)


# ----------------------------------------------------------------------------- merges
MERGE_CASES = [(100, 35, 0.4, 0.6), (100, 35, 0.05, 0.9), (50, 20, 0.0, 0.5), (50, 20, 0.5, 0.0),
               (3, 300, 0.9, 0.9), (1000, 64, 0.1, 0.1), (17, 1, 0.5, 0.5), (40, 200, 0.7, 0.02)]


@pytest.mark.parametrize("m,K,d1,d2", MERGE_CASES)
def test_add_sub_mul(gpu, m, K, d1, d2):
    p1, j1, x1 = rand_csr(m, K, d1, seed=11 + m, empty_rows=(1,) if m > 2 else ())
    p2, j2, x2 = rand_csr(m, K, d2, seed=12 + K, empty_rows=(1, 2) if m > 3 else ())
    keep1, keep2 = (x1.copy(), j1.copy(), p1.copy()), (x2.copy(), j2.copy(), p2.copy())
    for sub in (False, True):
        assert_list_equal(G.add_csr_elemwise(p1, p2, j1, j2, x1, x2, sub), O.add_csr_elemwise(p1, p2, j1, j2, x1, x2, sub))
    assert_list_equal(G.multiply_csr_elemwise(p1, p2, j1, j2, x1, x2), O.multiply_csr_elemwise(p1, p2, j1, j2, x1, x2))
    # inputs are never modified (test-operators.R:21-27 expect_unmodified)
    for a, b in zip(keep1 + keep2, (x1, j1, p1, x2, j2, p2)):
        np.testing.assert_array_equal(a, b)


def test_merge_special_values_and_fast_paths(gpu):
    p = np.array([0, 2, 2, 4], dtype=np.int32)
    j = np.array([1, 3, 0, 2], dtype=np.int32)
    x = np.array([1.5, -2.0, np.inf, 0.0])
    x2 = np.array([-1.5, np.nan, -np.inf, 0.0])
    # general path (different buffers): cancellation keeps an explicit 0; Inf-Inf = NaN; -0 handling
    for sub in (False, True):
        assert_list_equal(G.add_csr_elemwise(p, p.copy(), j, j.copy(), x, x2, sub),
                          O.add_csr_elemwise(p, p.copy(), j, j.copy(), x, x2, sub))
    assert_list_equal(G.multiply_csr_elemwise(p, p.copy(), j, j.copy(), x, x2),
                      O.multiply_csr_elemwise(p, p.copy(), j, j.copy(), x, x2))
    # B-only entries under subtraction are negated, -(0.0) = -0.0
    pb = np.array([0, 1, 3, 3], dtype=np.int32); jb = np.array([0, 1, 2], dtype=np.int32); xb = np.array([0.0, 5.0, -7.0])
    assert_list_equal(G.add_csr_elemwise(p, pb, j, jb, x, xb, True), O.add_csr_elemwise(p, pb, j, jb, x, xb, True))
    # identical-structure fast paths (pointer identity): aliased structure / all-empty result
    r = G.add_csr_elemwise(p, p, j, j, x, x2, False)
    assert r["indptr"] is p and r["indices"] is j
    assert_list_equal(r, O.add_csr_elemwise(p, p, j, j, x, x2, False))
    r = G.multiply_csr_elemwise(p, p, j, j, x, x2)
    assert r["indptr"] is p and r["indices"] is j
    assert_list_equal(r, O.multiply_csr_elemwise(p, p, j, j, x, x2))
    assert_list_equal(G.add_csr_elemwise(p, p, j, j, x, x, True), O.add_csr_elemwise(p, p, j, j, x, x, True))


@pytest.mark.parametrize("m,K,d1,d2", MERGE_CASES[:6])
def test_logical_or_xor_and(gpu, m, K, d1, d2):
    p1, j1, x1 = rand_csr(m, K, d1, seed=21 + m, dtype="l")
    p2, j2, x2 = rand_csr(m, K, d2, seed=22 + K, dtype="l")
    for xor in (False, True):
        assert_list_equal(G.logicalor_csr_elemwise(p1, p2, j1, j2, x1, x2, xor),
                          O.logicalor_csr_elemwise(p1, p2, j1, j2, x1, x2, xor))
    assert_list_equal(G.logicaland_csr_elemwise(p1, p2, j1, j2, x1, x2), O.logicaland_csr_elemwise(p1, p2, j1, j2, x1, x2))
    # fast path for logicals
    assert_list_equal(G.logicalor_csr_elemwise(p1, p1, j1, j1, x1, x1[::-1].copy(), False),
                      O.logicalor_csr_elemwise(p1, p1, j1, j1, x1, x1[::-1].copy(), False))


def _csr_with_row_lengths(lens, K, rng, dtype="d"):
    p = np.zeros(len(lens) + 1, dtype=np.int32); p[1:] = np.cumsum(lens)
    j = np.empty(int(p[-1]), dtype=np.int32)
    for r, l in enumerate(lens):
        if l:
            j[p[r]:p[r + 1]] = np.sort(rng.choice(K, size=int(l), replace=False)).astype(np.int32)
    x = rng.integers(0, 2, size=j.size).astype(np.int32) if dtype == "l" else rng.normal(size=j.size).round(3)
    return p, j, x


@pytest.mark.parametrize("m", [300, 2500])
def test_merge_long_and_very_long_row_pairs(gpu, m):
    """Row pairs that do not fit a lane group take the blocked merge (merge.hip count_row_slow / fill_row_slow); with >= 2048
    rows, pairs with a row of more than 1024 entries are left to merge_long_count_kernel / merge_long_fill_kernel (pieces by
    value, one workgroup per pair).  Lengths around every boundary (G, block multiples, 1024, 1025, several pieces), long
    against short / empty / long, heavy and no overlap; every operation, structure and values bit for bit."""
    rng = np.random.default_rng(77 + m)
    K = 40_000
    special = [0, 1, 63, 64, 65, 127, 128, 129, 200, 1023, 1024, 1025, 2047, 2048, 2049, 3000, 5000, 20_000, 39_999]
    l1 = rng.integers(0, 30, size=m); l2 = rng.integers(0, 30, size=m)
    rows = rng.choice(m, size=3 * len(special), replace=False)
    for q, r in enumerate(rows):
        a, b = special[q % len(special)], special[(q * 7 + 3) % len(special)]
        if q < len(special):
            l1[r], l2[r] = a, b                                   # long against whatever
        elif q < 2 * len(special):
            l1[r], l2[r] = b, a
        else:
            l1[r], l2[r] = a, a                                   # equally long
    if m >= 2048:                                                 # (the one-workgroup-per-pair kernels start at 2^21 entries)
        more = np.setdiff1d(np.arange(m), rows)[:90]
        l1[more] = rng.integers(8_000, 16_000, size=more.size)
        l2[more] = np.where(rng.random(more.size) < 0.5, rng.integers(8_000, 16_000, size=more.size), rng.integers(0, 40, size=more.size))
    p1, j1, x1 = _csr_with_row_lengths(l1, K, rng)
    p2, j2, x2 = _csr_with_row_lengths(l2, K, rng)
    assert m < 2048 or j1.size + j2.size >= 1 << 21
    r0, r1 = int(rows[-1]), int(rows[-2])                         # full overlap (same columns) / interleaved without any
    j2[p2[r0]:p2[r0 + 1]] = j1[p1[r0]:p1[r0] + (p2[r0 + 1] - p2[r0])] if l1[r0] >= l2[r0] else j2[p2[r0]:p2[r0 + 1]]
    for sub in (False, True):
        assert_list_equal(G.add_csr_elemwise(p1, p2, j1, j2, x1, x2, sub), O.add_csr_elemwise(p1, p2, j1, j2, x1, x2, sub))
    assert_list_equal(G.multiply_csr_elemwise(p1, p2, j1, j2, x1, x2), O.multiply_csr_elemwise(p1, p2, j1, j2, x1, x2))
    b1, b2 = (x1 > 0).astype(np.int32), (x2 > 0).astype(np.int32)
    for xor in (False, True):
        assert_list_equal(G.logicalor_csr_elemwise(p1, p2, j1, j2, b1, b2, xor), O.logicalor_csr_elemwise(p1, p2, j1, j2, b1, b2, xor))
    assert_list_equal(G.logicaland_csr_elemwise(p1, p2, j1, j2, b1, b2), O.logicaland_csr_elemwise(p1, p2, j1, j2, b1, b2))
    del r1


def test_merge_mid_size_overlap(gpu):
    # cfg4-shaped, scaled down: fixed 50/row, ~50% shared pattern
    m, K = 20_000, 20_000
    p1, j1, x1 = synth.csr_fixed(m, K, 50)
    p2, j2, x2 = synth.csr_overlapping(p1, j1, K, 50)
    for sub in (False, True):
        assert_list_equal(G.add_csr_elemwise(p1, p2, j1, j2, x1, x2, sub), O.add_csr_elemwise(p1, p2, j1, j2, x1, x2, sub))
    assert_list_equal(G.multiply_csr_elemwise(p1, p2, j1, j2, x1, x2), O.multiply_csr_elemwise(p1, p2, j1, j2, x1, x2))


# ----------------------------------------------------------------------------- gather
def test_merge_one_pass_kernel_vs_oracle(gpu):
    """mxd_csr_merge_fused (one pass, decoupled look-back over tile totals): every op, rows longer than a lane group,
    empty rows, several tiles, one operand empty — bit-exact against the oracle like the two-pass form."""
    from devmem import merge_fused_device
    cases = [(100, 35, 0.4, 0.6), (3, 300, 0.9, 0.9), (5000, 64, 0.1, 0.1), (40, 200, 0.7, 0.02), (50, 20, 0.0, 0.5),
             (9000, 900, 0.02, 0.03), (700, 3000, 0.05, 0.001)]
    for m, K, d1, d2 in cases:
        p1, j1, x1 = rand_csr(m, K, d1, seed=31 + m, empty_rows=(1,) if m > 2 else ())
        p2, j2, x2 = rand_csr(m, K, d2, seed=32 + K, empty_rows=(1, 2) if m > 3 else ())
        for op, ref in ((_lib.MX_OP_ADD, O.add_csr_elemwise(p1, p2, j1, j2, x1, x2, False)),
                        (_lib.MX_OP_SUB, O.add_csr_elemwise(p1, p2, j1, j2, x1, x2, True)),
                        (_lib.MX_OP_MUL, O.multiply_csr_elemwise(p1, p2, j1, j2, x1, x2))):
            gp, gj, gx = merge_fused_device(op, p1, j1, x1, p2, j2, x2)
            np.testing.assert_array_equal(gp, ref["indptr"])
            np.testing.assert_array_equal(gj, ref["indices"])
            np.testing.assert_array_equal(gx, ref["values"])
        l1 = np.random.default_rng(m).choice(np.array([0, 1, NA], dtype=np.int32), size=j1.size)
        l2 = np.random.default_rng(K).choice(np.array([0, 1, NA], dtype=np.int32), size=j2.size)
        for op, ref in ((_lib.MX_OP_OR, O.logicalor_csr_elemwise(p1, p2, j1, j2, l1, l2, False)),
                        (_lib.MX_OP_XOR, O.logicalor_csr_elemwise(p1, p2, j1, j2, l1, l2, True)),
                        (_lib.MX_OP_AND, O.logicaland_csr_elemwise(p1, p2, j1, j2, l1, l2))):
            gp, gj, gx = merge_fused_device(op, p1, j1, l1, p2, j2, l2)
            np.testing.assert_array_equal(gp, ref["indptr"])
            np.testing.assert_array_equal(gj, ref["indices"])
            np.testing.assert_array_equal(gx, ref["values"])


def test_copy_csr_rows(gpu):
    p, j, x = rand_csr(1000, 500, 0.1, seed=7, empty_rows=(10, 11))     # test-slice.R:6-16 fixture shape
    rng = np.random.default_rng(9)
    cases = [rng.integers(0, 1000, size=300).astype(np.int32),
             np.array([999, 999, 10, 0, 999], dtype=np.int32),          # repeats, first/last row (test-slice.R:282-292)
             np.arange(999, -1, -1, dtype=np.int32),                    # reversed sequence (test-slice.R:243-280)
             np.array([5], dtype=np.int32),
             np.array([10, 11, 10], dtype=np.int32),                    # selects no entries: three EMPTY vectors
             np.zeros(0, dtype=np.int32)]
    xl = rng.choice(np.array([0, 1, NA], dtype=np.int32), size=x.size)
    for rows in cases:
        assert_list_equal(G.copy_csr_rows_numeric(p, j, x, rows), O.copy_csr_rows_numeric(p, j, x, rows))
        assert_list_equal(G.copy_csr_rows_logical(p, j, xl, rows), O.copy_csr_rows_logical(p, j, xl, rows))
        assert_list_equal(G.copy_csr_rows_binary(p, j, rows), O.copy_csr_rows_binary(p, j, rows))


def test_copy_csr_rows_skewed_lengths(gpu):
    p, j, x = synth.csr_skewed(5000, 3000, 20, seed=5)
    rows = synth.rows_with_replacement(4000, 5000)
    assert_list_equal(G.copy_csr_rows_numeric(p, j, x, rows), O.copy_csr_rows_numeric(p, j, x, rows))


def test_check_is_seq(gpu):
    rng = np.random.default_rng(0)
    big = np.arange(7, 7 + 100_000, dtype=np.int32)
    cases = [[], [5], [3, 4, 5], [3, 5, 5], [3, 5, 4, 6], [5, 4, 3], big, big[::-1].copy()]
    broken = big.copy(); broken[[500, 501]] = broken[[501, 500]]         # same end points, not a sequence
    cases += [broken, broken[::-1].copy(), rng.integers(0, 100, size=1000).astype(np.int32)]
    for c in cases:
        assert G.check_is_seq(c) == O.check_is_seq(c)
        assert G.check_is_rev_seq(c) == O.check_is_rev_seq(c)


# ----------------------------------------------------------------------------- sort precondition (§8f rank 1)
def test_sort_indices_kat_and_random(gpu):
    p = np.array([0, 1, 4, 5, 6], dtype=np.int32)                      # tests/testthat/test-utilities.R:32-49
    j = np.array([4, 2, 1, 4, 1, 0], dtype=np.int32)
    x = np.array([-0.91, 0.14, -0.12, -0.12, 1.1, 0.66])
    assert not G.check_indices_are_sorted(p, j)
    G.sort_sparse_indices_inplace(p, j, x)
    assert j.tolist() == [4, 1, 2, 4, 1, 0] and x.tolist() == [-0.91, -0.12, 0.14, -0.12, 1.1, 0.66]
    assert G.check_indices_are_sorted(p, j)
    for dtype in ("d", "l", "n"):
        pp, jj, xx = rand_csr(300, 200, 0.2, seed=4, sorted_cols=False, dtype=dtype, empty_rows=(0, 7))
        js, xs = O.sort_sparse_indices(pp, jj, xx)
        assert G.check_indices_are_sorted(pp, jj) == O.check_indices_are_sorted(pp, jj)
        G.sort_sparse_indices_inplace(pp, jj, xx)
        np.testing.assert_array_equal(jj, js)
        if xx is not None:
            np.testing.assert_array_equal(xx, xs)
        assert G.check_indices_are_sorted(pp, jj)


def test_sort_rows_bitonic_and_rank_paths_with_duplicates(gpu):
    # rows of 2 .. 511 entries (bitonic network in LDS), of exactly 512 / 513 and ~1500 (rank sort), duplicates inside rows
    # (stable: ties keep their storage order, like the oracle), rows that are already sorted (left alone), empty rows
    rng = np.random.default_rng(21)
    lens = np.concatenate([[0, 1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1500, 0, 40],
                           rng.integers(0, 300, size=400)])
    p = np.zeros(lens.size + 1, dtype=np.int32)
    p[1:] = np.cumsum(lens)
    j = rng.integers(0, 200, size=p[-1]).astype(np.int32)              # 200 columns: plenty of duplicates
    for r in (5, 30, 77):                                              # some rows sorted to begin with
        j[p[r]:p[r + 1]].sort()
    x = rng.normal(size=p[-1])
    js, xs = O.sort_sparse_indices(p, j.copy(), x.copy())
    assert not G.check_indices_are_sorted(p, j)
    jj, xx = j.copy(), x.copy()
    G.sort_sparse_indices_inplace(p, jj, xx)
    np.testing.assert_array_equal(jj, js)
    np.testing.assert_array_equal(xx, xs)
    assert G.check_indices_are_sorted(p, jj)
    xl = rng.choice(np.array([0, 1, NA], dtype=np.int32), size=p[-1])
    jl, xls = O.sort_sparse_indices(p, j.copy(), xl.copy())
    jj, xx = j.copy(), xl.copy()
    G.sort_sparse_indices_inplace(p, jj, xx)
    np.testing.assert_array_equal(jj, jl)
    np.testing.assert_array_equal(xx, xls)


def test_check_indices_are_sorted_edges(gpu):
    # descents only AT row starts (sorted), one descent inside the last row, inside the first row, in a one-quad matrix,
    # array lengths that are not multiples of 4, rows separated by empty rows
    p = np.array([0, 3, 3, 6, 6, 6, 11], dtype=np.int32)
    j = np.array([5, 7, 9, 0, 1, 8, 2, 3, 3, 4, 6], dtype=np.int32)
    assert G.check_indices_are_sorted(p, j) and O.check_indices_are_sorted(p, j)
    for k in (1, 4, 10):                                               # a descent inside a row
        jb = j.copy()
        jb[k] = jb[k - 1] - 1
        assert not G.check_indices_are_sorted(p, jb) and not O.check_indices_are_sorted(p, jb)
    pp, jj, _ = synth.csr_fixed(20_000, 3000, 7, seed=9)               # 140 000 entries: several workgroups, ragged tail
    assert G.check_indices_are_sorted(pp, jj)
    for k in (6, 7, 69_999, 139_999):
        jb = jj.copy()
        jb[k], jb[k - 1] = jb[k - 1], jb[k]
        assert G.check_indices_are_sorted(pp, jb) == O.check_indices_are_sorted(pp, jb)
    assert G.check_indices_are_sorted(np.array([0, 1], dtype=np.int32), np.array([3], dtype=np.int32))
    assert not G.check_indices_are_sorted(np.array([0, 2, 5], dtype=np.int32), np.array([3, 4, 9, 1, 2], dtype=np.int32))
    assert G.check_indices_are_sorted(np.array([0, 2, 5], dtype=np.int32), np.array([3, 4, 1, 2, 9], dtype=np.int32))


# ----------------------------------------------------------------------------- SpMM kernel variants (device level)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("colmajor", [False, True])
@pytest.mark.parametrize("m,K,n,per_row,npanels", [
    (3000, 5000, 128, 32, 4), (3000, 5000, 128, 32, 1), (1000, 700, 64, 9, 3), (777, 900, 16, 20, 7),
    (515, 300, 132, 40, 5), (2050, 64, 256, 12, 2), (300, 4000, 8, 70, 16)])
def test_spmm_slab_kernel_vs_oracle(gpu, dtype, colmajor, m, K, n, per_row, npanels):
    from devmem import spmm_device
    p, j, x = synth.csr_fixed(m, K, per_row, seed=m + n)
    p = p.copy(); p[m // 2 + 1:] -= 0            # keep; empty rows come from the skewed case below
    B = synth.dense_normal(K, n, dtype=dtype)
    ref = O.tcrossprod_csr_dense(p, j, x, np.asfortranarray(B.T), 1, use_fma=True)
    for algo in (1, 2):
        got = spmm_device(p, j, x, B, colmajor, algo, rows_sorted=True, npanels=npanels)
        np.testing.assert_array_equal(got, ref)                     # same order + FMA => bitwise
    # panels forced off for rows of unknown order
    np.testing.assert_array_equal(spmm_device(p, j, x, B, colmajor, 2, rows_sorted=False, npanels=npanels), ref)


def test_spmm_slab_kernel_skewed_rows_and_unsorted(gpu):
    from devmem import spmm_device
    p, j, x = synth.csr_skewed(4000, 2500, 24, seed=3)               # empty rows, rows longer than a chunk
    B = synth.dense_normal(2500, 128)
    ref = O.tcrossprod_csr_dense(p, j, x, np.asfortranarray(B.T), 1, use_fma=True)
    for npanels in (1, 2, 5, 33):
        np.testing.assert_array_equal(spmm_device(p, j, x, B, True, 2, True, npanels), ref)
    pu, ju, xu = rand_csr(500, 300, 0.2, seed=8, sorted_cols=False)
    Bu = synth.dense_normal(300, 32)
    refu = O.tcrossprod_csr_dense(pu, ju, xu, np.asfortranarray(Bu.T), 1, use_fma=True)
    np.testing.assert_array_equal(spmm_device(pu, ju, xu, Bu, True, 2, rows_sorted=False), refu)
    np.testing.assert_array_equal(spmm_device(pu, ju, xu, Bu, False, 0, rows_sorted=False), refu)


# ----------------------------------------------------------------------------- §8(f) rank 2: column-filtering slices
def _eq_lists(g, o):
    assert set(g.keys()) == set(o.keys())
    for k in g:
        assert g[k].dtype == o[k].dtype and g[k].shape == o[k].shape, (k, g[k].dtype, o[k].dtype, g[k].shape, o[k].shape)
        np.testing.assert_array_equal(g[k], o[k])


def test_copy_csr_rows_col_seq(gpu):
    p, j, x = rand_csr(1000, 500, 0.1, seed=7, empty_rows=(10, 11))
    rng = np.random.default_rng(2)
    xl = rng.choice(np.array([0, 1, NA], dtype=np.int32), size=x.size)
    row_sets = [rng.integers(0, 1000, size=300).astype(np.int32), np.array([10, 11], dtype=np.int32),
                np.arange(999, -1, -1, dtype=np.int32), np.zeros(0, dtype=np.int32)]
    col_sets = [(np.arange(100, 201, dtype=np.int32), True), (np.arange(0, 500, dtype=np.int32), False),
                (np.array([499], dtype=np.int32), False), (np.arange(250, 120, -1, dtype=np.int32), True)]
    for rows in row_sets:
        for cols, index1 in col_sets:
            _eq_lists(G.copy_csr_rows_col_seq_numeric(p, j, x, rows, cols, index1),
                      O.copy_csr_rows_col_seq_numeric(p, j, x, rows, cols, index1))
            _eq_lists(G.copy_csr_rows_col_seq_logical(p, j, xl, rows, cols, index1),
                      O.copy_csr_rows_col_seq_logical(p, j, xl, rows, cols, index1))
            _eq_lists(G.copy_csr_rows_col_seq_binary(p, j, rows, cols, index1),
                      O.copy_csr_rows_col_seq_binary(p, j, rows, cols, index1))


def test_copy_csr_arbitrary(gpu):
    p, j, x = rand_csr(1000, 500, 0.1, seed=7, empty_rows=(10, 11))
    rng = np.random.default_rng(4)
    xl = rng.choice(np.array([0, 1, NA], dtype=np.int32), size=x.size)
    rows = rng.integers(0, 1000, size=400).astype(np.int32)
    col_sets = [rng.permutation(500)[:60].astype(np.int32),                        # unsorted, unique
                np.sort(rng.permutation(500)[:200]).astype(np.int32),              # sorted, unique
                np.array([70, 3, 3, 41, 0, 499, 12, 3, 70], dtype=np.int32),       # repeats, unsorted
                np.array([5, 5, 5, 9, 9, 300], dtype=np.int32),                    # repeats, sorted
                np.array([17], dtype=np.int32)]
    for cols in col_sets:
        for rr in (rows, np.array([10, 11, 10], dtype=np.int32)):
            _eq_lists(G.copy_csr_arbitrary_numeric(p, j, x, rr, cols), O.copy_csr_arbitrary_numeric(p, j, x, rr, cols))
            _eq_lists(G.copy_csr_arbitrary_logical(p, j, xl, rr, cols), O.copy_csr_arbitrary_logical(p, j, xl, rr, cols))
            _eq_lists(G.copy_csr_arbitrary_binary(p, j, rr, cols), O.copy_csr_arbitrary_binary(p, j, rr, cols))
    ps, js, xs = synth.csr_skewed(3000, 2000, 15, seed=9)
    cols = rng.permutation(2000)[:700].astype(np.int32)
    rows = synth.rows_with_replacement(2500, 3000)
    _eq_lists(G.copy_csr_arbitrary_numeric(ps, js, xs, rows, cols), O.copy_csr_arbitrary_numeric(ps, js, xs, rows, cols))


def test_reverse_rows_and_columns(gpu):
    for dtype in ("d", "l", "n"):
        p, j, x = rand_csr(300, 170, 0.15, seed=23, dtype=dtype, empty_rows=(0, 299, 100))
        if dtype == "d":
            _eq_lists(G.reverse_rows_numeric(p, j, x), O.reverse_rows_numeric(p, j, x))
        elif dtype == "l":
            _eq_lists(G.reverse_rows_logical(p, j, x), O.reverse_rows_logical(p, j, x))
        else:
            _eq_lists(G.reverse_rows_binary(p, j), O.reverse_rows_binary(p, j))
        jg, jo = j.copy(), j.copy()
        xg = None if x is None else x.copy()
        xo = None if x is None else x.copy()
        O.reverse_columns_inplace(p, jo, xo, 170)
        if dtype == "d":
            G.reverse_columns_inplace_numeric(p, jg, xg, 170)
        elif dtype == "l":
            G.reverse_columns_inplace_logical(p, jg, xg, 170)
        else:
            G.reverse_columns_inplace_binary(p, jg, 170)
        np.testing.assert_array_equal(jg, jo)
        if x is not None:
            np.testing.assert_array_equal(xg, xo)
    e = G.reverse_rows_numeric(np.zeros(4, dtype=np.int32), np.zeros(0, dtype=np.int32), np.zeros(0))
    assert e["indptr"].tolist() == [0, 0, 0, 0] and e["indices"].size == 0


# ----------------------------------------------------------------------------- §8(f) rank 3: cbind / rbind
def test_cbind_csr(gpu):
    rng = np.random.default_rng(8)
    for (m1, m2, d1, d2) in [(400, 400, 0.2, 0.3), (400, 250, 0.2, 0.3), (250, 400, 0.0, 0.3), (50, 50, 0.0, 0.0)]:
        p1, j1, x1 = rand_csr(m1, 130, d1, seed=51 + m2, empty_rows=(2,))
        p2, j2, x2 = rand_csr(m2, 90, d2, seed=52 + m1, empty_rows=(2, 3))
        j2s = (j2 + 130).astype(np.int32)
        _eq_lists(G.cbind_csr_numeric(p1, j1, x1, p2, j2s, x2), O.cbind_csr_numeric(p1, j1, x1, p2, j2s, x2))
        l1 = rng.choice(np.array([0, 1, NA], dtype=np.int32), size=x1.size)
        l2 = rng.choice(np.array([0, 1, NA], dtype=np.int32), size=x2.size)
        _eq_lists(G.cbind_csr_logical(p1, j1, l1, p2, j2s, l2), O.cbind_csr_logical(p1, j1, l1, p2, j2s, l2))
        _eq_lists(G.cbind_csr_binary(p1, j1, p2, j2s), O.cbind_csr_binary(p1, j1, p2, j2s))


def test_concat_csr_batch(gpu):
    p1, j1, x1 = rand_csr(40, 13, 0.3, seed=51, empty_rows=(2,))
    p2, j2, x2 = rand_csr(25, 13, 0.4, seed=52, empty_rows=(2, 3))
    xl = np.where(np.arange(j2.size) % 7 == 0, NA, (x2 > 0).astype(np.int32)).astype(np.int32)
    x1n = x1.copy(); x1n[::5] = np.nan
    objs = [(0, p1, j1, x1n, 40), (1, p2, j2, xl, 25), (2, p2, j2, None, 25),
            (3, None, np.array([2, 5], np.int32), np.array([1.5, np.nan]), 1),
            (4, None, np.array([1, 3], np.int32), np.array([NA, 7], np.int32), 1),
            (5, None, np.array([3, 4], np.int32), np.array([1, NA], np.int32), 1),
            (6, None, np.array([9], np.int32), None, 1), (0, np.zeros(4, np.int32), np.zeros(0, np.int32), np.zeros(0), 3)]
    for out_kind in (0, 1, 2):
        g, o = G.concat_csr_batch(objs, out_kind), O.concat_csr_batch(objs, out_kind)
        np.testing.assert_array_equal(g["indptr"], o["indptr"])
        np.testing.assert_array_equal(g["indices"], o["indices"])
        if out_kind == 2:
            assert g["values"] is None
        else:
            assert g["values"].dtype == o["values"].dtype
            np.testing.assert_array_equal(g["values"], o["values"])     # NaN == NaN positionally (assert_array_equal)


# ----------------------------------------------------------------------------- §8(f) rank 4
def test_csr_svec(gpu):
    rng = np.random.default_rng(62)
    for (m, K, dens, ny) in [(80, 60, 0.2, 20), (500, 3000, 0.05, 700), (30, 10, 1.0, 10), (64, 200, 0.3, 0)]:
        p, j, x = rand_csr(m, K, dens, seed=61 + m, empty_rows=(1,))
        yi = (np.sort(rng.permutation(K)[:ny]) + 1).astype(np.int32)
        yv = rng.normal(size=ny)
        yint = rng.integers(-4, 5, size=ny).astype(np.int32)
        ylg = rng.integers(0, 2, size=ny).astype(np.int32)
        if ny > 3:
            yint[2] = NA; ylg[1] = NA
        cases = [(G.matmul_csr_svec_numeric, O.matmul_csr_svec_numeric, (yi, yv)),
                 (G.matmul_csr_svec_integer, O.matmul_csr_svec_integer, (yi, yint)),
                 (G.matmul_csr_svec_logical, O.matmul_csr_svec_logical, (yi, ylg)),
                 (G.matmul_csr_svec_binary, O.matmul_csr_svec_binary, (yi,)),
                 (G.matmul_csr_svec_float32, O.matmul_csr_svec_float32, (yi, yv.astype(np.float32)))]
        for gf, of, args in cases:
            g, o = gf(p, j, x, *args), of(p, j, x, *args)
            np.testing.assert_array_equal(np.isnan(g), np.isnan(o))
            np.testing.assert_allclose(g[~np.isnan(g)], o[~np.isnan(o)], rtol=1e-11, atol=1e-12)


def test_csr_by_dense_elemwise(gpu):
    rng = np.random.default_rng(63)
    p, j, x = rand_csr(300, 70, 0.15, seed=64, empty_rows=(0, 299))
    D = rng.normal(size=(300, 70))
    np.testing.assert_array_equal(G.multiply_csr_by_dense_elemwise_double(p, j, x, D),
                                  O.multiply_csr_by_dense_elemwise_double(p, j, x, D))
    D32 = D.astype(np.float32)
    np.testing.assert_array_equal(G.multiply_csr_by_dense_elemwise_float32(p, j, x, D32),
                                  O.multiply_csr_by_dense_elemwise_float32(p, j, x, D32))
    Di = rng.integers(-3, 4, size=(300, 70)).astype(np.int32); Di[5, :] = NA
    for gf, of in [(G.multiply_csr_by_dense_elemwise_int, O.multiply_csr_by_dense_elemwise_int),
                   (G.multiply_csr_by_dense_elemwise_bool, O.multiply_csr_by_dense_elemwise_bool)]:
        np.testing.assert_array_equal(gf(p, j, x, Di), of(p, j, x, Di))
    xl = rng.choice(np.array([0, 1, NA], np.int32), size=x.size)
    Dl = rng.choice(np.array([0, 1, NA], np.int32), size=(300, 70))
    np.testing.assert_array_equal(G.logicaland_csr_by_dense_cpp(p, j, xl, Dl), O.logicaland_csr_by_dense_cpp(p, j, xl, Dl))


# ----------------------------------------------------------------------------- planned SpMM kernel (v3)
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("colmajor", [False, True])
@pytest.mark.parametrize("m,K,n,per_row,npanels", [
    (3000, 5000, 128, 32, 4), (3000, 5000, 128, 32, 1), (1000, 700, 64, 9, 3), (777, 900, 16, 20, 7),
    (515, 300, 132, 40, 5), (2050, 64, 256, 12, 2), (300, 4000, 8, 70, 16), (63, 50, 16, 5, 64)])
def test_spmm_planned_kernel_vs_oracle(gpu, dtype, colmajor, m, K, n, per_row, npanels):
    from devmem import spmm_planned_device
    p, j, x = synth.csr_fixed(m, K, per_row, seed=m + n)
    B = synth.dense_normal(K, n, dtype=dtype)
    ref = O.tcrossprod_csr_dense(p, j, x, np.asfortranarray(B.T), 1, use_fma=True).astype(np.float64)
    scale = (np.abs(csr_abs_dense(p, j, x, K)) @ np.abs(B.astype(np.float64)))
    tol = (1e-14 if dtype == np.float64 else 1e-6) * scale + 1e-300
    for sync in (0, 2):
        got = spmm_planned_device(p, j, x, B, colmajor, npanels=npanels, sync_mode=sync).astype(np.float64)
        assert got.shape == ref.shape
        assert (np.abs(got - ref) <= tol).all()


def csr_abs_dense(p, j, x, K):
    import scipy.sparse as sp
    return sp.csr_matrix((np.abs(x), j, p), shape=(p.size - 1, K)).toarray()


def test_spmm_planned_kernel_skewed_unsorted_special(gpu):
    from devmem import spmm_planned_device
    p, j, x = synth.csr_skewed(4000, 2500, 24, seed=3)               # empty rows, very long rows
    B = synth.dense_normal(2500, 128)
    ref = O.tcrossprod_csr_dense(p, j, x, np.asfortranarray(B.T), 1, use_fma=True)
    for npanels in (1, 2, 5, 33):
        got = spmm_planned_device(p, j, x, B, True, npanels=npanels)
        np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-11)
    pu, ju, xu = rand_csr(500, 300, 0.2, seed=8, sorted_cols=False)  # unsorted rows, duplicates-free
    Bu = synth.dense_normal(300, 32)
    refu = O.tcrossprod_csr_dense(pu, ju, xu, np.asfortranarray(Bu.T), 1, use_fma=True)
    np.testing.assert_allclose(spmm_planned_device(pu, ju, xu, Bu, False, npanels=3), refu, rtol=1e-12, atol=1e-12)
    # NaN / Inf in values and in B propagate like in the reference
    ps = np.array([0, 3, 3, 5], dtype=np.int32); js = np.array([1, 1, 0, 1, 0], dtype=np.int32)
    xs = np.array([2.0, 3.0, 1.0, np.inf, np.nan])
    Bs = np.zeros((2, 16)); Bs[0, :] = 1.0; Bs[1, :] = np.arange(16); Bs[0, 3] = np.inf
    got, refs = spmm_planned_device(ps, js, xs, Bs, True), O.tcrossprod_csr_dense(ps, js, xs, np.asfortranarray(Bs.T), 1)
    np.testing.assert_array_equal(np.isnan(got), np.isnan(refs))
    np.testing.assert_array_equal(got[~np.isnan(got)], refs[~np.isnan(refs)])


def test_spmm_planned_edge_shapes(gpu):
    from devmem import spmm_planned_device
    B = synth.dense_normal(40, 16)
    # all-empty matrix, single row, single entry, rows that are all padding in their octet
    z = spmm_planned_device(np.zeros(6, np.int32), np.zeros(0, np.int32), np.zeros(0), B, True)
    assert z.shape == (5, 16) and not z.any()
    p = np.array([0, 1], np.int32); j = np.array([39], np.int32); x = np.array([2.5])
    np.testing.assert_array_equal(spmm_planned_device(p, j, x, B, False), 2.5 * B[39:40])
    p = np.zeros(131, np.int32); p[70:] = 3                     # only row 69 has entries (second octet)
    j = np.array([0, 7, 39], np.int32); x = np.array([1.0, -1.0, 0.5])
    got = spmm_planned_device(p, j, x, B, True, npanels=7)
    ref = np.zeros((130, 16)); ref[69] = B[0] - B[7] + 0.5 * B[39]
    np.testing.assert_allclose(got, ref, rtol=1e-15, atol=1e-15)
    # duplicate column ids inside a row accumulate (SpMM does not need unique columns)
    p = np.array([0, 4], np.int32); j = np.array([3, 3, 3, 1], np.int32); x = np.array([1.0, 2.0, 3.0, 4.0])
    np.testing.assert_allclose(spmm_planned_device(p, j, x, B, True, npanels=2), (6 * B[3] + 4 * B[1])[None, :], rtol=1e-15)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("colmajor", [True, False])
def test_spmm_planned_balanced_bundles(gpu, dtype, colmajor):
    """Rows of very uneven length: octets switch to the dealt layout (per panel, the octet's entries are cut into 8
    equal pieces; long rows are shared by several lane groups and folded atomically).  Shapes cover a last octet that
    is cut by m, empty rows, one row far longer than the rest of its octet, and rows shorter than one batch."""
    from devmem import spmm_planned_device
    rng = np.random.default_rng(99)
    K, n = 500, 32
    for m in (64, 130, 1000, 1027):
        lens = np.floor(rng.lognormal(1.5, 1.3, size=m)).astype(np.int64)
        lens[rng.random(m) < 0.1] = 0
        lens[m // 2] = 700                                           # > K: duplicate columns, one dominant row
        p = np.zeros(m + 1, dtype=np.int32); p[1:] = np.cumsum(lens)
        j = rng.integers(0, K, size=int(p[-1]), dtype=np.int32)     # unsorted, repeats allowed (SpMM does not care)
        x = rng.uniform(-1, 1, size=int(p[-1])).round(3)
        B = synth.dense_normal(K, n, dtype=dtype)
        ref = np.zeros((m, n))
        np.add.at(ref, np.repeat(np.arange(m), lens), x[:, None] * B[j].astype(np.float64))
        for npanels in (1, 3):
            got = spmm_planned_device(p, j, x, B, colmajor, npanels=npanels)
            tol = 1e-12 if dtype == np.float64 else 2e-4
            np.testing.assert_allclose(got, ref, rtol=tol, atol=tol * 50)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("colmajor", [True, False])
def test_spmm_planned_dealt_layout_loses_no_update(gpu, dtype, colmajor):
    """The dealt layout's folds are plain LDS read-modify-writes in the f32 sweep except where two lane groups can fold the
    same row in the same step (plan_flag_kernel, spmm_plan.hip).  Small-integer operands make every sum exact in f32
    whatever the order, so a single lost update is a wrong integer: heavy-tailed rows (rows that span several pieces,
    whole pieces and whole panels of an octet), sorted and unsorted columns, 1 / 4 / 9 / 16 panels."""
    from devmem import spmm_planned_device
    rng = np.random.default_rng(2025)
    m, K, n = 64 * 120 + 17, 4000, 64
    lens = np.minimum(np.floor(rng.lognormal(2.5, 1.5, size=m)).astype(np.int64), 3500)
    lens[rng.random(m) < 0.05] = 0
    lens[[5, 64 * 7 + 63, 64 * 50, m - 1]] = [3000, 2500, 3999, 1200]
    p = np.zeros(m + 1, dtype=np.int32); p[1:] = np.cumsum(lens)
    nnz = int(p[-1])
    B = rng.integers(-3, 4, size=(K, n)).astype(dtype)
    x = rng.integers(-2, 3, size=nnz).astype(np.float64)
    for sort_rows in (True, False):
        j = rng.integers(0, K, size=nnz, dtype=np.int32)
        if sort_rows:
            for r in range(m):
                j[p[r]:p[r + 1]].sort()
        ref = np.zeros((m, n), dtype=np.int64)
        np.add.at(ref, np.repeat(np.arange(m), lens), x[:, None].astype(np.int64) * B[j].astype(np.int64))
        assert np.abs(ref).max() < 2 ** 23
        for npanels in (1, 4, 9, 16):
            for rep in range(2):
                got = spmm_planned_device(p, j, x, B, colmajor, npanels=npanels)
                assert np.array_equal(got.astype(np.int64), ref) and np.array_equal(got, got.astype(np.int64)), (sort_rows, npanels, rep)


def test_spmm_plan_limits_and_errors(gpu):
    """The planned kernel addresses a slab with 32-bit byte offsets: K >= 2^25 columns is refused with a message (AUTO
    never picks it there), a plan that was only sized cannot be run, and errors do not poison later calls."""
    import ctypes as C
    from devmem import Dev, spmm_planned_device
    from matrixextra_amd import _lib
    lib = _lib.load()
    p = np.array([0, 2, 3], np.int32); j = np.array([0, (1 << 25) + 3, 7], np.int32); x = np.array([1.0, 2.0, 3.0])
    dp, dj, dx = Dev(p), Dev(j), Dev(x)
    plan = C.c_void_p()
    rc = lib.mxd_spmm_plan_create(C.c_int(2), C.c_int((1 << 25) + 5), dp.ptr, dj.ptr, dx.ptr, C.c_int(0), None, C.byref(plan))
    assert rc != 0 and b"2^25" in lib.mx_last_error()
    assert not plan.value
    rc = lib.mxd_spmm_plan_run(None, C.c_int(4), None, C.c_size_t(4), None, C.c_size_t(4), C.c_int(_lib.MX_F64),
                               C.c_int(0), C.c_int(0), C.c_int(-1), None)
    assert rc != 0
    B = synth.dense_normal(40, 16)
    pp = np.array([0, 1], np.int32); jj = np.array([39], np.int32); xx = np.array([2.5])
    np.testing.assert_array_equal(spmm_planned_device(pp, jj, xx, B, False), 2.5 * B[39:40])   # still healthy


def test_spmm_plan_dealt_layout_keeps_skewed_plans_small(gpu):
    """Log-normal row lengths: with whole rows per lane group the plan is 1.8x the CSR; dealing every panel's entries
    in 8 equal pieces keeps it near 1x (at most 7 holes per panel and octet + the rounding to 32 steps)."""
    import torch
    from matrixextra_amd import device as D
    rng = np.random.default_rng(3)
    m, K = 64 * 400, 20_000
    lens = np.minimum(np.floor(rng.lognormal(np.log(48) - 0.5, 1.0, size=m)).astype(np.int64), 3000)
    p = np.zeros(m + 1, dtype=np.int64); p[1:] = np.cumsum(lens)
    j = rng.integers(0, K, size=int(p[-1]), dtype=np.int32)
    x = rng.uniform(-1, 1, size=int(p[-1]))
    A = D.DeviceCSR.from_host(p.astype(np.int32), j, x, K)
    A.plan(npanels=4)
    info = A.plan_info()
    assert info["padded_entries"] <= 1.15 * A.nnz, info
    B = torch.from_numpy(synth.dense_normal(K, 32)).cuda()
    got = D.spmm_planned(A, B, npanels=4).cpu().numpy()
    ref = np.zeros((m, 32))
    np.add.at(ref, np.repeat(np.arange(m), lens), x[:, None] * B.cpu().numpy()[j])
    np.testing.assert_allclose(got, ref, rtol=1e-11, atol=1e-11)


# ----------------------------------------------------------------------------- CSR (op) dense vector (§8f rank 4)
DV_FLAGS = {"mul": (1, 0, 0, 0, 0), "pow": (0, 1, 0, 0, 0), "div": (0, 0, 1, 0, 0), "mod": (0, 0, 0, 1, 0), "idiv": (0, 0, 0, 0, 1)}


def _dv_check(got, want, exact):
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    if exact:       # one IEEE multiplication / division: bit for bit, sign of zero included
        np.testing.assert_array_equal(got[ok].view(np.int64), want[ok].view(np.int64))
    else:           # %% %/% go through long double in the reference, ^ through libm's pow: last-bit agreement
        np.testing.assert_allclose(got[ok], want[ok], rtol=1e-13, atol=1e-300)


@pytest.mark.parametrize("m,K,dens", [(200, 37, 0.2), (1, 9, 0.9), (64, 1, 1.0), (513, 700, 0.02)])
def test_csr_by_dvec_all_ops_and_lengths(gpu, m, K, dens):
    """multiply_csr_by_dvec_no_NAs_numeric (src/operators.cpp:1604-2175): the four recycling branches x five
    operations x both operand orders (the kind of inputs test-operators.R:373-893 feeds it)."""
    p, j, x = rand_csr(m, K, dens, seed=m + K, empty_rows=(0,) if m > 3 else ())
    x = (x * 4).round(2)
    x[x == 0] = 1.5
    rng = np.random.default_rng(m * 7 + K)
    lens = sorted({m, m * K, 1, max(1, m // 2) if m % 2 == 0 else 1, 5, min(m * K, m + 3)})
    for ln in lens:
        v = (rng.uniform(0.5, 3.0, size=ln) * rng.choice([-1.0, 1.0], size=ln)).round(2)
        for opname, f in DV_FLAGS.items():
            for lhs in (True, False):
                got = G.multiply_csr_by_dvec_no_NAs_numeric(p, j, x, v, K, *f, lhs)
                want = O.multiply_csr_by_dvec_no_NAs_numeric(p, j, x, v, K, *f, lhs)
                assert got.dtype == np.float64 and got.shape == want.shape
                _dv_check(got, want, exact=opname in ("mul", "div"))


def test_csr_by_dvec_special_values(gpu):
    """R_pow / R_modulus / R_intdiv corner cases (src/operators.cpp:1482-1601): zeros, infinities, NaN, huge
    quotients, integer and non-integer exponents of negative bases."""
    vals = np.array([0.0, -0.0, 1.0, -1.0, 2.0, -2.0, 0.5, -0.5, 3.0, -3.0, np.inf, -np.inf, np.nan, 1e300, -1e300,
                     1e-300, 7.25, -7.25, 2.0 ** 70, -(2.0 ** 70)])
    n = vals.size
    p = np.arange(0, n * n + 1, n, dtype=np.int32)                  # n rows, every row holds all n columns
    j = np.tile(np.arange(n, dtype=np.int32), n)
    x = np.repeat(vals, n)                                          # row r has value vals[r] in every column
    v = np.tile(vals, n).reshape(n, n).T.reshape(-1, order="F")     # full matrix: entry (r, c) -> vals[c]
    for opname, f in DV_FLAGS.items():
        for lhs in (True, False):
            with np.errstate(all="ignore"):
                got = G.multiply_csr_by_dvec_no_NAs_numeric(p, j, x, v, n, *f, lhs)
            want = O.multiply_csr_by_dvec_no_NAs_numeric(p, j, x, v, n, *f, lhs)
            np.testing.assert_array_equal(np.isnan(got), np.isnan(want), err_msg=f"{opname} lhs={lhs}")
            ok = ~np.isnan(want)
            np.testing.assert_array_equal(np.isinf(got[ok]), np.isinf(want[ok]), err_msg=f"{opname} lhs={lhs}")
            fin = ok & np.isfinite(want)
            np.testing.assert_allclose(got[fin], want[fin], rtol=1e-13, atol=0, err_msg=f"{opname} lhs={lhs}")
            np.testing.assert_array_equal(np.signbit(got[ok]), np.signbit(want[ok]), err_msg=f"{opname} lhs={lhs}")


def test_csr_by_dvec_logical_and(gpu):
    """logicaland_csr_by_dvec_internal (src/operators.cpp:2177-2200): R's three-valued & with NA on either side."""
    p, j, xl = rand_csr(90, 23, 0.3, seed=12, dtype="l", empty_rows=(7,))
    rng = np.random.default_rng(13)
    for ln in (90, 90 * 23, 45, 1, 7, 100):
        vl = rng.integers(0, 2, size=ln).astype(np.int32)
        vl[rng.random(ln) < 0.2] = NA
        got = G.logicaland_csr_by_dvec_internal(p, j, xl, vl, 23)
        assert got.dtype == np.int32
        np.testing.assert_array_equal(got, O.logicaland_csr_by_dvec_internal(p, j, xl, vl, 23))


def test_csr_by_dvec_empty_inputs(gpu):
    p = np.zeros(6, dtype=np.int32); j = np.zeros(0, dtype=np.int32); x = np.zeros(0)
    assert G.multiply_csr_by_dvec_no_NAs_numeric(p, j, x, np.ones(5), 4, 1, 0, 0, 0, 0, 1).size == 0
    assert G.logicaland_csr_by_dvec_internal(p, j, np.zeros(0, np.int32), np.ones(5, np.int32), 4).size == 0
    p0 = np.zeros(1, dtype=np.int32)
    assert G.multiply_csr_by_dvec_no_NAs_numeric(p0, j, x, np.ones(1), 0, 0, 0, 1, 0, 0, 1).size == 0


# ----------------------------------------------------------------------------- round 3: one-stream sortedness, one-launch gather
def _sorted_ref(p, j):
    """numpy statement of check_is_sorted per row (misc.cpp:118-128): no descent inside a row"""
    lo, hi = int(p[0]), int(p[-1])
    if hi - lo < 2:
        return True
    d = np.zeros(hi, dtype=bool)
    d[lo + 1:hi] = j[lo + 1:hi] < j[lo:hi - 1]
    starts = p[:-1][(p[1:] > p[:-1])]
    d[starts] = False
    return not d.any()


@pytest.mark.parametrize("mean_len,m", [(1.5, 400_000), (3, 600_000), (50, 300_000), (120, 400_000), (6000, 2500)])
def test_rows_sorted_one_stream_kernel(gpu, mean_len, m):
    """The sortedness check streams the indices once and resolves row starts against an LDS bitmap (gather.hip): sizes at which
    a workgroup walks several sub-chunks (> 8.4 M entries), short rows (several rounds of row pointers per sub-chunk), long
    rows (sub-chunks without a row start), empty rows, descents planted at sub-chunk / workgroup boundaries, an indices
    array that is not 16-byte aligned, and a row-block view whose indptr[0] > 0."""
    from devmem import rows_sorted_device
    rng = np.random.default_rng(int(mean_len * 10) + m)
    lens = rng.poisson(mean_len, size=m).astype(np.int64)
    lens[rng.integers(0, m, size=m // 20)] = 0                          # empty rows, also runs of them
    lens[m // 2:m // 2 + 50] = 0
    p = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(lens, out=p[1:])
    nnz = int(p[-1])
    assert nnz < 2 ** 31
    p = p.astype(np.int32)
    K = 1 << 20
    # sorted rows: ascending inside a row by construction (cumulative gaps), random across rows
    gaps = rng.integers(1, 40, size=nnz).astype(np.int64)
    c = np.cumsum(gaps)
    row_of = np.repeat(np.arange(m), lens)
    base = np.concatenate([[0], c])[p[:-1]][row_of]                    # cumulative value before the row's first entry
    j = ((c - base) % K).astype(np.int32)
    # (c - base) grows inside a row and rows are far shorter than K / 40 on average ... except the long-row case: sort those
    if mean_len > 1000:
        for r in range(m):
            j[p[r]:p[r + 1]].sort()
    assert _sorted_ref(p, j)
    assert rows_sorted_device(p, j) is True
    for mis in (1, 2, 3):
        assert rows_sorted_device(p, j, misalign=mis) is True
    # one descent INSIDE a row, at positions spread over the array incl. the first / last entries and multiples of the
    # sub-chunk (4096 entries) and of a workgroup's chunk
    nq = (nnz + 3) // 4
    cq = -(-nq // 2048)                                                # quads per workgroup (gather.hip: RS_NB = 2048) ...
    cq = -(-cq // 1024) * 1024                                         # ... in whole sub-chunks of RS_SUB_Q = 1024 quads
    cand = {1, 2, nnz - 1, nnz // 2, 4096, 4095, 4097, 8192, 4 * cq, 4 * cq - 1, 4 * cq + 1, 12 * cq} | \
        set(rng.integers(1, nnz, size=12).tolist())
    tried = 0
    for k in sorted(cand):
        if not (0 < k < nnz):
            continue
        r = int(np.searchsorted(p, k, side="right") - 1)
        if k == p[r]:                                                  # first entry of its row: move inside the row if it has one
            if p[r + 1] - p[r] < 2:
                continue
            k += 1
        jb = j.copy()
        jb[k] = jb[k - 1] - 1 if jb[k - 1] > 0 else -1
        if jb[k] < 0:
            continue
        assert not _sorted_ref(p, jb)
        assert rows_sorted_device(p, jb, misalign=tried % 4) is False, f"descent at {k} (row {r}) not seen"
        tried += 1
    assert tried >= 8
    # row-block views with absolute offsets: indptr[0] > 0; entries before the view are garbage that must not count
    for r0 in (1, m // 3, m - 5):
        jb = j.copy()
        if p[r0] > 2:
            jb[:p[r0]] = jb[:p[r0]][::-1]                              # descending garbage before the view
        assert rows_sorted_device(p[r0:], jb) is True
        if p[-1] - p[r0] > 4:
            k = int(p[r0]) + 1
            r = int(np.searchsorted(p, k, side="right") - 1)
            if k > p[r]:
                jb[k] = jb[k - 1] - 1
                assert rows_sorted_device(p[r0:], jb) == _sorted_ref(p[r0:], jb)


def test_gather_one_launch_kernel(gpu):
    """mxd_csr_gather_fused against the oracle, bit for bit: capacity exact / generous / too small (then indptr and the
    total must still be exact and everything that fits must have been copied), numeric / logical / pattern values,
    rows of very uneven length, repeats, empty selections."""
    from devmem import gather_fused_device
    p, j, x = synth.csr_skewed(20_000, 5000, 25, seed=11, sigma=1.2)
    xl = np.random.default_rng(3).choice(np.array([0, 1, NA], dtype=np.int32), size=x.size)
    for r, seed in ((1, 1), (255, 2), (256, 3), (257, 4), (30_000, 5)):
        rows = synth.rows_with_replacement(r, 20_000, seed=seed)
        ref = O.copy_csr_rows_numeric(p, j, x, rows)
        total = int(ref["indices"].size)
        refp = ref["indptr"] if total else np.concatenate([[0], np.cumsum((p[1:] - p[:-1])[rows])]).astype(np.int32)
        for cap in (total, total + 1000, max(total // 2, 0), 0):
            for vals, dt, refv in ((x, _lib.MX_F64, ref["values"]), (xl, _lib.MX_LGL, None), (None, _lib.MX_NONE, None)):
                gp, gj, gx, nnz = gather_fused_device(p, j, vals, rows, cap, dt)
                assert nnz == total
                np.testing.assert_array_equal(gp, refp)
                if total == 0:
                    continue
                # rows that end within the capacity are copied completely
                fits = refp[1:] <= cap
                for t in np.nonzero(fits)[0][:: max(1, r // 300)]:
                    rr = rows[t]
                    np.testing.assert_array_equal(gj[refp[t]:refp[t + 1]], j[p[rr]:p[rr + 1]])
                    if vals is not None:
                        np.testing.assert_array_equal(gx[refp[t]:refp[t + 1]], vals[p[rr]:p[rr + 1]])
                if cap >= total:
                    np.testing.assert_array_equal(gj, ref["indices"])
                    if refv is not None:
                        np.testing.assert_array_equal(gx, refv)
    # nothing selected / rows without entries
    gp, gj, gx, nnz = gather_fused_device(p, j, x, np.zeros(0, dtype=np.int32), 10, _lib.MX_F64)
    assert nnz == 0 and gp.tolist() == [0]


# ----------------------------------------------------------------------------- round 3: CSR (op) vector, the NA route
def _is_na_real(a):
    """R's ISNA: a NaN whose low word is 1954 (arithmetic may set the quiet bit)"""
    return np.isnan(a) & ((a.view(np.uint64) & np.uint64(0xFFFFFFFF)) == 1954)


def _special_vector(rng, ln, opname):
    """a vector with the elements that send `CSR op vector` down the NA route (R/operators.R:981-988), ~12 % of them"""
    v = (rng.uniform(0.5, 3.0, size=ln) * rng.choice([-1.0, 1.0], size=ln)).round(2)
    if opname == "pow":
        v = np.abs(v)                                               # negatives are special under ^: placed explicitly below
    pool = [NA_REAL_F, np.nan]
    if opname == "mul":
        pool += [np.inf, -np.inf]
    if opname in ("div", "mod", "idiv", "pow"):
        pool += [0.0]
    if opname == "pow":
        pool += [-1.0, -2.5]
    hit = rng.random(ln) < 0.12
    if ln > 2 and not hit.any():
        hit[rng.integers(ln)] = True
    v[hit] = rng.choice(np.array(pool), size=int(hit.sum()))
    return v


NA_REAL_F = np.frombuffer(np.uint64(0x7FF00000000007A2).tobytes(), dtype=np.float64)[0]


@pytest.mark.parametrize("m,K,dens", [(200, 37, 0.2), (64, 5, 0.5), (513, 300, 0.02), (1, 9, 0.9)])
def test_csr_by_dvec_with_NAs_all_ops_and_lengths(gpu, m, K, dens):
    """multiply_csr_by_dvec_with_NAs (src/operators.cpp:2258-2856) against the oracle's restatement: structure bit for bit
    (pattern grows by every cell the recycled vector makes special; rows stay sorted), values NaN-for-NaN, NA-for-NA
    (NA_real_ and plain NaN are told apart, incl. the reference's swapped fills in the general branch), exact for * and /,
    1e-13 for %% %/% ^.  Lengths: == nrows, a divisor of nrows (branch A), the whole matrix (B), others (C)."""
    p, j, x = rand_csr(m, K, dens, seed=m + K, empty_rows=(0,) if m > 3 else ())
    x = (x * 4).round(2)
    x[x == 0] = 1.5
    rng = np.random.default_rng(m * 11 + K)
    lens = sorted({m, m * K, max(1, m // 2) if m % 2 == 0 else m, 5, min(m * K, m + 3), max(2, (m * K) // 3)})
    for ln in lens:
        for opname, f in DV_FLAGS.items():
            v = _special_vector(rng, ln, opname)
            want = O.multiply_csr_by_dvec_with_NAs(p, j, x, v, K, *f, True)
            with np.errstate(all="ignore"):
                got = G.multiply_csr_by_dvec_with_NAs(p, j, x, v, K, *f, True)
            what = f"{opname} len={ln}"
            if want["alias_structure"]:
                assert got["indptr"] is p and got["indices"] is j, what      # the INPUT objects, as the reference returns them
            else:
                np.testing.assert_array_equal(got["indptr"], want["indptr"], err_msg=what)
                np.testing.assert_array_equal(got["indices"], want["indices"], err_msg=what)
            gv, wv = got["values"], want["values"]
            assert gv.shape == wv.shape, what
            np.testing.assert_array_equal(np.isnan(gv), np.isnan(wv), err_msg=what)
            np.testing.assert_array_equal(_is_na_real(gv), _is_na_real(wv), err_msg=what + " (NA vs NaN)")
            ok = ~np.isnan(wv)
            if opname in ("mul", "div"):
                np.testing.assert_array_equal(gv[ok].view(np.int64), wv[ok].view(np.int64), err_msg=what)
            else:
                np.testing.assert_array_equal(np.isinf(gv[ok]), np.isinf(wv[ok]), err_msg=what)
                fin = ok & np.isfinite(wv)
                np.testing.assert_allclose(gv[fin], wv[fin], rtol=1e-13, atol=1e-300, err_msg=what)
            # rows sorted and unique
            gp, gj = got["indptr"], got["indices"]
            d = np.diff(gj) <= 0
            starts = np.zeros(gj.size, dtype=bool)
            starts[gp[1:-1][gp[1:-1] < gj.size]] = True
            assert not (d & ~starts[1:]).any(), what


def test_csr_by_dvec_with_NAs_errors_and_edge_cases(gpu):
    p, j, x = rand_csr(50, 20, 0.3, seed=3)
    v = np.ones(50); v[3] = np.nan
    # ^ / %% with the matrix on the right: the reference's internal error (operators.cpp:2274-2275)
    for f in (DV_FLAGS["pow"], DV_FLAGS["div"], DV_FLAGS["mod"]):
        with pytest.raises(_lib.MxError):
            G.multiply_csr_by_dvec_with_NAs(p, j, x, v, 20, *f, False)
    # a vector without any special element: branch A copies, branch C hands the input structure back
    v1 = np.full(50, 2.0)
    got = G.multiply_csr_by_dvec_with_NAs(p, j, x, v1, 20, *DV_FLAGS["mul"], True)
    np.testing.assert_array_equal(got["indptr"], p); np.testing.assert_array_equal(got["values"], x * 2.0)
    v2 = np.full(7, 2.0)
    got = G.multiply_csr_by_dvec_with_NAs(p, j, x, v2, 20, *DV_FLAGS["mul"], True)
    assert got["indptr"] is p and got["indices"] is j
    np.testing.assert_array_equal(got["values"], x * 2.0)
    # empty matrix, rows of a full-NA branch-A vector
    p0 = np.zeros(4, dtype=np.int32); j0 = np.zeros(0, dtype=np.int32); x0 = np.zeros(0)
    vna = np.array([1.0, NA_REAL_F, 1.0])
    got = G.multiply_csr_by_dvec_with_NAs(p0, j0, x0, vna, 6, *DV_FLAGS["mul"], True)
    want = O.multiply_csr_by_dvec_with_NAs(p0, j0, x0, vna, 6, *DV_FLAGS["mul"], True)
    np.testing.assert_array_equal(got["indptr"], want["indptr"]); np.testing.assert_array_equal(got["indices"], want["indices"])
    assert got["indptr"].tolist() == [0, 0, 6, 6] and _is_na_real(got["values"]).all()
    # the mirror of R's `X * v` takes the route by itself (R/operators.R:981-1131) and agrees with dense R arithmetic
    import matrixextra_amd as mx
    A = mx.dgRMatrix(p, j, x, (50, 20))
    vv = np.ones(50); vv[[2, 9]] = [np.inf, np.nan]
    R = A * vv
    dense = A.toarray()
    with np.errstate(all="ignore"):
        exp = dense * vv[:, None]
    got_d = R.toarray()
    np.testing.assert_array_equal(np.isnan(got_d), np.isnan(exp))
    np.testing.assert_array_equal(got_d[~np.isnan(exp)], exp[~np.isnan(exp)])
    assert R.p[-1] == p[-1] + 2 * 20 - (p[3] - p[2]) - (p[10] - p[9])    # two rows became full rows


# ----------------------------------------------------------------------------- round 3: kept plans, row ranges, the octet schedule
def test_spmm_plan_rows_and_auto_choice(gpu):
    """mxd_spmm_plan_create_auto + mxd_spmm_plan_run_rows (what the CSR cache and DeviceCSR keep): the whole matrix and
    row ranges of ONE plan against the oracle, on a matrix whose octets differ in length (so the sweep's schedule is not
    the natural order and some octets take the dealt layout); mxd_spmm_auto_algo's thresholds."""
    import ctypes as C
    from devmem import Dev
    lib = _lib.load()
    m, K, n = 5000, 3000, 64
    p, j, x = synth.csr_skewed(m, K, 20, seed=9, sigma=1.1)
    B = synth.dense_normal(K, n, seed=4)
    ref = np.zeros(m * n)
    O.gemm_csr_drm_as_drm(m, n, p, j, x, B.reshape(-1), n, ref, n, 4, True)
    ref = ref.reshape(m, n)
    dp, dj, dx, dB = Dev(p), Dev(j), Dev(x), Dev(B)
    plan = C.c_void_p()
    ready = C.c_int(0)
    _lib.check(lib.mxd_spmm_plan_create_auto(C.c_int(m), C.c_int(K), dp.ptr, dj.ptr, dx.ptr, C.c_int(0), None, C.byref(plan),
                                             C.byref(ready)))
    try:
        assert ready.value == 1
        for colmajor in (0, 1):
            ldc = m if colmajor else n
            for row0, nrows in ((0, m), (0, 640), (640, m - 640), (4992, 8), (1024, 1024)):
                dC = Dev(nbytes=m * n * 8)
                _lib.check(lib.mx_dev_memset(dC.ptr, 0xFF, C.c_size_t(dC.nbytes), None))
                off = row0 * 8 if colmajor else row0 * n * 8
                _lib.check(lib.mxd_spmm_plan_run_rows(plan, C.c_int(row0), C.c_int(nrows), C.c_int(n), dB.ptr, C.c_size_t(n),
                                                      C.c_void_p(dC.ptr.value + off), C.c_size_t(ldc), C.c_int(_lib.MX_F64),
                                                      C.c_int(colmajor), C.c_int(0), C.c_int(-1), None))
                _lib.check(lib.mx_stream_sync(None))
                got = dC.download(np.float64, (n, m) if colmajor else (m, n))
                got = got.T if colmajor else got
                np.testing.assert_allclose(got[row0:row0 + nrows], ref[row0:row0 + nrows], rtol=1e-12, atol=1e-12 * np.abs(ref).max())
                rest = np.ones(m, dtype=bool); rest[row0:row0 + nrows] = False
                assert np.isnan(got[rest]).all()                      # nothing outside the range was written
        with pytest.raises(_lib.MxError):                             # a range must start at a multiple of 64
            _lib.check(lib.mxd_spmm_plan_run_rows(plan, C.c_int(10), C.c_int(64), C.c_int(n), dB.ptr, C.c_size_t(n), dB.ptr,
                                                  C.c_size_t(n), C.c_int(_lib.MX_F64), C.c_int(0), C.c_int(0), C.c_int(-1), None))
    finally:
        lib.mxd_spmm_plan_destroy(plan)
    pick = C.c_int(-1)
    al = C.c_void_p(4096)
    for (mm, nn, KK, want) in ((1_000_000, 128, 100_000, 3), (10_000, 100, 10_000, 1), (100_000, 128, 100_000, 1), (1_000_000, 128, 1 << 25, 2)):
        _lib.check(lib.mxd_spmm_auto_algo(C.c_int(mm), C.c_int(nn), C.c_int(KK), C.c_int(_lib.MX_F64), al, C.c_size_t(nn), al,
                                          C.c_size_t(mm), C.c_int(1), C.byref(pick)))
        assert pick.value == want, (mm, nn, KK, pick.value)


@pytest.mark.parametrize("m,K,dens", [(1, 1, 1.0), (37, 5, 0.6), (700, 300, 0.05), (5000, 64, 0.9), (20000, 4000, 0.004)])
def test_remove_zero_valued_csr(gpu, m, K, dens):
    """remove_zero_valued_csr_{numeric,logical} (src/misc.cpp:553-698) — what `remove_zeros` runs after A - B has left
    explicit zeros behind: structure and values bit for bit the oracle's, for every lanes-per-row choice (row lengths from
    0.3 to 58 entries on average), with -0.0, NaN, NA_real_ (payload kept), and the reference's quirk for R logicals with
    remove_NAs (the zeros stay).  Nothing to remove: the INPUT objects come back."""
    rng = np.random.default_rng(m + K)
    p, j, x = rand_csr(m, K, dens, seed=m, empty_rows=(0,) if m > 30 else ())
    if x.size == 0:
        return
    x = x.copy()
    x[rng.random(x.size) < 0.3] = 0.0
    x[rng.random(x.size) < 0.05] = -0.0
    x[rng.random(x.size) < 0.1] = np.nan
    na_real = np.array([0x7FF00000000007A2], dtype=np.uint64).view(np.float64)[0]
    x[rng.random(x.size) < 0.05] = na_real
    xl = rng.choice(np.array([0, 1, NA], dtype=np.int32), size=x.size, p=[0.3, 0.5, 0.2])
    for rm in (False, True):
        assert_list_equal(G.remove_zero_valued_csr_numeric(p, j, x, rm), O.remove_zero_valued_csr_numeric(p, j, x, rm))
        assert_list_equal(G.remove_zero_valued_csr_logical(p, j, xl, rm), O.remove_zero_valued_csr_logical(p, j, xl, rm))
    ones = np.ones(x.size)
    r = G.remove_zero_valued_csr_numeric(p, j, ones, True)
    assert r["indptr"] is p and r["indices"] is j and r["values"] is ones
    only_nan = ones.copy(); only_nan[x.size // 2] = np.nan
    assert G.remove_zero_valued_csr_numeric(p, j, only_nan, False)["values"] is only_nan
    assert_list_equal(G.remove_zero_valued_csr_numeric(p, j, only_nan, True), O.remove_zero_valued_csr_numeric(p, j, only_nan, True))
    lz = np.ones(x.size, dtype=np.int32); lz[0] = 0                  # zeros trigger the rebuild but stay
    g = G.remove_zero_valued_csr_logical(p, j, lz, True)
    assert g["values"] is not lz
    assert_list_equal(g, O.remove_zero_valued_csr_logical(p, j, lz, True))
    allz = np.zeros(x.size)                                          # everything leaves: empty indices / values, indptr of zeros
    assert_list_equal(G.remove_zero_valued_csr_numeric(p, j, allz, False), O.remove_zero_valued_csr_numeric(p, j, allz, False))


def test_remove_zeros_after_a_subtraction(gpu):
    """the caller's sequence: X - Y keeps cancelled entries as explicit zeros (operators.cpp:477-495), remove_zeros drops them"""
    p1, j1, x1 = rand_csr(3000, 500, 0.05, seed=1)
    d = G.add_csr_elemwise(p1, p1.copy(), j1, j1.copy(), x1, x1.copy(), True)      # X - X: every entry an explicit zero
    assert d["indices"].size == j1.size and not d["values"].any()
    z = G.remove_zero_valued_csr_numeric(d["indptr"], d["indices"], d["values"], False)
    assert z["indices"].size == 0 and z["values"].size == 0 and not z["indptr"].any() and z["indptr"].size == p1.size


def test_check_valid_csr_matrix(gpu):
    """check_valid_csr_matrix (src/misc.cpp:970-1016): the reference's messages in the reference's order of checks"""
    p, j, _ = rand_csr(4000, 900, 0.02, seed=3)
    cases = [(p, j, 4000, 900), (p, j, 4000, int(j.max()))]
    jn = j.copy(); jn[j.size // 2] = -1
    cases.append((p, jn, 4000, 900))
    jna = j.copy(); jna[-1] = NA
    cases.append((p, jna, 4000, 900))
    pn = p.copy(); pn[1234] = NA
    cases.append((pn, j, 4000, 900))
    pd = p.copy(); pd[2000] = pd[2001] + 1
    cases.append((pd, j, 4000, 900))
    jb = j.copy(); jb[0] = 5000
    cases.append((pd, jb, 4000, 900))
    cases.append((np.zeros(1, dtype=np.int32), np.zeros(0, dtype=np.int32), 0, 0))
    cases.append((np.zeros(6, dtype=np.int32), np.zeros(0, dtype=np.int32), 5, 3))
    seen = set()
    for args in cases:
        g, o = G.check_valid_csr_matrix(*args), O.check_valid_csr_matrix(*args)
        assert g == o, (g, o)
        seen.add(g.get("err", ""))
    assert len(seen) == 5


@pytest.mark.parametrize("ncols,nrows,dens", [(1, 1, 1.0), (300, 200, 0.1), (5000, 4000, 0.02), (70000, 3000, 0.01)])
def test_matmul_rowvec_by_csc(gpu, ncols, nrows, dens):
    """matmul_rowvec_by_csc / _cscbin (src/matmul.cpp:643-684): float32 row vector x CSC, accumulated in float.  Against the
    oracle's loop to float tolerance (the lane-group kernel adds a column's terms in another order) and, for the flat
    kernel's sizes, bit for bit."""
    p, j, x = rand_csr(ncols, nrows, dens, seed=ncols, empty_rows=(0,) if ncols > 10 else ())
    v = np.random.default_rng(ncols + 1).normal(size=nrows).astype(np.float32)
    for got, want in ((G.matmul_rowvec_by_csc(v, p, j, x), O.matmul_rowvec_by_csc(v, p, j, x)),
                      (G.matmul_rowvec_by_cscbin(v, p, j), O.matmul_rowvec_by_cscbin(v, p, j))):
        assert got.shape == want.shape == (1, ncols) and got.dtype == np.float32
        scale = max(1.0, float(np.abs(want).max()))
        np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5 * scale)
    if j.size >= (1 << 20):      # the flat kernel (AUTO from 2^24 entries on; here by option): the reference's order of additions
        import ctypes as C
        lib = _lib.load()
        _lib.check(lib.mx_set_option(b"spmv_algo", C.c_int64(3)))
        try:
            np.testing.assert_array_equal(G.matmul_rowvec_by_csc(v, p, j, x), O.matmul_rowvec_by_csc(v, p, j, x))
            np.testing.assert_array_equal(G.matmul_rowvec_by_cscbin(v, p, j), O.matmul_rowvec_by_cscbin(v, p, j))
        finally:
            _lib.check(lib.mx_set_option(b"spmv_algo", C.c_int64(0)))


def test_per_thread_scratch_survives_a_stream_switch(gpu):
    """One thread, two torch streams, products queued back to back without a host wait in between: the library's per-thread
    scratch (SpMV slice table, the row-split kernel's panel cursors, the gather's look-back state) is shared by both
    launches — the second stream must wait for the first user (csrc/scan.hip scratch_acquire / scratch_done)."""
    import torch
    from matrixextra_amd import device as D
    m, K = 200_000, 20_000
    p, j, x = synth.csr_fixed(m, K, 64, seed=31)
    p2, j2, x2 = synth.csr_fixed(30_000, K, 300, seed=32)
    A, A2 = D.DeviceCSR.from_host(p, j, x, K), D.DeviceCSR.from_host(p2, j2, x2, K)
    v = torch.from_numpy(synth.dense_normal(K, 1, seed=33).reshape(-1)).cuda()
    B = torch.from_numpy(synth.dense_normal(K, 128, seed=34)).cuda()
    rows = torch.from_numpy(synth.rows_with_replacement(50_000, m)).cuda()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    ref_y = D.spmv(A, v, algo=3).clone()
    ref_y2 = D.spmv(A2, v, algo=3).clone()
    ref_C = D.spmm(A2, B, colmajor=False, algo=4, npanels=4).clone()
    ref_g = D.csr_gather_rows(A, rows)
    ref_g2 = D.csr_gather_rows(A2, rows[:20_000] % 30_000)
    # the planned kernel: a kept plan per matrix, but ONE per-thread packed copy of B shared by both sweeps; and the C-ABI's
    # own AUTO (keep_plan=False), whose plan buffers are per thread too
    Bs = [B, B * 2.0]
    ref_P = [D.spmm_planned(A, Bs[k], colmajor=True).clone() for k in (0, 1)]
    ref_Q = D.spmm(A, B, colmajor=True, algo=0, keep_plan=False).clone()
    torch.cuda.synchronize()
    for _ in range(5):
        with torch.cuda.stream(s1):
            y1 = D.spmv(A, v, algo=3)
            C1 = D.spmm(A2, B, colmajor=False, algo=4, npanels=4)
            g1 = D.csr_gather_rows(A, rows)
        with torch.cuda.stream(s2):
            y2 = D.spmv(A2, v, algo=3)
            C2 = D.spmm(A2, B, colmajor=False, algo=4, npanels=2)
            g2 = D.csr_gather_rows(A2, rows[:20_000] % 30_000)
        with torch.cuda.stream(s1):
            P1 = D.spmm_planned(A, Bs[0], colmajor=True)
            Q1 = D.spmm(A, B, colmajor=True, algo=0, keep_plan=False)
        with torch.cuda.stream(s2):
            P2 = D.spmm_planned(A, Bs[1], colmajor=True)
            Q2 = D.spmm(A, B, colmajor=True, algo=0, keep_plan=False)
        torch.cuda.synchronize()
        assert torch.equal(P1, ref_P[0]) and torch.equal(P2, ref_P[1])
        assert torch.allclose(Q1, ref_Q, rtol=1e-12, atol=1e-12) and torch.allclose(Q2, ref_Q, rtol=1e-12, atol=1e-12)
        assert torch.equal(y1, ref_y) and torch.equal(y2, ref_y2)
        assert torch.equal(C1, ref_C) and torch.equal(C2, ref_C)
        assert torch.equal(g1.indices, ref_g.indices) and torch.equal(g1.indptr, ref_g.indptr)
        assert torch.equal(g2.indices, ref_g2.indices) and torch.equal(g2.values, ref_g2.values)
