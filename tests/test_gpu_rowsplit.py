"""The row-split SpMM kernel (csrc/spmm_rowsplit.hip, MX_SPMM_ROWSPLIT) against the oracle's gemm_csr_drm_as_drm /
gemm_csr_drm_as_dcm (src/matmul.cpp:118-185): every segment count, column panels (sorted and unsorted rows), every
lane-group width (n = 2 .. 300), both layouts of C, f64 and f32, aligned and unaligned operands, rows longer than several
chunks, empty rows, m not a multiple of the rows a workgroup takes.  Tolerance 1e-12 (f64) / 1e-5 (f32) relative; bit for
bit where the kernel keeps the reference's storage-order FMA chain (one segment per row, B's rows filling the wavefront,
and — with panels — row-major C, where the chain continues from the value left in C)."""
import numpy as np
import pytest

from conftest import rand_csr
from devmem import spmm_device
from matrixextra_amd import _lib, exports as G, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
ROWSPLIT = 4


def _oracle(p, j, x, B, use_fma):
    """A %*% B with B (K x n) row-major: tcrossprod_csr_dense takes Y = t(B) column-major, i.e. the same bytes"""
    return O.tcrossprod_csr_dense(p, j, x, np.asfortranarray(B.T), 1, use_fma)


def _run(p, j, x, B, colmajor, S, P=1):
    return spmm_device(p, j, x, B, colmajor, ROWSPLIT, False, npanels=P, wg_per_cu=S)


SHAPES = [                      # m, K, n, density
    (37, 3000, 100, 0.3),       # ~900 entries / row: 15 chunks, the vignette's shape in small
    (1, 5000, 128, 0.5),        # one very long row
    (100, 50, 20, 0.4),         # test-matmul.R:108-114
    (64, 700, 16, 0.2), (65, 700, 64, 0.2), (9, 400, 130, 0.6), (23, 90, 256, 0.3), (10, 12, 300, 0.5),
    (130, 1000, 8, 0.1), (7, 64, 2, 1.0),
]


@pytest.mark.parametrize("S", [1, 2, 4, 8])
@pytest.mark.parametrize("colmajor", [False, True])
@pytest.mark.parametrize("m,K,n,dens", SHAPES)
def test_rowsplit_f64(gpu, m, K, n, dens, colmajor, S):
    p, j, x = rand_csr(m, K, dens, seed=m + 13 * n, sorted_cols=False, empty_rows=(0, m // 2) if m > 4 else ())
    B = np.random.default_rng(n + S).normal(size=(K, n))
    got = _run(p, j, x, B, colmajor, S)
    ref = _oracle(p, j, x, B, False)
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
    if S == 1 and n > 64:        # 64 lanes x 16 B per row of B, one wavefront per row: the storage-order FMA chain
        np.testing.assert_array_equal(got, _oracle(p, j, x, B, True))
    # run-to-run reproducible in every configuration (fixed-order LDS combine, no atomics)
    np.testing.assert_array_equal(got, _run(p, j, x, B, colmajor, S))


@pytest.mark.parametrize("P", [2, 5, 32])
@pytest.mark.parametrize("S", [1, 4])
@pytest.mark.parametrize("colmajor", [False, True])
@pytest.mark.parametrize("m,K,n,dens", [(37, 3000, 100, 0.3), (64, 700, 16, 0.2), (9, 400, 130, 0.6), (130, 1000, 8, 0.1), (1, 5000, 128, 0.5)])
def test_rowsplit_column_panels(gpu, m, K, n, dens, colmajor, S, P):
    """P launches over column panels: sorted rows are cut at the panel bounds; the result is the one-launch result up to
    rounding, and bitwise the storage-order chain when nothing reassociates (S = 1, G = 64, row-major C)"""
    p, j, x = rand_csr(m, K, dens, seed=m + n, sorted_cols=True, empty_rows=(0,) if m > 4 else ())
    B = np.random.default_rng(n + P).normal(size=(K, n))
    got = _run(p, j, x, B, colmajor, S, P)
    ref = _oracle(p, j, x, B, False)
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
    if S == 1 and n > 64 and not colmajor:
        np.testing.assert_array_equal(got, _oracle(p, j, x, B, True))
    np.testing.assert_array_equal(got, _run(p, j, x, B, colmajor, S, P))


def test_rowsplit_column_panels_with_unsorted_and_duplicate_columns(gpu):
    """rows that are not sorted by column are taken whole by the first launch (the cursor kernel's per-row check,
    src/misc.cpp:118-128's test): some rows sorted, some shuffled, some with repeated column ids (SpMM accumulates
    duplicates)"""
    m, K, n = 90, 2000, 100
    p, j, x = rand_csr(m, K, 0.2, seed=12, sorted_cols=True)
    rng = np.random.default_rng(5)
    for r in range(0, m, 3):                       # every third row shuffled
        s, e = p[r], p[r + 1]
        perm = rng.permutation(e - s)
        j[s:e], x[s:e] = j[s:e][perm], x[s:e][perm]
    for r in range(1, m, 7):                       # repeated ids, still non-decreasing
        s, e = p[r], p[r + 1]
        if e - s > 4:
            j[s + 2] = j[s + 1]
    B = rng.normal(size=(K, n))
    ref = _oracle(p, j, x, B, False)
    for P in (1, 3, 8):
        for S in (1, 2):
            for colmajor in (False, True):
                got = _run(p, j, x, B, colmajor, S, P)
                np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())
    np.testing.assert_array_equal(_run(p, j, x, B, False, 1, 4), _oracle(p, j, x, B, True))


@pytest.mark.parametrize("launches", ["0", "1"])
@pytest.mark.parametrize("S,colmajor,dtype", [(1, False, np.float64), (1, True, np.float64), (2, False, np.float64), (0, False, np.float64),
                                              (0, True, np.float64), (1, False, np.float32)])
def test_column_panels_in_one_launch_and_in_one_launch_per_panel(gpu, monkeypatch, S, colmajor, dtype, launches):
    """Both forms of the panels (spmm_rowsplit.hip: one panel-major launch whose workgroups hand C over with agent-scope
    accesses and a counter per row block; one launch per panel), forced by MXGPU_ROWSPLIT_LAUNCHES, on a grid of several
    rounds of the machine (20,000 rows = 2,500 .. 5,000 workgroups per panel, 5 panels, 2 passes): the same bits from both,
    the storage-order FMA chain for row-major C with one segment per row, and the same bits again on every repeat."""
    m, K, n = 20_000, 4_000, 160 if dtype == np.float64 else 320
    if S == 0:
        n = 16
    p, j, x = synth.csr_fixed(m, K, 24 if S == 0 else 96, seed=5)
    rng = np.random.default_rng(3)
    B = rng.normal(size=(K, n)).astype(dtype)
    monkeypatch.setenv("MXGPU_ROWSPLIT_LAUNCHES", launches)
    got = _run(p, j, x, B, colmajor, -1 if S == 0 else S, P=5)
    chain = (S in (0, 1)) and not colmajor
    want = _oracle(p, j, x, B.astype(np.float64) if dtype == np.float64 else B, True)
    if dtype == np.float32:
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)
    elif chain:                                    # one wavefront (n = 160) / one lane group (row groups) sums a row in storage order
        assert np.array_equal(got, want)
    else:
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
    for _ in range(3):
        assert np.array_equal(_run(p, j, x, B, colmajor, -1 if S == 0 else S, P=5), got)
    monkeypatch.setenv("MXGPU_ROWSPLIT_LAUNCHES", "1" if launches == "0" else "0")
    assert np.array_equal(_run(p, j, x, B, colmajor, -1 if S == 0 else S, P=5), got)


@pytest.mark.parametrize("S", [1, 4])
@pytest.mark.parametrize("colmajor", [False, True])
@pytest.mark.parametrize("m,K,n,dens", SHAPES[:8])
def test_rowsplit_f32(gpu, m, K, n, dens, colmajor, S):
    p, j, x = rand_csr(m, K, dens, seed=m + 5 * n, sorted_cols=True, empty_rows=(0,) if m > 4 else ())
    B = np.random.default_rng(n).normal(size=(K, n)).astype(np.float32)
    ref = _oracle(p, j, x, B, False)
    for P in (1, 3):
        got = _run(p, j, x, B, colmajor, S, P)
        assert got.dtype == np.float32
        np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())
        if S == 1 and n > 128 and not (colmajor and P > 1):
            np.testing.assert_array_equal(got, _oracle(p, j, x, B, True))


def test_rowsplit_unaligned_operands_take_the_scalar_path(gpu):
    """odd n -> rows of B not 16-byte aligned: VEC = 1, every shape through G = 64"""
    for (m, K, n) in ((41, 600, 33), (12, 2000, 101), (300, 40, 7)):
        p, j, x = rand_csr(m, K, 0.3, seed=n, sorted_cols=True)
        B = np.random.default_rng(n).normal(size=(K, n))
        for S in (1, 4):
            for colmajor in (False, True):
                for P in (1, 2):
                    got = _run(p, j, x, B, colmajor, S, P)
                    np.testing.assert_allclose(got, _oracle(p, j, x, B, False), rtol=1e-12, atol=1e-12)
                    if S == 1 and not (colmajor and P > 1):
                        np.testing.assert_array_equal(got, _oracle(p, j, x, B, True))


def test_rowsplit_nonfinite_values_propagate_like_the_reference(gpu):
    """NaN / Inf in A or B reach exactly the cells they reach in the reference's loop — in particular the idle entry slots
    of a partly filled load (entries past the segment's end) must not contribute a 0 * Inf"""
    m, K, n = 50, 300, 24
    p, j, x = rand_csr(m, K, 0.1, seed=3, sorted_cols=True)
    B = np.random.default_rng(4).normal(size=(K, n))
    B[j[0], :] = np.inf                         # the row of B that idle entry slots re-read
    B[j[5], 3] = np.nan
    x[7] = np.inf
    ref = _oracle(p, j, x, B, False)
    for S, P in ((1, 1), (2, 1), (8, 1), (1, 3)):
        got = _run(p, j, x, B, False, S, P)
        np.testing.assert_array_equal(np.isnan(got), np.isnan(ref))
        np.testing.assert_array_equal(np.isinf(got), np.isinf(ref))
        ok = np.isfinite(ref)
        np.testing.assert_allclose(got[ok], ref[ok], rtol=1e-12, atol=1e-12)


def test_rowsplit_auto_shape_and_unknown_nnz(gpu):
    """npanels = wg_per_cu = 0: segments and panels come from m, n, K and the mean row length — read from the device when
    the caller does not pass nnz (mxd_spmm_csr_dense_ex) — and the result does not depend on them beyond rounding"""
    p, j, x = synth.csr_fixed(200, 20_000, 900, seed=5)
    B = synth.dense_normal(20_000, 100, seed=6)              # 16 MB: several panels
    got = spmm_device(p, j, x, B, False, ROWSPLIT, False, npanels=0, wg_per_cu=0)
    np.testing.assert_allclose(got, _oracle(p, j, x, B, False), rtol=1e-12, atol=1e-11)
    assert _lib.load().mxd_spmm_last_kernel() == b"spmm_rowsplit_kernel"


def test_auto_families_around_the_published_workload(gpu):
    """AUTO's family: TILE (round 5; ROWSPLIT before) for dense 100 x 1e4 %*% CSC 1e4 x 1e4 (vignette Rmd:247-251), ROWWAVE for the reference's
    tiny test shapes (which keep their bitwise storage-order sums) and for callers that do not know nnz, PLANNED at the
    headline size; and the export (matmul_dense_csc_numeric, src/matmul.cpp:221-235) at a fifth of the vignette's size
    really runs the row-split kernel (2,000 rows do not fill the tile kernel's workgroups)"""
    import ctypes as C
    lib = _lib.load()
    pick = C.c_int(0)
    al = C.c_void_p(256)
    for (m, n, K, nnz, colmajor, want) in ((10_000, 100, 10_000, 5_000_000, 0, 5), (100, 20, 50, 2000, 1, 1), (1_000_000, 128, 100_000, 32_000_000, 1, 3),
                                           (10_000, 100, 10_000, -1, 0, 1)):
        _lib.check(lib.mxd_spmm_auto_algo2(C.c_int(m), C.c_int(n), C.c_int(K), C.c_int64(nnz), C.c_int(0), C.c_int(_lib.MX_F64), al, C.c_size_t(n), al,
                                           C.c_size_t(m if colmajor else n), C.c_int(colmajor), C.byref(pick)))
        assert pick.value == want, (m, n, K, nnz, pick.value)
    pc, ic, xc = synth.csr_fixed(2000, 10_000, 500, seed=7)
    X = np.asfortranarray(synth.dense_normal(100, 10_000, seed=8))
    got = G.matmul_dense_csc_numeric(X, pc, ic, xc, 1)
    assert lib.mxd_spmm_last_kernel() == b"spmm_rowsplit_kernel"
    ref = O.matmul_dense_csc(X, pc, ic, xc, O.max_threads(), False)
    np.testing.assert_allclose(got, ref, rtol=1e-12, atol=1e-12 * np.abs(ref).max())


# ------------------------------------------------------------------------------------------------ the row-group form
ROWGROUPS = -1          # wg_per_cu = -1: G lanes own one row of A, 64 / G rows per wavefront (spmm_rowgroup_kernel)
GROUP_SHAPES = [        # m, K, n, density — n <= 32 * VEC so that G < 64 (f64: n <= 64, f32: n <= 128)
    (1000, 300, 16, 0.03), (1000, 300, 16, 0.3), (517, 90, 8, 0.2), (64, 700, 32, 0.05), (65, 40, 64, 0.5), (7, 64, 2, 1.0),
    (1, 50, 16, 0.5), (513, 1000, 48, 0.02), (200, 30, 24, 0.4), (129, 2000, 64, 0.01),
]


@pytest.mark.parametrize("colmajor", [False, True])
@pytest.mark.parametrize("m,K,n,dens", GROUP_SHAPES)
def test_rowgroup_form_is_the_storage_order_fma_chain_f64(gpu, m, K, n, dens, colmajor):
    """every row is summed by its own lane group in storage order: bit for bit gemm_csr_drm_as_drm's chain with fused
    multiply-add (src/matmul.cpp:118-142) — unsorted rows, duplicate column ids, empty rows, rows longer than a group,
    Inf / NaN in A (entries past a row's end are dropped by a select, never multiplied in)"""
    p, j, x = rand_csr(m, K, dens, seed=3 * m + n, sorted_cols=False, empty_rows=(0, m // 2, m - 1) if m > 4 else ())
    if x.size > 10:
        x[1], x[x.size // 2], x[-2] = np.inf, np.nan, -np.inf
    if j.size > 6:
        j[3] = j[2]                                              # a repeated column id (may cross a row boundary: harmless)
    B = np.random.default_rng(n).normal(size=(K, n))
    got = _run(p, j, x, B, colmajor, ROWGROUPS)
    ref = _oracle(p, j, x, B, True)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    ok = ~np.isnan(ref)
    np.testing.assert_array_equal(got[ok], ref[ok])
    np.testing.assert_array_equal(np.nan_to_num(got), np.nan_to_num(_run(p, j, x, B, colmajor, ROWGROUPS)))


@pytest.mark.parametrize("colmajor", [False, True])
@pytest.mark.parametrize("m,K,n,dens", GROUP_SHAPES[:8] + [(300, 500, 128, 0.05), (90, 200, 96, 0.2)])
def test_rowgroup_form_f32(gpu, m, K, n, dens, colmajor):
    p, j, x = rand_csr(m, K, dens, seed=m + 7 * n, sorted_cols=True, empty_rows=(0,) if m > 4 else ())
    B = np.random.default_rng(n).normal(size=(K, n)).astype(np.float32)
    got = _run(p, j, x, B, colmajor, ROWGROUPS)
    assert got.dtype == np.float32
    if n % 4 == 0:                      # 16-byte rows of B: the row-group kernel itself — the f32 chain with alpha narrowed per entry
        np.testing.assert_array_equal(got, _oracle(p, j, x, B, True))
    else:                               # otherwise the call falls back to one wavefront per row (scalar accesses)
        np.testing.assert_allclose(got, _oracle(p, j, x, B, False), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("P", [2, 5])
@pytest.mark.parametrize("colmajor", [False, True])
def test_rowgroup_form_with_column_panels(gpu, colmajor, P):
    """forced panels: launch p continues from what launch p - 1 left in C — for row-major C the chain itself (bitwise)"""
    m, K, n = 700, 3000, 16
    p, j, x = rand_csr(m, K, 0.02, seed=4, sorted_cols=True, empty_rows=(0, 5))
    B = np.random.default_rng(2).normal(size=(K, n))
    got = _run(p, j, x, B, colmajor, ROWGROUPS, P)
    if not colmajor:
        np.testing.assert_array_equal(got, _oracle(p, j, x, B, True))
    else:
        np.testing.assert_allclose(got, _oracle(p, j, x, B, False), rtol=1e-12, atol=1e-12)


def test_auto_takes_the_rowgroup_form_for_many_short_rows(gpu):
    """m = 2e5 rows of 12 entries against a 16-column B: AUTO -> row-split family -> the row-group form; the result is the
    storage-order chain bit for bit, whatever form the heuristic took"""
    import torch
    from matrixextra_amd import device as D
    m, K, n = 200_000, 5000, 16
    p, j, x = synth.csr_fixed(m, K, 12, seed=8)
    A = D.DeviceCSR.from_host(p, j, x, K)
    B = synth.dense_normal(K, n, seed=9)
    got = D.spmm(A, torch.from_numpy(B).cuda(), keep_plan=False).cpu().numpy()
    assert _lib.load().mxd_spmm_last_kernel().decode() == "spmm_rowgroup_kernel"
    rows = np.r_[0:300, m - 300:m]
    pp = np.concatenate([[0], np.cumsum(np.diff(p)[rows])]).astype(np.int32)
    sel = np.concatenate([np.arange(p[r], p[r + 1]) for r in rows])
    np.testing.assert_array_equal(got[rows], _oracle(pp, j[sel], x[sel], B, True))
