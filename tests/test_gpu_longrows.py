"""The row-split kernel's LONG-ROWS path (csrc/spmm_rowsplit.hip: LongRows, spmm_longrows_kernel, spmm_longrows_combine_kernel):
with a matrix profile in scope whose longest row is at least two pieces long, rows longer than a piece are left out by the
product kernels, summed piece by piece into scratch rows and combined in order.  Against the oracle's gemm_csr_drm_as_drm /
_dcm (src/matmul.cpp:118-185) at 1e-12 (f64) / 1e-5 (f32) — a regrouping of the row's sum, like the segments — and EXACTLY on
small-integer operands, where a lost or doubled piece is a wrong integer; every lane-group width, segments, row groups,
column panels in both launch forms, both layouts of C, the same bits on every repeat."""
import ctypes as C

import numpy as np
import pytest
import torch

from matrixextra_amd import _lib, device as D
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def _matrix(m, K, rng, base_len, long_rows, sort=True):
    lens = rng.integers(0, 2 * base_len + 1, size=m)
    for r, l in long_rows.items():
        lens[r] = l
    lens = np.minimum(lens, K)
    p = np.zeros(m + 1, dtype=np.int32); p[1:] = np.cumsum(lens)
    j = np.empty(int(p[-1]), dtype=np.int32)
    for r in range(m):
        c = rng.choice(K, size=int(lens[r]), replace=False).astype(np.int32)
        j[p[r]:p[r + 1]] = np.sort(c) if sort else c
    return p, j, lens


def _long_counts():
    rows, pieces = C.c_longlong(0), C.c_longlong(0)
    _lib.check(_lib.load().mxd_debug_rowsplit_long_rows(C.byref(rows), C.byref(pieces)))
    return rows.value, pieces.value


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("colmajor", [False, True])
@pytest.mark.parametrize("n,S,P", [(100, 1, 1), (16, 1, 1), (64, 4, 1), (260, 1, 1), (16, -1, 1), (32, -1, 2), (128, 1, 3), (48, 2, 2)])
def test_long_rows_are_cut_into_pieces_and_combined_in_order(gpu, monkeypatch, dtype, colmajor, n, S, P):
    monkeypatch.setenv("MXGPU_LONG_PIECE", "128")
    rng = np.random.default_rng(11)
    m, K = 1203, 3000
    long_rows = {0: 128 * 2, 5: 128 * 2 + 1, 77: 2999, 640: 1500, 641: 700, m - 1: 1000, 300: 128, 301: 129}
    p, j, lens = _matrix(m, K, rng, 10, long_rows)
    expect_rows = int((lens > 128).sum())
    expect_pieces = int(np.ceil(lens[lens > 128] / 128).sum())
    for exact in (True, False):
        x = rng.integers(-2, 3, size=j.size).astype(np.float64) if exact else rng.uniform(-1, 1, size=j.size)
        B = (rng.integers(-3, 4, size=(K, n)) if exact else rng.normal(size=(K, n))).astype(dtype)
        A = D.DeviceCSR.from_host(p, j, x, K)
        Bd = torch.from_numpy(B).cuda()
        for launches in ("0", "1"):
            monkeypatch.setenv("MXGPU_ROWSPLIT_LAUNCHES", launches)
            got = D.spmm(A, Bd, colmajor=colmajor, algo=4, npanels=P, wg_per_cu=S).cpu().numpy()
            assert _long_counts() == (expect_rows, expect_pieces)
            want = O.tcrossprod_csr_dense(p, j, x, np.asfortranarray(B.T), 1, True)
            if exact:
                assert np.array_equal(got, want)
            else:
                tol = 1e-12 if dtype == np.float64 else 1e-5
                np.testing.assert_allclose(got, want, rtol=tol, atol=tol * 50)
            for _ in range(3):
                assert np.array_equal(D.spmm(A, Bd, colmajor=colmajor, algo=4, npanels=P, wg_per_cu=S).cpu().numpy(), got)
            if P == 1:
                break


def test_long_rows_path_stays_off_without_long_rows_or_without_a_profile(gpu, monkeypatch):
    """Rows of even length never pay the two extra launches; the bare C-ABI call (no profile in scope) keeps the kernel's
    bit-for-bit storage-order chain even when long rows exist."""
    from devmem import spmm_device
    rng = np.random.default_rng(5)
    m, K, n = 500, 2000, 128
    p, j, lens = _matrix(m, K, rng, 40, {})
    x = rng.uniform(-1, 1, size=j.size)
    B = rng.normal(size=(K, n))
    A = D.DeviceCSR.from_host(p, j, x, K)
    D.spmm(A, torch.from_numpy(B).cuda(), algo=4)
    assert _long_counts() == (0, 0)
    p, j, lens = _matrix(m, K, rng, 40, {3: 1900, 100: 1500})
    x = rng.uniform(-1, 1, size=j.size)
    got = spmm_device(p, j, x, B, False, 4, False, npanels=1, wg_per_cu=1)
    assert _long_counts() == (0, 0)
    assert np.array_equal(got, O.tcrossprod_csr_dense(p, j, x, np.asfortranarray(B.T), 1, True))
    A = D.DeviceCSR.from_host(p, j, x, K)                     # with the profile: the two long rows are cut
    got2 = D.spmm(A, torch.from_numpy(B).cuda(), algo=4).cpu().numpy()
    rows, pieces = _long_counts()
    piece = 128                                               # the default rule: ~6 mean rows, a power of two in [128, 1024]
    while piece < 1024 and piece < 6.0 * j.size / m:
        piece *= 2
    assert rows == 2 and pieces == int(np.ceil(1900 / piece) + np.ceil(1500 / piece))
    np.testing.assert_allclose(got2, got, rtol=1e-12, atol=1e-12)


def test_auto_with_long_rows_matches_the_oracle_at_size(gpu):
    """m = 1e5, 64 per row and four rows of 10,000 entries (tools/cliff_hunt.py's `giant`): AUTO (plan / profile kept) against
    sampled oracle rows, the long rows among them."""
    rng = np.random.default_rng(7)
    m, K, n = 100_000, 10_000, 64
    lens = np.full(m, 64, dtype=np.int64)
    giant = rng.integers(0, m, size=4)
    lens[giant] = 10_000
    p = np.zeros(m + 1, dtype=np.int64); p[1:] = np.cumsum(lens)
    row = np.repeat(np.arange(m, dtype=np.int64), lens)
    key = np.unique(row * K + rng.integers(0, K, size=row.size))
    row = key // K
    j = (key - row * K).astype(np.int32)
    p = np.zeros(m + 1, dtype=np.int64); np.cumsum(np.bincount(row, minlength=m), out=p[1:])
    x = rng.uniform(-1, 1, size=j.size)
    B = rng.normal(size=(K, n))
    A = D.DeviceCSR.from_host(p.astype(np.int32), j, x, K)
    got = D.spmm(A, torch.from_numpy(B).cuda()).cpu().numpy()
    rows = np.unique(np.concatenate([giant, rng.integers(0, m, size=200)]))
    for r in rows:
        s, e = p[r], p[r + 1]
        want = x[s:e] @ B[j[s:e]]
        np.testing.assert_allclose(got[r], want, rtol=1e-11, atol=1e-11)


def test_spmv_auto_with_the_profile_takes_the_flat_kernel_for_very_long_rows(gpu):
    """mxd_spmv_csr_dvec_ex2: with the matrix profile AUTO knows about rows of 16k entries or more (one lane group's tail)
    and runs the flat kernel from 2^20 entries on; flat and lane-group kernels both give the reference's sums
    (src/matmul.cpp:395-416), so the result is the oracle's whichever runs."""
    rng = np.random.default_rng(4)
    m, K = 20_000, 60_000
    lens = np.full(m, 60, dtype=np.int64)
    lens[[7, 9_000, m - 1]] = [50_000, 20_000, 30_000]
    p = np.zeros(m + 1, dtype=np.int32); p[1:] = np.cumsum(lens)
    j = np.empty(int(p[-1]), dtype=np.int32)
    for r in range(m):
        j[p[r]:p[r + 1]] = np.sort(rng.choice(K, size=int(lens[r]), replace=False))
    x = rng.uniform(-1, 1, size=j.size)
    v = rng.normal(size=K)
    A = D.DeviceCSR.from_host(p, j, x, K)
    assert A.nnz >= 1 << 20 and A.profile()[33] * A.profile()[34] >= 16384
    vd = torch.from_numpy(v).cuda()
    got_auto = D.spmv(A, vd).cpu().numpy()
    got_flat = D.spmv(A, vd, algo=3).cpu().numpy()
    got_group = D.spmv(A, vd, algo=1).cpu().numpy()
    assert np.array_equal(got_auto, got_flat)                      # AUTO ran the flat kernel (bitwise its sums)
    want = np.array([np.dot(x[p[r]:p[r + 1]], v[j[p[r]:p[r + 1]]]) for r in range(m)])
    np.testing.assert_allclose(got_auto, want, rtol=1e-11, atol=1e-11)
    np.testing.assert_allclose(got_group, want, rtol=1e-11, atol=1e-11)


def test_export_of_a_matrix_with_giant_rows(gpu):
    """The export level profiles the caller's host arrays itself (api.hip host_csr_profile) and fixes the piece length for
    the WHOLE product before it runs block by block: tcrossprod_csr_dense on 1e5 rows of 64 entries with four rows of
    10,000 against oracle rows, the giant ones among them.  AUTO's family for this product is the planned sweep, whose plan
    is left for its imbalance (one octet with a 10,000-entry row): the row-split kernel runs, with the piece length chosen
    while the host-side profile was in scope."""
    from matrixextra_amd import exports as G
    rng = np.random.default_rng(17)
    m, K, n = 100_000, 10_000, 128
    lens = np.full(m, 64, dtype=np.int64)
    giant = np.array([3, 40_000, 77_777, m - 1])
    lens[giant] = 10_000
    row = np.repeat(np.arange(m, dtype=np.int64), lens)
    key = np.unique(row * K + rng.integers(0, K, size=row.size))
    row = key // K
    j = (key - row * K).astype(np.int32)
    p = np.zeros(m + 1, dtype=np.int64); np.cumsum(np.bincount(row, minlength=m), out=p[1:])
    p = p.astype(np.int32)
    x = rng.uniform(-1, 1, size=j.size)
    Y = np.asfortranarray(rng.normal(size=(n, K)))                 # tcrossprod: X %*% t(Y)
    got = G.tcrossprod_csr_dense_numeric(p, j, x, Y)
    assert _lib.load().mxd_spmm_last_kernel().decode() == "spmm_rowsplit_kernel"
    # (the export runs the product in row blocks: the counts are those of the LAST block, which holds row m - 1)
    rows, pieces = _long_counts()
    assert rows >= 1 and pieces >= 10, (rows, pieces)
    for r in np.concatenate([giant, rng.integers(0, m, size=100)]):
        s, e = p[r], p[r + 1]
        np.testing.assert_allclose(got[r], x[s:e] @ Y[:, j[s:e]].T, rtol=1e-11, atol=1e-11)


def test_auto_leaves_a_plan_whose_longest_octet_outlasts_the_sweep(gpu):
    """Rows SORTED by length (longest first) against a narrow B: one octet of 64 long rows is one wavefront's work item and
    outlasts everything else of the planned sweep (tools/cliff_hunt.py: 1.98 ms against 0.33 for equal rows).
    mxd_spmm_plan_imbalance reports it and AUTO — the kept-plan path of DeviceCSR.spmm — runs the row-split kernel instead;
    the same rows in random order keep the plan.  Both against sampled oracle rows."""
    rng = np.random.default_rng(23)
    m, K, n = 200_000, 50_000, 32
    lens = np.minimum(np.floor(rng.lognormal(np.log(100) - 0.5, 1.0, size=m)).astype(np.int64), K)
    B = rng.normal(size=(K, n))
    Bd = torch.from_numpy(B).cuda()
    lib = _lib.load()
    for order, want_kernel in (("sorted", "spmm_rowsplit_kernel"), ("random", "spmm_plan_kernel")):
        l = np.sort(lens)[::-1].copy() if order == "sorted" else lens
        row = np.repeat(np.arange(m, dtype=np.int64), l)
        key = np.unique(row * K + rng.integers(0, K, size=row.size))
        row = key // K
        j = (key - row * K).astype(np.int32)
        p = np.zeros(m + 1, dtype=np.int64); np.cumsum(np.bincount(row, minlength=m), out=p[1:])
        x = rng.uniform(-1, 1, size=j.size)
        A = D.DeviceCSR.from_host(p.astype(np.int32), j, x, K)
        got = D.spmm(A, Bd).cpu().numpy()
        assert lib.mxd_spmm_last_kernel().decode() == want_kernel, order
        plan = A.auto_plan(0)
        assert plan is not None
        imb = C.c_double(0.0)
        _lib.check(lib.mxd_spmm_plan_imbalance(plan, C.c_int(n), C.c_int(_lib.MX_F64), C.byref(imb)))
        assert (imb.value > lib.mxd_spmm_plan_imbalance_limit(C.c_int(_lib.MX_F64))) == (order == "sorted"), (order, imb.value)
        for r in np.concatenate([[0, 1, 63, 64, m - 1], rng.integers(0, m, size=60)]):
            s, e = p[r], p[r + 1]
            np.testing.assert_allclose(got[r], x[s:e] @ B[j[s:e]], rtol=1e-11, atol=1e-11)
        del A


def test_export_spmv_with_very_long_rows_takes_the_flat_kernel(gpu):
    """matmul_csr_dvec through the export on 20,000 rows (below the 32k rows from which the exports take the flat kernel
    anyway) with rows of 50,000 / 20,000 entries: the host-side row pointers show them and the flat kernel runs — rows of
    at most 256 entries are the reference's loop bit for bit (src/matmul.cpp:395-416; the lane-group kernel regroups
    them), the long rows are summed by a wavefront."""
    from matrixextra_amd import exports as G
    rng = np.random.default_rng(8)
    m, K = 20_000, 60_000
    lens = np.full(m, 60, dtype=np.int64)
    lens[[11, 5_000, m - 2]] = [50_000, 20_000, 16_384]
    p = np.zeros(m + 1, dtype=np.int32); p[1:] = np.cumsum(lens)
    j = np.empty(int(p[-1]), dtype=np.int32)
    for r in range(m):
        j[p[r]:p[r + 1]] = np.sort(rng.choice(K, size=int(lens[r]), replace=False))
    x = rng.uniform(-1, 1, size=j.size)
    v = rng.normal(size=K)
    assert j.size >= 1 << 20
    short = lens <= 256
    got, want = G.matmul_csr_dvec_numeric(p, j, x, v), O.matmul_csr_dvec_numeric(p, j, x, v)
    assert np.array_equal(got[short], want[short])
    np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-11)
    vf = v.astype(np.float32)
    got, want = G.matmul_csr_dvec_float32(p, j, x, vf), O.matmul_csr_dvec_float32(p, j, x, vf)
    assert np.array_equal(got[short], want[short])
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=1e-3)


def _long_fit():
    rows, pieces, fit = C.c_longlong(0), C.c_longlong(0), C.c_int(0)
    _lib.check(_lib.load().mxd_debug_rowsplit_long_fit(C.byref(rows), C.byref(pieces), C.byref(fit)))
    return rows.value, pieces.value, fit.value


def test_scratch_is_sized_from_the_profile_and_a_wrong_profile_costs_speed_not_answers(gpu):
    """The long-rows scratch is sized from the profile's counts ([35] entries / [36] number of the rows longer than the
    canonical piece), not from nnz / piece; a kernel checks on the device that the launch's long rows fit.  A profile that
    understates them (another matrix's) leaves the rows to the product kernels: bit for bit the storage-order chain."""
    rng = np.random.default_rng(9)
    m, K, n = 900, 4000, 128                                     # (128 f64 columns: one wavefront per row of B, one row per wavefront)
    long_rows = {r: int(rng.integers(1100, 3900)) for r in rng.choice(m, size=40, replace=False)}
    p, j, lens = _matrix(m, K, rng, 20, long_rows)
    x = rng.uniform(-1, 1, size=j.size)
    B = rng.normal(size=(K, n))
    Bd = torch.from_numpy(B).cuda()
    A = D.DeviceCSR.from_host(p, j, x, K)
    prof = A.profile()
    mean = j.size / m
    piece = 128
    while piece < 1024 and piece < 6.0 * mean:
        piece *= 2
    nlong, elong = int((lens > piece).sum()), int(lens[lens > piece].sum())
    assert prof[37] == 1.0 and nlong <= prof[36] <= nlong * (1 + 1e-6) + 1 and elong <= prof[35] <= elong * (1 + 1e-6) + 1
    want = O.tcrossprod_csr_dense(p, j, x, np.asfortranarray(B.T), 1, True)
    got = D.spmm(A, Bd, algo=4, wg_per_cu=1, npanels=1).cpu().numpy()
    need_pieces = int(np.ceil(lens[lens > piece] / piece).sum())
    assert _long_fit() == (nlong, need_pieces, 1) and _long_counts() == (nlong, need_pieces)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
    # the same matrix with a profile that knows of one long row of 2,000 entries only: nothing fits, nothing is diverted
    forged = (C.c_float * len(prof))(*prof)
    forged[35], forged[36] = 2000.0, 1.0
    A._profile = forged
    got2 = D.spmm(A, Bd, algo=4, wg_per_cu=1, npanels=1).cpu().numpy()
    assert _long_fit() == (nlong, need_pieces, 0) and _long_counts() == (0, 0)
    assert np.array_equal(got2, want)                        # one wavefront per row, storage order: the FMA oracle's bits
    # ... and one that overstates them changes nothing
    forged[35], forged[36] = 4.0e6, 900.0
    got3 = D.spmm(A, Bd, algo=4, wg_per_cu=1, npanels=1).cpu().numpy()
    assert _long_fit()[2] == 1 and np.array_equal(got3, got)
